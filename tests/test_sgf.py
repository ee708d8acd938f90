"""SGF ingestion against golden vectors produced by the reference's own parser
(utils/sgf_dataIter.py:27-66) and SGF replay (game.py:233-304)."""
import os

import numpy as np
import pytest

from alphapig_amd import sgf
from alphapig_amd.game import Board, Game


class _Player(object):
    def __init__(self):
        self.resets = 0

    def reset_player(self):
        self.resets += 1


def test_parser_and_replay_match_reference(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "sgf.npz"))
    n_warn = 0
    for k in range(int(g["n"])):
        name, text = str(g["f%d_name" % k]), str(g["f%d_text" % k])
        rec = sgf.parse(text, name)
        assert rec["winner"] == int(g["f%d_winner" % k])
        assert rec["seq_list"] == [str(x) for x in g["f%d_seq_list" % k]]
        np.testing.assert_array_equal(np.array(rec["seq_num_list"]), g["f%d_seq_num" % k])
        with open(tmp_path / name, "w", newline="") as f:
            f.write(text)
        assert name in sgf.get_files_as_list(str(tmp_path))
        b = Board(width=15, height=15, n_in_row=5)
        pl = _Player()
        warning, winner, data = Game(b).start_self_play(pl, sgf_home=str(tmp_path), file_name=name)
        assert warning == int(g["f%d_warning" % k])
        if warning:
            n_warn += 1
            assert winner is None and data is None
            continue
        data = list(data)
        assert winner == int(g["f%d_replay_winner" % k]) and pl.resets == 1
        np.testing.assert_array_equal(np.stack([np.ascontiguousarray(d[0]) for d in data]).astype(np.uint8),
                                      g["f%d_states" % k])
        np.testing.assert_array_equal(np.stack([d[1] for d in data]), g["f%d_pis" % k])
        np.testing.assert_array_equal(np.array([d[2] for d in data]), g["f%d_zs" % k])
    assert n_warn == 1


def test_parser_errors():
    with pytest.raises(AssertionError):
        sgf.parse("(;SZ[15];B[hh])\n\n", "0001_Blank_.txt")
    with pytest.raises(ValueError):
        sgf.parse("(;SZ[15];B[hh])\n\n", "0001_Nobody_.sgf")
    with pytest.raises(ValueError):
        sgf.parse("(;GM[4];B[hh])\n\n", "0001_Blank_.sgf")      # no SZ[15]
