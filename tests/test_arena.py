"""Arena: M concurrent evaluation matches == the reference's sequential
`policy_evaluate` loop (train_mxnet.py:242-263 -> Game.start_play), match for match."""
import numpy as np

from alphapig_amd import mcts_pure
from alphapig_amd.arena import Arena, win_ratio
from alphapig_amd.game import Board, Game
from alphapig_amd.mcts_alphaZero import MCTSPlayer
from fakenet import fake_policy_value_batch, fake_policy_value_fn
from oracle.board_ref import RefBoard
from oracle.mcts_ref import RefMCTSPlayer, RefPureMCTSPlayer


def sequential_match(i, base, w, nrow, npl, pure_n):
    """The reference loop body with the drop-in classes (themselves pinned to golden traces)."""
    b = Board(width=w, height=w, n_in_row=nrow)
    np.random.seed(base + i)
    winner = Game(b).start_play(MCTSPlayer(fake_policy_value_fn, c_puct=5, n_playout=npl),
                                mcts_pure.MCTSPlayer(c_puct=5, n_playout=pure_n), start_player=i % 2, is_shown=0)
    return winner, [m for m, _ in b.history]


def oracle_match(i, base, w, nrow, npl, pure_n):
    b = RefBoard(w, w, nrow)
    b.init_board(i % 2)
    rs = np.random.RandomState(base + i)
    players = {1: RefMCTSPlayer(fake_policy_value_fn, 5, npl, 0, rng=rs), 2: RefPureMCTSPlayer(5, pure_n, rng=rs)}
    while True:
        mv = players[b.get_current_player()].get_action(b)
        b.do_move(int(mv))
        end, winner = b.game_end()
        if end:
            return winner, list(b.move_list)


def test_arena_equals_sequential_policy_evaluate():
    w, nrow, npl, pure_n, base, n = 8, 4, 30, 40, 700, 6
    res = Arena(fake_policy_value_batch, w, w, nrow, n_playout=npl, pure_mcts_playout_num=pure_n, base_seed=base,
                n_threads=2, max_concurrent=4).play(n)          # two blocks: 4 + 2
    assert [r.index for r in res] == list(range(n)) and [r.start_player for r in res] == [0, 1, 0, 1, 0, 1]
    for r in res:
        winner, moves = sequential_match(r.index, base, w, nrow, npl, pure_n)
        assert r.winner == winner
        assert list(r.moves) == moves
    for i in (0, 1):
        winner, moves = oracle_match(i, base, w, nrow, npl, pure_n)
        assert res[i].winner == winner and list(res[i].moves) == moves
    ratio = win_ratio(res)
    assert 0.0 <= ratio <= 1.0
    assert ratio == (sum(r.winner == 1 for r in res) + 0.5 * sum(r.winner == -1 for r in res)) / n
