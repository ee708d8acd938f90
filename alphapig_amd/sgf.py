"""SGF game-record ingestion: the bootstrap phase of the reference pipeline.

Drop-in for reference utils/sgf_dataIter.py:27-66 (`get_files_as_list`, `content_to_order`,
`get_data_from_files`) -- same parsing rules, quirks included:

  * the move list is the text after "SZ[15]" plus one more character, minus the last 4
    characters of the file; it is split on ';' and characters 2..3 of every item are the two
    board letters ('a'..'o'), move = 15 * index(first letter) + index(second letter);
  * the winner comes from the FILE NAME: the 5 characters after the first '_' are
    'Blank'/'blank' (player 1, black) or 'White'/'white' (player 2).

`replay` turns a record into training tuples exactly like the reference's SGF branch of
Game.start_self_play (game.py:233-304): pi = 0.99999 on the played move over a 1e-6 floor,
z from the file-name winner.
"""
import os

import numpy as np

LETTERS = "abcdefghijklmno"
_LOOKUP = {c: i for i, c in enumerate(LETTERS)}


def get_files_as_list(data_dir):
    return [f for f in os.listdir(data_dir) if f.endswith(".sgf") and os.path.isfile(os.path.join(data_dir, f))]


def content_to_order(sequence):
    seq_list = [item[2:4] for item in sequence.split(";")]
    return seq_list, [_LOOKUP[s[0]] * 15 + _LOOKUP[s[1]] for s in seq_list]


def parse(text, file_name):
    """-> {'winner', 'seq_list', 'seq_num_list', 'file_name'} (sgf_dataIter.py:45-66)."""
    assert file_name.endswith(".sgf"), "file: %s is not an SGF file" % file_name
    file_name.index("_.")                                   # the reference requires it too
    sequence = text[text.index("SZ[15]") + 7:-4]
    seq_list, seq_num_list = content_to_order(sequence)
    tag = file_name[file_name.index("_") + 1:file_name.index("_") + 6]
    if tag in ("Blank", "blank"):
        winner = 1
    elif tag in ("White", "white"):
        winner = 2
    else:
        raise ValueError("file name %r names no winner (Blank/White after the first '_')" % file_name)
    return {"winner": winner, "seq_list": seq_list, "seq_num_list": seq_num_list, "file_name": file_name}


def get_data_from_files(file_name, data_dir):
    with open(os.path.join(data_dir, file_name)) as f:
        return parse(f.read(), file_name)


def replay(board, record):
    """-> (warning, winner, [(state, pi, z)]) ; (1, None, None) on an illegal move, like the
    reference (game.py:265-270)."""
    board.init_board()
    size = board.width * board.height
    states, pis, movers = [], [], []
    for move in record["seq_num_list"]:
        pi = np.full(size, 0.000001)
        pi[move] = 0.99999
        states.append(board.current_state())
        pis.append(pi)
        movers.append(board.current_player)
        try:
            board.do_move(move)
        except Exception:
            return 1, None, None
    winner = record["winner"]
    z = np.zeros(len(movers))
    if winner != -1:
        mv = np.array(movers)
        z[mv == winner] = 1.0
        z[mv != winner] = -1.0
    return 0, winner, list(zip(states, pis, z))
