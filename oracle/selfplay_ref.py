"""ORACLE (test infrastructure): restatement of one self-play episode and of the 8-fold
dihedral augmentation.

start_self_play follows /root/reference/game_ai.py:70-139; equi_data follows
/root/reference/train_mxnet.py:115-135.
"""
import numpy as np

# game_ai.py:77 -- rows 0..6, columns 0..8 of the 15-wide board
BLANK_MOVES = [r * 15 + c for r in range(7) for c in range(9)]
WHITE_MOVES = range(0, 103)                                    # game_ai.py:78


def start_self_play(board, player, temp=1e-3, pyrandom=None):
    """-> (winner, [(state[9,H,W] f64, pi[HW] f64, z)]).  `pyrandom` is a random.Random
    standing in for the reference's global `random` module (game_ai.py:79-83)."""
    import random as _random
    rnd = pyrandom if pyrandom is not None else _random
    board.init_board()
    size = board.width * board.height
    states, pis, movers = [], [], []
    if rnd.random() < 0.09:
        while True:
            mb = rnd.choice(BLANK_MOVES)
            mw = rnd.choice(WHITE_MOVES)
            if mb != mw:
                break
        for mv in (mb, mw):
            pr = np.asarray([0.000001] * size)
            pr[mv] = 0.99999
            states.append(board.current_state())
            pis.append(pr)
            movers.append(board.current_player)
            board.do_move(mv)
    while True:
        move, move_probs = player.get_action(board, temp=temp, return_prob=1)
        states.append(board.current_state())
        pis.append(move_probs)
        movers.append(board.current_player)
        board.do_move(move)
        end, winner = board.game_end()
        if end:
            z = np.zeros(len(movers))
            if winner != -1:
                mv = np.array(movers)
                z[mv == winner] = 1.0
                z[mv != winner] = -1.0
            player.reset_player()
            return winner, list(zip(states, pis, z))


def equi_data(play_data, height, width):
    """train_mxnet.py:115-135: output order r1, r1f, r2, r2f, r3, r3f, id, idf per tuple."""
    out = []
    for state, pi, z in play_data:
        grid = np.flipud(np.asarray(pi).reshape(height, width))
        for i in (1, 2, 3, 4):
            es = np.stack([np.rot90(pl, i) for pl in state])
            ep = np.rot90(grid, i)
            out.append((es, np.flipud(ep).flatten(), z))
            es = np.stack([np.fliplr(pl) for pl in es])
            ep = np.fliplr(ep)
            out.append((es, np.flipud(ep).flatten(), z))
    return out
