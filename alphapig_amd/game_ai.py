"""Game_AI: drop-in for the reference's self-play episode driver (reference game_ai.py:11-139).

`Game_AI(board).start_self_play(player, is_shown=0, temp=1e-3)` plays one game of the given
MCTS player against itself and returns `(winner, zip(states, mcts_probs, winners_z))`, with
the reference's 9 % forced random two-ply opening (game_ai.py:77-111) drawn from Python's
global `random` (or the `pyrandom` passed in).
"""
from __future__ import print_function

import random as _random

import numpy as np

from .game import Game

# opening book of game_ai.py:77-78: black anywhere in rows 0-6 x columns 0-8, white in 0..102
OPENING_BLACK = [row * 15 + col for row in range(7) for col in range(9)]
OPENING_WHITE = range(0, 103)
OPENING_PROB = 0.09


def one_hot_pi(size, move):
    pi = np.full(size, 0.000001)
    pi[move] = 0.99999
    return pi


def draw_forced_opening(rnd):
    """-> (black, white) or None; consumes `rnd` exactly like game_ai.py:79-85."""
    if not rnd.random() < OPENING_PROB:
        return None
    while True:
        black = rnd.choice(OPENING_BLACK)
        white = rnd.choice(OPENING_WHITE)
        if black != white:
            return black, white


def outcome_z(movers, winner):
    z = np.zeros(len(movers))
    if winner != -1:
        movers = np.array(movers)
        z[movers == winner] = 1.0
        z[movers != winner] = -1.0
    return z


class Game_AI(Game):
    def __init__(self, board, pyrandom=None, **kwargs):
        Game.__init__(self, board, **kwargs)
        self._rnd = pyrandom if pyrandom is not None else _random

    def start_self_play(self, player, is_shown=0, temp=1e-3):
        board = self.board
        board.init_board()
        p1, p2 = board.players
        states, pis, movers = [], [], []

        def record_and_play(move, pi):
            states.append(board.current_state())
            pis.append(pi)
            movers.append(board.current_player)
            board.do_move(move)
            if is_shown:
                self.graphic(board, p1, p2)

        forced = draw_forced_opening(self._rnd)
        if forced is not None:
            for mv in forced:
                record_and_play(mv, one_hot_pi(self._boardSize, mv))
        while True:
            move, move_probs = player.get_action(board, temp=temp, return_prob=1)
            record_and_play(move, move_probs)
            end, winner = board.game_end()
            if end:
                player.reset_player()
                if is_shown:
                    print("Game end. Winner is player:", winner) if winner != -1 else print("Game end. Tie")
                return winner, zip(states, pis, outcome_z(movers, winner))
