"""Board and Game: drop-in for the reference's rules layer (reference game.py:21-230).

Same public surface (`Board(width=, height=, n_in_row=)`, `.init_board`, `.do_move`,
`.has_a_winner`, `.game_end`, `.current_state`, `.availables`, `.states`, `.history`,
`.current_player`, `.last_move`, `.players`, `move_to_location`, `location_to_move`;
`Game(board).start_play(p1, p2, start_player, is_shown)`), re-implemented on flat NumPy
cell arrays.  `current_state()` returns the 9-plane float64 tensor of game.py:68-94.
"""
from __future__ import print_function

import numpy as np


class Board(object):
    """Gomoku board; moves are `h * width + w` with row 0 at the bottom (game.py:46-56)."""

    def __init__(self, **kwargs):
        self.width = int(kwargs.get("width", 8))
        self.height = int(kwargs.get("height", 8))
        self.n_in_row = int(kwargs.get("n_in_row", 5))
        self.players = [1, 2]
        self.states = {}

    def init_board(self, start_player=0):
        if self.width < self.n_in_row or self.height < self.n_in_row:
            raise Exception("board width and height can not be less than {}".format(self.n_in_row))
        self.current_player = self.players[start_player]
        self.availables = list(range(self.width * self.height))
        self.states = {}
        self.history = []
        self.last_move = -1
        self._cells = np.zeros(self.width * self.height, dtype=np.int8)

    # -- cheap copy: MCTS copies the board once per playout (mcts_alphaZero.py:148)
    def __deepcopy__(self, memo):
        b = Board.__new__(Board)
        b.width, b.height, b.n_in_row = self.width, self.height, self.n_in_row
        b.players = [1, 2]
        b.current_player = self.current_player
        b.availables = list(self.availables)
        b.states = dict(self.states)
        b.history = list(self.history)
        b.last_move = self.last_move
        b._cells = self._cells.copy()
        return b

    def move_to_location(self, move):
        return [move // self.width, move % self.width]

    def location_to_move(self, location):
        if len(location) != 2:
            return -1
        move = location[0] * self.width + location[1]
        if move not in range(self.width * self.height):
            return -1
        return move

    def get_current_player(self):
        return self.current_player

    def do_move(self, move):
        self.availables.remove(move)            # ValueError on an illegal move, as the reference
        self.states[move] = self.current_player
        self._cells[move] = self.current_player
        self.history.append((move, self.current_player))
        self.current_player = 2 if self.current_player == 1 else 1
        self.last_move = move

    def _stone_grid(self, upto, own):
        """[W,H] 0/1 grid of the first `upto` plies placed by (own ? current : other) player."""
        g = np.zeros((self.width, self.height))
        for mv, who in self.history[:upto]:
            if (who == self.current_player) == own:
                g[mv // self.width, mv % self.height] = 1.0
        return g

    def current_state(self):
        """(9, W, H) float64 planes from the mover's perspective, rows flipped (game.py:68-94):
        planes 6/7 = own/opponent stones now, 4/5 one ply ago, 2/3, 0/1; plane 8 = colour."""
        planes = np.zeros((9, self.width, self.height))
        nply = len(self.history)
        for drop in range(4):
            if nply - drop <= 0:
                break
            planes[6 - 2 * drop] = self._stone_grid(nply - drop, True)
            planes[7 - 2 * drop] = self._stone_grid(nply - drop, False)
        if nply % 2 == 0:
            planes[8] = 1.0
        return planes[:, ::-1, :]

    def position_codes(self):
        """Compact leaf encoding consumed by the HIP plane encoder (include/alphapig_host.h)."""
        hw = self.width * self.height
        stride = (hw + 1 + 15) // 16 * 16
        codes = np.zeros(stride, dtype=np.uint8)
        nply = len(self.history)
        for k, (mv, who) in enumerate(self.history):
            age = min(nply - 1 - k, 3)
            codes[mv] = (1 if who == self.current_player else 5) + age
        codes[hw] = 1 if nply % 2 == 0 else 0
        return codes

    def has_a_winner(self):
        """(win, winner): any n_in_row run, overlines included (game.py:127-158)."""
        n = self.n_in_row
        if len(self.history) < n + 2:
            return False, -1
        grid = self._cells.reshape(self.height, self.width)
        H, W = self.height, self.width
        for who in (1, 2):
            m = grid == who
            if not m.any():
                continue
            runs = []
            if W >= n:
                runs.append(np.logical_and.reduce([m[:, k:W - n + 1 + k] for k in range(n)]))
            if H >= n:
                runs.append(np.logical_and.reduce([m[k:H - n + 1 + k, :] for k in range(n)]))
            if W >= n and H >= n:
                runs.append(np.logical_and.reduce([m[k:H - n + 1 + k, k:W - n + 1 + k] for k in range(n)]))
                runs.append(np.logical_and.reduce(
                    [m[k:H - n + 1 + k, n - 1 - k:W - k] for k in range(n)]))
            if any(r.any() for r in runs):
                return True, who
        return False, -1

    def game_end(self):
        win, winner = self.has_a_winner()
        if win:
            return True, winner
        if not len(self.availables):
            return True, -1
        return False, -1


class Game(object):
    """Match loop between two players exposing `get_action(board)` (game.py:173-230)."""

    def __init__(self, board, **kwargs):
        self.board = board
        self._boardSize = board.width * board.height

    def graphic(self, board, player1, player2):
        print("Player", player1, "with X".rjust(3))
        print("Player", player2, "with O".rjust(3))
        print()
        print("".join("{0:8}".format(x) for x in range(board.width)))
        print("\r\n")
        sym = {player1: "X", player2: "O"}
        for i in range(board.height - 1, -1, -1):
            row = "{0:4d}".format(i)
            for j in range(board.width):
                row += sym.get(board.states.get(i * board.width + j, -1), "_").center(8)
            print(row + "\r\n\r\n")

    def start_play(self, player1, player2, start_player=0, is_shown=1):
        if start_player not in (0, 1):
            raise Exception("start_player should be either 0 (player1 first) or 1 (player2 first)")
        self.board.init_board(start_player)
        p1, p2 = self.board.players
        player1.set_player_ind(p1)
        player2.set_player_ind(p2)
        seat = {p1: player1, p2: player2}
        if is_shown:
            self.graphic(self.board, player1.player, player2.player)
        while True:
            mover = seat[self.board.get_current_player()]
            self.board.do_move(mover.get_action(self.board))
            if is_shown:
                self.graphic(self.board, player1.player, player2.player)
            end, winner = self.board.game_end()
            if end:
                if is_shown:
                    print("Game end. Winner is", seat[winner]) if winner != -1 else print("Game end. Tie")
                return winner

    def start_self_play(self, player, is_shown=0, temp=1e-3, sgf_home=None, file_name=None):
        """SGF replay used by the first phase of the reference pipeline (game.py:233-304):
        -> (warning, winner, zip(states, mcts_probs, winners_z)).  MCTS self-play lives in
        game_ai.Game_AI.start_self_play."""
        from . import sgf
        warning, winner, data = sgf.replay(self.board, sgf.get_data_from_files(file_name, sgf_home))
        if warning:
            return warning, None, None
        player.reset_player()
        if is_shown:
            print("Game end. Winner is player:", winner)
        return warning, winner, zip(*[list(x) for x in zip(*data)])

