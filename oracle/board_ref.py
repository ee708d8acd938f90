"""ORACLE (test infrastructure): array-based restatement of the reference Board.

Follows /root/reference/game.py:21-170.  State is a flat int8 cell array plus the
ordered move list; `availables` stays an ascending python list with order-preserving
removal (game.py:41,120) because MCTS child order is the order of `availables` at
expansion time (mcts_alphaZero.py:39-41).
"""
import numpy as np


class RefBoard(object):
    def __init__(self, width=8, height=8, n_in_row=5):
        self.width = int(width)
        self.height = int(height)
        self.n_in_row = int(n_in_row)
        self.players = [1, 2]                                   # game.py:33

    # game.py:35-44
    def init_board(self, start_player=0):
        if self.width < self.n_in_row or self.height < self.n_in_row:
            raise Exception("board width and height can not be less than %d" % self.n_in_row)
        self.current_player = self.players[start_player]
        self.availables = list(range(self.width * self.height))
        self.cells = np.zeros(self.width * self.height, dtype=np.int8)
        self.move_list = []            # moves in order
        self.mover_list = []           # who played each
        self.last_move = -1

    def clone(self):
        c = RefBoard(self.width, self.height, self.n_in_row)
        c.current_player = self.current_player
        c.availables = list(self.availables)
        c.cells = self.cells.copy()
        c.move_list = list(self.move_list)
        c.mover_list = list(self.mover_list)
        c.last_move = self.last_move
        return c

    def __deepcopy__(self, memo):
        return self.clone()

    def get_current_player(self):                               # game.py:169
        return self.current_player

    # game.py:117-125
    def do_move(self, move):
        self.cells[move] = self.current_player
        self.move_list.append(move)
        self.mover_list.append(self.current_player)
        self.availables.remove(move)                             # ValueError if illegal
        self.current_player = 1 if self.current_player == 2 else 2
        self.last_move = move

    # game.py:68-94
    def current_state(self):
        H, W = self.height, self.width
        raw = np.zeros((9, W, H), dtype=np.float64)
        nply = len(self.move_list)
        if nply:
            for i in range(4):
                keep = nply - i                                  # first `keep` plies
                for k in range(max(keep, 0)):
                    m = self.move_list[k]
                    r, c = m // W, m % H
                    if self.mover_list[k] == self.current_player:
                        raw[8 - 2 * i - 2, r, c] = 1.0           # square_state[-2i-3]
                    else:
                        raw[8 - 2 * i - 1, r, c] = 1.0           # square_state[-2i-2]
                if keep == 0:
                    break
        if nply % 2 == 0:
            raw[8, :, :] = 1.0
        return np.ascontiguousarray(raw[:, ::-1, :])

    # game.py:96-115 (dead code in the reference, kept for the C_in=4 measurement shape)
    def current_state_old(self):
        H, W = self.height, self.width
        raw = np.zeros((4, W, H), dtype=np.float64)
        nply = len(self.move_list)
        for k in range(nply):
            m = self.move_list[k]
            r, c = m // W, m % H
            raw[0 if self.mover_list[k] == self.current_player else 1, r, c] = 1.0
        if nply:
            raw[2, self.last_move // W, self.last_move % H] = 1.0
        if nply % 2 == 0:
            raw[3, :, :] = 1.0
        return np.ascontiguousarray(raw[:, ::-1, :])

    # game.py:127-158.  The reference scans `list(set(all) - set(availables))`; the scan
    # order only matters if both colours own a line at once, which cannot happen when
    # play stops at the first win.  Scanned ascending here.
    def has_a_winner(self):
        W, H, n = self.width, self.height, self.n_in_row
        cells = self.cells
        if len(self.move_list) < n + 2:
            return False, -1
        for m in range(W * H):
            p = cells[m]
            if p == 0:
                continue
            h, w = m // W, m % W
            for (dh, dw) in ((0, 1), (1, 0), (1, 1), (1, -1)):
                if dw == 1 and not (w <= W - n):
                    continue
                if dw == -1 and not (w >= n - 1):
                    continue
                if dh == 1 and not (h <= H - n):
                    continue
                step = dh * W + dw
                ok = True
                for k in range(1, n):
                    if cells[m + k * step] != p:
                        ok = False
                        break
                if ok:
                    return True, int(p)
        return False, -1

    def game_end(self):                                         # game.py:160-167
        win, who = self.has_a_winner()
        if win:
            return True, who
        if not self.availables:
            return True, -1
        return False, -1
