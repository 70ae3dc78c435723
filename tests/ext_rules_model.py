"""TEST-ONLY: a second, independently written restatement of Azul in plain Python, used to cross-check the C oracle's EXTENDED rules
(row N4 of SURVEY.md 8f: 2P+1 factory displays, end-of-game bonuses, the short deal and the finite bag).

"Beyond the reference, parity unpinned": the reference implements none of these rules (azulnet/azul.py:19 deals five displays for
any number of players, the TODOs at azul.py:72,86 and tests/test_azul.py:14 say so, and its line bonuses are paid per round,
azul.py:266-288), so there is no reference behaviour to pin them to.  What can be done instead is done here: the rules are written
down twice, by different code in different styles -- oracle/azul_oracle.c follows the reference's colour-indexed arrays and loops,
this file keeps the wall in BOARD coordinates and scores runs on it -- and tests/test_ext_rules_model.py plays both against each other
on random streams, with the flags off (where the oracle is pinned to the reference, so the base rules of this model are pinned
through it) and with each flag on.

Rule source for the three rules: the published Azul rulebook (Plan B Games / Next Move, 2017), restated here from memory (no network
in the build container):
  * Setup: 5 factory displays for 2 players, 7 for 3, 9 for 4; the bag holds 100 tiles, 20 of each of the 5 colours; every display
    receives 4 tiles drawn from the bag.
  * Preparing the next round: refill the displays from the bag; when the bag is empty, refill it from the lid and continue; when
    bag and lid are both empty, start the round although not every display is filled.
  * End of the game: after the wall-tiling phase in which somebody completed a horizontal line; then +2 per complete horizontal
    line, +7 per complete vertical line, +10 per colour with all five tiles on the wall.

Randomness: a `random.Random` instance, i.e. CPython's own generator and its own `choice` / `choices` / `randrange`.
"""
import random

DISPLAYS_2P1, END_BONUS, SHORT_DEAL, FINITE_BAG = 1, 2, 4, 8
FLOOR_COST = (0, 1, 2, 4, 6, 8, 11, 14)          # cumulative cost of 0..7 tiles on the floor line


class Stuck(Exception):
    pass


class BoxEmpty(Exception):
    pass


class ModelGame:
    def __init__(self, players, first_player, pool, ext, rng):
        """first_player: "Random" or 1..players; pool: "Random" or "Lid"."""
        self.rng = rng
        self.P = players
        self.ext = ext
        self.D = 2 * players + 1 if ext & DISPLAYS_2P1 else 5
        self.factories = [[0] * 5 for _ in range(self.D)]
        self.middle = [0] * 5
        self.token_in_middle = False
        self.lines = [[(None, 0)] * 5 for _ in range(players)]       # per player and row: (colour, count); see set_lines for odd states
        self.odd_lines = None                                         # states with two colours on a row never arise from play
        self.board = [[[False] * 5 for _ in range(5)] for _ in range(players)]      # board[p][row][column]
        self.floor = [0] * players
        self.points = [0] * players
        self.to_move = 0
        self.starter = rng.choice(list(range(1, players + 1))) if first_player == "Random" else int(first_player)
        self.over = False
        self.rounds = 0
        self.pool = pool
        self.tracked = pool == "Lid" or bool(ext & FINITE_BAG)
        if pool == "Lid" and ext & FINITE_BAG:
            raise ValueError("the Lid pool already is a finite bag")
        self.bag = [20] * 5 if self.tracked else [0] * 5
        self.discard = [0] * 5
        self.times_first = [0] * players
        self.floor_paid = [0] * players
        self.best_tile = [0] * players
        self.done_rows = [0] * players
        self.done_colours = [0] * players
        self.done_columns = [0] * players

    # ---- dealing ---------------------------------------------------------------------------------------------------------
    def _draw(self):
        """One tile for a display, or None when nothing is left to draw and the short deal is allowed."""
        if not self.tracked:
            return self.rng.randrange(0, 5, 1)
        if sum(self.bag) == 0:
            self.bag, self.discard = self.discard, [0] * 5
        left = sum(self.bag)
        if left == 0:
            if self.ext & SHORT_DEAL:
                return None
            raise BoxEmpty()
        if self.pool == "Lid":
            colour = self.rng.choices([0, 1, 2, 3, 4], weights=[b / left for b in self.bag])[0]
        else:
            nth = self.rng.randrange(left)                # the nth tile of the bag, tiles sorted by colour
            colour = 0
            while nth >= self.bag[colour]:
                nth -= self.bag[colour]
                colour += 1
        self.bag[colour] -= 1
        return colour

    def new_round(self):
        self.to_move = self.starter
        self.times_first[self.starter - 1] += 1           # starter 0 (never chosen) counts for the last player, like a[-1]
        self.rounds += 1
        self.starter = 0
        self.middle = [0] * 5
        self.token_in_middle = True
        self.factories = [[0] * 5 for _ in range(self.D)]
        for f in self.factories:
            for _ in range(4):
                colour = self._draw()
                if colour is None:
                    return
                f[colour] += 1

    # ---- moves -----------------------------------------------------------------------------------------------------------
    def _me(self):
        return (self.to_move - 1) % self.P

    def _floor_add(self, n):
        me = self._me()
        self.floor[me] = min(7, self.floor[me] + n)

    def legal(self, source, colour, row):
        """source 0 = the middle, 1..D = a factory; row 0 = straight to the floor, 1..5 = a pattern line."""
        have = self.middle[colour] if source == 0 else self.factories[source - 1][colour]
        if have < 1:
            return False
        if row == 0:
            return True
        me = self._me()
        col_on_line, n = self.lines[me][row - 1]
        if n > 0 and col_on_line != colour:
            return False
        return not self.board[me][row - 1][(colour + row - 1) % 5]

    def play(self, source, colour, row):
        me = self._me()
        if source == 0:
            n = self.middle[colour]
            self.middle[colour] = 0
            if self.token_in_middle:
                self.token_in_middle = False
                self.starter = self.to_move
                self._floor_add(1)
        else:
            f = self.factories[source - 1]
            n = f[colour]
            f[colour] = 0
            for c in range(5):
                self.middle[c] += f[c]
                f[c] = 0
        if row == 0:
            spill = n
        else:
            _, have = self.lines[me][row - 1]
            room = row - have
            if n <= room:
                self.lines[me][row - 1] = (colour, have + n)
                spill = 0
            else:
                self.lines[me][row - 1] = (colour, row)
                spill = n - room
        if spill:
            self._floor_add(spill)
            if self.tracked:
                self.discard[colour] += spill         # discarded even when the floor line is full

    def round_over(self):
        return not self.token_in_middle and not any(self.middle) and not any(any(f) for f in self.factories)

    def somebody_finished_a_row(self):
        return any(all(r) for b in self.board for r in b)

    # ---- scoring ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _run(cells, at):
        lo = at
        while lo > 0 and cells[lo - 1]:
            lo -= 1
        hi = at
        while hi < 4 and cells[hi + 1]:
            hi += 1
        return hi - lo + 1

    def _tile(self, p, row, colour):
        b = self.board[p]
        col = (colour + row) % 5
        b[row][col] = True
        h = self._run(b[row], col)
        v = self._run([b[r][col] for r in range(5)], row)
        pts = 1 if (h == 1 and v == 1) else (h if h > 1 else 0) + (v if v > 1 else 0)
        self.best_tile[p] = max(self.best_tile[p], pts)
        extra = 0
        if all(b[row]):
            extra += 2
            self.done_rows[p] += 1
        if all(b[r][(colour + r) % 5] for r in range(5)):
            extra += 10
            self.done_colours[p] += 1
        if all(b[r][col] for r in range(5)):
            extra += 7
            self.done_columns[p] += 1
        return pts + (0 if self.ext & END_BONUS else extra)

    def score_round(self):
        for p in range(self.P):
            cost = FLOOR_COST[self.floor[p]]
            self.floor_paid[p] -= cost
            self.floor[p] = 0
            gained = 0
            for row in range(5):
                colour, n = self.lines[p][row]
                if n == row + 1:
                    self.lines[p][row] = (None, 0)
                    if self.tracked:
                        self.discard[colour] += row
                    gained += self._tile(p, row, colour)
            self.points[p] = max(0, self.points[p] - cost + gained)

    def final_bonus(self):
        for p in range(self.P):
            b = self.board[p]
            rows = sum(all(r) for r in b)
            cols = sum(all(b[r][c] for r in range(5)) for c in range(5))
            colours = sum(all(b[r][(k + r) % 5] for r in range(5)) for k in range(5))
            self.points[p] += 2 * rows + 7 * cols + 10 * colours

    def step(self, source, colour, row):
        assert not self.over and self.legal(source, colour, row)
        self.play(source, colour, row)
        if self.round_over():
            self.score_round()
            if self.somebody_finished_a_row():
                self.over = True
                if self.ext & END_BONUS:
                    self.final_bonus()
            else:
                self.new_round()
        else:
            self.to_move = self.to_move % self.P + 1

    # ---- the random agent and the flat loop ------------------------------------------------------------------------------
    def num_actions(self):
        return (self.D + 1) * 30

    def mask(self):
        S = self.D + 1
        return [self.legal(a % S, (a // S) % 5, a // (5 * S)) for a in range(self.num_actions())]

    def random_action(self, mask):
        floor_moves = self.num_actions() // 6
        w = [(0.01 if a < floor_moves else 1.0) * (1.0 if m else 0.0) for a, m in enumerate(mask)]
        try:
            return self.rng.choices(range(len(w)), weights=w)[0]
        except ValueError:
            raise Stuck()

    def snapshot(self):
        """Everything the oracle's oz_game holds, in its conventions (colour-indexed walls and lines)."""
        lines = [[[0] * 5 for _ in range(5)] for _ in range(self.P)]
        walls = [[[0] * 5 for _ in range(5)] for _ in range(self.P)]
        for p in range(self.P):
            for r in range(5):
                colour, n = self.lines[p][r]
                if n:
                    lines[p][r][colour] = n
                for c in range(5):
                    walls[p][r][c] = int(self.board[p][r][(c + r) % 5])
        return {"displays": [list(f) for f in self.factories], "center": list(self.middle) + [int(self.token_in_middle)],
                "pattern_lines": lines, "walls": walls, "floors": list(self.floor), "score": list(self.points),
                "current_player": self.to_move, "next_first_player": self.starter, "end_of_game": int(self.over),
                "turn_counter": self.rounds, "box": list(self.bag), "lid": list(self.discard),
                "first_player_stats": list(self.times_first), "floor_penalty": list(self.floor_paid), "max_combo": list(self.best_tile),
                "completed_lines": [[self.done_rows[p], self.done_colours[p], self.done_columns[p]] for p in range(self.P)]}


class ModelStream:
    """random.seed(seed); a game; new_round(); the random agent moves for every seat; a fresh game whenever one ends or nobody can move."""

    def __init__(self, seed, players, first_player, pool, ext):
        self.rng = random.Random(seed)
        self.args = (players, first_player, pool, ext)
        self.stuck = 0
        self.episodes = 0
        self._fresh()

    def _fresh(self):
        self.game = ModelGame(*self.args, self.rng)
        self.game.new_round()

    def advance(self):
        """One env move.  Returns (mask, action or -1, done in {0, 1, 2}, snapshot after the move and before a restart)."""
        g = self.game
        mask = g.mask()
        try:
            if g.over:
                raise Stuck()
            a = g.random_action(mask)
        except Stuck:
            self.stuck += 1
            snap = g.snapshot()
            self._fresh()
            return mask, -1, 2, snap
        S = g.D + 1
        g.step(a % S, (a // S) % 5, a // (5 * S))
        snap = g.snapshot()
        done = int(g.over)
        if done:
            self.episodes += 1
            self._fresh()
        return mask, a, done, snap
