// azul_tables.hpp -- host-side builders of the constant tables the kernels read.
#pragma once
#include "azul_common.hpp"

namespace az {

// RandomAgent (game_runner.py:87-97) hands random.choices the weights 0.01 (actions 0..29, pattern 0)
// or 1.0 (actions 30..179) times the legal mask.  CPython accumulates them left to right in fp64, and
// adding an exact 0.0 never changes the sum, so the cumulative weight after J legal floor moves and m
// legal pattern moves is T[J][m] with T[J][0] = T[J-1][0] + 0.01 and T[J][m] = T[J][m-1] + 1.0.
// build_weight_table() performs exactly those IEEE additions (this file is compiled without fast-math).
static inline void build_weight_table(double *T /* [31][151] */)
{
    double s = 0.0;
    for (int J = 0; J < 31; J++) {
        if (J > 0) s = s + 0.01;
        double c = s;
        T[J * 151] = c;
        for (int m = 1; m < 151; m++) { c = c + 1.0; T[J * 151 + m] = c; }
    }
}

// Compact form used by the kernels: Fr[J][b] = T[J][2^b] - 2^b (exact), S[J] = T[J][0]; then T[J][m] == m + Fr[J][floor(log2 m)] for every m.
// For an action space with `rows - 1` floor actions (the reference's game: rows = 31; the P-player rules on D displays, azul_rules_x.hpp:
// rows = 5 (D + 1) + 1, at most 5 (rows - 1) pattern moves), as pairs the kernels read with one 16-byte load, T_STRIDE = 9 per row:
//   pair 9 J + b, b < 8:   {Fr[J][b], S[J]}
//   pair 9 J + 8:          {fl(100 S[J]), 0.0}   -- the FLOOR-ONLY pair, read (as "binade 8": M = 0 counts as 256) when every legal action is
//                          a 0.01-weight floor move -- the late moves of many rounds, and EVERY move of a game whose pattern lines are locked
//                          for good (hazard H9).  CPython draws x = random() * S[J] and returns the smallest k with S[k] > x; S[k] is k additions
//                          of 0.01 (|100 S[k] - k| < 1e-13 for k <= 60), so with x100 = random() * fl(100 S[J]) -- within 1e-13 of 100 x -- the
//                          ordinal is floor(x100) + 1 wherever x100 is further than 1e-9 from an integer: the one-compare form of the pattern
//                          moves with (total, S) = (fl(100 S[J]), 0) and the ordinal counted from 0 instead of J.
// Returns false if the identity ever failed.
static inline bool build_sample_pairs(int rows, double *out /* [rows * 9 * 2] */)
{
    const int mmax = 5 * (rows - 1);
    bool ok = mmax < 256;
    double s = 0.0;
    for (int J = 0; J < rows; J++) {
        if (J > 0) s = s + 0.01;
        double fr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double c = s;
        for (int m = 1; m <= mmax; m++) {
            c = c + 1.0;
            const int b = 31 - __builtin_clz((unsigned)m);
            if (m == (1 << b)) fr[b] = c - (double)m;
            if ((double)m + fr[b] != c) ok = false;
        }
        for (int b = 0; b < 8; b++) { out[2 * (9 * J + b)] = fr[b]; out[2 * (9 * J + b) + 1] = s; }
        out[2 * (9 * J + 8)] = 100.0 * s;
        out[2 * (9 * J + 8) + 1] = 0.0;
    }
    return ok;
}

} // namespace az
