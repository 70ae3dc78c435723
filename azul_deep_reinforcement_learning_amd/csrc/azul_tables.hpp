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
// rows = 5 (D + 1) + 1, at most 5 (rows - 1) pattern moves), as the {Fr[J][b], S[J]} pairs the kernels read with one 16-byte load:
// out[2 (8 J + b)] = Fr[J][b], out[2 (8 J + b) + 1] = S[J].  Returns false if the identity ever failed.
static inline bool build_sample_pairs(int rows, double *out /* [rows * 8 * 2] */)
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
        for (int b = 0; b < 8; b++) { out[2 * (8 * J + b)] = fr[b]; out[2 * (8 * J + b) + 1] = s; }
    }
    return ok;
}

} // namespace az
