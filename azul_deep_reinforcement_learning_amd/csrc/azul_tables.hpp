// azul_tables.hpp -- host-side builders of the constant tables the kernels read.
#pragma once
#include "azul_core.hpp"

namespace az {

// RandomAgent (game_runner.py:87-97) hands random.choices the weights 0.01 (actions 0..29, pattern 0)
// or 1.0 (actions 30..179) times the legal mask.  CPython accumulates them left to right in fp64, and
// adding an exact 0.0 never changes the sum, so the cumulative weight after J legal floor moves and m
// legal pattern moves is one of 31*151 doubles: T[J][0] = T[J-1][0] + 0.01, T[J][m] = T[J][m-1] + 1.0.
// Built with the very IEEE additions CPython performs (this file is compiled without fast-math).
static inline void build_weight_table(double *T)
{
    double s = 0.0;
    for (int J = 0; J < T_ROWS; J++) {
        if (J > 0) s = s + 0.01;
        double c = s;
        T[J * T_COLS] = c;
        for (int m = 1; m < T_COLS; m++) { c = c + 1.0; T[J * T_COLS + m] = c; }
    }
}

} // namespace az
