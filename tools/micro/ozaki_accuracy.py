#!/usr/bin/env python3
"""VERDICT r03 item 9, the accuracy half of the gate: how many 7-bit slices per operand does an
integer-matrix-core contraction of BASELINE configs[4] (G = 200, R = 760, float32-exact table)
need to stay within 1e-10 of the float64 result?

Emulates the scheme on the CPU for a few draws: the table T[r][p] (float32 values) and the pair
weights w[p][d] = prefactor n_i n_j are cut per block of 64 pairs (the K extent of one
v_mfma_i32_16x16x64_i8) into S signed 7-bit slices against the block's largest magnitude
(per row r and block for T, per draw and block for w); slice products with i + j <= D - 1 are
accumulated exactly (int64 here, int32 on the device) and scaled back per block.  Reports the
error relative to the largest |xi| of a draw -- the floor of tests/util.py: assert_rel -- and
elementwise."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tabcorr_amd import synthetic          # noqa: E402
from oracle import tabcorr_oracle as oracle  # noqa: E402

BLOCK = 64
BITS = 7


def slices(values, n_slices):
    """values (..., n_blocks, BLOCK) -> integer slices (n_slices, ...) and the blocks' scales:
    values ~ scale * sum_s slice_s * 2^(-BITS (s + 1))."""
    top = np.max(np.abs(values), axis=-1, keepdims=True)
    exponent = np.where(top > 0, np.ceil(np.log2(np.where(top > 0, top, 1.0))) + 1, 0.0)
    scale = np.exp2(exponent)
    rest = values / scale                       # |rest| <= 1/2
    out = []
    for s in range(n_slices):
        rest = rest * 2.0 ** BITS
        piece = np.rint(rest)                   # |piece| <= 64: fits int8
        rest = rest - piece
        out.append(piece.astype(np.int64))
    return np.array(out), scale[..., 0]


def main():
    table = synthetic.synthetic_table(100, 1, (19, 40), 'auto', seed=9)
    matrix = table['tpcf_matrix'].astype(np.float32).astype(np.float64)    # float32-exact
    n_rows = 64                                   # (a sample of the 760 rows)
    rows = np.linspace(0, matrix.shape[0] - 1, n_rows).astype(int)
    theta = synthetic.zheng07_draws(6, seed=3)
    index_1, index_2, prefactor = oracle.pair_indices(len(table['gal_type']))
    n_pairs = len(index_1)
    pad = (-n_pairs) % BLOCK
    t = np.pad(matrix[rows], ((0, 0), (0, pad))).reshape(n_rows, -1, BLOCK)
    w = []
    for th in theta:
        model = oracle.Zheng07(th, False, None)
        ngal = oracle.mean_occupation(table, model, 10) * table['gal_type']['n_h']
        w.append(prefactor * ngal[index_1] * ngal[index_2])
    w = np.pad(np.array(w), ((0, 0), (0, pad))).reshape(len(theta), -1, BLOCK)
    exact = np.einsum('rbk,dbk->dr', t, w)
    print('table %s, %d pairs in %d blocks of %d; %d rows x %d draws sampled'
          % (matrix.shape, n_pairs, t.shape[1], BLOCK, n_rows, len(theta)))
    for n_slices in (4, 5, 6, 7):
        ts, t_scale = slices(t, n_slices)         # (S, r, b, k), (r, b)
        ws, w_scale = slices(w, n_slices)         # (S, d, b, k), (d, b)
        for diagonals in (n_slices, n_slices + 1):
            total = np.zeros_like(exact)
            products = 0
            for i in range(n_slices):
                for j in range(n_slices):
                    if i + j > diagonals - 1:
                        continue
                    products += 1
                    block = np.einsum('rbk,dbk->drb', ts[i], ws[j]).astype(np.float64)
                    total += np.einsum('drb,rb,db->dr', block, t_scale, w_scale) * \
                        2.0 ** (-BITS * (i + j + 2))
            error = np.abs(total - exact)
            print('%d slices per operand, slice products with i + j < %d (%2d products, %d '
                  'conversions per block): error / largest |xi| of the draw %.1e, largest '
                  'elementwise relative error %.1e'
                  % (n_slices, diagonals, products, diagonals,
                     np.max(error / np.max(np.abs(exact), axis=1, keepdims=True)),
                     np.max(error / np.abs(exact))))


if __name__ == '__main__':
    main()
