#!/usr/bin/env python
"""Statistics of the dropout hash (wav2vec2/variables.py::dropout_hash = csrc/train.h, the oct hash of round 6) on the index spaces the
model uses it on: keep-rate error at p = 0.1, correlation of the keep decisions at lags along a row, between rows (same column), on the
diagonals and between the eight positions of an oct, uniformity of the 16-bit values (chi-square on the high and the low byte).
CPU only:  python tools/dropout_hash_stats.py > profiles/r06_dropout_hash_stats.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
from wav2vec2 import variables as V

P = 0.1
print(f"# dropout hash statistics, p = {P}: maxima over the seeds / streams listed; 'noise' = 1 / sqrt(decisions), the standard deviation of an empty correlation")
for T, rows in ((768, 8192), (1504, 4096), (3072, 2048)):
    worst = np.zeros(7)
    cases = [(1, 16), (0xDEADBEEF12345, 17), (7, 18), (12345, 60), (2 ** 63 + 5, 110), (99, 3)]
    for seed, stream in cases:
        n = rows * T
        h = V.dropout_hash(seed, stream, n)
        keep = V.dropout_keep(seed, stream, n, P).astype(np.float64)
        thr = int(np.float32(P) * 65536.0)
        k = keep - keep.mean()
        v = k.var()
        K = k.reshape(rows, T)
        lag = max(abs((k[:-l] * k[l:]).mean() / v) for l in (1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 32, 64, T // 4, T // 2))
        col = max(abs((K[:-l] * K[l:]).mean() / v) for l in (1, 2, 3, 4, 8, 16))
        dg = max(abs((K[:-1, :-1] * K[1:, 1:]).mean() / v), abs((K[:-1, 1:] * K[1:, :-1]).mean() / v))
        Q = k.reshape(-1, 8)
        ino = max(abs((Q[:, a] * Q[:, b]).mean() / v) for a in range(8) for b in range(a + 1, 8))
        chi_hi = (((np.bincount((h >> 8).astype(np.int64), minlength=256) - n / 256) ** 2) / (n / 256)).sum()
        chi_lo = (((np.bincount((h & 0xFF).astype(np.int64), minlength=256) - n / 256) ** 2) / (n / 256)).sum()
        worst = np.maximum(worst, [abs(keep.mean() - (1 - thr / 65536.0)), lag, col, dg, ino, chi_hi, chi_lo])
    print(f"row length {T:5d} x {rows} rows ({rows * T / 1e6:.1f} M decisions per case, noise {1 / np.sqrt(rows * T):.1e}): keep-rate error {worst[0]:.1e}; "
          f"correlation: lags along a row {worst[1]:.1e}, between rows {worst[2]:.1e}, diagonals {worst[3]:.1e}, inside an oct (28 pairs) {worst[4]:.1e}; "
          f"chi-square (255 dof) high byte {worst[5]:.0f}, low byte {worst[6]:.0f}")
# the attention index space: (rows, T) with the padded stride and the column permutation
for T in (768, 1499):
    a = V.attention_keep(5, 16, 4096, T, P).astype(np.float64)
    k = a - a.mean()
    v = k.var()
    lag = max(abs((k[:, :-l] * k[:, l:]).mean() / v) for l in (1, 2, 3, 4, 8, 16))
    col = max(abs((k[:-l] * k[l:]).mean() / v) for l in (1, 2, 3))
    print(f"attention_keep T = {T}: keep rate {a.mean():.5f}; correlation along keys {lag:.1e}, along queries {col:.1e} (noise {1 / np.sqrt(a.size):.1e})")
