"""Tile configuration / split-K sweep of the weight-gradient GEMMs of a PNA layer (dW = dY^T X, K = rows), slab path.
    python tools/wgrad_sweep.py"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemm_bench as gb  # noqa: E402

if __name__ == '__main__':
    lib = gb.L.load()
    N, E, F = 8511, 16907, 200
    for name, M, Nn, K in (('W2   [F,F]  K=E', F, F, E), ('Wsd  [2F,F] K=N', 2 * F, F, N), ('W_h  [F,F]  K=N', F, F, N),
                           ('dQ   [64,F] K=E', 64, F, E)):
        us, tf, _ = gb.run(lib, 1, 0, M, Nn, K, -1, 0)
        print(f'{name}: auto {us:6.1f} us {tf:5.1f} TF', flush=True)
        best = []
        for cfg in (0, 2, 3, 4, 5, 7, 8):
            for splits in (4, 8, 12, 16, 24, 32, 48, 64):
                try:
                    us, tf, err = gb.run(lib, 1, 0, M, Nn, K, cfg, splits, reps=10)
                except AssertionError:
                    continue
                best.append((us, cfg, splits, err))
        best.sort()
        print('     best: ' + '  '.join(f'cfg{c}/s{s} {u:5.1f}us' for u, c, s, e in best[:8]), flush=True)
