import numpy as np, sys
Cout, P, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[5]) if len(sys.argv) > 5 else 2
n = N * P ** 3 * Cout
a, b = [np.fromfile(f, dtype=np.uint8) for f in sys.argv[3:5]]
oa, ob = a[:2 * n].view(np.float16).reshape(N, P, P, P, Cout).astype(np.float32), b[:2 * n].view(np.float16).reshape(N, P, P, P, Cout).astype(np.float32)
sa, sb = a[2 * n:].view(np.float64), b[2 * n:].view(np.float64)
d = np.abs(oa - ob)
print('out: max diff', d.max(), 'mismatching', int((d > 0).sum()), 'of', d.size, ' nan', int(np.isnan(oa).sum()), int(np.isnan(ob).sum()))
if d.max() > 0:
    idx = np.argwhere(d > 0)
    print('first mismatches (n, d, h, w, c):', idx[:12].tolist())
    print('per-item mismatch counts', (d > 0).sum(axis=(1, 2, 3, 4)).tolist())
    print('per-d mismatch counts', (d > 0).sum(axis=(0, 2, 3, 4)).tolist())
    print('per-channel mismatch counts', (d > 0).sum(axis=(0, 1, 2, 3)).tolist())
    print('per-w mismatch counts', (d > 0).sum(axis=(0, 1, 2, 4)).tolist()[:16])
    print('per-h mismatch counts', (d > 0).sum(axis=(0, 1, 3, 4)).tolist()[:16])
print('stats: max rel diff', float(np.max(np.abs(sa - sb) / (np.abs(sb) + 1e-6))), 'n', sa.size)
