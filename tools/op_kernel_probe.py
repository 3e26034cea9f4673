"""Which kernel variant does the launcher pick for a conv shape?  (GPU box)  python tools/op_kernel_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('FNN_KNOBS', '1')
from fast_nnunet_amd import capi

CASES = [
    # n, cin, cin2, cout, dims, k, stride
    (16, 16, 0, 16, (16, 64, 80), (1, 3, 3), (1, 1, 1)),
    (16, 16, 16, 16, (16, 64, 80), (1, 3, 3), (1, 1, 1)),
    (16, 48, 0, 16, (16, 64, 80), (1, 3, 3), (1, 1, 1)),
    (18, 16, 0, 16, (4, 96, 80), (1, 3, 3), (1, 1, 1)),
    (16, 16, 0, 16, (16, 64, 80), (1, 1, 1), (1, 1, 1)),
    (18, 16, 0, 16, (4, 96, 80), (1, 1, 1), (1, 1, 1)),
    (16, 16, 0, 16, (16, 64, 80), (3, 3, 1), (1, 1, 1)),
    (16, 16, 16, 16, (16, 64, 80), (3, 3, 1), (1, 1, 1)),
    (13, 176, 0, 16, (16, 64, 80), (1, 3, 3), (1, 1, 1)),
    (13, 176, 0, 16, (16, 64, 80), (1, 1, 1), (1, 1, 1)),
    (1, 16, 0, 16, (256, 192, 192), (3, 3, 3), (1, 1, 1)),
    (2, 16, 16, 16, (8, 16, 128), (1, 3, 3), (1, 1, 1)),
    (2, 16, 16, 16, (8, 16, 160), (1, 3, 3), (1, 1, 1)),
    (4, 32, 0, 32, (64, 64, 64), (3, 3, 3), (1, 1, 1)),
    (4, 16, 0, 32, (64, 64, 64), (3, 3, 3), (1, 1, 1)),
    (2, 32, 0, 16, (8, 64, 64), (3, 3, 3), (1, 1, 1)),
    (32, 32, 0, 32, (4, 32, 32), (3, 3, 3), (1, 1, 1)),
    (1, 16, 0, 48, (16, 32, 32), (1, 3, 3), (1, 1, 1)),
    (1, 16, 0, 16, (12, 20, 20), (1, 3, 3), (1, 1, 1)),
]
TCASES = [(1, 32, 16, (4, 8, 8), (2, 1, 1)), (1, 64, 32, (4, 8, 8), (1, 1, 2)), (1, 32, 16, (4, 8, 8), (1, 2, 1)), (1, 64, 64, (4, 8, 8), (2, 1, 1))]
rng = np.random.default_rng(0)
for n, cin, cin2, cout, dims, k, stride in CASES:
    x = rng.standard_normal((n, cin, *dims), dtype=np.float32)
    x2 = rng.standard_normal((n, cin2, *dims), dtype=np.float32) if cin2 else None
    w = rng.standard_normal((cout, cin + cin2, *k), dtype=np.float32) * 0.05
    capi.op_conv3d(x, w, None, k, stride, x2=x2)
    print((n, cin, cin2, cout, dims, k, stride), '->', capi.op_last_kernels(), flush=True)
for n, cin, cout, dims, stride in TCASES:
    x = rng.standard_normal((n, cin, *dims), dtype=np.float32)
    w = rng.standard_normal((cin, cout, *stride), dtype=np.float32) * 0.05
    capi.op_conv_transpose3d(x, w, None, stride)
    print('tconv', (n, cin, cout, dims, stride), '->', capi.op_last_kernels(), flush=True)
