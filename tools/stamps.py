"""Diagnostic: run one conv layer shape through fnn_op_conv3d of a -DFNN_STAMPS build (prints s_memtime segment means).
usage: [STRIDE=2 | STRIDE=1,2,2] python tools/stamps.py N CIN COUT D H W [kd kh kw] [cin2]"""
import os
os.environ.setdefault('FNN_KNOBS', '1')
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_nnunet_amd import capi
a = [int(v) for v in sys.argv[1:]]
n, cin, cout, d, h, w = a[:6]
k = tuple(a[6:9]) if len(a) >= 9 else (3, 3, 3)
cin2 = a[9] if len(a) >= 10 else 0
rng = np.random.default_rng(0)
x = rng.standard_normal((n, cin, d, h, w), dtype=np.float32)
wt = rng.standard_normal((cout, cin + cin2, *k), dtype=np.float32) * 0.05
kw = {}
if cin2:
    kw = dict(x2=rng.standard_normal((n, cin2, d, h, w), dtype=np.float32), gamma2=np.ones(cin2, np.float32),
              beta2=np.zeros(cin2, np.float32), slope2=0.01)
y = capi.op_conv3d(x, wt, np.zeros(cout, np.float32), k, (tuple(int(v) for v in os.environ['STRIDE'].split(',')) * 3)[:3] if 'STRIDE' in os.environ else (1, 1, 1), gamma=np.ones(cin, np.float32),
                   beta=np.zeros(cin, np.float32), slope=0.01, want_stats=True, **kw)
print('ok', y[0].shape)
