"""Diagnostic: one ZR conv layer under the timing-only switches of a -DFNN_STAMPS -DFNN_TMODE build
(`make EXP="-DFNN_STAMPS -DFNN_TMODE" libfnn_exp.so`, FNN_LIB=.../libfnn_exp.so).  Results are wrong by design.
usage: python tools/zr_tmode.py N CIN COUT D H W [cin2] -- prints launch time + stamp segments per mode"""
import os
os.environ.setdefault('FNN_KNOBS', '1')
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_nnunet_amd import capi
a = [int(v) for v in sys.argv[1:]]
n, cin, cout, d, h, w = a[:6]
cin2 = a[6] if len(a) >= 7 else 0
rng = np.random.default_rng(0)
x = rng.standard_normal((n, cin, d, h, w), dtype=np.float32)
wt = rng.standard_normal((cout, cin + cin2, 3, 3, 3), dtype=np.float32) * 0.05
kw = {}
if cin2:
    kw = dict(x2=rng.standard_normal((n, cin2, d, h, w), dtype=np.float32), gamma2=np.ones(cin2, np.float32),
              beta2=np.zeros(cin2, np.float32), slope2=0.01)
for mode in (0, 8, 4, 12, 0, 8):
    os.environ['FNN_ZR_TMODE'] = str(mode)
    sys.stderr.write(f'mode {mode:2d}: ')
    sys.stderr.flush()
    capi.op_conv3d(x, wt, np.zeros(cout, np.float32), (3, 3, 3), (1, 1, 1), gamma=np.ones(cin, np.float32),
                   beta=np.zeros(cin, np.float32), slope=0.01, want_stats=True, **kw)
