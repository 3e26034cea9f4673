"""Diagnostic: mean launch time of one conv layer shape (FNN_OP_TIME: 10 launches after 2 warm-ups, HIP events).
usage: [FNN_LIB=.../libfnn_exp.so] [LT_STRIDE=2] python tools/layer_time.py N CIN COUT D H W [cin2]   (3x3x3, stride 1 (LT_STRIDE: 2,2,2), fused-norm inputs; D H W = the INPUT size)"""
import os
os.environ.setdefault('FNN_KNOBS', '1')
os.environ['FNN_OP_TIME'] = '1'
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast_nnunet_amd import capi
a = [int(v) for v in sys.argv[1:]]
n, cin, cout, d, h, w = a[:6]
cin2 = a[6] if len(a) >= 7 else 0
rng = np.random.default_rng(0)
x = rng.standard_normal((1, cin, d, h, w), dtype=np.float32).repeat(n, 0)
wt = rng.standard_normal((cout, cin + cin2, 3, 3, 3), dtype=np.float32) * 0.05
kw = {}
if cin2:
    kw = dict(x2=rng.standard_normal((1, cin2, d, h, w), dtype=np.float32).repeat(n, 0), gamma2=np.ones(cin2, np.float32),
              beta2=np.zeros(cin2, np.float32), slope2=0.01)
sys.stderr.write(f'{os.path.basename(os.environ.get("FNN_LIB", "libfnn_hip.so"))}: ')
sys.stderr.flush()
st = int(os.environ.get('LT_STRIDE', '1'))
capi.op_conv3d(x, wt, np.zeros(cout, np.float32), (3, 3, 3), (st, st, st), gamma=np.ones(cin, np.float32),
               beta=np.zeros(cin, np.float32), slope=0.01, want_stats=True, **kw)
