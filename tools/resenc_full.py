import sys, time
sys.path[:0] = ['.', './tests', './tests/golden']
import torch
from oracle.topology import UNetSpec
from oracle.unet import synthetic_state_dict
from tests.test_gpu_predictor import _predictor
spec = UNetSpec('resenc', 1, 3, [16, 32, 64, 128, 160, 160], [(3, 3, 3)] * 6, [(1, 1, 1)] + [(2, 2, 2)] * 5, [1, 3, 4, 6, 6, 6], [1] * 5)
patch = (160, 160, 160)
sd = synthetic_state_dict(spec, 1)
p = _predictor(spec, patch, [sd], batch=2)
x = torch.randn(1, 200, 200, 200)
torch.cuda.synchronize(); t = time.time()
out = p.predict_sliding_window_return_logits(x)
torch.cuda.synchronize(); print('resenc 160^3 r=2: 8 patches', time.time() - t, 's', out.shape, float(out.float().abs().max()))
t = time.time(); out = p.predict_sliding_window_return_logits(x); torch.cuda.synchronize(); print('second', time.time() - t)
