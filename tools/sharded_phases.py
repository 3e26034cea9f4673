"""Diagnostic: wall time of the phases of one sharded step at world size 1 (RCCL group of one) - the per-phase split
bench.py --gpus N prints for N > 1, next to the single-GPU entry points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dev = torch.device('cuda', 0)
mirror = '--mirror' in sys.argv
predictor, sd, info = bench.build_predictor('bone_turbo_r2', dev, 32, 'fp16', 'f16', mirror)
vol = bench.synthetic_volume(512, dev)
from fast_nnunet_amd.dist import ShardedPredictor
r = ShardedPredictor(predictor, dist.group.WORLD)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('single-GPU logits     %.1f ms' % t(lambda: predictor.predict_sliding_window_return_logits(vol)))
print('single-GPU labels     %.1f ms' % t(lambda: predictor.predict_segmentation_from_preprocessed_data(vol)))
print('sharded labels        %.1f ms' % t(lambda: r.predict_segmentation_from_preprocessed_data(vol)))
print('sharded labels nogath %.1f ms' % t(lambda: r.predict_segmentation_from_preprocessed_data(vol, gather=False)))
ph = r.start_phases()
r.predict_segmentation_from_preprocessed_data(vol)
print('phases of one profiled step (device synchronised at every boundary):')
for k, v in ph.items():
    print(f'  {k:28s} {v:.1f}' if isinstance(v, float) else f'  {k:28s} {v}')
dist.destroy_process_group()
