"""Diagnostic: wall time of the phases of one sharded step at world size 1 (RCCL group of one)."""
import os, sys, time
sys.path.insert(0, '.')
import torch
import torch.distributed as dist
import bench
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dev = torch.device('cuda', 0)
predictor, sd, info = bench.build_predictor('bone_turbo_r2', dev, 32, 'fp16', 'f16', False)
vol = bench.synthetic_volume(512, dev)
from fast_nnunet_amd.dist import ShardedPredictor
r = ShardedPredictor(predictor, dist.group.WORLD)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('single-GPU logits     %.1f ms' % t(lambda: predictor.predict_sliding_window_return_logits(vol)))
print('single-GPU labels     %.1f ms' % t(lambda: predictor.predict_segmentation_from_preprocessed_data(vol) if hasattr(predictor, 'predict_segmentation_from_preprocessed_data') else None))
print('sharded labels        %.1f ms' % t(lambda: r.predict_segmentation_from_preprocessed_data(vol)))
print('sharded labels nogath %.1f ms' % t(lambda: r.predict_segmentation_from_preprocessed_data(vol, gather=False)))
x = vol.to(device=dev, dtype=torch.float32).contiguous()
dec, origins, owns = r._plan(x)
opts = predictor._opts()
print('plan                  %.1f ms' % t(lambda: r._plan(x)))
print('features fold         %.1f ms' % t(lambda: r._features_fold(x, dec, origins, opts, 0)))
got = r._features_fold(x, dec, origins, opts, 0)
own = owns[0]
lab = torch.empty(x.shape[1:], dtype=torch.uint8, device=dev)
eng = predictor._engine
print('gather_box labels     %.1f ms' % t(lambda: eng.gather_box(got[0].data_ptr(), got[1].data_ptr(), got[2], x.shape, opts, own[0], own[1], labels_ptr=lab.data_ptr(), fold=0)))
out = torch.empty((61, *x.shape[1:]), dtype=torch.half, device=dev)
print('gather_box logits     %.1f ms' % t(lambda: eng.gather_box(got[0].data_ptr(), got[1].data_ptr(), got[2], x.shape, opts, own[0], own[1], logits_ptr=out.data_ptr(), fold=0)))
dist.destroy_process_group()
