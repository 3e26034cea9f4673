import torch, time
x = torch.empty(2 * 1024**3, dtype=torch.float16, device='cuda')   # 4 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
gb = x.numel() * 2 / 1e9
ms = t(lambda: x.fill_(1.0)); print('fill  write %.2f TB/s' % (gb / ms))
ms = t(lambda: x.zero_()); print('zero  write %.2f TB/s' % (gb / ms))
ms = t(lambda: y.copy_(x)); print('copy  r+w %.2f TB/s (each way %.2f)' % (2 * gb / ms, gb / ms))
ms = t(lambda: x.sum()); print('sum   read %.2f TB/s' % (gb / ms))
ms = t(lambda: torch.add(x, y, out=y)); print('add 2r+1w %.2f TB/s' % (3 * gb / ms))
