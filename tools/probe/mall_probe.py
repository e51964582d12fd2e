"""Does the 256 MB Infinity Cache keep a freshly written tensor for the kernel that reads it next?  In-place add over buffers of
growing size (footprint = the buffer): effective bandwidth = 2 x bytes / time.  Probe only (torch kernels), not product code."""
import torch
dev = torch.device('cuda:0')
for mb in (16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 768, 1024, 2048):
    a = torch.zeros(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    for _ in range(5):
        a.add_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    e0.record()
    for _ in range(reps):
        a.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('%5d MB  %8.1f us  %6.2f TB/s' % (mb, ms * 1e3, 2 * mb * 1024 * 1024 / ms / 1e9))
    # producer -> consumer through a second tensor: b = a * 2 then c = b + 1 (b written, then read)
    b = torch.empty_like(a); c = torch.empty_like(a)
    for _ in range(3):
        torch.mul(a, 2.0, out=b); torch.add(b, 1.0, out=c)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        torch.mul(a, 2.0, out=b); torch.add(b, 1.0, out=c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('        chain a->b->c: %8.1f us per pair, %6.2f TB/s (4 x bytes)' % (ms * 1e3, 4 * mb * 1024 * 1024 / ms / 1e9))
    del a, b, c
