import torch, time
x = torch.empty(256, 3, 320, 320, dtype=torch.float32).pin_memory()
u8 = torch.empty(256, 320, 320, 3, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device='cuda'); d8 = torch.empty_like(u8, device='cuda')
for src, dst, name in ((x, d, 'fp32 batch 315 MB'), (u8, d8, 'uint8 batch 79 MB')):
    for _ in range(3): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f'{name}: {dt*1e3:.2f} ms  {src.numel()*src.element_size()/dt/1e9:.1f} GB/s')
