import torch, time
torch.cuda.init()
for mb in [32, 64, 128, 192, 256, 384, 512, 1024, 4096]:
    x = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    reps = max(5, 4096 // mb)
    t = time.perf_counter()
    for _ in range(reps): y.copy_(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    # read-only pass: sum
    for _ in range(3): x.sum()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): x.sum()
    torch.cuda.synchronize()
    ds = (time.perf_counter() - t) / reps
    print(f"{mb:5d} MB  copy {2*mb/1024/dt:7.1f} GB/s (r+w)   sum {mb/1024/ds:7.1f} GB/s (read)")
