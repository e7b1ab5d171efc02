import torch, time
for n in (1536, 5000, 10000):
    A = torch.randn(n, n + 500, device="cuda", dtype=torch.float64); G = A @ A.T
    for dt in (torch.float64, torch.float32):
        H = G.to(dt); torch.linalg.eigh(H); torch.cuda.synchronize()
        t = time.perf_counter(); w, V = torch.linalg.eigh(H); torch.cuda.synchronize()
        print(n, dt, f"{(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
    t = time.perf_counter(); w = torch.linalg.eigvalsh(G); torch.cuda.synchronize(); print(n, "eigvalsh f64", f"{(time.perf_counter()-t)*1e3:.1f} ms")
    t = time.perf_counter(); L = torch.linalg.cholesky(G); torch.cuda.synchronize(); print(n, "chol f64", f"{(time.perf_counter()-t)*1e3:.1f} ms")
    t = time.perf_counter(); Q, R = torch.linalg.qr(G); torch.cuda.synchronize(); print(n, "qr f64", f"{(time.perf_counter()-t)*1e3:.1f} ms")
    t = time.perf_counter(); C = G @ G; torch.cuda.synchronize(); print(n, "gemm f64", f"{(time.perf_counter()-t)*1e3:.1f} ms")
