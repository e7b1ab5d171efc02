"""Which library routine solves the order-1510 symmetric eigen-problem of the PCA pre-reduction fastest?  torch.linalg.eigh
(rocSOLVER syevd: 36 ms, launch-bound tridiagonalisation) against rocsolver_dsyevdj / dsyevj called through ctypes."""
import ctypes as C, time, sys, os
import numpy as np, torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1510
rs = C.CDLL("librocsolver.so")
rb = C.CDLL("librocblas.so")
h = C.c_void_p()
assert rb.rocblas_create_handle(C.byref(h)) == 0
stream = torch.cuda.current_stream().cuda_stream
rb.rocblas_set_stream(h, C.c_void_p(stream))
EVECT_ORIGINAL, FILL_LOWER, ESORT_ASC = 211, 122, 252      # rocblas_evect_original, rocblas_fill_lower, rocblas_esort_ascending
g = torch.Generator(device="cuda"); g.manual_seed(0)
B = torch.randn((20000, n), generator=g, device="cuda", dtype=torch.float64) * torch.logspace(0, -2, n, device="cuda", dtype=torch.float64)
M = (B.T @ B).contiguous()
ref_w = torch.linalg.eigvalsh(M)

def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out

ms, (w, V) = t(lambda: torch.linalg.eigh(M))
print(f"n {n}: torch.linalg.eigh {ms:.1f} ms")
info = torch.zeros(1, dtype=torch.int32, device="cuda")
D = torch.empty(n, dtype=torch.float64, device="cuda")
def dj():
    A = M.clone()
    rc = rs.rocsolver_dsyevdj(h, EVECT_ORIGINAL, FILL_LOWER, n, C.c_void_p(A.data_ptr()), n, C.c_void_p(D.data_ptr()), C.c_void_p(info.data_ptr()))
    assert rc == 0, rc
    return A
ms, A = t(dj)
print(f"n {n}: rocsolver_dsyevdj {ms:.1f} ms   max |w - w_ref| / w_max {float((D - ref_w).abs().max() / ref_w.max()):.1e}  info {int(info)}")
res = torch.zeros(1, dtype=torch.float64, device="cuda"); nsw = torch.zeros(1, dtype=torch.int32, device="cuda")
def jj():
    A = M.clone()
    rc = rs.rocsolver_dsyevj(h, ESORT_ASC, EVECT_ORIGINAL, FILL_LOWER, n, C.c_void_p(A.data_ptr()), n, C.c_double(0.0), C.c_void_p(res.data_ptr()), 100,
                             C.c_void_p(nsw.data_ptr()), C.c_void_p(D.data_ptr()), C.c_void_p(info.data_ptr()))
    assert rc == 0, rc
    return A
ms, A = t(jj, 1)
print(f"n {n}: rocsolver_dsyevj {ms:.1f} ms   sweeps {int(nsw)}  max |w - w_ref| / w_max {float((D - ref_w).abs().max() / ref_w.max()):.1e}  info {int(info)}")
