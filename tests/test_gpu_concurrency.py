"""Two engine contexts on two streams of one GPU, driven from two threads.  On this platform FFT-type kernels
return occasional wrong 64-byte pieces while the split-fp16 in-place streaming kernels of another stream share the chip (about one
Hilbert call in 80 before round 6; rocFFT itself: 72 % of calls, tools/thread_probe5.py).  The engine fences its transform entries
against the passes of every other context of the process (FenceGuard, csrc/eofx_abi.hip): this test is the former reproducer and
must see no difference."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_hilbert_beside_another_contexts_in_place_passes_is_bitwise_stable(ctx):
    import torch

    from xeofs_amd import engine

    rng = np.random.default_rng(0)
    fields = [(rng.standard_normal((n, 5)) @ rng.standard_normal((5, p)) + 0.2 * rng.standard_normal((n, p)) + 1.0).astype(np.float32)
              for n, p in ((400, 4096), (700, 2048), (300, 8192), (1000, 1000))]

    def transform(c, X):
        A, _ = engine.preprocess(c, X, True, False, None, in_place=True, for_hilbert=True)
        B, _ = engine.hilbert(c, A, "exp", 0.2)
        im = B.download()
        A.free(); B.free()
        return im

    serial = [transform(ctx, X) for X in fields]
    bad, errors = [], []
    stop = threading.Event()

    def victim(tid):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                c = engine.Context(0)
                for rep in range(150):
                    i = (rep + tid) % len(fields)
                    im = transform(c, fields[i])
                    if not np.array_equal(im, serial[i]):
                        bad.append((tid, rep, i, int((im != serial[i]).sum())))
                c.close()
        except BaseException as e:      # noqa: BLE001  (reported in the main thread)
            errors.append(e)

    def load(tid):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                c = engine.Context(0)
                X = fields[1]
                keep = engine.preprocess(c, X, True, False, None, in_place=True)[0]
                Zn = torch.randn(keep.n_pad, 64, device="cuda")
                Zn[keep.n:] = 0
                while not stop.is_set():
                    for _ in range(10):
                        Y = engine.panel_tmul(c, keep, Zn, prec="f16x3")
                        engine.panel_mul(c, keep, Y, prec="f16x3")
                c.synchronize()
                keep.free()
                c.close()
        except BaseException as e:      # noqa: BLE001
            errors.append(e)

    victims = [threading.Thread(target=victim, args=(t,)) for t in range(2)]
    loads = [threading.Thread(target=load, args=(t,)) for t in range(2)]
    for t in loads + victims:
        t.start()
    for t in victims:
        t.join()
    stop.set()
    for t in loads:
        t.join()
    assert not errors, errors
    assert not bad, f"{len(bad)} of 300 transforms differ from the serial result: {bad[:5]}"
