"""Varimax / Promax rotation of resident loadings (xeofs/linalg/_numpy/_rotation.py:6-187).

The p x m loadings stay on the GPU as a 64-wide panel.  Each iteration of the reference loop
(`basis = X R; transformed = X^H (basis * (|basis|^2 - alpha W)); U, s, VT = svd(transformed);
R = U VT`) is one fused pass over the panel (`eofx_panel_rot_step_f64`, float64 row arithmetic) plus an
m x m SVD on the host; `W = diag(R^T (X^T X) R)` comes from the m x m Gram matrix computed once.
Convergence test, error and outputs are the reference's.

More than 64 modes (up to 256; rare) keep the same driver on 128 / 256-wide panels: the same entry runs the
column-blocked variant of the step kernel (`rot_step_wide_kernel`), still one pass over the panel per iteration.
"""

from __future__ import annotations

import numpy as np

from . import engine

MAX_ROT_MODES = 256
FUSED_ROT_MODES = 64      # widest panel of the single-workgroup step kernel (wider ones: its column-blocked variant)


def _dev(a, like):
    torch = engine._torch()
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=like.device)


def _pad(M, L):
    out = np.zeros((L, L))
    m = M.shape[0]
    out[:m, :m] = M
    return out


def _rot_width(m):
    """panel width for m modes: the fused kernel's 32 / 64, else the next power of two (row normalisation needs one)"""
    return engine.panel_width(m) if m <= FUSED_ROT_MODES else (128 if m <= 128 else 256)


def _rot_step(ctx, X, R, aux, mode, power=1.0):
    return engine.panel_rot_step(ctx, X, R, aux, mode, power)


def promax(ctx, loadings: np.ndarray, power: int = 1, max_iter: int = 1000, rtol: float = 1e-8):
    """-> (rotated loadings [p, m] float32, rotation matrix [m, m], phi [m, m]) like `_promax`."""
    Xrot, p, m, rot_mat, phi = promax_panel(ctx, loadings, power=power, max_iter=max_iter, rtol=rtol)
    return engine.panel_export(ctx, Xrot, p, m), rot_mat, phi


def promax_panel(ctx, loadings: np.ndarray, power: int = 1, max_iter: int = 1000, rtol: float = 1e-8, col_scale=None):
    """Like `promax`, but the rotated loadings stay in HBM: -> (panel [rows_pad, L] on the device, p, m, rotation
    matrix, phi).  `col_scale` (m values) multiplies the columns of `loadings` on the device first (components ->
    loadings = components * sqrt(explained variance) without a host pass)."""
    loadings = np.ascontiguousarray(loadings, dtype=np.float32)
    p, m = loadings.shape
    if m < 2:
        raise ValueError("Cannot rotate {:} modes (columns), but must be 2 or more.".format(m))
    if m > MAX_ROT_MODES:
        raise NotImplementedError(f"rotation of more than {MAX_ROT_MODES} modes is not supported by this build")
    L = _rot_width(m)
    rows_pad = (p + 511) // 512 * 512
    Lp = engine.panel_import(ctx, loadings, rows_pad, L)           # loadings
    if col_scale is not None:
        D = np.zeros((L, L))
        D[np.arange(m), np.arange(m)] = np.asarray(col_scale, dtype=np.float64)
        Lp = engine.panel_matmul(ctx, Lp, _dev(D, Lp))
    Xn = engine.panel_row_normalize(ctx, Lp)                       # Kaiser-normalised rows
    S = engine.panel_gram(ctx, Xn).cpu().numpy()[:m, :m]           # X^T X
    S = 0.5 * (S + S.T)
    R = np.eye(m)
    alpha = 1.0 / p
    delta = 0.0
    for _ in range(int(max_iter)):
        delta_old = delta
        W = np.einsum("ij,ik,kj->j", R, S, R)                      # column sums of (X R)^2
        aux = np.zeros(L)
        aux[:m] = alpha * W
        G = _rot_step(ctx, Xn, _dev(_pad(R, L), Xn), _dev(aux, Xn), 0).cpu().numpy()[:m, :m]
        U, svals, VT = np.linalg.svd(G)
        R = U @ VT
        delta = float(np.sum(svals))
        if abs(delta - delta_old) / delta < rtol:
            break
    if abs(delta - delta_old) / delta > rtol:
        raise RuntimeError("Rotation process did not converge.")
    rot_mat = R
    phi = np.eye(m)
    if power != 1:
        # Promax: regress the powered, max-normalised varimax solution on itself (_rotation.py:57-84)
        B = engine.panel_matmul(ctx, Xn, _dev(_pad(R, L), Xn))     # X R (normalised, rotated)
        mx, mn = engine.panel_colminmax(ctx, B, p)
        cmax = np.maximum(np.abs(mx.cpu().numpy()), np.abs(mn.cpu().numpy()))[:m].astype(np.float64)
        aux = np.ones(L)
        aux[:m] = cmax
        XtP = _rot_step(ctx, Xn, _dev(_pad(R, L), Xn), _dev(aux, Xn), 1, float(power)).cpu().numpy()[:m, :m]
        XtX = R.T @ S @ R
        Lm = np.linalg.inv(XtX) @ XtP
        try:
            sigma_inv = np.diag(np.diag(np.linalg.inv(Lm.T @ Lm)))
        except np.linalg.LinAlgError:
            sigma_inv = np.diag(np.diag(np.linalg.pinv(Lm.T @ Lm)))
        Lm = Lm @ np.sqrt(sigma_inv)
        rot_mat = R @ Lm
        L_inv = np.linalg.inv(Lm)
        phi = L_inv @ L_inv.T
    Xrot = engine.panel_matmul(ctx, Lp, _dev(_pad(rot_mat, L), Lp))   # (h Xn) rot_mat = loadings rot_mat
    return Xrot, p, m, rot_mat, phi


def finish_on_device(ctx, Xrot, p, m):
    """Post-processing of the rotated loadings panel without host passes (eof_rotator.py:150-190): explained
    variance = column sums of squares (float64 Gram diagonal), order by it, unit-norm components, deterministic sign
    (utils/xarray_utils.py:273-301).  -> (components [p, m] float32 host, sorted / normalised / signed;
    expvar (unsorted), idx, sign (unsorted))."""
    L = Xrot.shape[1]
    expvar = np.diag(engine.panel_gram(ctx, Xrot).cpu().numpy())[:m].copy()
    idx = np.argsort(expvar)[::-1]
    mx, mn = engine.panel_colminmax(ctx, Xrot, p)
    mx, mn = mx.cpu().numpy()[:m].astype(np.float64), mn.cpu().numpy()[:m].astype(np.float64)
    sign = np.where(np.abs(mx) >= np.abs(mn), 1.0, -1.0)
    M = np.zeros((L, L))                              # column j of the output = column idx[j], scaled and signed
    M[idx, np.arange(m)] = sign[idx] / np.sqrt(expvar[idx])
    comps = engine.panel_export(ctx, engine.panel_matmul(ctx, Xrot, _dev(M, Xrot)), p, m)
    return comps, expvar, idx, sign


# ---------------------------------------------------------------------------------------------------------------------
# complex loadings (the rotation of ComplexEOF / HilbertEOF models, xeofs/single/eof_rotator.py:294-400; `_promax` /
# `_varimax` "also work for complex numbers", _rotation.py:16,105)
# ---------------------------------------------------------------------------------------------------------------------
CH = 32            # complex panels of the fused step: [Re (32 columns) | Im (32 columns)]
MAX_CROT_MODES = 128


def _cwidth(m):
    """complex columns per panel half: 32, 64 or 128 (panels of 64 / 128 / 256 real columns)"""
    return CH if m <= CH else (64 if m <= 64 else 128)


def _cembed(M, ch=CH):
    """real (2 ch) x (2 ch) matrix E with [Pr | Pi] @ E = [Re(P M) | Im(P M)] for a complex m x m' matrix M"""
    l, m = M.shape
    E = np.zeros((2 * ch, 2 * ch))
    E[:l, :m] = M.real
    E[ch:ch + l, :m] = -M.imag
    E[:l, ch:ch + m] = M.imag
    E[ch:ch + l, ch:ch + m] = M.real
    return E


def _cblocks(G, m, ch=CH):
    """X^H T (complex m x m) from the real product [Xr | Xi]^T [Tr | Ti]"""
    rr, ri = G[:m, :m], G[:m, ch:ch + m]
    ir, ii = G[ch:ch + m, :m], G[ch:ch + m, ch:ch + m]
    return (rr + ii) + 1j * (ri - ir)


def _crot_step(ctx, X, R, auxh, mode, power, ch, m):
    """One step of the complex loop on the [Re | Im] panel X: the complex m x m matrix X^H (b (|b|^2 - aux)) (mode 2) or
    b^H ((b / aux) (|b| / aux)^(power-1)) (mode 3), b = X R -- modes 2 / 3 of `eofx_panel_rot_step_f64` with R in its real
    embedding, for 32, 64 or 128 columns per half."""
    aux = np.zeros(2 * ch) if mode == 2 else np.ones(2 * ch)
    aux[:m] = aux[ch:ch + m] = auxh
    G = engine.panel_rot_step(ctx, X, _dev(_cembed(R, ch), X), _dev(aux, X), mode, float(power))
    return _cblocks(G.cpu().numpy(), m, ch)


def cpromax_panel(ctx, loadings: np.ndarray, power: int = 1, max_iter: int = 1000, rtol: float = 1e-8, col_scale=None):
    """`promax_panel` for complex loadings [p, m]: -> ([Re | Im] panel on the device, p, m, complex rotation
    matrix, complex phi).  Up to 32 modes every step of the reference loop is the same fused pass
    (`eofx_panel_rot_step_f64`, modes 2 / 3) with the complex m x m matrices in their real embedding; 33 .. 128 modes
    (rare) use 64 / 128-column halves: the same entry, column-blocked kernel."""
    loadings = np.asarray(loadings)
    p, m = loadings.shape
    if m < 2:
        raise ValueError("Cannot rotate {:} modes (columns), but must be 2 or more.".format(m))
    if m > MAX_CROT_MODES:
        raise NotImplementedError(f"rotation of more than {MAX_CROT_MODES} complex modes is not supported by this build")
    ch = _cwidth(m)
    L = 2 * ch
    rows_pad = (p + 511) // 512 * 512
    host = np.zeros((p, L), np.float32)
    host[:, :m], host[:, ch:ch + m] = loadings.real, loadings.imag
    Lp = engine.panel_import(ctx, host, rows_pad, L)
    del host
    if col_scale is not None:
        Lp = engine.panel_matmul(ctx, Lp, _dev(_cembed(np.diag(np.asarray(col_scale, dtype=np.float64)).astype(complex), ch), Lp))
    Xn = engine.panel_row_normalize(ctx, Lp)                       # Kaiser: rows / (sqrt(sum |x|^2) + eps)
    S = _cblocks(engine.panel_gram(ctx, Xn).cpu().numpy(), m, ch)  # X^H X
    S = 0.5 * (S + S.conj().T)
    R = np.eye(m, dtype=complex)
    alpha = 1.0 / p
    delta = 0.0
    for _ in range(int(max_iter)):
        delta_old = delta
        W = np.einsum("ij,ik,kj->j", R.conj(), S, R).real          # column sums of |X R|^2
        G = _crot_step(ctx, Xn, R, alpha * W, 2, 1.0, ch, m)
        U, svals, VT = np.linalg.svd(G)
        R = U @ VT
        delta = float(np.sum(svals))
        if abs(delta - delta_old) / delta < rtol:
            break
    if abs(delta - delta_old) / delta > rtol:
        raise RuntimeError("Rotation process did not converge.")
    rot_mat = R
    phi = np.eye(m, dtype=complex)
    if power != 1:
        B = engine.panel_matmul(ctx, Xn, _dev(_cembed(R, ch), Xn)) # X R (normalised, rotated)
        cmax = engine.cpanel_colabsmax(ctx, B, p).cpu().numpy()[:m].astype(np.float64)
        XtP = _crot_step(ctx, Xn, R, cmax, 3, float(power), ch, m)
        XtX = R.conj().T @ S @ R
        Lm = np.linalg.inv(XtX) @ XtP
        try:
            sigma_inv = np.diag(np.diag(np.linalg.inv(Lm.conj().T @ Lm)))
        except np.linalg.LinAlgError:
            sigma_inv = np.diag(np.diag(np.linalg.pinv(Lm.conj().T @ Lm)))
        Lm = Lm @ np.sqrt(sigma_inv)
        rot_mat = R @ Lm
        L_inv = np.linalg.inv(Lm)
        phi = L_inv @ L_inv.conj().T
    Xrot = engine.panel_matmul(ctx, Lp, _dev(_cembed(rot_mat, ch), Lp))   # (h Xn) rot_mat = loadings rot_mat
    return Xrot, p, m, rot_mat, phi


def cfinish_on_device(ctx, Xrot, p, m):
    """`finish_on_device` for a complex [Re | Im] panel: explained variance = column sums of |x|^2, order, unit-norm
    components, the reference's +-1 sign from numpy's lexicographic complex max / min (real part decides; ties on it are
    measure-zero).  -> (components [p, m] complex64 host, expvar (unsorted), idx, sign (unsorted))."""
    torch = engine._torch()
    CH = Xrot.shape[1] // 2
    Gd = np.diag(engine.panel_gram(ctx, Xrot).cpu().numpy())
    expvar = (Gd[:m] + Gd[CH:CH + m]).copy()
    idx = np.argsort(expvar)[::-1]
    amax, amin = engine.panel_colargminmax(ctx, Xrot, p)
    cols = torch.arange(m, device=Xrot.device)
    pick = lambda ix: (Xrot[ix[:m], cols].double().cpu().numpy(), Xrot[ix[:m], cols + CH].double().cpu().numpy())
    (mr, mi), (nr, ni) = pick(amax), pick(amin)
    sign = np.where(np.hypot(mr, mi) >= np.hypot(nr, ni), 1.0, -1.0)
    M = np.zeros((m, m), dtype=complex)                   # column j of the output = column idx[j], scaled and signed
    M[idx, np.arange(m)] = sign[idx] / np.sqrt(expvar[idx])
    out = engine.panel_matmul(ctx, Xrot, _dev(_cembed(M, CH), Xrot))[:p].cpu().numpy()
    comps = np.empty((p, m), np.complex64)
    comps.real, comps.imag = out[:, :m], out[:, CH:CH + m]
    return comps, expvar, idx, sign
