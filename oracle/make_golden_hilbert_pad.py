"""G9: pin the oracle's padded Hilbert transform (row R16) to outputs of the REFERENCE's own
functions, `_hilbert_transform_with_padding` and `_pad_exp` of /root/reference/xeofs/utils/hilbert_transform.py:40-114.

xeofs itself cannot be imported here (no xarray).  The two functions use numpy and scipy.signal.hilbert only; the file's
other top-level imports -- `import xarray as xr` (used by the public wrapper around them, which is NOT called) and
`from .data_types import DataArray` (a type annotation) -- are satisfied by empty placeholder modules so that the file can be
executed WHERE IT LIES.  Nothing of it is copied; what runs below is the reference's own code for the two functions.

    python oracle/make_golden_hilbert_pad.py

Writes tests/golden/g9_hilbert_pad.npz: seeded real fields with trends and offsets (odd and even lengths, one column and
several) and the reference's analytic signal for padding "exp" (decay 0.2 and 0.35) and padding None, plus `_pad_exp` itself.
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/xeofs/utils/hilbert_transform.py"

pkg = types.ModuleType("refpkg")
pkg.__path__ = []
dt = types.ModuleType("refpkg.data_types")
dt.DataArray = object
sys.modules.setdefault("xarray", types.ModuleType("xarray"))      # placeholder: never touched by the functions used here
sys.modules["refpkg"] = pkg
sys.modules["refpkg.data_types"] = dt
spec = importlib.util.spec_from_file_location("refpkg.hilbert_transform", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
cases = [(37, 3, 1), (64, 5, 2), (200, 4, 3), (1, 2, 4), (2, 3, 5), (129, 1, 6)]
for n, p, seed in cases:
    rng = np.random.default_rng(seed)
    t = np.arange(n)[:, None]
    y = (np.cos(0.31 * t + rng.uniform(0, 6, p)) * rng.uniform(0.5, 3, p) + 0.02 * t * rng.standard_normal(p)
         + rng.uniform(-40, 40, p) + 0.2 * rng.standard_normal((n, p)))
    key = f"n{n}_p{p}"
    out[f"{key}_y"] = y
    if n >= 2:
        out[f"{key}_pad02"] = ref._pad_exp(y.copy(), decay_factor=0.2)
        out[f"{key}_exp02"] = ref._hilbert_transform_with_padding(y.copy(), padding="exp", decay_factor=0.2)
        out[f"{key}_exp035"] = ref._hilbert_transform_with_padding(y.copy(), padding="exp", decay_factor=0.35)
    out[f"{key}_none"] = ref._hilbert_transform_with_padding(y.copy(), padding=None)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g9_hilbert_pad.npz"), **out)
print("wrote g9_hilbert_pad.npz:", sorted(out)[:6], "...", len(out), "arrays")
