"""xeofs_amd -- MI355X-native EOF / randomized-SVD engine behind the xeofs API.

`xeofs_amd.single.EOF` / `xeofs_amd.cross.MCA` mirror `xeofs.single.EOF` /
`xeofs.cross.MCA`; the numerics run in hand-written HIP kernels (libeofx.so, C ABI in
include/eofx.h).  There is no CPU fallback.
"""

__version__ = "0.1.0"

from . import cross, single, validation  # noqa: E402,F401
from .labelled import DataArray, Dataset  # noqa: E402,F401
