"""xeofs_amd.cross.MCA lives in cpcca.py (MCA = CPCCA with alpha = 1, xeofs/cross/mca.py:107)."""
from .cpcca import MCA  # noqa: F401
