from .mca import MCA  # noqa: F401
