from .decomposer import Decomposer  # noqa: F401
