from .cpcca import CCA, CPCCA, MCA, RDA  # noqa: F401
