from .complex_mca import ComplexMCA, ComplexMCARotator, HilbertMCA, HilbertMCARotator  # noqa: F401
from .cpcca import CCA, CPCCA, MCA, RDA  # noqa: F401
from .cpcca_rotator import CPCCARotator, MCARotator  # noqa: F401
