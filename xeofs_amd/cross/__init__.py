from .complex_mca import (ComplexCCA, ComplexCPCCA, ComplexCPCCARotator, ComplexMCA, ComplexMCARotator,  # noqa: F401
                          ComplexRDA, HilbertCCA, HilbertCPCCA, HilbertCPCCARotator, HilbertMCA, HilbertMCARotator,
                          HilbertRDA)
from .cpcca import CCA, CPCCA, MCA, RDA  # noqa: F401
from .cpcca_rotator import CPCCARotator, MCARotator  # noqa: F401
