from .eof import EOF, ComplexEOF, HilbertEOF  # noqa: F401
from .eof_rotator import ComplexEOFRotator, EOFRotator, HilbertEOFRotator  # noqa: F401
