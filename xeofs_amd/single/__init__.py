from .eof import EOF, ComplexEOF, HilbertEOF  # noqa: F401
