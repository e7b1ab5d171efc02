from .eof import EOF  # noqa: F401
