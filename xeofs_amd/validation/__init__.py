from .bootstrapper import EOFBootstrapper  # noqa: F401
