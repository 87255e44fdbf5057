from .version import __version__  # noqa: F401
