"""``snacc.version`` (ref:snacc/version.py) -> :mod:`snacc_amd.version`."""
from snacc_amd.version import __version__  # noqa: F401
