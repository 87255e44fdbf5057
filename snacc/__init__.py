"""``import snacc`` -- the reference's package name, served by the MI355X-native implementation.

The reference exports ``compressed_size``, ``compute_distance`` and ``__version__`` from package
``snacc`` (ref:snacc/__init__.py:1-2) and installs the console script ``snacc = snacc.cli:cli``
(ref:setup.py:115-117).  Everything here is a re-export of :mod:`snacc_amd`; there is no second
implementation."""
from snacc_amd import __version__, compressed_size, compute_distance, ncd  # noqa: F401
