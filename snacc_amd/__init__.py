"""snacc_amd -- MI355X-native drop-in for the lz4 all-pairs NCD hot path of alexsweeten/snacc.

Exports what ``import snacc`` exports (ref:snacc/__init__.py:1-2)."""
from .pairwise_ncd import compressed_size, compute_distance  # noqa: F401
from .version import __version__  # noqa: F401
