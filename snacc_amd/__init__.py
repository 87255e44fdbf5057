"""snacc_amd -- MI355X-native drop-in for the lz4 all-pairs NCD hot path of alexsweeten/snacc.

Exports what ``import snacc`` exports (ref:snacc/__init__.py:1-2), plus the sequence-level ``ncd(seq_i, seq_j,
compressor)`` convenience (not in the reference)."""
from .pairwise_ncd import compressed_size, compute_distance, ncd  # noqa: F401
from .version import __version__  # noqa: F401
