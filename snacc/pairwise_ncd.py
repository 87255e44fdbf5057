"""``snacc.pairwise_ncd`` (ref:snacc/pairwise_ncd.py:15,42,93) -> :mod:`snacc_amd.pairwise_ncd`."""
from snacc_amd.pairwise_ncd import (all_pairs, compressed_size, compute_distance,  # noqa: F401
                                    extract_sequences, ncd)
