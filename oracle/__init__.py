"""CPU oracle for the lz4 NCD hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``snacc_amd``) never does.

* ``lz4f_oracle.c``  -- C restatement of liblz4 1.9.3 ``LZ4F_compressFrame(prefs=NULL)``
  (size only), the arithmetic behind ``lz4framed.compress`` at
  ref:snacc/pairwise_ncd.py:80.
* ``loader.py``      -- ctypes loader of the C restatements + the generator of SURVEY.md 8c
  (``lcg_genome`` / ``lcg_mutant``) and multi-threaded pair loops for the tests and the
  ``cpu_baseline`` leg.  (The host logic of ref:snacc/pairwise_ncd.py / ref:snacc/cli.py is pinned
  by ``tests/golden/golden.json``, generated from the reference's own Python.)
* ``deflate_oracle.c`` / ``deflate_rules.c`` / ``deflate.py`` -- zlib 1.2.11 deflate sizes behind
  ``gzip.compress`` / ``zlib.compress`` (ref:snacc/pairwise_ncd.py:73-78): a window-faithful
  restatement, and the rules the GPU kernel applies, both pinned against the interpreter's zlib.
* ``liblz4_ref.py``  -- optional ctypes binding to the liblz4 1.9.3 *binary* of the
  image, used to pin the C restatement (differential fuzz).  Test-only.
"""
from .loader import lib, build, lz4f_size, lz4f_size_pair, lcg_genome, lcg_mutant  # noqa: F401
