"""GPU parity tests: the HIP path, through the C-ABI, against the oracle on the same seeded
inputs, against the committed golden vectors, and -- at BASELINE.json's full 1 Mbp size --
through size-independent properties.  Bit-exact everywhere (integer sizes); NCD floats are
derived on the host from equal integers and compared exactly (tolerance of north_star: 1e-6)."""
import numpy as np
import pytest

from conftest import lcg_bytes, materialise_cli_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from snacc_amd import hip_backend
    hip_backend.load()                       # raises if libsnacc_hip.so is missing: no silent fallback
    return hip_backend


def _check_all(hip, oracle, seqs, **opts):
    with hip.HipContext(0, **opts) as ctx:
        ctx.upload(seqs)
        s = ctx.singles()
        p = ctx.pairs()
        packed = ctx.num_packed
    exp_s = np.array([oracle.lz4f_size(x) for x in seqs], dtype=np.uint32)
    exp_p = np.array([[oracle.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
    assert np.array_equal(s, exp_s)
    assert np.array_equal(p, exp_p)
    return packed


def test_phase_a_on_demand_per_row_and_spread_over_the_card(hip, oracle_mod):
    """Round 4: (a) phase A spreads its N jobs over every wave of the card and runs the two-lane loop (fast_spec=0: the
    one-lane loop; fast_lanes: the old packed geometry) -- same sizes and the same snapshots (the pairs that start from them);
    (b) option defer_singles: the upload leaves phase A to the calls that need it -- snk_singles_rows for a block of rows,
    snk_pairs / snk_pairs_list for the prefixes they compute -- on a set that mixes 2-bit sequences (with and without
    exception sites), byte-kernel sequences and one-shot inputs; (c) snk_upload_times reports the upload's stages."""
    o = oracle_mod
    nrun = np.frombuffer(bytes(o.lcg_genome(41, 150000)).replace(b"ACGTA", b"ACNTA"), dtype=np.uint8)
    seqs = [o.lcg_genome(31 + i, 140000 + 9001 * i) for i in range(5)] + [nrun, o.lcg_genome(51, 30000),
            np.frombuffer(lcg_bytes(32, 90000, b"ACDEFGHIKLMNPQRSTVWY"), dtype=np.uint8), o.lcg_genome(52, 70000)]
    n = len(seqs)
    exp_s = np.array([o.lz4f_size(x) for x in seqs], dtype=np.uint32)
    exp_p = np.array([[o.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
    pure = [o.lcg_genome(61 + i, 200000 + 4099 * i) for i in range(7)]
    exp_ps = np.array([o.lz4f_size(x) for x in pure], dtype=np.uint32)
    exp_pp = np.array([[o.lz4f_size_pair(a, b) for b in pure] for a in pure], dtype=np.uint32)
    for opts in ({}, {"fast_spec": 0}, {"fast_lanes": 3}, {"fast_lanes": 20, "fast_spec": 0}):     # (a set with exceptions holds 83 chains per CU: 4 x 21 do not fit)
        with hip.HipContext(0, **opts) as ctx:
            ctx.upload(pure)
            assert np.array_equal(ctx.singles(), exp_ps), opts
            assert np.array_equal(ctx.pairs(), exp_pp), opts
            t = ctx.upload_times()
            assert t["total"] > 0 and t["singles"] > 0 and abs(sum(v for k, v in t.items() if k != "total") - t["total"]) < 0.05
        with hip.HipContext(0, **opts) as ctx:
            ctx.upload(seqs)
            assert np.array_equal(ctx.singles(), exp_s), opts
            assert np.array_equal(ctx.pairs(), exp_p), opts
    with hip.HipContext(0, defer_singles=1) as ctx:
        ctx.upload(seqs)
        assert ctx.upload_times()["singles"] == 0.0
        assert np.array_equal(ctx.singles_rows(2, 5), exp_s[2:5])             # a block of rows: their phase A runs now
        assert np.array_equal(ctx.pairs(2, 5), exp_p[2:5])
        assert np.array_equal(ctx.pairs(5, 7), exp_p[5:7])                    # rows nobody asked the singles of: pairs run it
        ij = [(8, 0), (0, 8), (7, 7), (1, 6)]
        assert ctx.pairs_list(ij).tolist() == [int(exp_p[i, j]) for i, j in ij]      # ... and so does a pair list, per prefix
        assert np.array_equal(ctx.singles(), exp_s)                           # the rest
        assert np.array_equal(ctx.pairs(), exp_p)
    with hip.HipContext(0, defer_singles=1) as ctx:                           # gzip / zlib never run the lz4 pass
        ctx.upload(seqs[:4])
        from oracle import deflate as dfl
        assert [int(v) for v in ctx.deflate_singles("zlib")] == [dfl.zlib_size(x) for x in seqs[:4]]
        assert ctx.upload_times()["singles"] == 0.0
        assert np.array_equal(ctx.singles(), exp_s[:4])                       # (still there when asked for)


def test_tiny_and_empty_inputs(hip, oracle_mod):
    seqs = [b"ACGT" * 10, b"ACGTTGCA" * 3, b"A", b"", b"ACGTN" * 5, b"GATTACA" * 1000, b"ACGTACGTACGTA", b"ACGTACGTACGT"]
    _check_all(hip, oracle_mod, seqs)


def test_lcg_100k_fast_and_generic_paths(hip, oracle_mod):
    seqs = [oracle_mod.lcg_genome(1 + i, 100000) for i in range(6)]
    assert _check_all(hip, oracle_mod, seqs) == 6
    _check_all(hip, oracle_mod, seqs, force_generic=1)                      # tight-loop byte kernel (compact table)
    _check_all(hip, oracle_mod, seqs, force_generic=1, bytes_compact=0)     # same, full 4096-slot table
    _check_all(hip, oracle_mod, seqs, force_generic=1, bytes_legacy=1)      # legacy u32-table byte kernel
    _check_all(hip, oracle_mod, seqs, force_generic=1, bytes_lanes=4, bytes_waves=3)


def test_ragged_lengths_around_block_edges(hip, oracle_mod):
    lens = [65536, 65537, 131072, 200001, 30000, 35536, 12, 65535 + 65536, 65548, 4, 196608]
    seqs = [oracle_mod.lcg_genome(11 + i, n) for i, n in enumerate(lens)]
    _check_all(hip, oracle_mod, seqs)


def test_last_blocks_of_every_short_length_after_the_loops_own_turn_around(hip, oracle_mod):
    """(round 4) The two-lane loop takes a block step on its own when another full step of the same frame follows -- at least 13
    bytes, not the frame's end, no snapshot to write -- and goes back to the head of the wave loop otherwise.  Pairs whose
    stream is 3 blocks plus 0 .. 14 bytes (and a few more) stand on both sides of every one of those conditions; singles of the same
    lengths are the snapshot side (phase A writes its snapshot at the last block edge)."""
    x = oracle_mod.lcg_genome(77, 70000)
    tails = [0, 1, 4, 5, 11, 12, 13, 14, 15, 100, 65535]
    seqs = [x] + [oracle_mod.lcg_genome(78 + d, 3 * 65536 - 70000 + d) for d in tails]
    seqs += [oracle_mod.lcg_genome(60, 2 * 65536 + d) for d in (0, 12, 13)]
    assert _check_all(hip, oracle_mod, seqs) == len(seqs)


def test_mixed_alphabets_raw_blocks_and_n_runs(hip, oracle_mod):
    rng = np.random.default_rng(7)
    o = oracle_mod
    nrun = np.concatenate([o.lcg_genome(22, 70000), np.frombuffer(b"N" * 500, dtype=np.uint8), o.lcg_genome(23, 70000)])
    seqs = [o.lcg_genome(21, 150000), rng.integers(0, 256, 140000, dtype=np.uint8), nrun,
            np.tile(o.lcg_genome(24, 700), 300),
            np.frombuffer(lcg_bytes(31, 90000, b"ACDEFGHIKLMNPQRSTVWY"), dtype=np.uint8),
            np.frombuffer(bytes(o.lcg_genome(25, 120000)).lower(), dtype=np.uint8)]
    # the genome with one N run stays on the 2-bit kernel (an exception site), and so does (round 4) the lower-case genome of this
    # upper-case set: one stretch of the other case, however long
    assert _check_all(hip, oracle_mod, seqs) == 4
    assert _check_all(hip, oracle_mod, seqs, exc_limit=0) == 2
    _check_all(hip, oracle_mod, seqs, bytes_legacy=1)


def test_byte_kernel_on_everything(hip, oracle_mod):
    """force_generic routes the ACGT inputs through the byte kernel too: ragged block edges,
    long matches, low complexity, related genomes."""
    o = oracle_mod
    lens = [65537, 131072, 200001, 70000, 65535 + 65536, 65548, 196608]
    seqs = [o.lcg_genome(11 + i, n) for i, n in enumerate(lens)]
    rep = np.tile(o.lcg_genome(32, 5000), 40)
    seqs += [np.tile(o.lcg_genome(31, 37), 3000), rep, o.lcg_mutant(rep, 5),
             np.frombuffer(b"A" * 150000, dtype=np.uint8), np.frombuffer(b"AC" * 60000, dtype=np.uint8),
             np.frombuffer(b"N" * 1000 + bytes(o.lcg_genome(77, 90000)) + b"n" * 3000, dtype=np.uint8)]
    _check_all(hip, oracle_mod, seqs, force_generic=1)
    _check_all(hip, oracle_mod, seqs, force_generic=1, bytes_compact=0)


def test_byte_kernels_with_tables_in_global_memory(hip, oracle_mod):
    """bytes_gt: every lane of every wave runs a chain whose table (liblz4's own layout, u32 positions) lies in
    global memory.  The same sets as the LDS forms -- ragged block edges, long matches, low complexity, seam
    strings outside the compact set, raw blocks, protein -- full table and both compact capacities."""
    o = oracle_mod
    lens = [65537, 131072, 200001, 70000, 65535 + 65536, 65548, 196608]
    seqs = [o.lcg_genome(11 + i, n) for i, n in enumerate(lens)]
    rep = np.tile(o.lcg_genome(32, 5000), 40)
    seqs += [np.tile(o.lcg_genome(31, 37), 3000), rep, o.lcg_mutant(rep, 5),
             np.frombuffer(b"A" * 150000, dtype=np.uint8), np.frombuffer(b"AC" * 60000, dtype=np.uint8),
             np.frombuffer(b"N" * 1000 + bytes(o.lcg_genome(77, 90000)) + b"n" * 3000, dtype=np.uint8),
             np.frombuffer(bytes(o.lcg_genome(78, 66000)) + b"NNNN", dtype=np.uint8),
             np.frombuffer(b"NNNN" + bytes(o.lcg_genome(79, 80000)), dtype=np.uint8)]
    _check_all(hip, o, seqs, force_generic=1, bytes_gt=2)                       # compact, 1024 slots
    _check_all(hip, o, seqs, force_generic=1, bytes_compact=0, bytes_gt=4)      # full table, hashes taken from the window
    soft = [np.frombuffer(bytes(x[:40000]) + bytes(x[40000:]).lower(), dtype=np.uint8) for x in seqs[:6]]
    with hip.HipContext(0, force_generic=1, bytes_gt=3) as ctx:                 # compact, 2048 slots (both cases)
        ctx.upload(soft)
        assert 1024 < ctx.num_compact_hashes <= 2048
        p = ctx.pairs()
    assert np.array_equal(p, np.array([[o.lz4f_size_pair(a, b) for b in soft] for a in soft], dtype=np.uint32))
    rng = np.random.default_rng(7)
    mixed = [o.lcg_genome(21, 150000), rng.integers(0, 256, 140000, dtype=np.uint8),
             np.frombuffer(lcg_bytes(31, 90000, b"ACDEFGHIKLMNPQRSTVWY"), dtype=np.uint8),
             np.frombuffer(lcg_bytes(32, 70001, b"ACDEFGHIKLMNPQRSTVWY"), dtype=np.uint8)]
    _check_all(hip, o, mixed, bytes_gt=1)
    _check_all(hip, o, mixed, bytes_gt=8, bytes_gt_wgs=2)


def test_byte_kernels_with_two_lanes_per_chain(hip, oracle_mod):
    """bytes_spec: the LDS byte kernels' slot-stream loop with a speculative partner lane per chain (the second lane
    probes 5 bytes ahead; its probe counts when the first lane's match ends there).  The same sets as the one-lane
    forms -- ragged block edges, long matches, low complexity, seam strings outside the compact set, raw blocks,
    soft-masked text, protein -- both settings of the option against the oracle, and against each other."""
    o = oracle_mod
    lens = [65537, 131072, 200001, 70000, 65535 + 65536, 65548, 196608]
    seqs = [o.lcg_genome(11 + i, n) for i, n in enumerate(lens)]
    rep = np.tile(o.lcg_genome(32, 5000), 40)
    seqs += [np.tile(o.lcg_genome(31, 37), 3000), rep, o.lcg_mutant(rep, 5),
             np.frombuffer(b"A" * 150000, dtype=np.uint8), np.frombuffer(b"AC" * 60000, dtype=np.uint8),
             np.frombuffer(b"N" * 1000 + bytes(o.lcg_genome(77, 90000)) + b"n" * 3000, dtype=np.uint8),
             np.frombuffer(bytes(o.lcg_genome(78, 66000)) + b"NNNN", dtype=np.uint8),
             np.frombuffer(b"NNNN" + bytes(o.lcg_genome(79, 80000)), dtype=np.uint8)]
    for spec in (0, 1):
        _check_all(hip, o, seqs, force_generic=1, bytes_spec=spec)                               # compact, 1024 slots
        _check_all(hip, o, seqs, force_generic=1, bytes_spec=spec, cbytes_lanes=3, cbytes_waves=2)   # few chains: trips in which no second lane counts
        _check_all(hip, o, seqs[:8], force_generic=1, bytes_compact=0, bytes_gt=0, bytes_spec=spec)  # full table in LDS
    soft = [np.frombuffer(bytes(x[:40000]) + bytes(x[40000:]).lower(), dtype=np.uint8) for x in seqs[:6]]
    exp = np.array([[o.lz4f_size_pair(a, b) for b in soft] for a in soft], dtype=np.uint32)
    for spec in (0, 1):
        with hip.HipContext(0, force_generic=1, bytes_gt=0, bytes_spec=spec) as ctx:            # compact, 2048 slots (both cases)
            ctx.upload(soft)
            assert 1024 < ctx.num_compact_hashes <= 2048
            assert np.array_equal(ctx.pairs(), exp)
    rng = np.random.default_rng(7)
    mixed = [o.lcg_genome(21, 150000), rng.integers(0, 256, 140000, dtype=np.uint8),
             np.frombuffer(lcg_bytes(31, 90000, b"ACDEFGHIKLMNPQRSTVWY"), dtype=np.uint8),
             np.frombuffer(lcg_bytes(32, 70001, b"ACDEFGHIKLMNPQRSTVWY"), dtype=np.uint8)]
    _check_all(hip, o, mixed, bytes_gt=0, bytes_spec=1)


def test_global_table_byte_kernel_is_the_default_for_large_full_table_launches(hip, oracle_mod):
    """Protein sequences (full table): a launch with more jobs than two rounds of the LDS kernel takes the
    global-table kernel by itself.  144 x 144 pairs at one workgroup of 64 chains per CU and launch = two
    launches of the kernel; the whole matrix equals the LDS kernel's, a sample equals the oracle's sizes."""
    o = oracle_mod
    n = 144
    seqs = [np.frombuffer(lcg_bytes(500 + i, 66000 + 37 * (i % 5), b"ACDEFGHIKLMNPQRSTVWY"), dtype=np.uint8) for i in range(n)]
    with hip.HipContext(0, bytes_gt=0) as ctx:
        ctx.upload(seqs)
        lds = ctx.pairs()
        ms_lds = ctx.last_pairs_ms()
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        auto = ctx.pairs()
        ms_auto = ctx.last_pairs_ms()
    with hip.HipContext(0, bytes_gt=1) as ctx:
        ctx.upload(seqs)
        two = ctx.pairs()
    assert np.array_equal(lds, auto) and np.array_equal(lds, two)
    for i, j in [(0, 0), (0, 1), (1, 0), (n - 1, n - 1), (n - 1, 0), (77, 5), (5, 77), (100, 143)]:
        assert int(auto[i, j]) == o.lz4f_size_pair(seqs[i], seqs[j])
    print(f"144 x 144 protein pairs: LDS tables {ms_lds:.1f} ms, global tables {ms_auto:.1f} ms")
    assert ms_auto < ms_lds


def test_compact_byte_kernel_seam_strings(hip, oracle_mod):
    """Mostly-ACGT genomes with N runs: the byte kernel runs with its compact table (<= 1024 distinct
    5-byte hashes).  Sequence ends/starts are chosen so that the 5-byte strings spanning the x/y seam
    hash OUTSIDE the resident set (chain-private slots), including identical seam strings."""
    o = oracle_mod

    def g(seed, n):
        return bytes(o.lcg_genome(seed, n))
    seqs = [
        g(1, 90000) + b"NNNN",                         # ends in N: seam strings NNNN+?, ...
        b"NNNN" + g(2, 80000),                         # starts with N: with the one above -> NNNNN x4 (shared slot)
        g(3, 70000) + b"N" * 300 + g(4, 40000) + b"RY",
        b"K" + g(5, 100000),
        g(6, 120000),                                   # pure ACGT (2-bit path with itself)
        g(7, 66000) + b"NNN",
        b"NN" + g(8, 66000) + b"NNNNN" + g(9, 3000),
        g(10, 65536),                                   # exactly one block
        b"N" * 70000,
    ]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        assert 894 <= ctx.num_compact_hashes <= 1024
        s, p = ctx.singles(), ctx.pairs()
    exp_s = np.array([o.lz4f_size(x) for x in seqs], dtype=np.uint32)
    exp_p = np.array([[o.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
    assert np.array_equal(s, exp_s)
    assert np.array_equal(p, exp_p)
    with hip.HipContext(0, bytes_compact=0) as ctx:     # the full table gives the same
        ctx.upload(seqs)
        assert ctx.num_compact_hashes == 0
        assert np.array_equal(ctx.pairs(), exp_p)


def test_compact_2048_soft_masked_genomes(hip, oracle_mod):
    """Soft-masked genomes (upper + lower case + N runs): ~1900 distinct 5-byte hashes -> the
    2048-slot compact table."""
    o = oracle_mod
    rng = np.random.default_rng(8)

    def soft(seed, n):
        g = bytearray(bytes(o.lcg_genome(seed, n)))
        for start in rng.integers(0, n - 3000, 12):
            ln = int(rng.integers(50, 2500))
            g[start:start + ln] = bytes(g[start:start + ln]).lower()
        for start in rng.integers(0, n - 500, 3):
            g[start:start + int(rng.integers(1, 300))] = b"N" * int(rng.integers(1, 300))
        return bytes(g)
    seqs = [soft(1, 120000), soft(2, 90000), soft(3, 140000), bytes(o.lcg_genome(4, 100000)).lower(),
            bytes(o.lcg_genome(5, 100000))]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        assert 1024 < ctx.num_compact_hashes <= 2048
        s, p = ctx.singles(), ctx.pairs()
    assert np.array_equal(s, np.array([o.lz4f_size(x) for x in seqs], dtype=np.uint32))
    assert np.array_equal(p, np.array([[o.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32))


def test_oneshot_small_genomes(hip, oracle_mod):
    """Pairs of <= 64 KiB (viral / mitochondrial sizes) use liblz4's one-shot mode (13-bit hash of 4
    bytes): tight-loop kernel with compact table, with the full 8192-slot table, and the legacy
    kernel must all agree with the oracle.  Includes sums of exactly 65536 and seam strings with N."""
    o = oracle_mod
    rng = np.random.default_rng(4)
    anc = o.lcg_genome(50, 10700)
    seqs = [bytes(anc)] + [bytes(o.lcg_mutant(anc, 60 + i)) for i in range(5)]
    seqs += [bytes(o.lcg_genome(70, 16569)), bytes(o.lcg_genome(71, 32768)), bytes(o.lcg_genome(72, 32768)),
             bytes(o.lcg_genome(73, 29000)) + b"NNN", b"NN" + bytes(o.lcg_genome(74, 5000)), b"ACGT" * 3, b"A" * 20000,
             bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), 9000)), b"", b"ACGTACGTACGTA"]
    for opts in ({}, {"bytes_compact": 0}, {"bytes_legacy": 1}, {"cos_lanes": 5, "cos_waves": 3}):
        _check_all(hip, oracle_mod, seqs, **opts)
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        assert ctx.num_packed >= 8


def test_compact_falls_back_when_too_many_hashes(hip, oracle_mod):
    rng = np.random.default_rng(3)
    seqs = [oracle_mod.lcg_genome(1, 100000), rng.integers(0, 256, 100000, dtype=np.uint8)]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        assert ctx.num_compact_hashes == 0              # > 1024 distinct hashes: full table


def test_long_matches_and_low_complexity(hip, oracle_mod):
    o = oracle_mod
    rep = np.tile(o.lcg_genome(32, 5000), 40)
    seqs = [np.tile(o.lcg_genome(31, 37), 5000), rep, o.lcg_mutant(rep, 5),
            np.frombuffer(b"A" * 300000, dtype=np.uint8), o.lcg_genome(33, 250000),
            np.frombuffer(b"AC" * 100000, dtype=np.uint8)]
    assert _check_all(hip, oracle_mod, seqs) == 6


def test_long_matches_that_run_up_to_and_across_the_seam(hip, oracle_mod):
    """The extension of a match beyond 12 bases reads 128 bases a step where both runs lie inside one sequence, with the next
    step's loads issued ahead, and 16 a step across the seam (round 4).  y starts with a copy of x's last 300 .. 1300 bases -- the
    candidate's run walks through the end of x into y's first bases -- for every residue of len(x) mod 4, with the copy broken
    after 13 .. 700 bases, next to x's whose tail is a tandem repeat (runs of both sides inside one sequence, every phase)."""
    o = oracle_mod
    rng = np.random.default_rng(4128)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    for k in range(8):
        lx = 20000 + k                                           # every residue mod 4, twice
        x = o.lcg_genome(900 + k, lx).copy()
        if k >= 4:
            unit = int(rng.integers(3, 90)); x[-2000:] = np.tile(x[-2000:-2000 + unit], 2000 // unit + 1)[:2000]
        tail = int(rng.integers(300, 1300))
        y = np.concatenate([x[-tail:], o.lcg_genome(950 + k, 5000 + 3 * k)])
        cut = int(rng.integers(13, 700))
        y[cut] = acgt[(np.flatnonzero(acgt == y[cut])[0] + 1) % 4]          # the copy ends here
        seqs += [x, y]
    assert _check_all(hip, oracle_mod, seqs) == len(seqs)
    assert _check_all(hip, oracle_mod, seqs, fast_asm=0, fast_spec=0) == len(seqs)


def _tandem(rng, n, unit_len, rate):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    a = np.tile(rng.choice(acgt, unit_len), n // unit_len + 1)[:n].copy()
    hit = rng.random(n) < rate
    a[hit] = rng.choice(acgt, int(hit.sum()))
    return a


def test_blocks_ending_in_short_matches(hip, oracle_mod):
    """Regression (found by tools/gpu_fuzz.py): a 64 KiB block whose last match is 11..31 bases
    long slid the 2-bit kernel's cursor reservoir right before the block closed and left its
    look-ahead word stale; the next block then parsed wrong bases.  Mutated tandem repeats end
    blocks that way often; the single sizes (x part) and the pair sizes (y part, both loop
    instantiations) must equal the oracle's."""
    rng = np.random.default_rng(2510)
    seqs = [_tandem(rng, int(rng.integers(197000, 262000)), int(rng.integers(150, 3000)), rate)
            for rate in (0.001, 0.02, 0.02, 0.05) for _ in range(12)]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        assert ctx.num_packed == len(seqs)
        s = ctx.singles()
        ij = np.array([(i, (7 * i + 3) % len(seqs)) for i in range(len(seqs))], dtype=np.int32)
        p = ctx.pairs_list(ij)
    exp_s = np.array([oracle_mod.lz4f_size(x) for x in seqs], dtype=np.uint32)
    exp_p = np.array([oracle_mod.lz4f_size_pair(seqs[i], seqs[j]) for i, j in ij], dtype=np.uint32)
    assert np.array_equal(s, exp_s), np.flatnonzero(s != exp_s)
    assert np.array_equal(p, exp_p), np.flatnonzero(p != exp_p)


def test_hand_scheduled_steady_loop_equals_its_cxx_statement(hip, oracle_mod):
    """The 2-bit kernel's steady loop exists twice: hand-scheduled gfx950 code (the product) and the
    C++ statement of the same dataflow (option fast_asm=0; also what tests/test_kernel_emu.py runs
    on the CPU).  Both must equal the oracle on inputs that hit every service exit: block ends,
    seams at every phase of len(x) mod 4, long matches, long literal runs, stream starts, tandem
    repeats (owed put into the slot being read), relatives."""
    o = oracle_mod
    rng = np.random.default_rng(99)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    anc = o.lcg_genome(61, 150001)
    seqs = [o.lcg_genome(51 + k, n) for k, n in enumerate([100001, 100002, 100003, 65530, 65537, 131075, 70000, 40000])]
    seqs += [anc, o.lcg_mutant(anc, 3), _tandem(rng, 140000, 37, 0.0), _tandem(rng, 150003, 900, 0.02),
             np.frombuffer(b"A" * 100000, dtype=np.uint8), np.frombuffer(b"AC" * 50001, dtype=np.uint8),
             rng.choice(acgt, 90000, p=[0.85, 0.05, 0.05, 0.05]),
             np.repeat(rng.choice(acgt, 5000), rng.integers(1, 60, 5000))[:120000].copy()]
    exp_s = np.array([o.lz4f_size(x) for x in seqs], dtype=np.uint32)
    exp_p = np.array([[o.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
    # (round 3) ... and both exist in two forms: two lanes per chain (fast_spec=1, the default: the second lane probes 5 bases
    # ahead in the same trip and counts when the first lane's match ends there) and one lane per chain (fast_spec=0)
    # (round 4) ... and the C++ statement with THREE lanes per chain (fast_spec=3: the third lane probes 10 bases ahead and counts
    # when two 5-base matches follow each other; fast_spec=36: 6 bases ahead, counts when the first lane's match ends there) --
    # built and measured negative (profiles/r04_third_lane.json), kept exact
    for opts in ({}, {"fast_asm": 0}, {"fast_spec": 0}, {"fast_spec": 0, "fast_asm": 0}, {"fast_lanes": 3, "fast_waves": 5},
                 {"fast_spec": 3}, {"fast_spec": 36}, {"fast_spec": 3, "fast_lanes": 7, "fast_waves": 3}, {"fast_spec": 36, "fast_lanes": 4}):
        with hip.HipContext(0, **opts) as ctx:
            ctx.upload(seqs)
            assert ctx.num_packed == len(seqs)
            s, p = ctx.singles(), ctx.pairs()
        assert np.array_equal(s, exp_s), (opts, np.flatnonzero(s != exp_s))
        assert np.array_equal(p, exp_p), (opts, np.argwhere(p != exp_p)[:8].tolist())


def _with_exceptions(rng, a, runs, singles):
    a = np.array(a, dtype=np.uint8, copy=True)
    n = len(a)
    for _ in range(runs):
        s0 = int(rng.integers(0, n))
        a[s0:s0 + int(rng.integers(1, 700))] = ord("N")
    for _ in range(singles):
        a[int(rng.integers(0, n))] = rng.choice(np.frombuffer(b"NRYKMSWacgtn", dtype=np.uint8))
    return a


def test_sequences_with_a_few_exceptions_stay_on_the_2bit_kernel(hip, oracle_mod):
    """N runs and scattered IUPAC codes (ref:snacc/pairwise_ncd.py:31-36: every residue is compressed as it is) no
    longer demote a genome to the byte kernel: the few places go through the 2-bit kernel's byte-accurate
    general path (sentinel entries + overflow table).  Hand-scheduled and C++ steady loop, prefix snapshots,
    exceptions at the stream start, at a seam, across a block edge; and next to a sequence that does go to
    the byte kernel (snapshots in liblz4's layout both ways)."""
    o = oracle_mod
    rng = np.random.default_rng(17)
    g = [o.lcg_genome(200 + k, n) for k, n in enumerate([150000, 131072, 90001, 70000, 200003])]
    seqs = [_with_exceptions(rng, g[0], 2, 3), g[1], _with_exceptions(rng, g[2], 1, 0), _with_exceptions(rng, g[3], 0, 5),
            _with_exceptions(rng, g[4], 3, 6), _with_exceptions(rng, o.lcg_mutant(g[0], 9), 1, 2),
            _tandem(rng, 140000, 300, 0.01)]
    seqs[2][:40] = ord("N")
    seqs[3][-30:] = ord("N")
    seqs[4][65530:65545] = ord("N")
    seqs[6][70000:70300] = ord("N")
    heavy = _with_exceptions(rng, o.lcg_genome(210, 120000), 0, 9000)          # too many places: byte kernel
    for extra, packed in (([], 7), ([heavy], 7)):
        ss = seqs + extra
        exp_s = np.array([o.lz4f_size(x) for x in ss], dtype=np.uint32)
        exp_p = np.array([[o.lz4f_size_pair(a, b) for b in ss] for a in ss], dtype=np.uint32)
        for asm, spec in ((1, 1), (0, 1), (1, 0), (0, 0)):           # hand-scheduled / C++ statement; two lanes per chain / one
            with hip.HipContext(0, fast_asm=asm, fast_spec=spec, exc_limit=8192, fast_lanes=5, fast_waves=2) as ctx:
                ctx.upload(ss)
                assert ctx.num_packed == packed
                s, p = ctx.singles(), ctx.pairs()
            assert np.array_equal(s, exp_s), (asm, spec, np.flatnonzero(s != exp_s))
            assert np.array_equal(p, exp_p), (asm, spec, np.argwhere(p != exp_p)[:8].tolist())
    with hip.HipContext(0, exc_limit=0) as ctx:          # the option off: such sequences take the byte kernel, same sizes
        ctx.upload(seqs)
        assert ctx.num_packed == 1
        assert np.array_equal(ctx.pairs(), exp_p[:7, :7])


def test_clean_pairs_of_a_set_with_exceptions_run_on_the_pure_kernel(hip, oracle_mod):
    """(round 4) In a set where only some sequences carry exceptions the pairs of two clean sequences are the first 2-bit jobs of
    the launch's list and run on the pure kernel (option split_clean: 1 = when they fill the card 16 times, 2 = always, 0 = never);
    the others on the exception kernels.  Same
    matrix either way and as the oracle's -- with N runs / IUPAC codes only (84 chains) and with a soft-masked sequence in the set
    (83 chains, the pure kernel on that geometry), row tiles and a pair list included."""
    o = oracle_mod
    rng = np.random.default_rng(77)
    for with_soft in (False, True):
        seqs = [o.lcg_genome(300 + i, 140000 + 3001 * i) for i in range(9)]
        for i in (1, 4, 6):
            a = seqs[i].copy()
            for p0 in rng.integers(1000, a.size - 1000, 5):
                a[p0:p0 + int(rng.integers(1, 120))] = ord("N")
            a[rng.integers(0, a.size, 6)] = rng.choice(np.frombuffer(b"RYKMSW", dtype=np.uint8), 6)
            seqs[i] = a
        if with_soft:
            a = seqs[7].copy(); a[50000:50700] |= 0x20; seqs[7] = a
        exp = np.array([[o.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
        got = {}
        for split in (2, 0):
            with hip.HipContext(0, split_clean=split) as ctx:
                ctx.upload(seqs)
                assert ctx.num_packed == len(seqs)
                assert ctx.fast_chains() == (83 if with_soft else 84)
                got[split] = ctx.pairs()
                tile = ctx.pairs(2, 6)
                ij = np.array([(0, 2), (2, 0), (1, 4), (3, 5), (7, 8), (8, 7), (4, 4)], dtype=np.int32)
                lst = ctx.pairs_list(ij)
            assert np.array_equal(got[split], exp), (with_soft, split, np.argwhere(got[split] != exp)[:6].tolist())
            assert np.array_equal(tile, exp[2:6])
            assert np.array_equal(lst, np.array([exp[i, j] for i, j in ij], dtype=np.uint32))


def test_tiles_of_sets_with_exceptions_go_out_in_whole_workgroups(hip, oracle_mod):
    """(round 4) A dense tile of a set with exceptions runs as the largest multiple of the workgroup's chains, then the rest, so
    that the lanes of a wave keep sharing their suffix: 100 sequences of ~70 kbp -- 84 + 16 rows with N runs only, 83 + 17 with a
    soft-masked sequence in the set -- every size against the oracle, whole matrix and a row range."""
    o = oracle_mod
    rng = np.random.default_rng(913)
    base = [o.lcg_genome(500 + i, 66000 + 97 * i) for i in range(100)]
    for with_soft in (False, True):
        seqs = []
        for i, a in enumerate(base):
            a = a.copy()
            if i % 3 == 0:
                p0 = int(rng.integers(1000, a.size - 1000)); a[p0:p0 + int(rng.integers(1, 90))] = ord("N")
            if with_soft and i == 50:
                a[20000:20600] |= 0x20
            seqs.append(a)
        exp = np.array([[o.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
        with hip.HipContext(0) as ctx:
            ctx.upload(seqs)
            assert ctx.num_packed == 100 and ctx.fast_chains() == (83 if with_soft else 84)
            got = ctx.pairs()
            part = ctx.pairs(3, 98)
        assert np.array_equal(got, exp), (with_soft, np.argwhere(got != exp)[:6].tolist())
        assert np.array_equal(part, exp[3:98])


def test_fast_chains_reports_the_workgroup_geometry(hip):
    """snk_fast_chains: lanes x waves of a 2-bit kernel workgroup; fast_lanes = 0 (the default) takes as many chains as the
    160 KiB of LDS hold beside the slot LUT: 84 at 4 waves, and explicit settings are reported as given or refused."""
    with hip.HipContext(0) as ctx:
        assert ctx.fast_chains() == 84
    with hip.HipContext(0, fast_waves=3) as ctx:
        assert ctx.fast_chains() == 84                   # 28 x 3
    with hip.HipContext(0, fast_lanes=5, fast_waves=2) as ctx:
        assert ctx.fast_chains() == 10
    with hip.HipContext(0, fast_lanes=30, fast_waves=4) as ctx:
        with pytest.raises(hip.HipBackendError):
            ctx.fast_chains()


def test_soft_masked_genomes_stay_on_the_2bit_kernel(hip, oracle_mod):
    """Lower-case stretches (soft-masked genomes) are runs of exceptions for the 2-bit kernel: up to a quarter of a
    sequence's 16-base granules may be flagged by default.  Sizes equal the oracle's; both loops; next to a pure genome."""
    o = oracle_mod
    rng = np.random.default_rng(29)

    def soft(a, pct, run=500):
        a = a.copy()
        for s0 in rng.integers(0, len(a) - run - 200, max(1, len(a) * pct // 100 // run)):
            a[s0:s0 + int(rng.integers(run // 2, run * 3 // 2))] |= 0x20
        return a

    g = [o.lcg_genome(400 + k, n) for k, n in enumerate([300000, 200001, 131072, 150000, 262144])]
    seqs = [soft(g[0], 5), soft(g[1], 15, 300), g[2], soft(g[3], 1), soft(o.lcg_mutant(g[0], 7), 8), soft(g[4], 3)]
    seqs[1][:900] |= 0x20
    seqs[3][-500:] |= 0x20
    seqs[5][65300:66000] |= 0x20
    exp_s = np.array([o.lz4f_size(x) for x in seqs], dtype=np.uint32)
    exp_p = np.array([[o.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
    for opts in ({}, {"fast_asm": 0}, {"fast_lanes": 4, "fast_waves": 2}):
        with hip.HipContext(0, **opts) as ctx:
            ctx.upload(seqs)
            assert ctx.num_packed == len(seqs)           # all on the 2-bit kernel
            s, p = ctx.singles(), ctx.pairs()
        assert np.array_equal(s, exp_s), (opts, np.flatnonzero(s != exp_s))
        assert np.array_equal(p, exp_p), (opts, np.argwhere(p != exp_p)[:8].tolist())


def test_exceptions_at_the_bench_shape_1mbp_84_chains(hip, oracle_mod):
    """The exception machinery at the shape the kernel is benchmarked on: 128 genomes of 1 Mbp -- soft-masked (1 / 5 / 20 %
    lower case in stretches of ~500 bases, some with n runs and IUPAC codes inside the stretches), scattered IUPAC codes
    (100 and 1000 per Mbp), N runs, relatives in which the same region is masked in one genome and not in the other, and
    pure ones -- 84 rows x 128 columns (84 chains per workgroup, every lane of a wave at work, sites gathered, the steady
    loop's other-case mode, mask windows), EVERY size compared with the oracle.  Also far chains on the pure pairs'
    geometry (option far_lanes: a measured negative, kept for reproduction) must not change a size."""
    from oracle.loader import pairs_mt
    o = oracle_mod
    rng = np.random.default_rng(84)
    n, L = 128, 1_000_000
    iupac = np.frombuffer(b"RYKMSWBDHVNryn", dtype=np.uint8)
    anc = [o.lcg_genome(8400 + a, L) for a in range(4)]
    seqs = []
    for i in range(n):
        a = (o.lcg_mutant(anc[i % 4], 500 + i) if i % 3 else o.lcg_genome(8500 + i, L)).copy()
        kind = i % 8
        if kind in (1, 2, 3):                                   # soft-masked: 1 / 5 / 20 %
            pct = (1, 5, 20)[kind - 1]
            for s0 in rng.integers(0, L - 800, max(1, L * pct // 100 // 500)):
                a[s0:s0 + int(rng.integers(300, 700))] |= 0x20
            if i % 16 == kind:                                  # ... with n runs and IUPAC codes inside the stretches
                low = np.flatnonzero(a & 0x20)
                for p0 in rng.choice(low, 12):
                    a[p0:p0 + int(rng.integers(1, 25))] = ord("n")
                a[rng.choice(low, 20)] = rng.choice(iupac, 20)
        elif kind == 4:
            a[rng.integers(0, L, 100)] = rng.choice(iupac, 100)
        elif kind == 5:
            a[rng.integers(0, L, 1000)] = rng.choice(iupac, 1000)
        elif kind == 6:
            for s0 in rng.integers(0, L - 2000, 10):
                a[s0:s0 + int(rng.integers(10, 1500))] = ord("N")
        seqs.append(a)
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        assert ctx.num_packed == n and ctx.fast_chains() == 83      # (a set with exceptions: the other case's LUT takes the 84th chain's room)
        s, p = ctx.singles(), ctx.pairs(0, 84)
    assert np.array_equal(s[:4], np.array([o.lz4f_size(x) for x in seqs[:4]], dtype=np.uint32))
    want = pairs_mt(seqs, 0, 84, _threads())
    assert np.array_equal(p, want), np.argwhere(p != want)[:8].tolist()


def test_far_chains_option_changes_no_size(hip, oracle_mod):
    """far_lanes / far_waves (extra waves whose chains keep their tables in global memory: profiles/r03_far_chains.json, a
    measured negative, off by default): 300 pure genomes of 100 kbp (90 000 pairs: a launch large enough for the far waves to
    be used), every size equal to the oracle's with 8 x 4 far chains."""
    from oracle.loader import pairs_mt
    o = oracle_mod
    seqs = [o.lcg_genome(1 + i, 100_000) for i in range(300)]
    with hip.HipContext(0, far_lanes=8, far_waves=4) as ctx:
        ctx.upload(seqs)
        p = ctx.pairs()
    assert np.array_equal(p, pairs_mt(seqs, 0, 300, _threads()))


def test_lower_case_sets_run_on_the_2bit_kernel(hip, oracle_mod):
    """A set whose letters are acgt runs on the 2-bit kernel like an upper-case one (its LUTs are made from liblz4's
    hashes of the lower-case 5-mers; upper-case stretches are then the exceptions); the case goes by the set's majority,
    and a context that has served one case serves the other after the next upload."""
    o = oracle_mod
    rng = np.random.default_rng(31)
    g = [o.lcg_genome(500 + k, n) for k, n in enumerate([200000, 131072, 90001, 262144, 70000])]
    lower = [np.frombuffer(bytes(x).lower(), dtype=np.uint8).copy() for x in g]
    lower[1][5000:5600] &= 0xDF                           # an upper-case stretch in a lower-case genome
    lower[3][70000:70050] = ord("n")
    lower[3][131000:131300] &= 0xDF
    upper = [x.copy() for x in g[:3]]
    with hip.HipContext(0, fast_lanes=4, fast_waves=2) as ctx:
        for seqs in (lower, upper, lower[:2] + [g[4]]):   # (the last: two lower-case genomes and an upper-case one: majority lower)
            exp_s = np.array([o.lz4f_size(x) for x in seqs], dtype=np.uint32)
            exp_p = np.array([[o.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
            ctx.upload(seqs)
            s, p = ctx.singles(), ctx.pairs()
            assert np.array_equal(s, exp_s), np.flatnonzero(s != exp_s)
            assert np.array_equal(p, exp_p), np.argwhere(p != exp_p)[:8].tolist()
    with hip.HipContext(0) as ctx:
        ctx.upload(lower)
        assert ctx.num_packed == len(lower)               # all five on the 2-bit kernel
        ctx.upload(lower[:2] + [g[4]])
        assert ctx.num_packed == 3                        # (round 4) the upper-case genome of a lower-case set stays on the 2-bit kernel too: one long
                                                          # stretch of the other case, walked in the steady loop's other-case mode


def test_related_genomes_same_ancestor(hip, oracle_mod):
    o = oracle_mod
    anc = o.lcg_genome(40, 180000)
    seqs = [anc] + [o.lcg_mutant(anc, 50 + i) for i in range(4)] + [anc[::-1].copy()]
    _check_all(hip, oracle_mod, seqs)


def test_golden_frame_sizes_through_cabi(hip, golden, oracle_mod):
    """The liblz4 1.9.3 sizes recorded in golden.json, computed by the GPU."""
    o = oracle_mod
    g = golden["liblz4_frame_sizes"]
    row = g["lcg_seed1_2_mut3"][1]                     # n = 100 000
    x, y = o.lcg_genome(1, row["n"]), o.lcg_genome(2, row["n"])
    z = o.lcg_mutant(x, 3)
    with hip.HipContext(0) as ctx:
        ctx.upload([x, y, z])
        assert ctx.singles().tolist() == [row["x"], row["y"], row["z"]]
        got = ctx.pairs_list([(0, 1), (1, 0), (0, 0), (0, 2), (2, 0)]).tolist()
        assert got == [row["xy"], row["yx"], row["xx"], row["xz"], row["zx"]]
        rag = g["lcg_ragged"]
        ctx.upload([o.lcg_genome(r["seed"], r["n"]) for r in rag])
        assert ctx.singles().tolist() == [r["size"] for r in rag]
        oth = g["other_alphabets"]
        ctx.upload([lcg_bytes(r["seed"], r["n"], bytes.fromhex(r["alphabet_hex"])) for r in oth])
        assert ctx.singles().tolist() == [r["size"] for r in oth]


def test_content_size_option_adds_eight_bytes(hip, oracle_mod):
    seqs = [oracle_mod.lcg_genome(3, 70000), b"ACGT" * 10]
    with hip.HipContext(0) as a, hip.HipContext(0, content_size=1) as b:
        a.upload(seqs)
        b.upload(seqs)
        assert np.array_equal(a.singles() + 8, b.singles())
        assert np.array_equal(a.pairs() + 8, b.pairs())


def test_row_tiles_and_pair_lists_agree_with_full_matrix(hip, oracle_mod):
    seqs = [oracle_mod.lcg_genome(60 + i, 70000 + 1111 * i) for i in range(9)]
    with hip.HipContext(0, fast_lanes=3, fast_waves=2) as ctx:
        ctx.upload(seqs)
        full = ctx.pairs()
        tiles = np.concatenate([ctx.pairs(0, 2), ctx.pairs(2, 7), ctx.pairs(7, 9)])
        assert np.array_equal(full, tiles)
        ij = [(i, j) for i in range(9) for j in range(9)][::-1]
        assert np.array_equal(ctx.pairs_list(ij), full[::-1, ::-1].reshape(-1))
    exp = np.array([[oracle_mod.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
    assert np.array_equal(full, exp)


def test_interleaved_async_launches_and_schedules(hip, oracle_mod):
    """snk_pairs_device from two streams, interleaved, on a dense set (no job list on the device) and
    on a mixed set (the job list buffer is shared per context: the second launch must wait for the
    first one's copy); static round-robin and atomic-queue hand-out of batches; per-launch times."""
    import torch
    o = oracle_mod
    dense = [o.lcg_genome(90 + k, 70000 + 31000 * k) for k in range(7)]
    mixed = dense[:5] + [np.frombuffer(bytes(o.lcg_genome(99, 90000)).replace(b"ACGTA", b"ACNTA"), dtype=np.uint8),
                         o.lcg_genome(98, 30000)]
    dev = torch.device("cuda", 0)
    for seqs in (dense, mixed):
        n = len(seqs)
        exp = np.array([[o.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)
        for dyn in (0, 1):
            with hip.HipContext(0, fast_dynamic=dyn, fast_lanes=3, fast_waves=2) as ctx:
                ctx.upload(seqs)
                ctx.pairs_ms_log()                       # the per-launch log starts at the first call
                s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
                t1 = torch.zeros((4, n), dtype=torch.int32, device=dev)
                t2 = torch.zeros((n - 4, n), dtype=torch.int32, device=dev)
                t3 = torch.zeros((2, n), dtype=torch.int32, device=dev)
                ctx.pairs_device(0, 4, t1.data_ptr(), s1.cuda_stream)
                ctx.pairs_device(4, n, t2.data_ptr(), s2.cuda_stream)
                ctx.pairs_device(1, 3, t3.data_ptr(), s1.cuda_stream)
                ctx.sync(s1.cuda_stream)
                ctx.sync(s2.cuda_stream)
                ms = ctx.pairs_ms_log()
                assert len(ms) == 3 and all(v > 0 for v in ms)
                assert ctx.pairs_ms_log() == []
                got = np.concatenate([t1.cpu().numpy(), t2.cpu().numpy()]).view(np.uint32)
                assert np.array_equal(got, exp), (dyn, np.argwhere(got != exp)[:6].tolist())
                assert np.array_equal(t3.cpu().numpy().view(np.uint32), exp[1:3])


def test_sets_beyond_one_upload_are_processed_in_blocks(hip, oracle_mod, tmp_path, monkeypatch):
    """The device arenas use 32-bit offsets (about 4.29 GB of residues per upload).  A larger set -- here
    forced by a small arena_limit -- goes through snacc_amd.cli.blocked_sizes: groups of files, every
    pair of groups uploaded once.  The CSV equals the one-upload run's, for lz4 and gzip."""
    from click.testing import CliRunner
    from conftest import write_fasta
    from snacc_amd import cli as cli_mod
    d = tmp_path / "fa"
    d.mkdir()
    for k in range(7):
        write_fasta(d / f"g{k}.fasta", [("r", bytes(oracle_mod.lcg_genome(120 + k, 60000 + 17000 * k)).decode())])
    monkeypatch.chdir(tmp_path)
    with hip.HipContext(0, arena_limit=200000) as ctx:
        with pytest.raises(hip.ArenaTooBig):
            ctx.upload_fasta(sorted(d.iterdir()))
    for codec in ("lz4", "gzip"):
        texts = []
        for limit in (None, "420000"):
            if limit:
                monkeypatch.setenv("SNACC_ARENA_LIMIT", limit)
            else:
                monkeypatch.delenv("SNACC_ARENA_LIMIT", raising=False)
            out = tmp_path / f"{codec}_{limit}.csv"
            res = CliRunner().invoke(cli_mod.cli, [str(d), "-o", str(out), "-c", codec, "--no-show-progress", "--no-log"])
            assert res.exit_code == 0, (res.output, res.exception)
            texts.append(out.read_text())
        assert texts[0] == texts[1], codec
    monkeypatch.delenv("SNACC_ARENA_LIMIT", raising=False)


def test_python_api_single_items(hip, golden, tmp_path):
    """compressed_size(path | (path, path), "lz4") -- the reference's granularity."""
    from snacc_amd import compressed_size
    p = tmp_path / "sample_crlf.fa"
    with open(p, "w", newline="") as f:
        f.write(">derice\r\nACTGACTAGCTAGCTAACTG\r\n>sanka\r\nGCATCGTAGCTAGCTACGAT\r\n"
                ">junior\r\nCATCGATCGTACGTACGTAG\r\n>yul\r\nATCGATCGATCGTACGATCG")
    g = golden["sample_fa"]
    assert compressed_size(p, "lz4") == (p, g["sizes_single"]["lz4"])
    assert compressed_size((p, p), "lz4") == ((p, p), g["sizes_selfpair"]["lz4"])
    assert compressed_size(p, "lz4", reverse_complement=True)[1] == g["sizes_single_rc_lz4"]


def test_sequence_level_ncd_wrapper(hip, oracle_mod):
    """north_star's ``compute_distance(seq_i, seq_j, compressor)`` convenience: ``snacc_amd.ncd`` on two extracted
    sequences = the reference's four-integer formula (ref:snacc/pairwise_ncd.py:93-111) on the sizes of ref:...:69-90."""
    from snacc_amd import compute_distance, ncd
    o = oracle_mod
    x, y = o.lcg_genome(61, 90_000), o.lcg_mutant(o.lcg_genome(61, 90_000), 5)
    sx, sy = o.lz4f_size(x) + 33, o.lz4f_size(y) + 33
    want = compute_distance(sx, sy, o.lz4f_size_pair(x, y) + 33, o.lz4f_size_pair(y, x) + 33)
    assert ncd(bytes(x), bytes(y), "lz4") == want
    assert ncd(bytes(x).decode(), bytes(y).decode()) == want            # str or bytes; lz4 is the default here
    import gzip
    gx, gy = bytes(x[:5000]), bytes(y[:5000])
    c = lambda b: len(gzip.compress(b)) + 33                            # noqa: E731
    assert ncd(gx, gy, "gzip") == compute_distance(c(gx), c(gy), c(gx + gy), c(gy + gx))


def test_python_api_from_a_thread_pool(hip, oracle_mod, tmp_path):
    """The reference's callers run compressed_size from a ThreadPoolExecutor (ref:snacc/cli.py:104-129):
    8 threads on the shared device context must each get the sizes of THEIR sequences."""
    import concurrent.futures
    from conftest import write_fasta
    from snacc_amd import compressed_size
    seqs, files = [], []
    for k in range(8):
        sq = bytes(oracle_mod.lcg_genome(70 + k, 30000 + 9000 * k))
        f = tmp_path / f"t{k}.fa"
        write_fasta(f, [("r", sq.decode())])
        seqs.append(sq)
        files.append(f)
    keys = [f for f in files] + [(files[i], files[(3 * i + 1) % 8]) for i in range(8)] * 2
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        got = dict(ex.map(lambda k: compressed_size(k, "lz4"), keys))
    for k in keys:
        if type(k) == tuple:
            exp = oracle_mod.lz4f_size_pair(seqs[files.index(k[0])], seqs[files.index(k[1])])
        else:
            exp = oracle_mod.lz4f_size(seqs[files.index(k)])
        assert got[k] == exp + 33, k


@pytest.mark.parametrize("set_name", ["acgt_small", "ragged_blocks"])
@pytest.mark.parametrize("rc", [False, True])
def test_cli_lz4_csv_equals_reference_cli(hip, golden, oracle_mod, tmp_path, monkeypatch, set_name, rc):
    """`snacc <dir> -c lz4 [-r]` end to end on the GPU == the CSV the reference CLI wrote."""
    from click.testing import CliRunner
    from snacc_amd.cli import cli
    spec = golden["cli_lz4"]["sets"][set_name]
    d = materialise_cli_set(oracle_mod, spec, tmp_path / "fa")
    out = tmp_path / "out.csv"
    monkeypatch.chdir(tmp_path)
    args = [str(d), "-o", str(out), "-c", "lz4", "-n", "1", "--no-show-progress"] + (["-r"] if rc else [])
    res = CliRunner().invoke(cli, args)
    assert res.exit_code == 0, res.output
    want = golden["cli_lz4"]["outputs"][set_name]["csv_rc" if rc else "csv"].replace("{DIR}", str(d))
    assert out.read_text() == want
    log = (tmp_path / "out.md").read_text()
    assert "* Compression method: lz4" in log and f"* Reverse complement: {rc}" in log


def test_full_size_1mbp_properties(hip, golden, oracle_mod):
    """BASELINE.json size (1 Mbp genomes): oracle on a sample of pairs, plus size-independent
    properties over ALL pairs: the 2-bit kernel and the byte kernel (independent code) agree,
    results do not depend on the launch shape, and the NCD matrix is exactly symmetric."""
    o = oracle_mod
    n, L = 12, 1_000_000
    seqs = [o.lcg_genome(1 + i, L) for i in range(n)]
    seqs[5] = o.lcg_mutant(seqs[0], 3)
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        s = ctx.singles()
        p = ctx.pairs()
    with hip.HipContext(0, fast_lanes=10, fast_waves=8) as ctx:
        ctx.upload(seqs)
        assert np.array_equal(ctx.pairs(), p)
    with hip.HipContext(0, force_generic=1) as ctx:
        ctx.upload(seqs)
        assert np.array_equal(ctx.singles(), s)
        assert np.array_equal(ctx.pairs(0, 3), p[:3])           # byte kernel == 2-bit kernel
    with hip.HipContext(0, force_generic=1, bytes_legacy=1) as ctx:
        ctx.upload(seqs[:4])
        assert np.array_equal(ctx.pairs(0, 1), p[:1, :4])       # legacy byte kernel too
    row = golden["liblz4_frame_sizes"]["lcg_seed1_2_mut3"][2]
    assert (int(s[0]), int(s[1]), int(s[5])) == (row["x"], row["y"], row["z"])
    assert (int(p[0, 1]), int(p[1, 0]), int(p[0, 0]), int(p[0, 5]), int(p[5, 0])) == \
           (row["xy"], row["yx"], row["xx"], row["xz"], row["zx"])
    for i, j in [(2, 3), (7, 2), (11, 11), (4, 9), (9, 4), (6, 0)]:
        assert int(p[i, j]) == o.lz4f_size_pair(seqs[i], seqs[j])
    from snacc_amd.matrix import ncd_matrix
    from snacc_amd.pairwise_ncd import compute_distance
    m = ncd_matrix(s.astype(np.int64) + 33, p.astype(np.int64) + 33)
    assert np.array_equal(m, m.T)
    assert m[0, 1] == 0.999350356687629 and m[0, 0] == 1.000420899139182
    assert abs(m[3, 8] - compute_distance(int(s[3]) + 33, int(s[8]) + 33, int(p[3, 8]) + 33, int(p[8, 3]) + 33)) <= 1e-6


def _threads():
    import os
    return max(1, min(len(os.sched_getaffinity(0)), 32))


def test_config2_full_matrix_256x100kbp(hip, oracle_mod):
    """BASELINE.json configs[1]: all 65 536 ordered pairs + 256 singles of 256 x 100 kbp genomes,
    every size compared with the oracle."""
    from oracle.loader import pairs_mt
    n, L = 256, 100_000
    seqs = [oracle_mod.lcg_genome(1 + i, L) for i in range(n)]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        s, p = ctx.singles(), ctx.pairs()
    assert np.array_equal(s, np.array([oracle_mod.lz4f_size(x) for x in seqs], dtype=np.uint32))
    assert np.array_equal(p, pairs_mt(seqs, 0, n, _threads()))


def test_config3_sample_1024x1mbp(hip, oracle_mod):
    """BASELINE.json configs[2] at full size: the whole 1024 x 1024 matrix on the GPU; first row,
    first column, diagonal and a >= 1 % uniform sample of the pairs checked against the oracle."""
    from oracle.loader import pairs_list_mt
    n, L = 1024, 1_000_000
    seqs = [oracle_mod.lcg_genome(1 + i, L) for i in range(n)]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        s, p = ctx.singles(), ctx.pairs()
    rng = np.random.default_rng(2026)
    idx = np.arange(n)
    ij = np.concatenate([np.stack([np.zeros(n, int), idx], 1), np.stack([idx, np.zeros(n, int)], 1),
                         np.stack([idx, idx], 1), rng.integers(0, n, (10600, 2))])
    want = pairs_list_mt(seqs, ij, _threads())
    assert np.array_equal(p[ij[:, 0], ij[:, 1]], want)
    for g in (0, 1, 511, 1023):
        assert int(s[g]) == oracle_mod.lz4f_size(seqs[g])
    from snacc_amd.matrix import ncd_matrix
    m = ncd_matrix(s.astype(np.int64) + 33, p.astype(np.int64) + 33)
    assert np.array_equal(m, m.T) and m[0, 1] == 0.999350356687629


def test_config5_substitute_5mbp_reverse_complement_cli(hip, oracle_mod, tmp_path, monkeypatch):
    """BASELINE.json configs[4] names 92 E. coli genomes that are not in the reference;
    substitute: 10 synthetic ~5 Mbp multi-record genomes (2 % mutants of two ancestors), `-c lz4 -r`
    through the CLI, every matrix cell compared with the oracle-derived NCD."""
    from click.testing import CliRunner
    from conftest import write_fasta
    from oracle.loader import pairs_mt
    from snacc_amd import fasta
    from snacc_amd.cli import cli
    from snacc_amd.matrix import ncd_matrix
    o = oracle_mod
    anc = [o.lcg_genome(900 + a, 5_000_000 + 12345 * a) for a in range(2)]
    d = tmp_path / "fa"
    d.mkdir()
    seqs_rc = []
    for i in range(10):
        g = bytes(anc[i % 2] if i < 2 else o.lcg_mutant(anc[i % 2], 1000 + i)).decode()
        cut = [0, len(g) // 3 + 17 * i, 2 * len(g) // 3, len(g)]                 # three records per file
        recs = [(f"g{i}_c{k}", g[cut[k]:cut[k + 1]]) for k in range(3)]
        write_fasta(d / f"g{i:02d}.fna", recs, width=70)
        seqs_rc.append("".join(fasta.reverse_complement(r) for _, r in recs).encode())
    out = tmp_path / "out.csv"
    monkeypatch.chdir(tmp_path)
    res = CliRunner().invoke(cli, [str(d), "-o", str(out), "-c", "lz4", "-r", "--no-show-progress", "--no-log"])
    assert res.exit_code == 0, res.output
    got = np.loadtxt(out, delimiter=",", skiprows=1, usecols=range(1, 11))
    singles = np.array([o.lz4f_size(s_) for s_ in seqs_rc], dtype=np.int64) + 33
    pairs = pairs_mt([np.frombuffer(s_, dtype=np.uint8) for s_ in seqs_rc], 0, 10, _threads()).astype(np.int64) + 33
    want = ncd_matrix(singles, pairs)
    assert np.array_equal(got, want)                # repr round-trips: exact, well inside the 1e-6 of north_star


def test_config5_at_its_own_size_92_genomes_reverse_complement_cli(hip, oracle_mod, tmp_path, monkeypatch):
    """BASELINE.json configs[4] at its real size on one GPU: 92 genomes of ~5 Mbp (tests/ecoli_like.py: 2 % mutants
    of four ancestors, 1 .. 4 records per file, N runs, IUPAC codes, one soft-masked stretch), `snacc -c lz4 -r`
    through the CLI (native FASTA ingest, per-record reverse complement, one upload, 92 x 92 ordered pairs on the
    GPU).  First row, first column, the diagonal and a sample of > 5 % of the ordered pairs are compared with the
    oracle, as NCD values (exact: computed from equal integers) -- ref:snacc/pairwise_ncd.py:29-36, ref:snacc/cli.py:120-136."""
    import ecoli_like as ec
    from click.testing import CliRunner
    from oracle.loader import pairs_list_mt
    from snacc_amd.cli import cli
    from snacc_amd.pairwise_ncd import compute_distance
    o = oracle_mod
    n = ec.N_GENOMES
    anc = [o.lcg_genome(7000 + a, ec.ancestor_length(a)) for a in range(ec.N_ANCESTORS)]
    d = tmp_path / "st131"
    d.mkdir()
    seqs = []
    for i in range(n):
        recs = ec.records(i, ec.make_genome(o, i, ancestors=anc))
        ec.write_fasta_fast(d / f"g{i:02d}.fna", recs)
        seqs.append(ec.expected_sequence(recs, reverse_complement=True))
    assert [len(s) for s in seqs] == ec.lengths() and sum(len(s) for s in seqs) > 440_000_000
    out = tmp_path / "out.csv"
    monkeypatch.chdir(tmp_path)
    res = CliRunner().invoke(cli, [str(d), "-o", str(out), "-c", "lz4", "-r", "--no-show-progress", "--no-log"])
    assert res.exit_code == 0, res.output
    got = np.loadtxt(out, delimiter=",", skiprows=1, usecols=range(1, n + 1))
    assert got.shape == (n, n) and np.array_equal(got, got.T)
    rng = np.random.default_rng(92)
    cells = {(0, j) for j in range(n)} | {(i, i) for i in range(n)}
    while len(cells) < 2 * n - 1 + 230:                         # + 230 cells = 460 ordered pairs > 5 % of 8464
        i, j = (int(v) for v in rng.integers(0, n, 2))
        cells.add((min(i, j), max(i, j)))
    cells = sorted(cells)
    ij = np.array([(i, j) for i, j in cells] + [(j, i) for i, j in cells], dtype=np.int32)
    sz = pairs_list_mt(seqs, ij, _threads()).astype(np.int64) + 33
    single = {g: o.lz4f_size(seqs[g]) + 33 for g in {int(v) for v in ij.ravel()}}
    for k, (i, j) in enumerate(cells):
        want = compute_distance(single[i], single[j], int(sz[k]), int(sz[len(cells) + k]))
        assert got[i, j] == want, (i, j, got[i, j], want)


def test_edge_cases_single_sequence_and_duplicates(hip, oracle_mod):
    o = oracle_mod
    g = o.lcg_genome(5, 150000)
    with hip.HipContext(0) as ctx:
        ctx.upload([g])
        assert ctx.pairs().tolist() == [[o.lz4f_size_pair(g, g)]]
        ctx.upload([g, g.copy(), g.copy()])
        p = ctx.pairs()
        assert (p == p[0, 0]).all() and int(p[0, 0]) == o.lz4f_size_pair(g, g)
        ctx.upload([])
        assert ctx.singles().size == 0 and ctx.pairs().size == 0


def test_long_genomes_large_offsets(hip, oracle_mod):
    """12.5 Mbp genomes: stream positions beyond 2^24, 380 blocks per pair, arena offsets in the
    MB range -- on the 2-bit kernel (one genome with an N: an exception), and with that genome on the (compact) byte kernel."""
    o = oracle_mod
    a, b = o.lcg_genome(301, 12_500_000), o.lcg_genome(302, 12_345_678)
    c = b.copy()
    c[7_000_000] = ord("N")
    seqs = [a, b, c]
    exp_s = [o.lz4f_size(x) for x in seqs]
    exp_p = [[o.lz4f_size_pair(x, y) for y in seqs] for x in seqs]
    # default: the one N is an exception on the 2-bit kernel; exc_limit=0: that genome takes the (compact) byte kernel
    for opts, packed in (({}, 3), ({"exc_limit": 0}, 2)):
        with hip.HipContext(0, **opts) as ctx:
            ctx.upload(seqs)
            assert ctx.num_packed == packed
            s, p = ctx.singles(), ctx.pairs()
        assert s.tolist() == exp_s
        assert p.tolist() == exp_p


def test_emitted_frames_bytes_and_roundtrip(hip, golden, oracle_mod, tmp_path):
    """the GPU emits the LZ4 frames themselves.  Bytes equal the liblz4 1.9.3 binary's
    (when present), the golden frame strings, and every frame decodes back to its input."""
    from lz4_decode import decode_frame
    from oracle import liblz4_ref
    o = oracle_mod
    rng = np.random.default_rng(21)
    rep = np.tile(o.lcg_genome(32, 900), 200)
    seqs = [b"", b"ACGT" * 10, bytes(o.lcg_genome(1, 100000)), bytes(o.lcg_genome(2, 70001)), bytes(rep),
            bytes(rng.integers(0, 256, 90000, dtype=np.uint8)), b"N" * 40000 + bytes(o.lcg_genome(3, 50000)),
            bytes(o.lcg_genome(4, 30000)), b"A" * 70000, b"ACGTACGTACGTA"]
    n = len(seqs)
    items = [(i, -1) for i in range(n)] + [(i, j) for i in range(n) for j in range(n)]
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        frames = ctx.frames(items)
        singles, pairs = ctx.singles(), ctx.pairs()
    fr = golden["liblz4_frame_sizes"]["frames_hex"]
    assert frames[0].hex() == fr["empty"] and frames[1].hex() == fr["ACGTx10"]
    for (i, j), f in zip(items, frames):
        data = seqs[i] + (seqs[j] if j >= 0 else b"")
        assert len(f) == (int(singles[i]) if j < 0 else int(pairs[i, j]))          # the size kernels agree
        assert decode_frame(f) == data, (i, j)
        if liblz4_ref.available():
            assert f == liblz4_ref.compress_frame(data), (i, j)


def test_emitted_frames_equal_liblz4_by_hash(hip):
    """Frame BYTES pinned on any box: sha256 + length of what `snk_frames_list` emits against the
    hashes of the liblz4 1.9.3 binary's frames for the same generator-defined inputs
    (tests/golden/frame_hashes.json, made by tests/golden/make_frame_hashes.py) -- with the
    NULL-preferences header and with the content-size field (option content_size /
    SNACC_LZ4_CONTENT_SIZE=1).  ref:snacc/pairwise_ncd.py:80-88."""
    import hashlib
    import json
    from pathlib import Path
    from frame_items import ITEMS, build_sequences
    gold = json.loads((Path(__file__).parent / "golden" / "frame_hashes.json").read_text())
    seqs = build_sequences()
    assert [tuple(f["item"]) for f in gold["frames"]] == ITEMS
    for cs in (0, 1):
        with hip.HipContext(0, content_size=cs) as ctx:
            ctx.upload(seqs)
            frames = ctx.frames(ITEMS)
            singles, pairs = ctx.singles(), ctx.pairs()
        for f, g in zip(frames, gold["frames"]):
            i, j = g["item"]
            assert len(f) == (g["len_content_size"] if cs else g["len"]), (cs, i, j)
            assert hashlib.sha256(f).hexdigest() == (g["sha256_content_size"] if cs else g["sha256"]), (cs, i, j)
            assert len(f) == (int(singles[i]) if j < 0 else int(pairs[i, j])), (cs, i, j)      # the size kernels agree


def test_content_size_from_environment_and_cli(hip, oracle_mod, tmp_path, monkeypatch):
    """SNACC_LZ4_CONTENT_SIZE=1 / --lz4-content-size: every lz4 size grows by 8 (non-empty input)."""
    from click.testing import CliRunner
    from conftest import write_fasta
    from snacc_amd import cli as cli_mod
    seqs = [bytes(oracle_mod.lcg_genome(81 + k, 20000 + 30000 * k)) for k in range(3)]
    monkeypatch.setenv("SNACC_LZ4_CONTENT_SIZE", "1")
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        s1, p1 = ctx.singles(), ctx.pairs()
    monkeypatch.setenv("SNACC_LZ4_CONTENT_SIZE", "0")
    with hip.HipContext(0) as ctx:
        ctx.upload(seqs)
        s0, p0 = ctx.singles(), ctx.pairs()
    assert np.array_equal(s1, s0 + 8) and np.array_equal(p1, p0 + 8)
    d = tmp_path / "fa"
    d.mkdir()
    for k, sq in enumerate(seqs):
        write_fasta(d / f"g{k}.fasta", [("r", sq.decode())])
    monkeypatch.chdir(tmp_path)
    outs = []
    for flag in ("--no-lz4-content-size", "--lz4-content-size"):
        out = tmp_path / (flag.strip("-") + ".csv")
        res = CliRunner().invoke(cli_mod.cli, [str(d), "-o", str(out), "-c", "lz4", "--no-show-progress", "--no-log", flag])
        assert res.exit_code == 0, res.output
        outs.append(np.loadtxt(out, delimiter=",", skiprows=1, usecols=(1, 2, 3)))
    from snacc_amd.matrix import ncd_matrix
    for m, (s, p) in zip(outs, ((s0, p0), (s1, p1))):
        assert np.array_equal(m, ncd_matrix(s.astype(np.int64) + 33, p.astype(np.int64) + 33))


def test_save_compression_lz4_cli_and_api(hip, oracle_mod, tmp_path, monkeypatch):
    """`-s` with `-c lz4`: one .lz4 blob per file and per ordered pair, named as the reference names
    them, each a valid frame of the right content; compressed_size(..., save_directory=...) likewise."""
    from click.testing import CliRunner
    from conftest import write_fasta
    from lz4_decode import decode_frame
    from snacc_amd import compressed_size
    from snacc_amd.cli import cli
    o = oracle_mod
    d = tmp_path / "fa"
    d.mkdir()
    seqs = {}
    for i, nbases in enumerate((30000, 70000, 5000)):
        g = bytes(o.lcg_genome(80 + i, nbases)).decode()
        write_fasta(d / f"s{i}.fa", [(f"s{i}", g)])
        seqs[f"s{i}"] = g.encode()
    blobs = tmp_path / "blobs"
    blobs.mkdir()
    monkeypatch.chdir(tmp_path)
    res = CliRunner().invoke(cli, [str(d), "-o", "o.csv", "-c", "lz4", "-s", str(blobs), "--no-show-progress", "--no-log"])
    assert res.exit_code == 0, res.output
    names = sorted(p.name for p in blobs.iterdir())
    want = sorted([f"s{i}.fa.lz4" for i in range(3)] + [f"s{i}s{j}.fa.lz4" for i in range(3) for j in range(3)])
    assert names == want
    assert decode_frame((blobs / "s1.fa.lz4").read_bytes()) == seqs["s1"]
    assert decode_frame((blobs / "s0s1.fa.lz4").read_bytes()) == seqs["s0"] + seqs["s1"]
    key, size = compressed_size((d / "s2.fa", d / "s0.fa"), "lz4", save_directory=blobs)
    blob = (blobs / "s2s0.fa.lz4").read_bytes()
    assert size == len(blob) + 33 == o.lz4f_size_pair(seqs["s2"], seqs["s0"]) + 33
    assert decode_frame(blob) == seqs["s2"] + seqs["s0"]


def test_rccl_single_rank_smoke(hip, oracle_mod, tmp_path):
    """The only RCCL that one GPU allows: a process group of ONE rank on the nccl backend, through the
    product's own rendezvous, asynchronous all-gather of a device tile the kernel has just written, wait,
    allgather_check and barrier (snacc_amd/distributed.py).  It proves the library is there and that the
    calls, dtypes and stream ordering are accepted; it says nothing about scaling."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29731', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')\n"
        "import torch, torch.distributed as dist\n"
        "from snacc_amd import distributed as sd\n"
        "dev = torch.device('cuda', 0)\n"
        "sd.init_process_group('nccl', dev)\n"
        "torch.cuda.set_device(0)\n"
        "import oracle\n"
        "from snacc_amd.hip_backend import HipContext\n"
        "seqs = [oracle.lcg_genome(300 + k, 70000 + 9000 * k) for k in range(5)]\n"
        "ctx = HipContext(0); ctx.upload(seqs)\n"
        "tile = torch.zeros((5, 5), dtype=torch.int32, device=dev)\n"
        "torch.cuda.current_stream(dev).synchronize()\n"
        "ctx.pairs_device(0, 5, tile.data_ptr(), None); ctx.sync(None)\n"
        "g, work = sd.gather_tile(tile, 1, async_op=True, force_collective=True)\n"
        "work.wait(); torch.cuda.synchronize()\n"
        "assert sd.allgather_check(g, tile, 0, 1)\n"
        "exp = np.array([[oracle.lz4f_size_pair(a, b) for b in seqs] for a in seqs], dtype=np.uint32)\n"
        "assert np.array_equal(g.cpu().numpy().view(np.uint32), exp)\n"
        "dist.barrier(); dist.destroy_process_group(); ctx.close(); print('rccl-ok')\n" % str(tmp_path.parent.parent)
    )
    from pathlib import Path
    root = str(Path(__file__).resolve().parents[1])
    code = code.replace(repr(str(tmp_path.parent.parent)), repr(root))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root)
    assert res.returncode == 0 and "rccl-ok" in res.stdout, (res.stdout[-2000:], res.stderr[-3000:])


@pytest.mark.parametrize("seed", [424243, 424245, 424248])       # a pure-ACGT set, an "anything" set, a soft-masked set
def test_fuzz_seeds_through_every_kernel_configuration(hip, oracle_mod, seed):
    """Three seeds of the randomised stress run (tools/gpu_fuzz.py: ragged lengths around the block edges, N runs, soft-masked
    stretches, proteins, random bytes, tandem repeats, relatives with indels) through all 19 kernel configurations -- every
    kernel family, one / two / three lanes per chain, hand-scheduled loops and their C++ statements, tables in LDS and in
    global memory, phase A on demand -- all singles and all ordered pairs against the oracle.  (Thousands of further seeds
    run outside the suite every round; these three keep the harness itself under the driver's eyes.)"""
    from fuzzgen_lz4 import CONFIGS, one
    profile, n, lens, info, bad = one(seed)
    assert len(CONFIGS) == 19 and not bad, (profile, n, lens, info, bad)


def test_the_drivers_smoke_entry_point(hip):
    """`__graft_entry__.smoke()` is what the driver runs on the card before the bench; its expectations (which kernel each of its
    sequences runs on) follow the admission rule, so it is part of the suite (round 4: it was not, and the rule changed under it)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(root, "__graft_entry__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.smoke()


def test_bench_line_keeps_the_driver_contract():
    """`python bench.py` prints ONE JSON line with the keys the driver and the judge read (metric, value, unit, n_gpus, steps,
    warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline, cpu_baseline); a
    small run of it, as a child process."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--genomes", "128", "--steps", "2", "--warmup", "1",
                          "--cpu-seconds", "1", "--no-cli-wall"], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "u8" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["parity_spot_check"] is True
    # the measured full matrix with its serial share, and the strong-scaling figure beside the weak one
    m = d["matrix"]
    assert m["symmetric"] is True and 0 < m["fixed_s"] < m["matrix_wall_s"] and m["upload_stages_s"]["total"] > 0
    assert abs(m["fixed_s"] - (m["upload_and_singles_s"] + m["ncd_assembly_s"])) < 1e-9
    assert d["strong"]["n_gpus"] == 1 and d["strong"]["pair_compressions_per_s"] > 0
    # the secondary legs run under the same clock, each with a true spot check against the oracle
    sec = d["secondary"]
    assert sorted(sec) == ["gzip", "markov", "related", "softmask5", "zlib"], sec
    for name, leg in sec.items():
        assert "error" not in leg, (name, leg)
        assert leg["parity_spot_check"] is True and leg["pair_compressions_per_s"] > 0 and leg["kernel_ms_avg"] > 0, (name, leg)
        assert abs(leg["roofline"]["frac"] - leg["roofline"]["achieved"] / 8000.0) < 1e-12 and "traffic" in leg["roofline"]
