"""Stand-in for BASELINE.json configs[4] ("92 E. coli ST131 genomes, lz4 + --reverse-complement").

The 92 FASTA files are not in the reference (its publication_data holds a 10 x 10 lzma matrix only, SURVEY.md
8d), so the set is synthetic but shaped like a collection of assemblies of one sequence type: 92 genomes of
4.4 .. 5.3 Mbp, 2 % point mutants of four ancestors, cut into 1 .. 4 records (chromosome + plasmids / contigs),
some with runs of N between contigs, some with scattered IUPAC ambiguity codes, one with a soft-masked stretch.
Everything is a function of the genome's index: the GPU test (tests/test_gpu_parity.py) builds the set at full
size, the gloo test (tests/test_distributed_gloo.py) uses its 92 lengths for the row weights and the set at
1/1000 scale for the sizes.
"""
import numpy as np

N_GENOMES = 92
N_ANCESTORS = 4
IUPAC = b"RYKMSWBDHVN"


def ancestor_length(a, scale=1):
    return (4_600_000 + 211_111 * a) // scale


def genome_length(i, scale=1):
    return ancestor_length(i % N_ANCESTORS, scale) - ((i * 7919) % 150_000) // scale


def lengths(scale=1):
    return [genome_length(i, scale) for i in range(N_GENOMES)]


def make_genome(oracle, i, scale=1, ancestors=None):
    """uint8 array of genome i (record boundaries are applied by :func:`records`)."""
    a = i % N_ANCESTORS
    anc = ancestors[a] if ancestors is not None else oracle.lcg_genome(7000 + a, ancestor_length(a, scale))
    g = (anc if i < N_ANCESTORS else oracle.lcg_mutant(anc, 2000 + i))[: genome_length(i, scale)].copy()
    n = len(g)
    rng = np.random.default_rng(50_000 + i)
    if i % 5 == 0:                                   # scaffold gaps: a few runs of N
        for _ in range(1 + i % 4):
            p = int(rng.integers(1000 // min(scale, 100), n - 3000 // min(scale, 100)))
            g[p:p + int(rng.integers(10, max(11, 2000 // scale)))] = ord("N")
    if i % 7 == 0:                                   # ambiguity codes of a consensus call
        for p in rng.integers(0, n, max(2, 40 // min(scale, 20))):
            g[p] = IUPAC[int(rng.integers(0, len(IUPAC)))]
    if i == 33:                                      # one soft-masked stretch (a repeat annotated in lower case)
        p = n // 2
        g[p:p + max(20, 700 // scale)] |= 0x20
    return g


def records(i, g):
    """[(title, bytes)] -- 1 .. 4 records per file."""
    n = len(g)
    k = 1 + i % 4
    cuts = [0] + [n - (k - r) * (n // 40) - 13 * i for r in range(1, k)] + [n]
    return [(f"g{i:02d}_rec{r} synthetic", bytes(g[cuts[r]:cuts[r + 1]])) for r in range(k)]


def write_fasta_fast(path, recs, width=70):
    """80-column-style FASTA without a Python loop per line."""
    with open(path, "wb") as f:
        for title, seq in recs:
            f.write(b">" + title.encode() + b"\n")
            a = np.frombuffer(seq, dtype=np.uint8)
            full = len(a) // width * width
            if full:
                body = np.empty((full // width, width + 1), dtype=np.uint8)
                body[:, :width] = a[:full].reshape(-1, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if full < len(a):
                f.write(a[full:].tobytes() + b"\n")


def expected_sequence(recs, reverse_complement):
    """What ref:snacc/pairwise_ncd.py:31-36 builds for the file: records concatenated, each one reverse-complemented
    on its own under -r (Python statement of the rules: snacc_amd/fasta.py)."""
    from snacc_amd import fasta
    parts = [fasta.reverse_complement(s.decode()) if reverse_complement else s.decode() for _, s in recs]
    return np.frombuffer("".join(parts).encode(), dtype=np.uint8)
