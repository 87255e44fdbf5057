"""Generator-defined inputs of the frame-byte fixtures (tests/golden/frame_hashes.json): ragged
lengths, raw blocks, long matches, one-shot frames, N runs.  Pure Python / numpy, no oracle."""
import numpy as np

LCG_A, LCG_C = 6364136223846793005, 1442695040888963407


def lcg_genome(seed, n):
    s, out = seed, bytearray(n)
    for i in range(n):
        s = (s * LCG_A + LCG_C) & ((1 << 64) - 1)
        out[i] = b"ACGT"[(s >> 33) & 3]
    return bytes(out)


def lcg_bytes(seed, n, alphabet):
    s, out, k = seed, bytearray(n), len(alphabet)
    for i in range(n):
        s = (s * LCG_A + LCG_C) & ((1 << 64) - 1)
        out[i] = alphabet[(s >> 33) % k]
    return bytes(out)


def build_sequences():
    g1 = lcg_genome(101, 100000)
    rep = (lcg_genome(102, 900) * 200)[:170000]
    return [
        b"",                                                   # 0 empty frame
        b"ACGT" * 10,                                          # 1 tiny one-shot
        g1,                                                    # 2 linked, two blocks
        lcg_genome(103, 70001),                                # 3 ragged
        rep,                                                   # 4 long matches (length-extension bytes)
        lcg_bytes(104, 90000, bytes(range(256))),              # 5 incompressible: raw blocks
        b"N" * 40000 + lcg_genome(105, 50000),                 # 6 N run
        lcg_genome(106, 30000),                                # 7 one-shot (<= 64 KiB)
        b"A" * 70000,                                          # 8 homopolymer
        b"ACGTACGTACGTA",                                      # 9 13 bytes: the shortest compressed block
        lcg_genome(107, 65536),                                # 10 exactly one block
        lcg_bytes(108, 66000, b"ACDEFGHIKLMNPQRSTVWY"),        # 11 protein alphabet
    ]


ITEMS = [(i, -1) for i in range(12)] + [(2, 3), (3, 2), (2, 2), (4, 2), (2, 4), (5, 2), (2, 5), (6, 7), (7, 6),
                                        (7, 7), (8, 8), (9, 9), (10, 10), (10, 3), (11, 2), (1, 9), (0, 2), (2, 0),
                                        (5, 5), (4, 4)]
