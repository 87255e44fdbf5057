"""Native FASTA ingest (snk_fasta_extract*) against the Python statement of the
same rules (snacc_amd/fasta.py) and against the reference's recorded outputs.  Host only."""
import os

import numpy as np
import pytest

from snacc_amd import fasta, hip_backend


@pytest.fixture(scope="module", autouse=True)
def _built():
    hip_backend.build()
    hip_backend.load()


def _both(path, rc):
    return hip_backend.fasta_extract(path, rc), fasta.read_sequence(path, rc).encode("utf-8")


def test_sample_fa_matches_reference(golden, tmp_path):
    p = tmp_path / "sample.fa"
    with open(p, "w", newline="") as f:
        f.write(">derice\r\nACTGACTAGCTAGCTAACTG\r\n>sanka\r\nGCATCGTAGCTAGCTACGAT\r\n"
                ">junior\r\nCATCGATCGTACGTACGTAG\r\n>yul\r\nATCGATCGATCGTACGATCG")
    assert hip_backend.fasta_extract(p).decode() == golden["sample_fa"]["extract"]
    assert hip_backend.fasta_extract(p, True).decode() == golden["sample_fa"]["extract_rc"]


@pytest.mark.parametrize("text", [
    "; comment\n\n>r1 desc\nAC GT\nacgtn \t\n\n>r2\n>r3\nNNRY\n",
    ">a\rACGT\rTTGA\r>b\rGG\r",                       # old-Mac line ends
    ">a\r\nAC\r\nGT\r\n\r\n>b\r\nMRWSYKVHDBXN-*\r\n",
    "junk\n>x\nACGU\n>y\nacgu\n",                     # RNA records
    ">only header\n",
    ">h\nACGT",                                       # no trailing newline
    "text >not a header\n>h\n>ACGT in a header\nA>C\n",
])
@pytest.mark.parametrize("rc", [False, True])
def test_tricky_files(tmp_path, text, rc):
    p = tmp_path / "t.fa"
    with open(p, "w", newline="") as f:
        f.write(text)
    want = fasta.read_sequence(p, rc).encode()
    if not want:
        with pytest.raises(ValueError, match="No sequence extracted"):
            hip_backend.fasta_extract(p, rc)
    else:
        assert hip_backend.fasta_extract(p, rc) == want


def test_errors(tmp_path):
    p = tmp_path / "mixed.fa"
    p.write_text(">m\nACGTU\n")
    assert hip_backend.fasta_extract(p) == b"ACGTU"
    with pytest.raises(ValueError, match="Mixed RNA/DNA"):
        hip_backend.fasta_extract(p, True)
    with pytest.raises(hip_backend.HipBackendError, match="cannot open"):
        hip_backend.fasta_extract(tmp_path / "missing.fa")
    q = tmp_path / "nohdr.fa"
    q.write_text("ACGT\nACGT\n")
    with pytest.raises(ValueError, match=str(q)):
        hip_backend.fasta_extract(q)


def test_fuzz_against_python_reader(tmp_path):
    rng = np.random.default_rng(11)
    alphabet = list("ACGTNacgtnRYKMSWBDHVXU -*\t")
    for it in range(60):
        nl = ["\n", "\r\n", "\r"][it % 3]
        parts = []
        if it % 5 == 0:
            parts.append("preamble line" + nl)
        for r in range(int(rng.integers(1, 5))):
            parts.append(">rec%d some title" % r + nl)
            for _ in range(int(rng.integers(0, 6))):
                ln = int(rng.integers(0, 90))
                line = "".join(rng.choice(alphabet, ln))
                if "U" in line and it % 2:
                    line = line.replace("T", "A").replace("t", "a")
                parts.append(line + nl)
        text = "".join(parts)
        if it % 7 == 0:
            text = text.rstrip("\r\n")
        p = tmp_path / f"f{it}.fa"
        with open(p, "w", newline="") as f:
            f.write(text)
        for rc in (False, True):
            try:
                want = fasta.read_sequence(p, rc).encode()
            except ValueError:
                with pytest.raises(ValueError):
                    hip_backend.fasta_extract(p, rc)
                continue
            if want:
                assert hip_backend.fasta_extract(p, rc) == want, (it, rc)
            else:
                with pytest.raises(ValueError):
                    hip_backend.fasta_extract(p, rc)


def test_extract_many_threads(tmp_path, oracle_mod):
    import ctypes
    L = hip_backend.load()
    paths = []
    for i in range(9):
        p = tmp_path / f"g{i}.fasta"
        seq = bytes(oracle_mod.lcg_genome(i + 1, 5000 + 777 * i)).decode()
        with open(p, "w") as f:
            f.write(f">g{i}\n" + "\n".join(seq[k:k + 70] for k in range(0, len(seq), 70)) + "\n")
        paths.append(p)
    n = len(paths)
    arr = (ctypes.c_char_p * n)(*[os.fsencode(str(p)) for p in paths])
    outs = (ctypes.c_void_p * n)()
    lens = (ctypes.c_uint64 * n)()
    assert L.snk_fasta_extract_many(n, arr, 0, 4, outs, lens) == 0
    for i in range(n):
        assert ctypes.string_at(outs[i], lens[i]) == bytes(oracle_mod.lcg_genome(i + 1, 5000 + 777 * i))
        L.snk_free(outs[i])
    bad = tmp_path / "bad.fa"
    bad.write_text("nothing\n")
    arr2 = (ctypes.c_char_p * 2)(os.fsencode(str(paths[0])), os.fsencode(str(bad)))
    outs2 = (ctypes.c_void_p * 2)()
    lens2 = (ctypes.c_uint64 * 2)()
    assert L.snk_fasta_extract_many(2, arr2, 0, 2, outs2, lens2) == hip_backend.E_EMPTY
    assert outs2[0] is None and outs2[1] is None
    assert b"bad.fa" in L.snk_fasta_last_error()


# Cases worked out BY HAND from the documented behaviour of Biopython's plain FASTA reader
# (Bio.SeqIO.FastaIO.SimpleFastaParser, the reader behind SeqIO.parse(path, "fasta") at
# ref:snacc/pairwise_ncd.py:32) and of Seq.reverse_complement (ref:snacc/pairwise_ncd.py:34).
# Each row: (rule, file text, concatenated sequence, the same with -r).
DOCUMENTED_RULES = [
    ("blank lines inside a record are dropped: every line is rstrip()-ed before joining",
     ">a\nAC\n\n\nGT\n", "ACGT", "ACGT"),
    ("a header is a line whose FIRST character is '>': after leading blanks it is sequence text, "
     "and spaces are removed from the joined sequence",
     ">a\nAC\n >b\nGT\n", "AC>bGT", "ACv>GT"),          # lower-case b is the IUPAC code B: complement v
    ("text before the first '>' line (comments, stray residues) is skipped",
     "# produced by tool X\nACGTACGT\n\n>a\nTTG\n", "TTG", "CAA"),
    ("no validation of residues: digits, '*' and '-' stay, and the complement leaves them unchanged",
     ">a\nAC12*-GT\n", "AC12*-GT", "AC-*21GT"),
    ("text mode with universal newlines: a lone '\\r' ends a line",
     ">a\rAC\rGT\r>b\rTT\r", "ACGTTT", "ACGTAA"),
    ("only trailing white space is stripped and only spaces are removed: an inner tab stays",
     ">a\nAC\tGT \n", "AC\tGT", "AC\tGT"),
    ("a record may be empty; the file still yields the residues of the others",
     ">a\n>b\nAC\n>c\n", "AC", "GT"),
    ("-r reverse-complements every record on its own, records stay in file order "
     "(ref:snacc/pairwise_ncd.py:33-36)",
     ">a\nAAC\n>b\nGGT\n", "AACGGT", "GTTACC"),
    ("case is preserved, also by the complement", ">a\nacgtN\n", "acgtN", "Nacgt"),
    ("IUPAC ambiguity codes complement as M<->K R<->Y V<->B H<->D, W S N fixed",
     ">a\nMRWSYKVHDBN\n", "MRWSYKVHDBN", "NVHDBMRSWYK"),
    ("the title line is not sequence, whatever it contains", ">ACGT ACGT>>\nTG\n", "TG", "CA"),
    ("CRLF files: the reference's own fixture style", ">a\r\nAC\r\nGT\r\n", "ACGT", "ACGT"),
]


@pytest.mark.parametrize("rule,text,seq,seq_rc", DOCUMENTED_RULES, ids=[r[0][:40] for r in DOCUMENTED_RULES])
def test_documented_reader_rules(tmp_path, rule, text, seq, seq_rc):
    p = tmp_path / "case.fa"
    with open(p, "w", newline="") as f:
        f.write(text)
    assert fasta.read_sequence(p, False) == seq, rule
    assert fasta.read_sequence(p, True) == seq_rc, rule
    assert hip_backend.fasta_extract(p, False) == seq.encode(), rule
    assert hip_backend.fasta_extract(p, True) == seq_rc.encode(), rule
