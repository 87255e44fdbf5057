"""FASTA ingest for the NCD hot path (replaces ``Bio.SeqIO.parse`` + ``Seq.reverse_complement``
at ref:snacc/pairwise_ncd.py:32-36; Biopython is not a dependency here).

Semantics restated from Biopython's plain FASTA reader (``SimpleFastaParser``):
text mode with universal newlines, everything before the first ``>`` line is
skipped, each sequence line is ``rstrip()``-ed and joined, then spaces and ``\\r``
are removed.  Reverse complement follows ``Bio.Seq``: IUPAC ambiguous DNA table,
case preserved, unknown characters unchanged; a sequence with ``U`` and no ``T``
is complemented as RNA; ``T`` and ``U`` together raise ``ValueError``.
(Biopython itself is absent from this image, so this restatement is checked against the
reference's own fixture and against cases worked out by hand from Biopython's documented
reader -- tests/test_fasta_native.py::test_documented_reader_rules, tests/test_host_logic.py.)
"""
from pathlib import Path

_DNA = {"A": "T", "C": "G", "G": "C", "T": "A", "M": "K", "R": "Y", "W": "W", "S": "S",
        "Y": "R", "K": "M", "V": "B", "H": "D", "D": "H", "B": "V", "X": "X", "N": "N"}
_RNA = {"A": "U", "C": "G", "G": "C", "U": "A", "M": "K", "R": "Y", "W": "W", "S": "S",
        "Y": "R", "K": "M", "V": "B", "H": "D", "D": "H", "B": "V", "X": "X", "N": "N"}


def _table(mapping):
    keys = "".join(mapping.keys())
    vals = "".join(mapping.values())
    return str.maketrans(keys + keys.lower(), vals + vals.lower())


_DNA_TABLE = _table(_DNA)
_RNA_TABLE = _table(_RNA)


def reverse_complement(seq):
    """``str(Seq(seq).reverse_complement())``."""
    has_u = "U" in seq or "u" in seq
    has_t = "T" in seq or "t" in seq
    if has_u and has_t:
        raise ValueError("Mixed RNA/DNA found")
    table = _RNA_TABLE if has_u else _DNA_TABLE
    return seq.translate(table)[::-1]


def read_fasta_records(path):
    """Yield ``(title, sequence)`` for every record of a FASTA file."""
    with open(Path(path), "r") as handle:
        text = handle.read()
    start = 0 if text.startswith(">") else text.find("\n>")
    if start < 0:
        return
    if start:
        start += 1
    # records are delimited by lines that start with '>'
    body = text[start:]
    for chunk in ("\n" + body).split("\n>")[1:]:
        nl = chunk.find("\n")
        if nl < 0:
            title, rest = chunk, ""
        else:
            title, rest = chunk[:nl], chunk[nl + 1:]
        seq = "".join(line.rstrip() for line in rest.split("\n"))
        yield title.rstrip(), seq.replace(" ", "").replace("\r", "")


def read_sequence(path, reverse_complement_records=False):
    """Concatenated residues of all records (each record reverse-complemented on its own when
    asked), exactly what ref:snacc/pairwise_ncd.py:31-36 builds for one file."""
    parts = []
    for _title, seq in read_fasta_records(path):
        parts.append(reverse_complement(seq) if reverse_complement_records else seq)
    return "".join(parts)
