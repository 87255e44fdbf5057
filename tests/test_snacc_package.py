"""The reference's import name: ``import snacc`` / ``snacc.cli`` / ``snacc.pairwise_ncd`` resolve to the
MI355X-native implementation (ref:snacc/__init__.py:1-2, ref:setup.py:115-117), and single-item calls
on the shared device context are serialised (the reference's callers use a thread pool,
ref:snacc/cli.py:104-129)."""
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_import_by_the_reference_name():
    import snacc
    import snacc.cli
    import snacc.pairwise_ncd
    import snacc_amd
    assert snacc.compressed_size is snacc_amd.compressed_size
    assert snacc.compute_distance is snacc_amd.compute_distance
    assert snacc.__version__ == snacc_amd.__version__
    assert snacc.pairwise_ncd.extract_sequences is snacc_amd.pairwise_ncd.extract_sequences
    assert snacc.cli.cli is snacc_amd.cli.cli
    # the notebook KAT of the reference: sizes -> NCD
    assert snacc.compute_distance(1174721, 1173133, 1242873, 1242873) == 0.05936728806244206


def test_module_entry_point_and_console_script_name():
    out = subprocess.run([sys.executable, "-m", "snacc.cli", "--help"], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0 and "--compression" in out.stdout and "--reverse_complement" in out.stdout
    setup_py = (ROOT / "setup.py").read_text()
    assert "snacc=snacc.cli:cli" in setup_py and '"snacc"' in setup_py


def test_single_item_calls_hold_the_context_lock(monkeypatch, tmp_path):
    """compressed_size(..., 'lz4') from a thread pool: upload + launch + read-back of one call never
    interleave with another's (checked with a fake context that records overlap)."""
    from snacc_amd import pairwise_ncd as pn

    class FakeCtx:
        def __init__(self):
            self.busy = 0
            self.overlaps = 0
            self.resident = None

        def upload(self, seqs):
            self.busy += 1
            if self.busy > 1:
                self.overlaps += 1
            self.resident = [bytes(s) for s in seqs]
            time.sleep(0.002)

        def _finish(self, value):
            time.sleep(0.002)
            self.busy -= 1
            return value

        def singles(self):
            return self._finish([len(self.resident[0])])

        def pairs_list(self, items):
            return self._finish([len(self.resident[0]) + len(self.resident[1])])

    fake = FakeCtx()
    monkeypatch.setattr(pn, "_ctx", fake)
    files = []
    for k in range(6):
        f = tmp_path / f"g{k}.fa"
        f.write_text(">r\n" + "ACGT" * (5 + k) + "\n")
        files.append(f)
    results, errors = {}, []

    def work(k):
        try:
            for i in range(4):
                key = files[k] if i % 2 == 0 else (files[k], files[(k + 1) % 6])
                results[(k, i)] = pn.compressed_size(key, "lz4")[1]
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and fake.overlaps == 0 and fake.busy == 0
    for (k, i), v in results.items():        # every call saw ITS OWN sequences
        n = 4 * (5 + k) if i % 2 == 0 else 4 * (5 + k) + 4 * (5 + (k + 1) % 6)
        assert v == n + 33
