import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def derived_fasta(tmp_path_factory):
    """golden-derived FASTA pre-images of the reference's k=2 goldens (tests/golden/derive_fasta.py)."""
    sys.path.insert(0, GOLDEN)
    import derive_fasta

    d = tmp_path_factory.mktemp("derived")
    out = {}
    for name in ("seq1", "seq2"):
        rows = derive_fasta.parse_golden(os.path.join(GOLDEN, f"out-{name}.cfrk"))
        reads, tail = derive_fasta.derive(rows, 5)
        p = str(d / f"{name}.fasta")
        derive_fasta.write_fasta(reads, tail, p)
        out[name] = p
    return out
