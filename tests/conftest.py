import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:      # helper modules beside the tests (ocr_compare)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _library_is_built():
    """the package's host modules call into libta_hip.so (C ABI; loads without a GPU): build it once if a
    fresh checkout has none (hipcc cross-compiles gfx950 here)"""
    if not os.path.exists(os.path.join(REPO, "text_alignment_amd", "libta_hip.so")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def native(_library_is_built):
    from text_alignment_amd import _native
    return _native


def load_golden(name):
    with open(os.path.join(GOLDEN, name), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def unrle(pairs):
    out = []
    for op, cnt in pairs:
        out.extend([op] * cnt)
    return out


# callable scoring functions named by tests/golden/nw_kat.json ("scoring_fn")
_VOW = set("aeiouy")
SCORING_FNS = {
    "vowel_class": lambda a, b: 6 if a == b else (1 if (a in _VOW) == (b in _VOW) else -5),
    "ord_distance": lambda a, b: 5 - abs(ord(a) - ord(b)),
}


def kat_scoring(case):
    if "scoring_fn" in case:
        return [SCORING_FNS[case["scoring_fn"]]] + list(case["scoring_gaps"])
    return case["scoring"]
