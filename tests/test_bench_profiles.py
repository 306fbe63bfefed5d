"""bench.py reads a few measured constants (HBM traffic, MFMA pipe busy, mode agreement) from the rocprof summaries kept
under profiles/: it must cite the NEWEST round's copy of each -- a line that names an older round than profiles/ holds
is wrong the day a kernel changes."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib
    import sys
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    return importlib.import_module("bench")


@pytest.mark.parametrize("stem", ["nw2_hbm_traffic.json", "nw_hbm_traffic.json", "ocr_pmc_mfma.json",
                                  "ocr_hbm_traffic.json", "ocr_mode_agreement.json", "valu_issue_rates.txt"])
def test_profile_files_cited_are_the_newest_round_present(stem):
    bench = _bench()
    name, path = bench._profile_file(stem)
    assert name is not None and os.path.exists(path)
    rounds = [int(m.group(1)) for f in os.listdir(os.path.join(REPO, "profiles"))
              for m in [re.match(r"r(\d+)_" + re.escape(stem) + "$", f)] if m]
    assert int(re.match(r"r(\d+)_", name).group(1)) == max(rounds)


def test_no_round_is_spelled_out_in_bench_py():
    """every profile the line cites goes through _profile_file: no literal profiles/rNN_ path in the code"""
    with open(os.path.join(REPO, "bench.py")) as f:
        src = f.read()
    code = "\n".join(re.sub(r"\s+# .*$", "", ln) for ln in src.splitlines() if not ln.lstrip().startswith("#"))
    hits = re.findall(r"[\"']r0\d_[a-z0-9_]+\.(?:json|txt|csv)", code) + re.findall(r"profiles/r0\d_[a-z0-9_]+", code)
    # (peak_is strings name the microbenchmark records that established a hardware rate: documentation, not data read here)
    hits = [h for h in hits if "mfma_f64" not in h]
    assert not hits, hits


def test_traffic_sources_on_a_line_name_the_newest_round():
    bench = _bench()
    t, src = bench.measured_traffic(4096, 4096, 4096, "nw_score_kernel")
    if src is not None:
        assert src == "profiles/" + bench._profile_file("nw2_hbm_traffic.json")[0]
    b, bsrc = bench.measured_mfma_busy("lstm_seq4_kernel")
    if bsrc is not None:
        assert bsrc == "profiles/" + bench._profile_file("ocr_pmc_mfma.json")[0]
