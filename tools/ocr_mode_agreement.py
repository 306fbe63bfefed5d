"""Free-running parity of the line recogniser against oracle/ocr_ref_f64.py, per precision mode, on
the model AS SPECIFIED (seeds 7001 / 7002, U(-0.5, 0.5) weights) at the
benchmark's widths (800 .. 2000 columns): per line the max-abs logit / probability error and whether
the decoded (t, class) list is identical; summary per (model, mode).  A second pair of models is the
page tests' text-like variant (40 classes, blank favoured: thousands of characters come out), where
character agreement is the informative number.

Usage: python tools/ocr_mode_agreement.py [lines per model] [out.json]
"""
import json
import os
import sys
import zlib
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import ocr_ref_f64 as R
from text_alignment_amd import ocr


def _models():
    for seed, no in ((7001, 96), (7002, 64)):
        yield "%d/spec" % seed, R.synthetic_model(seed, no=no)
    for seed in (7001, 7002):
        om = R.synthetic_model(seed, no=40)
        om.W2[0, 0] += 4.0                          # tests/test_page_gpu.py's text-like variant
        yield "%d/textlike40" % seed, om


def measure(nlines=24, modes=None):
    modes = modes or list(ocr.PRECISIONS)
    out = {}
    for name, om in _models():
        rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)        # (hash() of a str changes from run to run)
        widths = [800, 2000] + [int(w) for w in rng.integers(800, 2001, size=nlines - 2)]
        lines = [R.synthetic_line(8000 + k, width=w) for k, w in enumerate(widths)]
        t0 = time.time()
        refs = [R.recognise(om, xs) for xs in lines]
        t_ref = time.time() - t0
        for mode in modes:
            rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), precision=mode)
            dec, probs, logits, states = rec.recognise(lines, want_probs=True)
            ez = [float(np.abs(logits[k] - refs[k]["logits"]).max()) for k in range(nlines)]
            ep = [float(np.abs(probs[k] - refs[k]["probs"]).max()) for k in range(nlines)]
            eh = [float(np.abs(states[k] - refs[k]["states"]).max()) for k in range(nlines)]
            same = [dec[k] == refs[k]["decoded"] for k in range(nlines)]
            nchar = sum(len(r["decoded"]) for r in refs)
            hit = sum(len(set(dec[k]) & set(refs[k]["decoded"])) for k in range(nlines))
            out["%s/%s" % (name, mode)] = dict(
                model=name, classes=om.no, mode=mode, lines=nlines, widths=[min(widths), max(widths)],
                logit_err_max=max(ez), logit_err_median=float(np.median(ez)),
                logit_err_p90=float(np.quantile(ez, 0.9)),
                prob_err_max=max(ep), prob_err_median=float(np.median(ep)), state_err_max=max(eh),
                lines_within_1e3=int(sum(e < 1e-3 for e in ez)),
                lines_decode_identical=int(sum(same)),
                chars_ref=nchar, chars_agree=hit, oracle_seconds=round(t_ref, 2))
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    res = measure(n)
    for k, v in res.items():
        print(k, json.dumps(v))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(res, f, indent=1)
