"""Ad-hoc fuzz of the float64 recogniser against the float64 restatement (oracle/ocr_ref_f64.py): ragged batches of short
random lines (1 .. a few hundred timesteps, row counts that are not multiples of the projection's 16-row tiles or of a
group), random spec-shaped models -- free-running logits within TOL of the restatement on every line, decode equal.
    python tools/fuzz_ocr_oracle.py [rounds] [seed] [4 | 16 | both]
(tools/fuzz_ocr_kernels.py compares the launch shapes with each other; this one pins them to the checker.  The restatement
runs ~28 000 timesteps per second on one core, so a round is kept under ~20 000 timesteps.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import ocr_ref_f64 as R
from text_alignment_amd import ocr

TOL = 1e-4          # the bound tests/test_ocr_gpu.py::test_spec_model_benchmark_widths_free_running_f64 asserts on the logits
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
groups = {"4": (4,), "16": (16,), "both": (4, 16)}[sys.argv[3] if len(sys.argv) > 3 else "both"]
bad, worst, nlines, nsteps = 0, 0.0, 0, 0
for rnd in range(rounds):
    no = int(rng.choice([3, 17, 40, 96, 128]))
    om = R.synthetic_model(int(rng.integers(1, 10 ** 6)), no=no)
    rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), precision="f64")
    n = int(rng.choice([1, 3, 4, 5, 15, 16, 17, 31, 64, 65]))
    hi = int(rng.choice([1, 3, 17, 40, 300]))
    hi = max(1, min(hi, 20000 // n))
    lines = []
    for _ in range(n):
        T = int(rng.integers(1, hi + 1))
        kind = rng.integers(0, 3)
        if kind == 0:                                   # ink / paper
            x = (rng.random((T, 48)) < rng.uniform(0.05, 0.6)).astype(np.float32)
        elif kind == 1:                                 # greys
            x = rng.random((T, 48)).astype(np.float32)
        else:                                           # mostly paper, a few columns of ink (exact zeros in the projection's inputs)
            x = np.zeros((T, 48), dtype=np.float32)
            x[rng.random(T) < 0.3] = rng.random(48).astype(np.float32)
        lines.append(x)
    refs = []
    for x in lines:             # (not R.recognise: its llocs scale divides by T - 32, and T = 32 is a length like any other here)
        z, p = R.softmax_layer(om, R.bilstm_states(om, x.astype(np.float64)))
        refs.append({"logits": z, "decoded": R.translate_back(p)})
    for G in groups:
        ocr.FORCE_GROUP = G
        dec, probs, logits, states = rec.recognise(lines, want_probs=True)
        for k in range(n):
            err = float(np.abs(logits[k] - refs[k]["logits"]).max())
            worst = max(worst, err)
            if not err < TOL or dec[k] != refs[k]["decoded"]:
                bad += 1
                print("MISMATCH round %d: line %d of %d (T %d), classes %d, groups of %d: logit error %.3g, decode equal %s"
                      % (rnd, k, n, lines[k].shape[0], no, G, err, dec[k] == refs[k]["decoded"]), flush=True)
    nlines += n
    nsteps += sum(x.shape[0] for x in lines)
    print("round %d: %d lines up to %d steps, %d classes: ok so far (%d mismatches, worst logit error %.3g)"
          % (rnd, n, hi, no, bad, worst), flush=True)
print("fuzz finished: %d lines, %d timesteps, groups of %s: %d mismatches, worst logit error %.3g (bound %.0e)"
      % (nlines, nsteps, " and ".join(map(str, groups)), bad, worst, TOL))
sys.exit(1 if bad else 0)
