"""sha256 of the recurrence kernel's outputs on seeded ragged lines (both modes): two builds whose
kernels accumulate in the same order print the same digests.  python tools/ocr_hash.py [nlines]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from text_alignment_amd import ocr
from tools import switches                               # noqa: E402

switches.apply()             # TA_* environment variables -> the product modules' attributes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(77)
lines = [(rng.random((int(rng.integers(1, 400)), 48)) < 0.3).astype(np.float32) for _ in range(n)]
lines[3] = lines[3][:1]
lines[4] = lines[4][:2]
for prec in ocr.PRECISIONS:
    model = ocr.LineModel.random(5, no=40)
    rec = ocr.LineRecognizer(model, precision=prec)
    st = rec.prepare(lines)
    rec.run(st, want_logits=True)
    torch.cuda.synchronize()
    h = hashlib.sha256(st["hout"].cpu().numpy().tobytes()).hexdigest()
    z = hashlib.sha256(st["logits"].cpu().numpy().tobytes()).hexdigest()
    print(prec, "hout", h[:16], "logits", z[:16], flush=True)
