"""Run only the OCR kernels a few times (profiling helper).
Usage: python tools/ocr_only.py [nlines] [split|f32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synthetic_lines
from text_alignment_amd import ocr
from tools import switches                               # noqa: E402

switches.apply()             # TA_* environment variables -> the product modules' attributes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
prec = sys.argv[2] if len(sys.argv) > 2 else ocr.DEFAULT_PRECISION
rec = ocr.LineRecognizer(ocr.LineModel.random(7001, no=96), precision=prec)
st = rec.prepare(synthetic_lines(n, 8000))
for _ in range(3):
    rec.run(st)
torch.cuda.synchronize()
for name, kw in (("recurrence", dict(output=False, decode=False)), ("all", {})):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        rec.run(st, **kw)
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.3f ms per pass" % (name, e0.elapsed_time(e1) / 3))
print("done", st["rows"], prec)
