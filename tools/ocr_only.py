"""Run only the OCR kernels a few times (profiling helper)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from text_alignment_amd import ocr
from bench import synthetic_lines
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
rec = ocr.LineRecognizer(ocr.LineModel.random(7001, no=96))
st = rec.prepare(synthetic_lines(n, 8000))
for _ in range(3):
    rec.run(st)
torch.cuda.synchronize()
print("done", st["rows"])
