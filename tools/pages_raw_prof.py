"""cProfile of process_batch on raw uint8 strips (device line normaliser in front)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools import pages_bench as pb
from text_alignment_amd import alignToOCR as atocr
from tools import switches                               # noqa: E402

switches.apply()             # TA_* environment variables -> the product modules' attributes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rec = pb.make_recognizer()
pages, trs = zip(*[pb.make_page(5100 + k, raw=True) for k in range(n)])
atocr.process_batch(list(pages), list(trs), rec, pb.PARAMS)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
atocr.process_batch(list(pages), list(trs), rec, pb.PARAMS)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
