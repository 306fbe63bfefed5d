"""cProfile of find_lines_all / process_batch on whole page images (single page thread):
python tools/pages_img_profile.py [npages]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools import pages_bench as pb
from text_alignment_amd import alignToOCR as atocr
from text_alignment_amd import textAlignPreprocessing as preproc
from tools import switches                               # noqa: E402

switches.apply()             # TA_* environment variables -> the product modules' attributes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
preproc.PAGE_THREADS = 1
rec = pb.make_recognizer()
pages = [pb.RawPage(pb.make_page_image(9100 + k)) for k in range(n)]
trs = [pb.page_meta(100 + k)[1] for k in range(n)]
atocr.process_batch(pages, trs, rec, pb.PARAMS)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
atocr.process_batch(pages, trs, rec, pb.PARAMS)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
