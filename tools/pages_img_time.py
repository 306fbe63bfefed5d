"""process_batch on whole page images, timed:
python tools/pages_img_time.py [npages] [pages per device batch] [page threads] [interpreter switch interval, s]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools import pages_bench as pb
from text_alignment_amd import alignToOCR as atocr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
from text_alignment_amd import textAlignPreprocessing as preproc
from tools import switches                               # noqa: E402

switches.apply()             # TA_* environment variables -> the product modules' attributes
if len(sys.argv) > 2:
    preproc.PAGES_PER_BATCH = int(sys.argv[2])
if len(sys.argv) > 3:
    preproc.PAGE_THREADS = int(sys.argv[3])
if len(sys.argv) > 4:
    sys.setswitchinterval(float(sys.argv[4]))            # seconds a thread may hold the interpreter against a waiting one
rec = pb.make_recognizer()
pages = [pb.RawPage(pb.make_page_image(9100 + k)) for k in range(n)]
trs = [pb.page_meta(100 + k)[1] for k in range(n)]
atocr.process_batch(pages, trs, rec, pb.PARAMS)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    found = atocr.find_lines_all(pages)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    atocr.process_batch(pages, trs, rec, pb.PARAMS)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    best = min(best, t2 - t1)
    print("switch %.4f chunk %d threads %d batch %d: find_lines_all %.1f ms, process_batch %.1f ms = %.0f pages/s"
          % (sys.getswitchinterval(), atocr.PIPELINE_CHUNK_PAGES_IMAGES, preproc.PAGE_THREADS, preproc.PAGES_PER_BATCH, 1e3 * (t1 - t0), 1e3 * (t2 - t1), n / (t2 - t1)), flush=True)
