"""The calling thread's stages of ONE process_batch pass, in order, with their wall time (a stage's time INCLUDES any
wait for the device or for the page threads inside it): where the host side of the chunk pipeline spends a call.
python tools/pages_host_stages.py [npages] [--images | --raw | --rows pinned|device]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools import pages_bench as pb, switches
from text_alignment_amd import alignToOCR as atocr, page as page_mod

switches.apply()
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
rec = pb.make_recognizer()
if "--images" in sys.argv:
    pages, trs = [pb.RawPage(pb.make_page_image(9100 + k)) for k in range(n)], [pb.page_meta(100 + k)[1] for k in range(n)]
elif "--rows" in sys.argv:
    pages, trs, _blocks = pb.make_pages_in_blocks([100 + k for k in range(n)], sys.argv[sys.argv.index("--rows") + 1])
else:
    raw = "--raw" in sys.argv
    pages, trs = zip(*[pb.make_page(100 + k + (5000 if raw else 0), raw=raw) for k in range(n)])
log = []


def wrap(owner, name, label=None):
    fn = getattr(owner, name)

    def timed(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            log.append((label or name, t0, time.perf_counter()))
    setattr(owner, name, timed)


for name in ("_pb_begin", "_pb_launch", "_pb_transcripts", "_pb_finish_a", "_pb_finish_b", "find_lines_all"):
    wrap(atocr, name)
wrap(page_mod, "prepared_lines")
wrap(type(rec), "prepare", "rec.prepare")
wrap(type(rec), "complete", "rec.complete")
wrap(type(rec), "run", "rec.run")
for _ in range(3):
    atocr.process_batch(list(pages), list(trs), rec, pb.PARAMS)
torch.cuda.synchronize()
for rep in range(2):
    del log[:]
    w0 = atocr.WAIT_SECONDS[0]
    t0 = time.perf_counter()
    atocr.process_batch(list(pages), list(trs), rec, pb.PARAMS)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("pass %d: %.2f ms, of which waiting for the device in the later stages %.2f ms" % (rep, 1e3 * (t1 - t0), 1e3 * (atocr.WAIT_SECONDS[0] - w0)))
    outer = [e for e in log if e[0].startswith("_pb_")]
    for name, a, b in sorted(outer, key=lambda e: e[1]):
        inner = ", ".join("%s %.2f" % (nm, 1e3 * (y - x)) for nm, x, y in log if not nm.startswith("_pb_") and x >= a and y <= b)
        print("   %7.2f ms  %-16s %6.2f ms   %s" % (1e3 * (a - t0), name, 1e3 * (b - a), inner))
    tot = {}
    for name, a, b in outer:
        tot[name] = tot.get(name, 0.0) + b - a
    print("   totals: " + ", ".join("%s %.2f" % (k, 1e3 * v) for k, v in tot.items()))
