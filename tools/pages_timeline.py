"""Device timeline of ONE process_batch pass on synthetic pages (torch profiler): every device interval with its
stream, start and duration -- what overlaps what in the chunk pipeline.   python tools/pages_timeline.py [npages] [--raw | --images | --rows pinned|device]
[--gaps: only the device's idle stretches of 0.2 ms and more]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

from tools import pages_bench as pb
from text_alignment_amd import alignToOCR as atocr
from tools import switches                               # noqa: E402

switches.apply()             # TA_* environment variables -> the product modules' attributes

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
rec = pb.make_recognizer()
raw = "--raw" in sys.argv
if "--images" in sys.argv:
    pages = [pb.RawPage(pb.make_page_image(9100 + k)) for k in range(n)]
    trs = [pb.page_meta(100 + k)[1] for k in range(n)]
elif "--rows" in sys.argv:
    pages, trs, _blocks = pb.make_pages_in_blocks([100 + k for k in range(n)], sys.argv[sys.argv.index("--rows") + 1])
else:
    pages, trs = zip(*[pb.make_page(100 + k + (5000 if raw else 0), raw=raw) for k in range(n)])
for _ in range(3):
    atocr.process_batch(list(pages), list(trs), rec, pb.PARAMS)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    atocr.process_batch(list(pages), list(trs), rec, pb.PARAMS)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if getattr(e, "device_type", None) is not None and "CUDA" in str(e.device_type)]
ev.sort(key=lambda e: e.time_range.start)
t0 = ev[0].time_range.start
if "--gaps" in sys.argv:
    # the idle stretches of the device (no interval of any stream running) of at least 0.2 ms, with what ran before and after
    end, last, idle = ev[0].time_range.end, ev[0], 0.0
    for e in ev[1:]:
        if e.time_range.start > end:
            gap = e.time_range.start - end
            idle += gap
            if gap >= 200:
                print("%9.3f ms  idle %7.3f ms   after %-42s before %s" % ((end - t0) / 1e3, gap / 1e3, last.name[:42], e.name[:42]))
        if e.time_range.end > end:
            end, last = e.time_range.end, e
    print("span %.3f ms, idle %.3f ms" % ((end - t0) / 1e3, idle / 1e3))
    sys.exit(0)
for e in ev:
    d = e.time_range.end - e.time_range.start
    if d >= 30:
        print("%9.3f ms  +%8.3f ms  %s" % ((e.time_range.start - t0) / 1e3, d / 1e3, e.name[:70]))
print("span %.3f ms" % ((max(e.time_range.end for e in ev) - t0) / 1e3))
