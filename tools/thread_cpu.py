"""Which threads of the process burn host CPU during process_batch passes: per-thread utime + stime deltas from
/proc/self/task (main interpreter thread, copy pool, the HIP / HSA runtime's helpers).   python tools/thread_cpu.py [npages] [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools import pages_bench as pb, switches
from text_alignment_amd import alignToOCR as atocr

print("switches:", switches.apply())


def snapshot():
    out = {}
    hz = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            with open("/proc/self/task/%s/stat" % tid) as f:
                s = f.read()
            comm = s[s.index("(") + 1:s.rindex(")")]
            rest = s[s.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / hz)
        except (OSError, ValueError):
            pass
    return out


n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows = sys.argv[2] if len(sys.argv) > 2 else "pinned"
rec = pb.make_recognizer()
if rows == "numpy":
    pages, trs = (list(v) for v in zip(*[pb.make_page(100 + k) for k in range(n)]))
else:
    pages, trs, blocks = pb.make_pages_in_blocks([100 + k for k in range(n)], rows)
for _ in range(3):
    atocr.process_batch(pages, trs, rec, pb.PARAMS)
torch.cuda.synchronize()
reps = 40
a, t0 = snapshot(), time.perf_counter()
for _ in range(reps):
    atocr.process_batch(pages, trs, rec, pb.PARAMS)
torch.cuda.synchronize()
wall, b = time.perf_counter() - t0, snapshot()
print("%d passes of %d pages (%s rows): wall %.1f ms per pass; main tid %d" % (reps, n, rows, 1e3 * wall / reps, os.getpid()))
for tid, (comm, t) in sorted(b.items(), key=lambda kv: -(kv[1][1] - a.get(kv[0], ("", 0))[1])):
    d = t - a.get(tid, ("", 0))[1]
    if d > 0:
        print("  tid %7d %-18s %.2f ms per pass (%.2f of wall)" % (tid, comm, 1e3 * d / reps, d / wall))
