"""Timing probe (results are NOT valid recogniser outputs): the recurrence launched as C group classes on C
streams, each followed on its stream by an output-layer launch over as many rows as the class owns -- what a
class-split run() would cost.  python tools/ocr_overlap_probe.py [nlines] [fractions of groups per class, longest first]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic_lines
from text_alignment_amd import _native, ocr
from tools import switches                               # noqa: E402

switches.apply()             # TA_* environment variables -> the product modules' attributes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
fr = [float(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.125, 0.125, 0.25, 0.5]
rec = ocr.LineRecognizer(ocr.LineModel.random(7001, no=96))
st = rec.prepare(synthetic_lines(n, 8000))
lib = _native.lib
ng = st["ngroups"]
T = st["T_host"]
order = np.argsort(-T, kind="stable")
cuts = [0] + [int(round(ng * v)) for v in np.cumsum(fr)]
cuts[-1] = ng
rows_of = [int(T[order[16 * cuts[c]:16 * cuts[c + 1]]].sum()) for c in range(len(fr))]
row0 = np.concatenate([[0], np.cumsum(rows_of)])
print("groups per class", np.diff(cuts).tolist(), "rows per class", rows_of)
ndummy = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # streams made (and used once) before the class streams
dummies = [torch.cuda.Stream() for _ in range(ndummy)]
for s_ in dummies:
    with torch.cuda.stream(s_):
        torch.zeros(8, device="cuda").add_(1)
torch.cuda.synchronize()
prio = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0] * len(fr)
streams = [torch.cuda.Stream(priority=prio[i]) for i in range(len(fr))]
main_stream = torch.cuda.Stream() if os.environ.get("PROBE_MAIN_SIDE") else None


def plain():
    rec.run(st)


def split():
    cs = torch.cuda.current_stream()
    for c, s in enumerate(streams):
        s.wait_stream(cs)
        h = s.cuda_stream
        gl = st["group_lines"].data_ptr() + 16 * 4 * cuts[c]
        _native.check(lib.ta_lstm_forward(st["x"].data_ptr(), st["row_off"].data_ptr(), st["T"].data_ptr(), gl,
                                          cuts[c + 1] - cuts[c], rec.wp.data_ptr(), rec.peep.data_ptr(),
                                          st["hout"].data_ptr(), rec.mode, None, None, None, h), "fwd")
        r0 = int(row0[c])
        _native.check(lib.ta_lstm_output(st["hout"].data_ptr() + r0 * 800, rows_of[c], rec.w2p.data_ptr(), rec.model.no,
                                         None, None, st["summary"].data_ptr() + r0 * 16, h), "out")
    for s in streams:
        cs.wait_stream(s)
    _native.check(lib.ta_decode_summary(st["summary"].data_ptr(), st["row_off"].data_ptr(), st["T"].data_ptr(), st["n"],
                                        ocr.THRESHOLD, st["dec_t"].data_ptr(), st["dec_c"].data_ptr(),
                                        st["dec_n"].data_ptr(), st["row_off"].data_ptr(), cs.cuda_stream), "dec")


import contextlib
ctx = torch.cuda.stream(main_stream) if main_stream is not None else contextlib.nullcontext()
ctx.__enter__()
for name, fn in (("one launch each", plain), ("class split", split), ("one launch each", plain), ("class split", split)):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-16s %.3f ms per pass" % (name, e0.elapsed_time(e1) / 5), flush=True)
