"""Does a two-phase batch gain from being run as two halves, the traceback of the first half beside the score fill of the
second (two streams)?   python tools/nw_halves_probe.py [nprob n m] ...
Prints per shape: one launch pair (the product), halves staggered, halves side by side.

Measured on MI355X (round 4): no.  1024 x 2048^2: 1.36 / 1.59 / 1.50 ms;  2048 x 2048^2: 2.35 / 2.67 / 2.48;
4096 x 4096^2: 13.82 / 14.62 / 13.87;  only 512 x 4096^2 gains side by side (2.45 / 2.60 / 2.24) -- both phases are bound
by VALU issue once the chip is full, so a half's traceback beside the other half's fill only shares the same issue slots."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from text_alignment_amd import _native                   # noqa: E402
from text_alignment_amd import textSeqCompare as tsc     # noqa: E402


def call(b, k0, k1, fill, tb, stream):
    flags = (_native.TA_NW_FILL if fill else 0) | (_native.TA_NW_TRACEBACK if tb else 0)
    if b.codes8:
        flags |= _native.TA_NW_CODES8
    flags |= b.phase1_flags()
    rc = _native.lib.ta_nw2_batch(
        b.t_codes.data_ptr(), b.t_off.data_ptr() + 8 * k0, b.o_codes.data_ptr(), b.o_off.data_ptr() + 8 * k0, k1 - k0,
        b.params.data_ptr() + 4 * b.params_stride * k0, b.params_stride, b.ws.data_ptr(), b.ws_off.data_ptr() + 8 * k0,
        b.ops.data_ptr(), b.ops_off.data_ptr() + 8 * k0, b.ops_len.data_ptr() + 4 * k0,
        b.max_n, b.max_m, b.score_bound, flags, stream.cuda_stream)
    _native.check(rc, "ta_nw2_batch")


def main():
    shapes = [tuple(int(v) for v in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)] or \
        [(1024, 2048, 2048), (512, 4096, 4096), (2048, 2048, 2048), (4096, 4096, 4096)]
    dev = torch.device("cuda:0")
    s1, s2 = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=-1)
    main_s = torch.cuda.current_stream(dev)
    for nprob, n, m in shapes:
        b, _ = bench.make_nw_batch(tsc, nprob, n, m, 1234, two_phase=True)
        b.run()
        torch.cuda.synchronize()
        want = [r.copy() for r in b.results()[:8]] + [r.copy() for r in b.results()[-8:]]
        h = nprob // 2

        def one():
            b.run()

        def staggered():
            s1.wait_stream(main_s); s2.wait_stream(main_s)
            call(b, 0, h, True, False, s1)
            ev = torch.cuda.Event(); ev.record(s1)
            call(b, 0, h, False, True, s1)
            s2.wait_event(ev)
            call(b, h, nprob, True, True, s2)
            main_s.wait_stream(s1); main_s.wait_stream(s2)

        def side_by_side():
            s1.wait_stream(main_s); s2.wait_stream(main_s)
            call(b, 0, h, True, True, s1)
            call(b, h, nprob, True, True, s2)
            main_s.wait_stream(s1); main_s.wait_stream(s2)

        def round_shaped(cut):
            # the fill exactly as one launch would dispatch it (part A, then part B right behind on the same stream); the
            # traceback of part A on a second stream as soon as A's fill is done -- beside B's fill, which leaves most of the
            # chip idle when A was a whole round of resident workgroups -- and the traceback of part B behind its fill
            def fn():
                s1.wait_stream(main_s); s2.wait_stream(main_s)
                call(b, 0, cut, True, False, s1)
                ev = torch.cuda.Event(); ev.record(s1)
                call(b, cut, nprob, True, True, s1)
                s2.wait_event(ev)
                call(b, 0, cut, False, True, s2)
                main_s.wait_stream(s1); main_s.wait_stream(s2)
            return fn

        out = []
        cuts = [c for c in (512, 640, 768, 896) if c < nprob] if nprob <= 2048 else []
        for name, fn in [("one launch pair", one), ("halves staggered", staggered), ("halves side by side", side_by_side)] + \
                [("first %d then the rest, tracebacks overlapped" % c, round_shaped(c)) for c in cuts]:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            got = b.results()
            ok = all(np.array_equal(a, c) for a, c in zip(want, got[:8] + got[-8:]))
            out.append("%s %.3f ms%s" % (name, float(np.median(ts)), "" if ok else " (WRONG)"))
        print("%5d x %d x %d: %s" % (nprob, n, m, ";  ".join(out)), flush=True)


if __name__ == "__main__":
    main()
