"""Device time of the two float64 kernels of the recogniser, each launched alone on the bench's OCR workload (1 920 synthetic
lines, groups of four): lstm_xproj_f64_kernel and lstm_seq4_f64_kernel, mean of 5 after 2 warm-ups, and a checksum of Gx /
hout (variants built by tools/f64_variants.sh must agree to the bit).  TA_HIP_LIB selects the library.

    python tools/xproj_time.py [nlines]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402
from text_alignment_amd import _native, ocr            # noqa: E402


def main():
    nlines = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
    rec = ocr.LineRecognizer(ocr.LineModel.random(7001, no=96), precision="f64")
    st = rec.prepare(bench.synthetic_lines(nlines, 8000))
    lib, rows = _native.lib, int(st["rows"])
    gx = torch.empty(lib.ta_lstm_f64_gx_bytes(rows) // 8, dtype=torch.float64, device=rec.device)
    stream = torch.cuda.current_stream().cuda_stream
    G = st["group_size"]
    assert G == 4

    def xproj():
        _native.check(lib.ta_lstm_xproj_f64(st["x"].data_ptr(), rows, rec.wx64.data_ptr(), gx.data_ptr(), stream), "xproj")

    def seq():
        _native.check(lib.ta_lstm_forward_f64_g4(gx.data_ptr(), 0, rows, st["row_off"].data_ptr(), st["T"].data_ptr(),
                                                 st["group_lines"].data_ptr(), st["ngroups"], rec.wh64g4.data_ptr(),
                                                 rec.peep64.data_ptr(), st["hout"].data_ptr(), None, None, None, None, stream), "seq4")
    out = {}
    for name, fn in (("xproj", xproj), ("seq4", seq)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()
        for r in range(5):
            fn()
            ev[r + 1].record()
        torch.cuda.synchronize()
        out[name] = [ev[r].elapsed_time(ev[r + 1]) for r in range(5)]
    # checksums ON THE DEVICE (Gx alone is 17.7 GB: a host copy + its bytes object was 35 GB of host memory, and a box short
    # of it aborted the tool): sums of the raw 64-bit / 32-bit patterns, xor-folded -- equal bits give equal words
    def word(t, as_type):
        v = t.view(as_type).reshape(-1)
        return "%016x" % ((int(v.sum().item()) ^ int(v[::7].sum().item()) * 31) & 0xFFFFFFFFFFFFFFFF)
    hg = word(gx, torch.int64)
    hh = word(st["hout"], torch.int32)
    print("%d lines, %d rows (%s): xproj %.3f ms (min %.3f)  seq4 %.3f ms (min %.3f)  gx %s hout %s"
          % (nlines, rows, os.path.basename(os.environ.get("TA_HIP_LIB", "libta_hip.so")), np.mean(out["xproj"]), min(out["xproj"]),
             np.mean(out["seq4"]), min(out["seq4"]), hg, hh))


if __name__ == "__main__":
    main()
