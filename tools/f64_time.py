"""Device time of the float64 recurrence mode on the bench's OCR workload (1 920 synthetic lines): the input projection +
recurrence (`rec.run(lstm=True)`), mean of 3 after 2 warm-ups.  With a library built with -DTA_F64_PROFILE (TA_HIP_LIB
selects it) also the cycle counters of wave 0: tiles (MFMAs + cell update), data movement, barrier.

    python tools/f64_time.py [nlines]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402
from text_alignment_amd import _native, ocr            # noqa: E402
from tools import switches                               # noqa: E402

switches.apply()             # TA_* environment variables -> the product modules' attributes


def main():
    nlines = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
    rec = ocr.LineRecognizer(ocr.LineModel.random(7001, no=96), precision="f64")
    st = rec.prepare(bench.synthetic_lines(nlines, 8000))
    for _ in range(2):
        rec.run(st, lstm=True, output=False, decode=False)
    torch.cuda.synchronize()
    lib = _native.lib
    prof = getattr(lib, "ta_lstm_f64_profile", None)
    buf = (ctypes.c_ulonglong * 4)()
    if prof is not None:
        prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
        prof(buf, 1)
    reps = 3
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        rec.run(st, lstm=True, output=False, decode=False)
        ev[r + 1].record()
    torch.cuda.synchronize()
    ms = np.mean([ev[r].elapsed_time(ev[r + 1]) for r in range(reps)])
    print(f"{nlines} lines, {int(st['rows'])} timesteps: xproj + recurrence {ms:.2f} ms "
          f"({os.environ.get('TA_HIP_LIB', 'libta_hip.so')})")
    if prof is not None:
        prof(buf, 0)
        comp, move, bar, steps = [float(v) for v in buf]
        print(f"  wave 0, per step (cycles): tiles (MFMAs + cell) {comp / steps:.0f}, LDS / store / next loads "
              f"{move / steps:.0f}, barrier {bar / steps:.0f}; total {(comp + move + bar) / steps:.0f}")


if __name__ == "__main__":
    main()
