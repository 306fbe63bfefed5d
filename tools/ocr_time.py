"""Times K3 / K4 / K5 of the line recogniser in both modes.  Usage: python tools/ocr_time.py [nlines]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


class A(object):
    ocr_lines = int(sys.argv[1]) if len(sys.argv) > 1 else 1920


for prec in ("f32", "split"):
    r = bench.bench_ocr(A, 0, precision=prec)
    print(prec, "lines/s %.0f" % r["lines_per_s"], r["ms"], "frac of f32 MFMA peak %.3f" % r["roofline"]["frac"])
