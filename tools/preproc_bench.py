"""Times the device page preprocessing + line finding (csrc/ta_preproc.hip) against the host
checker (oracle/preproc_ref.py) on a synthetic page.  python tools/preproc_bench.py [nlines]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch
    from test_preprocessing import _synthetic_page
    from text_alignment_amd import preproc_gpu as G
    from oracle import preproc_ref as H
    nlines = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    img, _ = _synthetic_page(nlines, angle=1.5)
    G.find_lines(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = G.find_lines(img)
    torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / 3
    t1 = time.perf_counter()
    H.find_lines(img)
    host = time.perf_counter() - t1
    print({"page": img.shape, "strips": len(out[3]), "device_s": dev, "host_s_one_core": host})


if __name__ == "__main__":
    main()
