for c in 16 32 64; do echo chunk_images=$c; TA_PAGE_CHUNK_IMAGES=$c python tools/pages_img_time.py 64 2>&1 | grep -v amdgpu | tail -2; done
for c in 8 16 32; do echo chunk=$c; TA_PAGE_CHUNK=$c python - <<EOF2 2>&1 | grep -v amdgpu | tail -1
import sys; sys.path.insert(0,'.')
from tools import pages_bench as pb
import time, torch, numpy as np
from text_alignment_amd import alignToOCR as atocr
rec = pb.make_recognizer()
pages, trs = zip(*[pb.make_page(100 + k) for k in range(64)])
rp, rt = zip(*[pb.make_page(5100 + k, raw=True) for k in range(64)])
for name, P, T in (("norm", pages, trs), ("raw", rp, rt)):
    for _ in range(3): atocr.process_batch(list(P), list(T), rec, pb.PARAMS)
    torch.cuda.synchronize(); ts=[]
    for _ in range(10):
        t0=time.perf_counter(); atocr.process_batch(list(P), list(T), rec, pb.PARAMS); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    print(name, 64/np.median(ts), end="  ")
print()
EOF2
done
