#!/bin/bash
# Evidence of the page-image leg (round 6, after the preprocessing stages became library calls): kernel trace of
# process_batch on 32 page images, the stage clocks, the host-stage timeline, the device's idle stretches and the A/B of
# where the pages lie.  Writes gpurun_out/ev_images/; copy what is to be judged into profiles/.
#   bash tools/evidence_images.sh
set -eo pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/ev_images
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_pages_images" -o pg -- python3 "$REPO/tools/pages_img_time.py" 32 8 1 > "$OUT/kt_pages_images.log" 2>&1
echo "page images kernel trace done"
cp $(find "$OUT/kt_pages_images" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats_pages_images.csv"
cd "$REPO"
export TA_BIND=1
{ echo "# TA_BIND=1 python tools/pages_ab.py 64 8 --images all   (bound to the GPU's NUMA node, as bench.py binds a rank)"; python tools/pages_ab.py 64 8 --images all 2>&1 | grep -v -i "warn\|amdgpu.ids"; } > "$OUT/pages_ab_images.txt"
{ echo "# TA_BIND=1 python tools/pages_img_stages.py 64"; python tools/pages_img_stages.py 64 2>&1 | grep -v -i "warn\|amdgpu.ids"; } > "$OUT/page_images_stages.txt"
{ echo "# TA_BIND=1 python tools/pages_host_stages.py 64 --images"; python tools/pages_host_stages.py 64 --images 2>&1 | grep -v -i "warn\|amdgpu.ids"; } > "$OUT/page_images_host_stages.txt"
{ echo "# TA_BIND=1 python tools/pages_timeline.py 64 --images --gaps"; python tools/pages_timeline.py 64 --images --gaps 2>&1 | grep -v -i "warn\|amdgpu.ids"; } > "$OUT/page_images_idle.txt"
tail -4 "$OUT/pages_ab_images.txt"
