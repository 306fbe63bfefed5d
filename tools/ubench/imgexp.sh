for cfg in "8 2 0.005" "8 2 0.0005" "8 2 0.0001" "4 4 0.0005" "8 4 0.0005"; do python tools/pages_img_time.py 64 $cfg 2>&1 | tail -3; done
for cfg in "8 2 0.005" "8 2 0.0005" "4 4 0.0005"; do TA_PAGE_CHUNK_IMAGES=16 python tools/pages_img_time.py 64 $cfg 2>&1 | tail -3; done
