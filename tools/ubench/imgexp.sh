for cfg in "16 0" "16 1" "16 2" "32 1" "16 0" "16 1" "32 0" "32 1"; do
  set -- $cfg
  echo -n "chunk $1 ahead $2: "; TA_PAGE_CHUNK_IMAGES=$1 TA_PB_LINES_AHEAD=$2 python tools/pages_ab.py 64 8 --images 2>&1 | tail -1
done
