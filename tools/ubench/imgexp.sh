for cfg in "16 8 2" "16 4 4" "32 8 4" "24 8 3" "32 4 4" "16 8 2" "32 8 4" "16 4 4"; do
  set -- $cfg
  echo -n "chunk $1 batch $2 threads $3: "; TA_PAGE_CHUNK_IMAGES=$1 TA_PP_BATCH=$2 TA_PP_THREADS=$3 python tools/pages_ab.py 64 8 --images 2>&1 | tail -1
done
