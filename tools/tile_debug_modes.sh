for m in 0 1 2 4 6; do echo "mode $m"; VQA_TILE_DEBUG=$m timeout -k 10 100 python3 tools/bench_hbm.py 20 8 64 2>&1 | grep tiled; done
