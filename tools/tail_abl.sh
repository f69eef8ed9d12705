for m in 1 3 5 7 9 17 33 65 129 193 15 47 255; do echo "mode $m"; ST_TAIL_MODE=$m timeout -k 10 100 python tools/tail_bench.py 2>&1 | cut -c1-60; done
