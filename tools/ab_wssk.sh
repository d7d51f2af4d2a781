run() { timeout 400 python tools/bench_wssk.py $N conn=1 mode=lut < /dev/null 2>/dev/null | grep -o "\"us_levels\": [0-9]*\|\"tile_rounds\": [0-9]*" | tr '\n' ' '; echo; }
timeout 900 python -m pytest tests/test_gpu_wssk.py -m gpu -x -q < /dev/null 2>&1 | tail -1
N=512
echo 512; run; run
N=1024
echo 1024; run
