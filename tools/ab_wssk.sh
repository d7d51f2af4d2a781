run() { timeout 400 python tools/bench_wssk.py $N conn=1 mode=lut < /dev/null 2>/dev/null | grep -o "\"us_levels\": [0-9]*"; }
timeout 900 python -m pytest tests/test_gpu_wssk.py -m gpu -x -q < /dev/null 2>&1 | tail -1
N=512
echo 512 default; run; run; run
