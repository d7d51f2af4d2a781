run() { timeout 400 python tools/bench_wssk.py $N conn=1 mode=lut < /dev/null 2>/dev/null | grep -o "\"us_levels\": [0-9]*"; }
timeout 600 python -m pytest tests/test_gpu_wssk.py -m gpu -x -q -k "generation0" < /dev/null 2>&1 | tail -2
N=1024
echo 1024 merge; IVX_SK_SORT=merge run
echo 1024 pairs256 then merge; IVX_SK_PAIR_CHUNKS=128 run
N=512
echo 512 merge; IVX_SK_SORT=merge run
echo 512 default; run
