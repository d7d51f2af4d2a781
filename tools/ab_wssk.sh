run() { timeout 400 python tools/bench_wssk.py $N conn=1 mode=lut < /dev/null 2>/dev/null | grep -o "\"us_levels\": [0-9]*"; }
IVX_SK_LOCAL=1 timeout 900 python -m pytest tests/test_gpu_wssk.py -m gpu -x -q < /dev/null 2>&1 | tail -1
N=512
echo 512 default; run
echo 512 local; IVX_SK_LOCAL=1 run; IVX_SK_LOCAL=1 run
echo 512 local wgs256; IVX_SK_LOCAL=1 IVX_SK_LOCAL_WGS=256 run
echo 512 local wgs64; IVX_SK_LOCAL=1 IVX_SK_LOCAL_WGS=64 run
