run() { timeout 400 python tools/bench_wssk.py $N conn=1 mode=lut < /dev/null 2>/dev/null | grep -o "\"us_levels\": [0-9]*"; }
N=1024
echo 1024 default; run
echo 1024 per_wg256; IVX_SK_PER_WG=256 run
echo 1024 per_wg256 res2; IVX_SK_PER_WG=256 IVX_SK_RES_PER_CU=2 run
echo 1024 per_wg256 res4; IVX_SK_PER_WG=256 IVX_SK_RES_PER_CU=4 run
echo 1024 per_wg512 res4; IVX_SK_PER_WG=512 IVX_SK_RES_PER_CU=4 run
echo 1024 per_wg128 res4; IVX_SK_PER_WG=128 IVX_SK_RES_PER_CU=4 run
