cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-tests_all}
mkdir -p $O
export TMPDIR=/tmp
timeout -k 5 2400 python -m pytest tests -m gpu -q -W ignore -x < /dev/null > $O/gpu_tests_full.txt 2>&1
tail -5 $O/gpu_tests_full.txt
