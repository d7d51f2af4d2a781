cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g
mkdir -p $O
df -h /tmp | tail -1 > $O/tmpfs.txt
IVX_HOST_TIMING=1 timeout -k 5 600 python - > $O/host_timing.txt 2>&1 <<'PY'
import sys, os, tempfile, time
sys.path.insert(0, ".")
import numpy as np
from scipy.ndimage import generate_binary_structure
from bench import synth_v512
from invesalius3_amd import watershed_process as wp
from tools.bench_wsift import markers_for
import warnings
warnings.simplefilter("ignore")
img = synth_v512((512, 512, 512))
mk = markers_for(img).astype(np.int16)
fd, tfile = tempfile.mkstemp(suffix=".dat"); os.close(fd)
np.memmap(tfile, shape=img.shape, dtype="uint8", mode="w+").flush()
s6 = generate_binary_structure(3, 1)
for alg in ("Watershed IFT", "Watershed"):
    for rep in range(3):
        print("----", alg, rep, file=sys.stderr)
        t = time.perf_counter()
        wp.do_watershed(img, mk, tfile, img.shape, s6, alg, (3, 3, 3), True, 300, 400, None)
        print("total %.2f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
PY
tail -60 $O/host_timing.txt; cat $O/tmpfs.txt
