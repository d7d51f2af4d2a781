"""A/B builds of the same ABI: python tools/build_variant.py <name> -DFOO=1 ... -> invesalius3_amd/libivx_<name>.so (objects under
invesalius3_amd/build_<name>/); run with IVX_LIB_PATH=invesalius3_amd/libivx_<name>.so."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from invesalius3_amd import build as b

name, defs = sys.argv[1], sys.argv[2:]
obj = os.path.join(b.HERE, "build_" + name)
os.makedirs(obj, exist_ok=True)
hipcc = b._hipcc()
objs, jobs = [], []
for src in b.sources():
    o = os.path.join(obj, os.path.basename(src)[:-4] + ".o")
    objs.append(o)
    jobs.append([hipcc, *b.FLAGS, *defs, "-c", src, "-o", o])
with ThreadPoolExecutor(8) as ex:
    for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
        if r.returncode:
            raise SystemExit(r.stderr[-4000:])
lib = os.path.join(b.HERE, "libivx_%s.so" % name)
subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=" + b.ARCH, *objs, "-ldl", "-o", lib], check=True)
print(lib)
