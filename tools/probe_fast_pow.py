import numpy as np, sys
sys.path.insert(0,'.')
from invesalius3_amd import _lib as L
from invesalius3_amd.device import DeviceBuffer, c64
from oracle import oracle
oracle.build()
n = 1 << 23
rng = np.random.default_rng(17)
x = rng.random(n, dtype=np.float32)
x[: n // 4] = (1.0 - rng.random(n // 4) ** 4).astype(np.float32)
x[n // 4: n // 2] = (rng.random(n // 4) ** 6).astype(np.float32)
y = rng.uniform(0.01, 64.0, n).astype(np.float32)
y[::3] = rng.choice(np.array([0.5, 1.5, 2.0, 3.0, 3.3, 8.0, 64.0], np.float32), len(y[::3]))
x[(x > 0) & (x < 2.0 ** -24)] = 2.0 ** -24
want = oracle.powf_array(x, y).astype(np.float64)
dx_, dy_, do_ = DeviceBuffer(n * 4), DeviceBuffer(n * 4), DeviceBuffer(n * 4)
dx_.upload(x); dy_.upload(y)
out = {}
for variant in (2, 3):
    L.check(L.lib().ivx_dev_powf(dx_.ptr, dy_.ptr, do_.ptr, c64(n), variant, None)); L.synchronize()
    out[variant] = do_.download((n,), np.float32).astype(np.float64)
fast, rel = out[2], out[3]
err = np.abs(fast - want)
m = fast > 1e-30
used = err[m] / (rel[m] * fast[m])
i = np.argmax(used)
print("worst ratio", used.max(), "x", x[m][i], "y", y[m][i], "fast", fast[m][i], "want", want[m][i], "rel", rel[m][i])
yl = np.abs(y[m] * np.log2(x[m].astype(np.float64)))
relerr = err[m] / fast[m]
for lo, hi in ((0,1),(1,4),(4,16),(16,64),(64,160)):
    s = (yl>=lo)&(yl<hi)
    if s.any(): print(lo,hi, "max relerr %.3e  = %.2f ulp(2^-23);  relerr/(|y|+4)/2^-21 max %.3f" % (relerr[s].max(), relerr[s].max()/2**-23, (relerr[s]/((yl[s]+4)*2**-21)).max()))
