import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from scipy.ndimage import generate_binary_structure
from invesalius3_amd import invesalius_rs as ff
img = np.full((32, 64, 128), 100, np.int16)
out = np.zeros(img.shape, np.uint8)
t = time.time()
try:
    ff.floodfill_threshold(img, [(0, 0, 0)], 50, 150, 1, generate_binary_structure(3, 3), out)
    print("ok", out.sum(), img.size, time.time() - t)
except Exception as e:
    print("ERR", e, time.time() - t)
