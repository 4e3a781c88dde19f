import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
hpgv = importlib.import_module("hpg-variant_amd")
V, N = 16384, 10000
rng = np.random.default_rng(1)
data = rng.choice(np.array([0, 1, 2, 255], np.uint8), size=(V, N), p=[0.5, 0.35, 0.14, 0.01])
e = hpgv.Engine(0)
ts = []
for r in range(4):
    t0 = time.perf_counter(); e.epi_set_dataset(data, N // 2, N // 2); ts.append(time.perf_counter() - t0)
print("set_dataset (164 MB) ms:", [round(x * 1e3, 1) for x in ts])
e.close()
