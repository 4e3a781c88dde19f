import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
hpgv = importlib.import_module("hpg-variant_amd")
V, N, K = 16384, 10000, 10
rng = np.random.default_rng(1)
nA = nU = N // 2
data = rng.choice(np.array([0, 1, 2, 255], np.uint8), size=(V, N), p=[0.5, 0.35, 0.14, 0.01])
e = hpgv.Engine(0)
t0 = time.perf_counter(); e.epi_set_dataset(data, nA, nU); t1 = time.perf_counter()
ts = []
for r in range(5):
    fold = np.empty(N, np.int32)
    fold[rng.permutation(nA)] = np.arange(nA) % K
    fold[nA + rng.permutation(nU)] = np.arange(nU) % K
    t2 = time.perf_counter(); e.epi_set_folds(fold, K); ts.append(time.perf_counter() - t2)
print("set_dataset %.1f ms, set_folds %s ms" % ((t1 - t0) * 1e3, [round(x * 1e3, 2) for x in ts]))
e.close()
