import os, sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, exaconstit_amd.lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
props = np.loadtxt("tests/golden/refdata/props_cp_voce.txt").ravel()
rng = np.random.default_rng(20240928); q = rng.standard_normal((N**3, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
dts = np.array([0.1] * 2)
d = L.Driver.synthetic(N, props, q.ravel(), dts, assembly=1, nrls=True, order=2, bbar=True, newton=(25, 5e-5, 5e-10), krylov=(1000, 1e-7, 1e-27))
for ti in range(1, 3):
    t0 = time.time(); ok = d.step(ti); torch.cuda.synchronize(); tm = d.timers()
    print("step", ti, ok, round(time.time() - t0, 3), [list(map(int, a)) for a in d.stats()], d.avgs(0, 6)[-1, 2], {k: round(float(v), 1) for k, v in tm.items()}, flush=True)
