"""The drop-in (MFEM adapter) route alone: python scripts/adapter_route.py [--n 128] [--model fcc_voce] [--steps 20] [--iters 20] [--solve-steps 0]
Kinematic drive (default) or a real solve to the state, then exa_driver_bench_adapter_route; one JSON line.  EXA_LIB selects a timing variant."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=128); ap.add_argument("--model", default="fcc_voce"); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--iters", type=int, default=20); ap.add_argument("--solve-steps", type=int, default=0); ap.add_argument("--repeat", type=int, default=1)
a = ap.parse_args()
import exaconstit_amd.lib as L
PREP = [0.005, 0.195, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1]
xt, sl = a.model.split("_", 1)
pfile = {"voce": "props_cp_voce.txt", "voce_nl": "props_cp_vocenl.txt", "kmdd": "props_cp_mts.txt"}[sl]
props = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", pfile)).ravel()
rng = np.random.default_rng(20240928)
quats = rng.standard_normal((a.n ** 3, 4)); quats /= np.linalg.norm(quats, axis=1, keepdims=True)
sched = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", "custom_dt.txt")).ravel()[:max(a.solve_steps, 1)] if a.solve_steps > 0 else np.array(PREP)
d = L.Driver.synthetic(a.n, props, quats.ravel(), sched, bcc=(xt == "bcc"), slip={"voce": 0, "voce_nl": 1, "kmdd": 2}[sl])
if a.solve_steps > 0:
    for ti in range(1, a.solve_steps + 1):
        assert d.step(ti, commit=ti < a.solve_steps)
else:
    d.bench_prepare(PREP)
out = None
for _ in range(a.repeat):
    out = d.bench_adapter_route(a.steps, a.iters)
out.update(lib=L.LIB_PATH, kernel_build_id=L.exa_kernel_build_id().decode(), n=a.n, model=a.model, state="solve %d steps" % a.solve_steps if a.solve_steps else "kinematic")
print(json.dumps(out))
d.close()
