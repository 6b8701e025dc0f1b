"""Diagnostic (GPU): does a point's local-solver evaluation count persist from one load step to the next?  If it does, ordering the elements
by the previous step's counts would put the lanes that need the extra evaluation into the same waves (a wave pays for its slowest lane).
Prints, for consecutive steps of a real Newton/PCG solve: P(n_k > mode | n_{k-1} > mode), the mean wave maximum of the natural element order
and of the order sorted by the previous step's per-element maximum / sum.
usage: python scripts/nfev_persistence_study.py [N] [steps] [model]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import exaconstit_amd.lib as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
MODEL = sys.argv[3] if len(sys.argv) > 3 else "fcc_voce"
xt, sl = MODEL.split("_", 1)
pfile = {"voce": "props_cp_voce.txt", "voce_nl": "props_cp_vocenl.txt", "kmdd": "props_cp_mts.txt"}[sl]
mk = dict(bcc=(xt == "bcc"), slip={"voce": 0, "voce_nl": 1, "kmdd": 2}[sl], temp_k=298.0)
props = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", pfile)).ravel()
rng = np.random.default_rng(20240928)
q = rng.standard_normal((N ** 3, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
sched = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", "custom_dt.txt")).ravel()[:STEPS]
drv = L.Driver.synthetic(N, props, q.ravel(), sched, **mk)


def wave_max(nf, order=None):
    a = nf if order is None else nf[order]
    E64 = (a.shape[0] // 64) * 64
    return float(a[:E64].reshape(-1, 64, 8).max(axis=1).mean())      # wave = 64 elements x one point index


prev = None
for ti in range(1, STEPS + 1):
    assert drv.step(ti)
    nf = drv.qf_component(0, 3).reshape(-1, 8)      # after the step the converged launch's state is the begin-of-step state
    row = {"step": ti, "mean": float(nf.mean()), "natural": wave_max(nf)}
    mode = np.bincount(nf.astype(int).ravel()).argmax()
    row["mode"] = int(mode); row["frac_above"] = float((nf > mode).mean())
    if prev is not None:
        pm = np.bincount(prev.astype(int).ravel()).argmax()
        a = prev > pm; b = nf > mode
        row["P(above|prev above)"] = float((a & b).sum() / max(a.sum(), 1)); row["P(above|prev not)"] = float((~a & b).sum() / max((~a).sum(), 1))
        row["sorted_by_prev_elem_max"] = wave_max(nf, np.argsort(prev.max(axis=1), kind="stable"))
        row["sorted_by_prev_elem_sum"] = wave_max(nf, np.argsort(prev.sum(axis=1), kind="stable"))
        row["sorted_by_own_elem_sum (bound)"] = wave_max(nf, np.argsort(nf.sum(axis=1), kind="stable"))
    print(json.dumps(row), flush=True)
    prev = nf
