"""Diagnostic (GPU): how the local-solver evaluation counts (state slot 3, nFEval) are distributed over the lanes of a wave for different
lane -> (element, point) mappings of the constitutive launch, on the bench's kinematic state and on real Newton/PCG solves.
A wave pays for its slowest lane: the figure of merit is the mean over waves of the per-wave maximum.
usage: python scripts/nfev_mapping_study.py [N_kin] [N_solve] [solve_steps] [model]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import exaconstit_amd.lib as L

N_KIN = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N_SOL = int(sys.argv[2]) if len(sys.argv) > 2 else 64
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 14
MODEL = sys.argv[4] if len(sys.argv) > 4 else "fcc_voce"
PREP_DTS = [0.005, 0.195, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1]
xt, sl = MODEL.split("_", 1)
pfile = {"voce": "props_cp_voce.txt", "voce_nl": "props_cp_vocenl.txt", "kmdd": "props_cp_mts.txt"}[sl]
mk = dict(bcc=(xt == "bcc"), slip={"voce": 0, "voce_nl": 1, "kmdd": 2}[sl], temp_k=298.0)
props = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", pfile)).ravel()


def quats_for(N):
    rng = np.random.default_rng(20240928)
    q = rng.standard_normal((N ** 3, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.ravel()


def wave_stats(nf, Q=8):
    """nf: [E][Q].  mean over waves of max over the wave's lanes, for lane sets of ne elements x nq points."""
    E = nf.shape[0]; E64 = (E // 64) * 64
    a = nf[:E64]
    out = {"mean": float(nf.mean()), "max": int(nf.max())}
    for ne in (64, 32, 16, 8):
        nq = 64 // ne
        out[f"{ne}x{nq}"] = float(a.reshape(-1, ne, Q // nq, nq).max(axis=(1, 3)).mean())
    # fraction of elements whose points do not all share one count, fraction of elements holding a point above the mode
    mode = np.bincount(nf.astype(int).ravel()).argmax()
    out["mode"] = int(mode)
    out["frac_points_above_mode"] = float((nf > mode).mean())
    out["frac_elems_with_point_above_mode"] = float((nf.max(axis=1) > mode).mean())
    out["frac_elems_mixed"] = float((nf.max(axis=1) != nf.min(axis=1)).mean())
    return out


res = {"model": MODEL}
drv = L.Driver.synthetic(N_KIN, props, quats_for(N_KIN), np.array(PREP_DTS), **mk)
drv.bench_prepare(PREP_DTS)
m = drv.bench_model(5)
nf = drv.qf_component(1, 3).reshape(-1, 8)
res["kinematic_state"] = {"N": N_KIN, "kernel_ms": m["kernel_ms"] / 5, **wave_stats(nf)}
drv.close()
print(json.dumps(res["kinematic_state"]), flush=True)

sched = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", "custom_dt.txt")).ravel()[:STEPS]
drv = L.Driver.synthetic(N_SOL, props, quats_for(N_SOL), sched, **mk)
rows = []
for ti in range(1, STEPS + 1):
    drv.reset = None
    L.exa_driver_reset_timers(drv.h)
    assert drv.step(ti)
    tm = drv.timers(); nw, kr, mc = drv.stats()
    nf = drv.qf_component(0, 3).reshape(-1, 8)   # after UpdateModel the end-of-step state is the begin-of-step one
    row = {"step": ti, "model_calls": int(mc[-1]), "kernel_ms_per_call": tm["model_ms"] / max(int(mc[-1]), 1),
           "qpt_per_s_in_kernel": 8 * N_SOL ** 3 * int(mc[-1]) / (tm["model_ms"] * 1e-3), **wave_stats(nf)}
    rows.append(row)
    print(json.dumps(row), flush=True)
res["solve"] = {"N": N_SOL, "rows": rows}
drv.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"nfev_mapping_{MODEL}_{N_KIN}_{N_SOL}.json"), "w"), indent=1)
