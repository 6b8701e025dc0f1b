"""Rates of BASELINE config 5's two hot kernels on one MI355X (64^3 hex RVE, p = 2, B-bar, element assembly, FCC Voce): constitutive launch and PCG iteration on the
kinematically driven plastic state, with the same definitions as bench.py (928 algorithmic B/qpt for the constitutive launch; the matrix-free p = 2 action streams
18 16-byte pairs per point = 288 B/qpt + the element-average gradients and L-vector gather / scatter).  One JSON object on stdout (NOT the contract line of bench.py):
    python scripts/bench_config5.py [N=64] > gpurun_out/r05_config5_rates.json"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exaconstit_amd.lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
props = np.loadtxt(os.path.join(ROOT, "tests/golden/refdata/props_cp_voce.txt")).ravel()
PREP = [0.005, 0.195, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1]
rng = np.random.default_rng(20240928); q = rng.standard_normal((N ** 3, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
d = L.Driver.synthetic(N, props, q.ravel(), np.array(PREP), assembly=1, order=2, bbar=True, nrls=True)
d.bench_prepare(PREP[:1], advance=False); d.bench_model(2); me = d.bench_model(20)
d.bench_prepare(PREP); d.bench_model(30); m = d.bench_model(50)
d.bench_pcg(20); pc = d.bench_pcg(200)
P = L.exa_driver_local_qpts(d.h); h = d.nfev_hist()
kms = m["kernel_ms"] / 50; it_ms = pc["pcg_ms"] / pc["iters"]; ap_ms = pc["apply_ms"] / 200
out = {"workload": f"BASELINE config 5 shape on one GPU: {N}^3 hex RVE p=2 ({P} qpts, {L.exa_driver_local_dofs(d.h)} dofs), B-bar, element assembly served matrix-free, FCC Voce, kinematic plastic state",
       "constitutive": {"kernel_ms": kms, "qpt_updates_per_s": P / (kms * 1e-3), "frac_of_hbm_peak_on_928_B_per_qpt": 928.0 * P / (kms * 1e-3) / 8e12,
                        "elastic_kernel_ms": me["kernel_ms"] / 20, "nfev_mean": float((h * np.arange(64)).sum() / h.sum()), "failed_points": m["failed"]},
       "pcg": {"ms_per_iter": it_ms, "iters_per_s": 1e3 / it_ms, "action_ms": ap_ms, "action_gbs_on_288_B_per_qpt": 288.0 * P / (ap_ms * 1e-3) / 1e9,
               "action_frac_of_hbm_peak": 288.0 * P / (ap_ms * 1e-3) / 8e12},
       "library": {"kernel_build_id": L.exa_kernel_build_id().decode()}}
print(json.dumps(out))
d.close()
