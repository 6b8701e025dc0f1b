"""Per-launch time of the fused constitutive kernel at p = 2 (config-5 shape: N^3 elements, 27 points each) on the kinematically driven plastic state, beside
p = 1 at the same point count: python scripts/p2_model_time.py [N=64]
Round 5, one MI355X, 7.08 M points: p = 2 plastic 2.59 ms (0.366 ns/pt), elastic 2.25 ms; p = 1 (96^3) plastic 1.74 ms (0.246 ns/pt), elastic 1.28 ms: the 27-node
gathers and their sum-factorised contractions, repeated by each of an element's 27 points, cost ~0.9 ms of the p = 2 launch.  An element-level pre-pass
(gather once per element, write L per point) would trade them for ~144 B/pt of HBM traffic (estimate 2.6 -> 2.0 ms); not built - in a config-5 solve the
constitutive launches are 1 % of the GPU time (PCG action 0.73 ms x thousands of iterations per step)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exaconstit_amd.lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
props = np.loadtxt(os.path.join(ROOT, "tests/golden/refdata/props_cp_voce.txt")).ravel()
PREP = [0.005, 0.195, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1]
for order, n, bbar, asm in ((2, N, True, 1), (2, N, False, 0), (1, int(round(N * 1.5)), False, 0)):
    rng = np.random.default_rng(20240928); q = rng.standard_normal((n ** 3, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    d = L.Driver.synthetic(n, props, q.ravel(), np.array(PREP), assembly=asm, order=order, bbar=bbar)
    d.bench_prepare(PREP[:1], advance=False); d.bench_model(2); me = d.bench_model(10)
    d.bench_prepare(PREP); d.bench_model(20); m = d.bench_model(20)
    P = L.exa_driver_local_qpts(d.h); h = d.nfev_hist()
    mean = float((h * np.arange(64)).sum() / h.sum())
    print(f"p={order} N={n} bbar={bbar}: {P} points  plastic {m['kernel_ms'] / 20:.3f} ms = {m['kernel_ms'] / 20 / P * 1e6:.4f} ns/pt   elastic {me['kernel_ms'] / 10:.3f} ms = {me['kernel_ms'] / 10 / P * 1e6:.4f} ns/pt   nfev mean {mean:.2f}  failed {m['failed']}", flush=True)
    d.close()
