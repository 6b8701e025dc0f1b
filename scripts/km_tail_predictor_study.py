"""Is there a predictor of the REMAINING evaluations of a point that the capped Kocks-Mecking launch hands to the dense tail launch?  (the tail
is bound by divergence - an active tail wave pays its slowest lane.)  CPU study with a traced build of the oracle (-DECM_TRACE prints every trust-region
iteration): kinematically driven FCC (0) / BCC (1) Kocks-Mecking RVE of N^3 elements in the plastic regime, cap K; features available at the hand-over
(accepted residual, trust radius, Newton-step length, last trial rejected) against the evaluations the point still needs; figure of merit = mean over
64-lane waves of the maximum remaining count when the list is ordered by the predictor (random order = today, sorted by the truth = bound).
    python scripts/km_tail_predictor_study.py [xtal=0] [N=8] [K=6]
Result (round 5, FCC, N = 8, K = 6: 590 of 4 096 points listed, remaining mean 4.0 / max 8): random order 7.64, sorted by the truth 4.22, by log10(res/tol) 6.22,
by a linear fit of the four features 6.00 (correlation 0.80), 4 / 8 coarse buckets of the fit 6.44 / 6.33: about a fifth of the tail's evaluation work, i.e.
~5 % of the FCC launch - not built (FCC Kocks-Mecking is not a BASELINE configuration; BCC lists 1.7 % of its points)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hipref  # noqa: E402
import orc  # noqa: E402

if os.environ.get("KM_STUDY_WORKER"):      # traced pass in a child process (its stderr is the trace)
    z = np.load(os.environ["KM_STUDY_WORKER"]); xtal = int(sys.argv[1])
    lib = C.CDLL(os.environ["KM_STUDY_LIB"]); p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    P = int(z["E"]) * int(z["Q"]); props = z["props"].copy()
    s1 = np.zeros(6 * P); sv1 = np.zeros(28 * P); cm = np.zeros(36 * P)
    a = [z[k].copy() for k in ("J", "G", "ve", "s0", "sv0")]
    lib.orc_model_setup(xtal, 2, p(props), len(props), int(z["Q"]), int(z["E"]), int(z["n"]), 28, C.c_double(0.1), C.c_double(298.0), p(a[0]), p(a[1]), p(a[2]), p(a[3]), p(a[4]),
                        p(s1), p(sv1), p(cm), None, 1, 0, 0)
    np.save(os.environ["KM_STUDY_WORKER"] + ".nfev.npy", sv1.reshape(P, 28)[:, 3])
    sys.exit(0)

xtal = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
tmp = tempfile.mkdtemp(prefix="km_study_")
lib_trace = os.path.join(tmp, "liboracle_trace.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-fopenmp", "-DECM_TRACE", "-Wno-unused-variable", "-shared", "-o", lib_trace, os.path.join(ROOT, "oracle", "oracle_capi.cpp")])
orc.build()
props = np.loadtxt(os.path.join(orc.REFDATA, "props_cp_mts.txt")).ravel()
rve = hipref.make_rve(orc, N); P = rve["E"] * rve["Q"]
quats = hipref.random_quats(rve["E"])
hist = np.zeros(26); orc.lib().orc_hist_init(xtal, 2, orc._p(props), len(props), orc._p(hist))
sv0 = np.tile(np.concatenate([hist, [1.0, 0.0]]), P).reshape(P, 28); sv0[:, 9:13] = np.repeat(quats, rve["Q"], axis=0); sv0 = sv0.ravel().copy(); s0 = np.zeros(6 * P)
v = hipref.velocity_field(rve); ve = hipref.l_to_e(rve, v); x = rve["X"].copy()
s1 = np.zeros(6 * P); sv1 = np.zeros(28 * P); cm = np.zeros(36 * P); J = np.zeros(9 * P)
for dt in [0.005, 0.195, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1]:      # the bench's kinematic preparation
    x = x + v * dt; orc.lib().orc_jacobians(1, rve["E"], orc._p(hipref.l_to_e(rve, x)), orc._p(J))
    orc.lib().orc_model_setup(xtal, 2, orc._p(props), len(props), rve["Q"], rve["E"], rve["n"], 28, C.c_double(dt), C.c_double(298.0), orc._p(J), orc._p(rve["G"]), orc._p(ve),
                              orc._p(s0), orc._p(sv0), orc._p(s1), orc._p(sv1), orc._p(cm), None, 1, 0, 0)
    s0[:] = s1; sv0[:] = sv1
state = os.path.join(tmp, "state.npz")
np.savez(state, J=J, ve=ve, s0=s0, sv0=sv0, G=rve["G"], Q=rve["Q"], E=rve["E"], n=rve["n"], props=props)
r = subprocess.run([sys.executable, __file__, str(xtal)], env=dict(os.environ, KM_STUDY_WORKER=state, KM_STUDY_LIB=lib_trace), capture_output=True, text=True, check=True)
nfev = np.load(state + ".nfev.npy").astype(int)
pat = re.compile(r"it (\d+) res (\S+) res0 (\S+) delta (\S+) nr (\S+) sd (\S+) use_nr (\d) reject (\d)")
rows = [pat.match(l) for l in r.stderr.splitlines() if l.startswith("it ")]
rows = [(int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), int(m.group(8))) for m in rows]
pos = 0; pts = []
for n in nfev:      # a point prints one line per iteration that did not converge: nfev - 2 lines
    k = max(n - 2, 0); pts.append(rows[pos:pos + k]); pos += k
assert pos == len(rows)
tol = props[2]
listed = [i for i in range(len(nfev)) if nfev[i] > K]
rem = np.array([nfev[i] - K for i in listed])
print(f"{len(nfev)} points, {len(listed)} listed at cap {K} ({100 * len(listed) / len(nfev):.1f} %); remaining evaluations mean {rem.mean():.2f} max {rem.max()}  hist {np.bincount(rem)}")
feat = []
for i in listed:
    it, res, res0, delta, nr, rej = pts[i][K - 2]
    feat.append((np.log10((res0 if rej else res) / tol), np.log10(delta), np.log10(max(nr, 1e-300)), rej))
feat = np.array(feat)


def wave_cost(order, W=64):
    rr = rem[order]; nw = len(rr) // W
    return float(np.mean([rr[w * W:(w + 1) * W].max() for w in range(nw)])) if nw else float("nan")


rng = np.random.default_rng(0)
print(f"mean wave maximum: random order {np.mean([wave_cost(rng.permutation(len(rem))) for _ in range(50)]):.2f} | sorted by the truth {wave_cost(np.argsort(rem)):.2f}")
for name, col in (("log10(res/tol)", 0), ("log10(delta)", 1), ("log10|Newton step|", 2)):
    print(f"  sorted by {name:20s} {wave_cost(np.argsort(feat[:, col])):.2f}   correlation {np.corrcoef(feat[:, col], rem)[0, 1]:+.3f}")
A = np.c_[feat, np.ones(len(rem))]
w, *_ = np.linalg.lstsq(A, rem, rcond=None); pred = A @ w
print(f"  sorted by the linear fit      {wave_cost(np.argsort(pred)):.2f}   correlation {np.corrcoef(pred, rem)[0, 1]:+.3f}   weights {np.round(w, 3)}")
for B in (2, 4, 8):
    b = np.digitize(pred, np.quantile(pred, np.linspace(0, 1, B + 1)[1:-1]))
    print(f"  {B} buckets of the fit (list order inside a bucket): {wave_cost(np.argsort(b, kind='stable')):.2f}")
