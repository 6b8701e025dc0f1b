"""Would an evaluation WITHOUT the Jacobian on rejected dog-leg trials shorten the elastic-plastic transition launches?  (Verdict of round 5, item 6: "only r is
needed to reject".)  CPU study with a traced build of the oracle (-DECM_TRACE prints every trust-region iteration with its reject flag): kinematically driven
FCC Voce RVE of N^3 elements, the ten preparation passes of the bench (the schedule's first ten steps: elastic, transition, plastic).  Per pass:
  * evaluations per point (mean / max), share of the trust-region iterations whose trial was rejected,
  * per 64-point wave (lane = element, a wave = 64 consecutive elements at one quadrature point - the element-blocked launch's mapping): the share of
    wave-iterations in which EVERY still-active lane rejects (the only ones in which a wave could skip the Jacobian half of the evaluation: a wave issues
    an instruction if one lane needs it) and the share in which at least one lane rejects (what a separate r-only code path would be issued for).
    python scripts/transition_reject_study.py [N=8]
Result (round 6, N = 8, 4 096 points; printed by the script, kept in profiles/r06_transition_reject_study.txt)."""
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

if os.environ.get("TR_STUDY_WORKER"):      # one traced pass in a child process (its stderr is the trace)
    z = np.load(os.environ["TR_STUDY_WORKER"])
    lib = C.CDLL(os.environ["TR_STUDY_LIB"]); p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    P = int(z["E"]) * int(z["Q"]); props = z["props"].copy()
    s1 = np.zeros(6 * P); sv1 = np.zeros(28 * P); cm = np.zeros(36 * P)
    a = [z[k].copy() for k in ("J", "G", "ve", "s0", "sv0")]
    os.environ["OMP_NUM_THREADS"] = "1"
    lib.orc_model_setup(0, 0, p(props), len(props), int(z["Q"]), int(z["E"]), int(z["n"]), 28, C.c_double(float(z["dt"])), C.c_double(298.0), p(a[0]), p(a[1]), p(a[2]), p(a[3]), p(a[4]),
                        p(s1), p(sv1), p(cm), None, 1, 0, 0)
    np.savez(os.environ["TR_STUDY_WORKER"] + ".out.npz", s1=s1, sv1=sv1)
    sys.exit(0)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tmp = tempfile.mkdtemp(prefix="tr_study_")
lib_trace = os.path.join(tmp, "liboracle_trace.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-DECM_TRACE", "-Wno-unused-variable", "-Wno-unknown-pragmas", "-shared", "-o", lib_trace, os.path.join(ROOT, "oracle", "oracle_capi.cpp")])
orc.build()
props = np.loadtxt(os.path.join(orc.REFDATA, "props_cp_voce.txt")).ravel()
rve = hipref.make_rve(orc, N); E, Q = rve["E"], rve["Q"]; P = E * Q
quats = hipref.random_quats(E)
hist = np.zeros(26); orc.lib().orc_hist_init(0, 0, orc._p(props), len(props), orc._p(hist))
sv0 = np.tile(np.concatenate([hist, [1.0, 0.0]]), P).reshape(P, 28); sv0[:, 9:13] = np.repeat(quats, Q, axis=0); sv0 = sv0.ravel().copy(); s0 = np.zeros(6 * P)
v = hipref.velocity_field(rve); ve = hipref.l_to_e(rve, v); x = rve["X"].copy()
J = np.zeros(9 * P)
pat = re.compile(r"it (\d+) res (\S+) res0 (\S+) delta (\S+) nr (\S+) sd (\S+) use_nr (\d) reject (\d)")
print(f"FCC Voce, {N}^3 elements, {P} points; wave = 64 consecutive elements at one quadrature point")
print("pass    dt   nfev mean  max | iterations  rejected  | wave-iterations  all active lanes reject  some lane rejects")
tot = np.zeros(5)
for ip, dt in enumerate([0.005, 0.195, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1]):      # the bench's kinematic preparation
    x = x + v * dt; orc.lib().orc_jacobians(1, E, orc._p(hipref.l_to_e(rve, x)), orc._p(J))
    state = os.path.join(tmp, "state.npz")
    np.savez(state, J=J, ve=ve, s0=s0, sv0=sv0, G=rve["G"], Q=Q, E=E, n=rve["n"], props=props, dt=dt)
    r = subprocess.run([sys.executable, __file__], env=dict(os.environ, TR_STUDY_WORKER=state, TR_STUDY_LIB=lib_trace, OMP_NUM_THREADS="1"), capture_output=True, text=True, check=True)
    out = np.load(state + ".out.npz"); s0 = out["s1"].copy(); sv0 = out["sv1"].copy()
    nfev = sv0.reshape(P, 28)[:, 3].astype(int)
    rows = [pat.match(l) for l in r.stderr.splitlines() if l.startswith("it ")]
    rej = [int(m.group(8)) for m in rows]
    pos = 0; per_pt = []
    for n in nfev:      # a point prints one line per iteration that did not converge: nfev - 2 lines (the first evaluation is the initial (r, J), the last one converges)
        k = max(n - 2, 0); per_pt.append(rej[pos:pos + k]); pos += k
    assert pos == len(rows), (pos, len(rows))
    # point index = e * Q + q (reference layout); a wave of the element-blocked launch = elements [64 b, 64 b + 64) at one q
    n_it = n_rej = w_it = w_all = w_some = 0
    for b in range((E + 63) // 64):
        for q in range(Q):
            lanes = [per_pt[e * Q + q] for e in range(64 * b, min(64 * b + 64, E))]
            for it in range(max(len(l) for l in lanes)):
                act = [l[it] for l in lanes if len(l) > it]
                w_it += 1; w_all += int(all(act)); w_some += int(any(act))
    for l in per_pt:
        n_it += len(l); n_rej += sum(l)
    tot += (n_it, n_rej, w_it, w_all, w_some)
    print(f"{ip + 1:4d} {dt:6.3f}   {nfev.mean():6.2f} {nfev.max():4d} | {n_it:10d}  {n_rej:7d} ({100.0 * n_rej / max(n_it, 1):4.1f} %) | {w_it:10d}  {w_all:8d} ({100.0 * w_all / max(w_it, 1):4.1f} %)  {w_some:8d} ({100.0 * w_some / max(w_it, 1):4.1f} %)")
print(f"all passes: {int(tot[1])} of {int(tot[0])} non-converging iterations rejected ({100 * tot[1] / tot[0]:.2f} %); wave-iterations in which every active lane rejects: "
      f"{int(tot[3])} of {int(tot[2])} ({100 * tot[3] / tot[2]:.2f} %), in which some lane rejects: {int(tot[4])} ({100 * tot[4] / tot[2]:.2f} %)")
