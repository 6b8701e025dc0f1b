"""Diagnostic (GPU + oracle): agreement of the local-solver evaluation count (state slot 3, ExaCMech's nFEval) between the HIP kernel and the
oracle, point by point, on a kinematically driven RVE through the elastic-plastic transition.  Both sides start every pass from the SAME
begin-of-step state (the GPU's), so a difference is a difference of the iteration path of that pass, not an accumulated one.
The kernel's Newton step is two block Gauss-Seidel sweeps on the exact diagonal-block inverses (relative step error ~1e-8) where the library
solves the 8 x 8 system exactly, and the kernel compares squared norms: the counts can differ where an iterate lands within ~1e-8 of a
decision threshold of the trust-region logic.
usage: python scripts/nfev_agreement.py [N] > profiles/rNN_nfev_agreement.txt"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import exaconstit_amd.lib as L
import hipref
import orc
from hipref import ptr

orc.build()
dev = hipref.Dev()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
MODELS = [("fcc_voce", L.EXA_FCC_VOCE, 0, 0, "props_cp_voce.txt"), ("bcc_voce", L.EXA_BCC_VOCE, 1, 0, "props_cp_voce.txt"),
          ("fcc_voce_nl", L.EXA_FCC_VOCE_NL, 0, 1, "props_cp_vocenl.txt"), ("fcc_kmdd", L.EXA_FCC_KMDD, 0, 2, "props_cp_mts.txt"),
          ("bcc_kmdd", L.EXA_BCC_KMDD, 1, 2, "props_cp_mts.txt")]
rve = hipref.make_rve(orc, N)
E, Q, n = rve["E"], rve["Q"], rve["n"]; P = E * Q
quats = hipref.random_quats(E)
v = hipref.velocity_field(rve); ve = hipref.l_to_e(rve, v)
orc.lib().orc_set_threads(8)
print(f"# {N}^3 elements, {P} points per pass, 10 passes (dt 0.005, 0.195, 8 x 0.1); lines: model, pass, GPU mean, oracle mean, points that differ, by how much")
for name, mid, xtal, kin, pfile in MODELS:
    props = np.loadtxt(os.path.join(orc.REFDATA, pfile)).ravel()
    ctx = L.Context(mid, props, 298.0, 1, E)
    d_q = dev.up(quats.ravel()); sv0 = dev.zeros(28 * P); sv1 = dev.zeros(28 * P); s0 = dev.zeros(6 * P); s1 = dev.zeros(6 * P); cm = dev.zeros(36 * P); dJ = dev.zeros(9 * P)
    ctx.check(L.exa_init_state(ctx.h, ptr(sv0), ptr(d_q), None))
    d_ve = dev.up(ve); x = rve["X"].copy()
    tot = 0; diff = 0; hist = {}
    for ip, dt in enumerate([0.005, 0.195] + [0.1] * 8):
        x = x + v * dt
        d_xe = dev.up(hipref.l_to_e(rve, x))
        ctx.check(L.exa_jacobians(ctx.h, ptr(d_xe), ptr(dJ), None))
        ctx.check(L.exa_model_setup(ctx.h, dt, ptr(dJ), ptr(d_ve), ptr(s0), ptr(sv0), ptr(s1), ptr(sv1), ptr(cm), None))
        assert ctx.check(L.exa_model_status(ctx.h, None)) == 0
        nf_g = sv1.cpu().numpy().reshape(P, 28)[:, 3]
        J = dJ.cpu().numpy(); h_s0 = s0.cpu().numpy(); h_sv0 = sv0.cpu().numpy()
        o_s1 = np.zeros(6 * P); o_sv1 = np.zeros(28 * P); o_cm = np.zeros(36 * P)
        nfail = orc.lib().orc_model_setup(xtal, kin, orc._p(props), len(props), Q, E, n, 28, C.c_double(dt), C.c_double(298.0), orc._p(J), orc._p(rve["G"]), orc._p(ve),
                                          orc._p(h_s0), orc._p(h_sv0), orc._p(o_s1), orc._p(o_sv1), orc._p(o_cm), None, 1, 0, 0)
        assert nfail == 0
        nf_o = o_sv1.reshape(P, 28)[:, 3]
        d = (nf_g - nf_o).astype(int)
        nd = int(np.count_nonzero(d)); tot += P; diff += nd
        for k, c in zip(*np.unique(d[d != 0], return_counts=True)):
            hist[int(k)] = hist.get(int(k), 0) + int(c)
        print(f"{name:12s} pass {ip + 1:2d}  mean {nf_g.mean():6.3f} {nf_o.mean():6.3f}  differ {nd:7d} = {100.0 * nd / P:7.4f} %  " + " ".join(f"{int(k):+d}:{int(c)}" for k, c in zip(*np.unique(d[d != 0], return_counts=True))))
        s0, s1 = s1, s0; sv0, sv1 = sv1, sv0
    print(f"{name:12s} TOTAL agreement {100.0 * (1 - diff / tot):8.4f} %  ({diff} of {tot} point-passes differ; GPU - oracle histogram {dict(sorted(hist.items()))})")
    ctx.close()
