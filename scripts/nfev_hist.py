"""Diagnostic: distribution of local-solver function evaluations (state slot 3) on a kinematically driven RVE, and the
wave-level divergence cost (mean over waves of max-per-wave / mean-per-lane)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import exaconstit_amd.lib as L
import hipref, orc
from hipref import ptr
orc.build()
dev = hipref.Dev()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
MODEL = sys.argv[2] if len(sys.argv) > 2 else "fcc_voce"
MID = {"fcc_voce": (L.EXA_FCC_VOCE, "props_cp_voce.txt"), "bcc_voce": (L.EXA_BCC_VOCE, "props_cp_voce.txt"), "fcc_kmdd": (L.EXA_FCC_KMDD, "props_cp_mts.txt"), "bcc_kmdd": (L.EXA_BCC_KMDD, "props_cp_mts.txt")}[MODEL]
rve = hipref.make_rve(orc, N)
P = rve["E"] * 8
props = np.loadtxt(os.path.join(orc.REFDATA, MID[1])).ravel()
ctx = L.Context(MID[0], props, 298.0, 1, rve["E"])
quats = hipref.random_quats(rve["E"])
d_q = dev.up(quats.ravel()); sv0 = dev.zeros(28 * P); sv1 = dev.zeros(28 * P); s0 = dev.zeros(6 * P); s1 = dev.zeros(6 * P); cm = dev.zeros(36 * P)
ctx.check(L.exa_init_state(ctx.h, ptr(sv0), ptr(d_q), None))
v = hipref.velocity_field(rve); ve = dev.up(hipref.l_to_e(rve, v)); x = rve["X"].copy()
dJ = dev.zeros(9 * P)
for dt in [0.005, 0.195] + [0.1] * 8:
    x = x + v * dt
    ctx.check(L.exa_jacobians(ctx.h, ptr(dev.up(hipref.l_to_e(rve, x))), ptr(dJ), None))
    ctx.check(L.exa_model_setup(ctx.h, dt, ptr(dJ), ptr(ve), ptr(s0), ptr(sv0), ptr(s1), ptr(sv1), ptr(cm), None))
    assert ctx.check(L.exa_model_status(ctx.h, None)) == 0
    nf = sv1.cpu().numpy().reshape(P, 28)[:, 3]
    w = nf[: (P // 64) * 64].reshape(-1, 64)
    print(f"dt {dt:5.3f} nfev mean {nf.mean():5.2f} max {nf.max():3.0f} hist {np.bincount(nf.astype(int))[:14]}  wave max/mean {w.max(axis=1).mean() / nf.mean():.2f}")
    s0, s1 = s1, s0; sv0, sv1 = sv1, sv0
