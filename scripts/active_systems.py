"""Diagnostic: number of slip systems with a non-zero slip rate per point after a kinematic drive into the plastic regime
(Kocks-Mecking models).  Measured: 8-9 of 12 on average, 10-11 for the busiest lane of a wave - compacting the active
systems per lane would not shorten the kinetics loop."""
import os, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import exaconstit_amd.lib as L
import hipref, orc
from hipref import ptr
orc.build()
dev = hipref.Dev()
N = 24
for MODEL in ("bcc_kmdd", "fcc_kmdd"):
    MID = {"fcc_kmdd": (L.EXA_FCC_KMDD, "props_cp_mts.txt"), "bcc_kmdd": (L.EXA_BCC_KMDD, "props_cp_mts.txt")}[MODEL]
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
        keep = dev.up(hipref.l_to_e(rve, x))
        ctx.check(L.exa_jacobians(ctx.h, ptr(keep), ptr(dJ), None))
        ctx.check(L.exa_model_setup(ctx.h, dt, ptr(dJ), ptr(ve), ptr(s0), ptr(sv0), ptr(s1), ptr(sv1), ptr(cm), None))
        assert ctx.check(L.exa_model_status(ctx.h, None)) == 0
        s0, s1 = s1, s0; sv0, sv1 = sv1, sv0
    g = sv0.cpu().numpy().reshape(P, 28)[:, 14:26]
    act = (g != 0).sum(axis=1)
    big = (np.abs(g) > 1e-12 * np.abs(g).max(axis=1, keepdims=True)).sum(axis=1)
    w = act[: (P // 64) * 64].reshape(-1, 64)
    print(MODEL, "nonzero slip rates per point: hist", np.bincount(act, minlength=13), "mean", act.mean(), "wave-max mean", w.max(axis=1).mean(), "| significant (>1e-12 of max):", np.bincount(big, minlength=13))
    ctx.close()
