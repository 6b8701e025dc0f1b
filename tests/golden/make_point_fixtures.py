"""Generates tests/golden/point_fixtures/<model>.npz: per-quadrature-point input/output vectors of ModelSetup, produced
by the CPU oracle AFTER it was pinned to the reference's golden curves (tests/test_oracle_golden.py).  The GPU test
tests/test_gpu_point_fixtures.py replays them through the C ABI without the oracle.

Per model: a distorted 2^3-element RVE (64 quadrature points, seeded orientations) driven kinematically through 6 steps; recorded at
step 0 (elastic), 2 (elastic-plastic transition) and 5 (plastic flow): Jacobians, E-vector velocity, dt, begin-of-step stress / state and
the oracle's end-of-step stress (6), state (28) and tangent (36, reference layout); xe = end-of-step nodal coordinates (E-vector).
Usage: python tests/golden/make_point_fixtures.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hipref  # noqa: E402
import orc  # noqa: E402

CASES = [("fcc_voce", 0, 0, "props_cp_voce.txt", 0), ("bcc_voce", 1, 0, "props_cp_voce.txt", 2), ("fcc_voce_nl", 0, 1, "props_cp_vocenl.txt", 1),
         ("bcc_voce_nl", 1, 1, "props_cp_vocenl.txt", 3), ("fcc_kmdd", 0, 2, "props_cp_mts.txt", 4), ("bcc_kmdd", 1, 2, "props_cp_mts.txt", 5)]
# Property variants (round 5): every power-law form of the Voce kinetics the device code carries (x^9, x^19, x^99, the rolled loop for another
# integer exponent, exp(xn log|t|) for a non-integer one), a Voce-NL hardening exponent != 1, and Kocks-Mecking with p, q != 1.
# (name, xtal, kin, property file, model id, {property index: value})
M_FORMS = [("m0p1", 0.1), ("m0p05", 0.05), ("m0p01", 0.01), ("m1o31", 1.0 / 31.0), ("m0p03", 0.03)]
VARIANTS = [(f"{c[0]}_{tag}", c[1], c[2], c[3], c[4], {7: m}) for c in (CASES[0], CASES[3]) for tag, m in M_FORMS]
VARIANTS += [("fcc_voce_nl_mp0p7", 0, 1, "props_cp_vocenl.txt", 1, {12: 0.7}),
             ("fcc_kmdd_p0p8_q1p4", 0, 2, "props_cp_mts.txt", 4, {10: 0.8, 11: 1.4}), ("bcc_kmdd_p0p8_q1p4", 1, 2, "props_cp_mts.txt", 5, {10: 0.8, 11: 1.4})]
DTS = [0.005, 0.195, 0.1, 0.1, 0.2, 0.4]
RECORD = (0, 2, 5)

if __name__ == "__main__":
    orc.build()
    out_dir = os.path.join(HERE, "point_fixtures")
    os.makedirs(out_dir, exist_ok=True)
    for name, xtal, kin, pfile, model, overrides in [c + ({},) for c in CASES] + VARIANTS:
        props = np.loadtxt(os.path.join(orc.REFDATA, pfile)).ravel()
        for idx, val in overrides.items():
            props[idx] = val
        rve = hipref.make_rve(orc, 2, distort=0.2, seed=11)
        P = rve["E"] * rve["Q"]
        quats = hipref.random_quats(rve["E"], seed=2024 + model)
        hist = np.zeros(26); orc.lib().orc_hist_init(xtal, kin, orc._p(props), len(props), orc._p(hist))
        sv0 = np.tile(np.concatenate([hist, [1.0, 0.0]]), P).reshape(P, 28)
        sv0[:, 9:13] = np.repeat(quats, rve["Q"], axis=0)
        sv0 = sv0.ravel().copy(); s0 = np.zeros(6 * P)
        v_nodes = hipref.velocity_field(rve, seed=5)
        vel_e = hipref.l_to_e(rve, v_nodes)
        x = rve["X"].copy()
        rec = dict(model=model, xtal=xtal, kin=kin, props=props, E=rve["E"], Q=rve["Q"], vel_e=vel_e, steps=np.array(RECORD))
        for step, dt in enumerate(DTS):
            x = x + v_nodes * dt
            xe = hipref.l_to_e(rve, x)
            J = np.zeros(9 * P); orc.lib().orc_jacobians(1, rve["E"], orc._p(xe), orc._p(J))
            s1 = np.zeros(6 * P); sv1 = np.zeros(28 * P); cm = np.zeros(36 * P)
            nf = orc.lib().orc_model_setup(xtal, kin, orc._p(props), len(props), rve["Q"], rve["E"], rve["n"], 28, C.c_double(dt), C.c_double(298.0),
                                           orc._p(J), orc._p(rve["G"]), orc._p(vel_e), orc._p(s0), orc._p(sv0), orc._p(s1), orc._p(sv1), orc._p(cm), None, 1, 0, 0)
            assert nf == 0
            if step in RECORD:
                rec.update({f"dt_{step}": dt, f"J_{step}": J, f"xe_{step}": xe, f"s0_{step}": s0.copy(), f"sv0_{step}": sv0.copy(), f"s1_{step}": s1.copy(),
                            f"sv1_{step}": sv1.copy(), f"cm_{step}": cm.copy()})
            s0, sv0 = s1, sv1
        assert np.abs(sv0.reshape(P, 28)[:, 14:26]).sum(axis=1).min() > 0      # ends in plastic flow
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print(name, "written,", P, "points x", len(RECORD), "steps")
