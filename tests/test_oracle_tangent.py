"""The material tangent d sigma / d eps (ddsdde, what AssembleGradPA / AssembleEA consume) checked against central differences of the
stress update itself.  The golden curves cannot see a wrong tangent (it only steers Newton), and GPU-vs-oracle comparisons only show that
two implementations of the same formula agree; this pins the formula: for all six crystal models, in the elastic regime and after the
elastic-plastic transition, tangent * v == [sigma(D + h v) - sigma(D - h v)] / (2 h dt) to 1e-5 along the five isochoric directions v, and for
the mean stress along the volumetric direction (eps = D dt with engineering shear strains, fixed spin W).  The deviatoric response to a volume
change - terms of relative size |sigma'| / K - is not part of the evptn tangent; the test bounds what is left out."""
import ctypes as C

import numpy as np
import pytest

MODELS = [("fcc_voce", 0, 0, "props_cp_voce.txt"), ("bcc_voce", 1, 0, "props_cp_voce.txt"), ("fcc_voce_nl", 0, 1, "props_cp_vocenl.txt"),
          ("bcc_voce_nl", 1, 1, "props_cp_vocenl.txt"), ("fcc_kmdd", 0, 2, "props_cp_mts.txt"), ("bcc_kmdd", 1, 2, "props_cp_mts.txt")]


def point_update(orc, xtal, kin, props, dt, L, state, want_tangent=True):
    """One call of the oracle's getResponseSngl on a copy of `state` = (hist26, vol, e_int, stress6); returns (stress6, tangent 6x6 row-major, new state)."""
    hist, vol, e_int, sig = [np.array(x, dtype=np.float64).copy() for x in state]
    D = 0.5 * (L + L.T)
    W = 0.5 * (L - L.T)
    tr = np.trace(D)
    d = np.array([D[0, 0] - tr / 3, D[1, 1] - tr / 3, D[2, 2] - tr / 3, D[1, 2], D[0, 2], D[0, 1], tr])
    w = np.array([W[2, 1], W[0, 2], W[1, 0]])
    v1 = vol[0] * np.exp(tr * dt)
    vr = np.array([vol[0], v1, (v1 - vol[0]) / (dt * 0.5 * (vol[0] + v1)), v1 - vol[0]])
    p = -(sig[0] + sig[1] + sig[2]) / 3.0
    sp = np.array([sig[0] + p, sig[1] + p, sig[2] + p, sig[3], sig[4], sig[5], p])
    tk = C.c_double(298.0); sdd = np.zeros(2); mt = np.zeros(36)
    rc = orc.lib().orc_point_response(xtal, kin, orc._p(props), len(props), C.c_double(dt), orc._p(d), orc._p(w), orc._p(vr), orc._p(e_int), orc._p(sp),
                                      orc._p(hist), C.byref(tk), orc._p(sdd), orc._p(mt) if want_tangent else None, 0, 0)
    assert rc == 0
    s = np.array([sp[0] - sp[6], sp[1] - sp[6], sp[2] - sp[6], sp[3], sp[4], sp[5]])
    return s, mt.reshape(6, 6).copy(), (hist, np.array([v1]), e_int, s)


def voigt_rate(j, h):
    """velocity gradient (symmetric) whose engineering Voigt strain rate is h in component j (11,22,33,23,13,12)"""
    L = np.zeros((3, 3))
    if j < 3:
        L[j, j] = h
    else:
        a, b = [(1, 2), (0, 2), (0, 1)][j - 3]
        L[a, b] = L[b, a] = 0.5 * h
    return L


@pytest.mark.parametrize("name,xtal,kin,pfile", MODELS)
def test_tangent_is_the_derivative_of_the_stress_update(oracle, name, xtal, kin, pfile):
    import os
    orc = oracle
    props = np.loadtxt(os.path.join(orc.REFDATA, pfile)).ravel()
    hist = np.zeros(26)
    orc.lib().orc_hist_init(xtal, kin, orc._p(props), len(props), orc._p(hist))
    q = np.array([0.83, 0.31, -0.36, 0.29]); hist[9:13] = q / np.linalg.norm(q)
    state = (hist, np.array([1.0]), np.array([0.0]), np.zeros(6))
    L0 = np.diag([-0.42e-3, -0.47e-3, 1.0e-3]) + 2.0e-4 * np.array([[0, 0.6, -0.4], [0.1, 0, 0.3], [0.5, -0.2, 0]])
    checked = 0
    for step, dt in enumerate([0.005, 0.1, 0.1, 0.1, 0.1, 0.1, 0.2, 0.2, 0.4]):
        if step in (0, 4, 8):      # elastic, early plastic, developed plastic flow
            s0, Ct, _ = point_update(orc, xtal, kin, props, dt, L0, state)
            h = 1.0e-7             # strain-rate perturbation (strain 1e-8 .. 4e-8)
            # five isochoric directions: there the tangent is the exact derivative of the update
            dirs = [np.array(v, dtype=np.float64) for v in ((1, -1, 0, 0, 0, 0), (1, 1, -2, 0, 0, 0), (0, 0, 0, 1, 0, 0), (0, 0, 0, 0, 1, 0), (0, 0, 0, 0, 0, 1))]
            vol = np.array([1.0, 1.0, 1.0, 0, 0, 0])

            def fd_along(v):
                Lp = sum(voigt_rate(j, h * v[j]) for j in range(6))
                sp, _, _ = point_update(orc, xtal, kin, props, dt, L0 + Lp, state, False)
                sm, _, _ = point_update(orc, xtal, kin, props, dt, L0 - Lp, state, False)
                return (sp - sm) / (2.0 * h * dt)
            for v in dirs:
                fd = fd_along(v)
                assert np.linalg.norm(Ct @ v - fd) < 1.0e-5 * np.linalg.norm(fd), (name, step, v, np.linalg.norm(Ct @ v - fd) / np.linalg.norm(fd))
            # volumetric direction: bulk response exact to 1e-5 on the mean stress; the deviatoric stress changes by -(sigma' d eps_v)-type
            # terms of relative size |sigma'| / K (Cauchy = Kirchhoff / J, strain state e = a_V E) that the evptn tangent leaves out
            fd = fd_along(vol)
            # (bulk modulus K v of the library's EOS bookkeeping against -v dp/dv = K / v of p = K (1/v - 1): O(eps_v) apart)
            assert abs((Ct @ vol)[:3].mean() - fd[:3].mean()) < 3.0e-4 * abs(fd[:3].mean())
            assert np.linalg.norm(Ct @ vol - fd) < 2.0 * np.linalg.norm(s0) + 3.0e-4 * np.linalg.norm(fd)
            checked += 1
        _, _, state = point_update(orc, xtal, kin, props, dt, L0, state)
    assert checked == 3
    assert state[0][0] > 0.0           # the history ends in plastic flow (effective shear rate > 0)
