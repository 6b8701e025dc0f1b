"""The reference's self-consistency unit tests, restated on the oracle (no MFEM):
  test/mechanics_test.cpp:51-178   PA gradient action == dense B^T C B action        (rel L2 < 1e-14)
  test/mechanics_test.cpp:184-303  PA residual == dense AssembleElementVector        (rel L2 < 2e-14)
  test/mechanics_test.cpp:310-461  EA matrices applied == dense action
  test/grad_test.cpp:88-102,182-195  grad_calc of an affine field gives F = [[3,3,4],[4,3,3],[3,4,3]] (||diff||/size < 3e-15)
on a 2x2x2 mesh, orders 1..3, C = all-ones or cubic (100/75/50), x = 1..N.
"""
import ctypes as C

import numpy as np
import pytest

import hipref
from hipref import rel_l2


def _cubic(P):
    c = np.zeros((6, 6))
    c[:3, :3] = 75.0
    np.fill_diagonal(c, 100.0)
    c[3, 3] = c[4, 4] = c[5, 5] = 50.0
    return np.tile(c.T.ravel(), P)


@pytest.mark.parametrize("p", [1, 2, 3])
@pytest.mark.parametrize("cmat", ["ones", "cubic"])
def test_pa_equals_ea_equals_dense(oracle, p, cmat):
    orc = oracle
    rve = hipref.make_rve(orc, 2, p=p)
    E, Q, n = rve["E"], rve["Q"], rve["n"]
    P = E * Q
    xe = hipref.l_to_e(rve, rve["X"])
    J = np.zeros(9 * P); orc.lib().orc_jacobians(p, E, orc._p(xe), orc._p(J))
    Cm = np.ones(36 * P) if cmat == "ones" else _cubic(P)
    x = np.arange(1, 3 * n * E + 1, dtype=np.float64)
    dt = 1.0
    C4 = np.zeros(81 * P); D4 = np.zeros(81 * P)
    orc.lib().orc_transform_4d(C.c_int64(P), orc._p(Cm), orc._p(C4))
    orc.lib().orc_assemble_grad_pa(Q, E, C.c_double(dt), orc._p(rve["W"]), orc._p(J), orc._p(C4), orc._p(D4))
    y_pa = np.zeros(3 * n * E)
    orc.lib().orc_add_mult_grad_pa(Q, E, n, orc._p(rve["G"]), orc._p(D4), orc._p(x), orc._p(y_pa))
    emat = np.zeros(9 * n * n * E)
    orc.lib().orc_assemble_ea(Q, E, n, C.c_double(dt), orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(Cm), orc._p(emat))
    y_ea = np.zeros(3 * n * E)
    orc.lib().orc_ea_mult(E, n, orc._p(emat), orc._p(x), orc._p(y_ea))
    assert rel_l2(y_pa, y_ea) < 1e-13
    d_pa = np.zeros(3 * n * E); d_ea = np.zeros(3 * n * E)
    orc.lib().orc_assemble_grad_diag_pa(Q, E, n, C.c_double(dt), orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(Cm), orc._p(d_pa))
    orc.lib().orc_ea_diag(E, n, orc._p(emat), orc._p(d_ea))
    assert rel_l2(d_pa, d_ea) < 1e-13


@pytest.mark.parametrize("p", [1, 2, 3, 6])      # 6: the order the reference's own vector test runs at (test/mechanics_test.cpp:188)
def test_pa_residual_equals_dense(oracle, p):
    orc = oracle
    rve = hipref.make_rve(orc, 2, p=p)
    E, Q, n = rve["E"], rve["Q"], rve["n"]
    P = E * Q
    xe = hipref.l_to_e(rve, rve["X"])
    J = np.zeros(9 * P); orc.lib().orc_jacobians(p, E, orc._p(xe), orc._p(J))
    sig = np.ones(6 * P)
    dmat = np.zeros(9 * P); orc.lib().orc_assemble_pa(Q, E, orc._p(rve["W"]), orc._p(J), orc._p(sig), orc._p(dmat))
    y1 = np.zeros(3 * n * E); orc.lib().orc_add_mult_pa(Q, E, n, orc._p(rve["G"]), orc._p(dmat), orc._p(y1))
    y2 = np.zeros(3 * n * E); orc.lib().orc_element_vector(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(sig), orc._p(y2))
    assert rel_l2(y1, y2) < 2e-14


@pytest.mark.parametrize("p", [1, 2, 3])
def test_affine_field_gradient(oracle, p):
    orc = oracle
    rve = hipref.make_rve(orc, 2, p=p)
    E, Q, n, NN = rve["E"], rve["Q"], rve["n"], rve["NN"]
    P = E * Q
    X = rve["X"].reshape(3, NN)
    A = np.array([[2.0, 3.0, 4.0], [4.0, 2.0, 3.0], [3.0, 4.0, 2.0]])
    cur = X + A @ X
    xe = hipref.l_to_e(rve, rve["X"])
    J = np.zeros(9 * P); orc.lib().orc_jacobians(p, E, orc._p(xe), orc._p(J))
    F = np.zeros(9 * P)
    orc.lib().orc_grad_calc(Q, E, n, orc._p(J), orc._p(rve["G"]), orc._p(hipref.l_to_e(rve, cur.ravel())), orc._p(F))
    want = np.tile((A + np.eye(3)).T.ravel(), P)   # column-major (q,t)
    assert np.linalg.norm(F - want) / F.size < 3e-15


def test_quadrature_and_partition_of_unity(oracle):
    orc = oracle
    for p in (1, 2, 3):
        rve = hipref.make_rve(orc, 1, p=p)
        assert abs(rve["W"].sum() - 1.0) < 1e-14
        G = rve["G"].reshape(rve["Q"], 3, rve["n"])
        assert np.max(np.abs(G.sum(axis=2))) < 1e-12


def _bbar_dense(rve, J, eDS, Cm, dt, e):
    """B-bar^T C B-bar of element e assembled with numpy from GenerateGradBarMatrix (reference src/mechanics_model.cpp:845-877)."""
    n, Q = rve["n"], rve["Q"]
    G = rve["G"].reshape(Q, 3, n)
    M = np.zeros((3 * n, 3 * n))
    ed = eDS.reshape(-1, 3, n)[e]
    for q in range(Q):
        Jq = J.reshape(-1, 3, 3)[q + Q * e].T          # J(i,j) stored column-major
        DS = G[q].T @ np.linalg.inv(Jq)                # (n,3): dN/dx
        B = np.zeros((3 * n, 6))
        for a in range(n):
            b = (ed[:, a] - DS[a]) / 3.0
            B[a] = [b[0] + DS[a, 0], b[0], b[0], 0, DS[a, 2], DS[a, 1]]
            B[a + n] = [b[1], b[1] + DS[a, 1], b[1], DS[a, 2], 0, DS[a, 0]]
            B[a + 2 * n] = [b[2], b[2], b[2] + DS[a, 2], DS[a, 1], DS[a, 0], 0]
        C = Cm.reshape(-1, 6, 6)[q + Q * e].T
        M += dt * rve["W"][q] * np.linalg.det(Jq) * B @ C @ B.T
    return M


@pytest.mark.parametrize("p", [1, 2, 3])      # 3: the order of the reference's ICExaNLFIntegratorEATest (test/mechanics_test.cpp:471)
def test_bbar_ea_and_residual(oracle, p):
    """ICExaNLFIntegratorEATest / ICExaNLFIntegratorPAVecTest (reference test/mechanics_test.cpp:468-746)."""
    orc = oracle
    rve = hipref.make_rve(orc, 2, p=p, distort=0.2)
    E, Q, n = rve["E"], rve["Q"], rve["n"]
    P = E * Q
    xe = hipref.l_to_e(rve, rve["X"])
    J = np.zeros(9 * P); orc.lib().orc_jacobians(p, E, orc._p(xe), orc._p(J))
    eDS = np.zeros(3 * n * E)
    orc.lib().orc_element_eds(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS))
    Cm = _cubic(P)
    dt = 0.3
    emat = np.zeros(9 * n * n * E)
    orc.lib().orc_assemble_ea_bbar(Q, E, n, C.c_double(dt), orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS), orc._p(Cm), orc._p(emat))
    for e in (0, E - 1):
        M = _bbar_dense(rve, J, eDS, Cm, dt, e)
        got = emat.reshape(E, 3 * n, 3 * n)[e].T
        assert rel_l2(got, M) < 1e-13
    sig = np.arange(1, 6 * P + 1, dtype=np.float64) / P
    y1 = np.zeros(3 * n * E); y2 = np.zeros(3 * n * E)
    orc.lib().orc_add_mult_pa_bbar(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS), orc._p(sig), orc._p(y1))
    orc.lib().orc_element_vector_bbar(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS), orc._p(sig), orc._p(y2))
    assert rel_l2(y1, y2) < 2e-14
    # the volumetric part of B-bar integrates the element-average gradient: for a constant hydrostatic stress both integrators agree
    hyd = np.tile([1.0, 1.0, 1.0, 0, 0, 0], P)
    y3 = np.zeros(3 * n * E); y4 = np.zeros(3 * n * E)
    orc.lib().orc_add_mult_pa_bbar(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS), orc._p(hyd), orc._p(y3))
    orc.lib().orc_element_vector(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(hyd), orc._p(y4))
    assert rel_l2(y3, y4) < 1e-12


def test_bbar_residual_order6(oracle):
    """ICExaNLFIntegratorPAVecTest at the reference's own order (6, test/mechanics_test.cpp:630): matrix-free B-bar residual == dense one."""
    orc = oracle
    p = 6
    rve = hipref.make_rve(orc, 2, p=p)
    E, Q, n = rve["E"], rve["Q"], rve["n"]
    P = E * Q
    xe = hipref.l_to_e(rve, rve["X"])
    J = np.zeros(9 * P); orc.lib().orc_jacobians(p, E, orc._p(xe), orc._p(J))
    eDS = np.zeros(3 * n * E)
    orc.lib().orc_element_eds(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS))
    sig = np.ones(6 * P)
    y1 = np.zeros(3 * n * E); y2 = np.zeros(3 * n * E)
    orc.lib().orc_add_mult_pa_bbar(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS), orc._p(sig), orc._p(y1))
    orc.lib().orc_element_vector_bbar(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS), orc._p(sig), orc._p(y2))
    assert rel_l2(y1, y2) < 2e-14


def test_bbar_nrls_case_runs(oracle):
    """config-5-like settings on a tiny RVE: B-bar + element assembly + NRLS (reference workflows/Stage3 options_master.toml:83,90,95)."""
    orc = oracle
    case = orc.load_case("voce_ea.toml")
    case.update(nx=4, ny=4, nz=4, elem_grain=np.arange(64, dtype=np.int32) % 125, integ=1, nl_solver=1, additional_avgs=False)
    out = orc.run_case(case, nsteps=3)
    assert out["failed"] == 0
    full = dict(case); full.update(integ=0, nl_solver=0)
    ref = orc.run_case(full, nsteps=3)
    # B-bar only changes the volumetric sampling: same average stress to a few percent on this coarse mesh, not identical
    d = np.abs(out["avg_stress"][:, 2] / ref["avg_stress"][:, 2] - 1.0)
    assert 1e-9 < d.max() < 5e-2


def test_ea_is_pa_of_transposed_tangent(oracle):
    """Reference fact worth pinning: with a NON-symmetric tangent the reference's element-assembly operator equals its
    partial-assembly operator applied to the transposed tangent (AssembleEA src/mechanics_integrators.cpp:893-960 vs
    AssembleGradPA/AddMultGradPA :425-511,592-620); the reference's own equivalence tests only use symmetric C."""
    orc = oracle
    rve = hipref.make_rve(orc, 2, p=1, distort=0.2)
    E, Q, n = rve["E"], rve["Q"], rve["n"]
    P = E * Q
    xe = hipref.l_to_e(rve, rve["X"])
    J = np.zeros(9 * P); orc.lib().orc_jacobians(1, E, orc._p(xe), orc._p(J))
    rng = np.random.default_rng(0)
    Cb = rng.uniform(-1, 1, (P, 6, 6))
    x = rng.uniform(-1, 1, 3 * n * E)
    C4 = np.zeros(81 * P); D4 = np.zeros(81 * P)
    orc.lib().orc_transform_4d(C.c_int64(P), orc._p(Cb.ravel().copy()), orc._p(C4))
    orc.lib().orc_assemble_grad_pa(Q, E, C.c_double(1.0), orc._p(rve["W"]), orc._p(J), orc._p(C4), orc._p(D4))
    y_pa = np.zeros(3 * n * E); orc.lib().orc_add_mult_grad_pa(Q, E, n, orc._p(rve["G"]), orc._p(D4), orc._p(x), orc._p(y_pa))
    for Cuse, same in ((Cb, False), (Cb.transpose(0, 2, 1), True)):
        emat = np.zeros(9 * n * n * E)
        orc.lib().orc_assemble_ea(Q, E, n, C.c_double(1.0), orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(np.ascontiguousarray(Cuse).ravel()), orc._p(emat))
        y_ea = np.zeros(3 * n * E); orc.lib().orc_ea_mult(E, n, orc._p(emat), orc._p(x), orc._p(y_ea))
        assert (rel_l2(y_pa, y_ea) < 1e-13) == same
