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


@pytest.mark.parametrize("p", [1, 2, 3])
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
