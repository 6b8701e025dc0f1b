"""The reference's own integrator unit tests (test/mechanics_test.cpp) at the reference's own orders, on the HIP path through the C ABI:
  :54   ExaNLFIntegratorPATest       order 3   PA gradient action == assembled action          C = ones / cubic(100,75,50), x = 1..N
  :187  ExaNLFIntegratorPAVecTest    order 6   PA residual == dense AssembleElementVector      sigma = ones
  :313  ExaNLFIntegratorEATest       order 3   element matrices == dense AssembleElementGrad
  :471  ICExaNLFIntegratorEATest     order 3   B-bar element matrices == dense B-bar matrices
  :630  ICExaNLFIntegratorPAVecTest  order 6   B-bar PA residual == dense B-bar element vector
on the reference's mesh (2 x 2 x 2 hexahedra of the unit cube).  The dense side is the oracle's restatement of AssembleElementVector /
AssembleElementGrad (oracle/fem_port.hpp), the matrix-free side runs on the GPU; tolerances are the reference's (1e-14 class, here 1e-12
relative in L2 because the two sides sum in different orders).  Orders above 2 take the run-time-order kernels (gen_kernels.hip)."""
import ctypes as C

import numpy as np
import pytest

import hipref
from hipref import ptr, rel_l2

pytestmark = pytest.mark.gpu


def _cubic(P):
    c = np.zeros((6, 6))
    c[:3, :3] = 75.0
    np.fill_diagonal(c, 100.0)
    c[3, 3] = c[4, 4] = c[5, 5] = 50.0
    return np.tile(c.T.ravel(), P)


def _setup(orc, p, assembly, integ, distort=0.0):
    import exaconstit_amd.lib as L
    rve = hipref.make_rve(orc, 2, p=p, distort=distort)
    E, Q, n = rve["E"], rve["Q"], rve["n"]
    props = np.loadtxt(orc.REFDATA + "/props_cp_voce.txt").ravel()
    ctx = L.Context(L.EXA_FCC_VOCE, props, 298.0, p, E, assembly=assembly, integ=integ)
    assert (ctx.n, ctx.Q) == (n, Q)
    G, W = ctx.shape_table()
    assert rel_l2(G, rve["G"]) < 1e-12 and rel_l2(W, rve["W"]) < 1e-13
    dev = hipref.Dev()
    xe = hipref.l_to_e(rve, rve["X"])
    P = E * Q
    J = np.zeros(9 * P); orc.lib().orc_jacobians(p, E, orc._p(xe), orc._p(J))
    d_J = dev.zeros(9 * P); d_xe = dev.up(xe)
    ctx.check(L.exa_jacobians(ctx.h, ptr(d_xe), ptr(d_J), None))
    assert rel_l2(d_J.cpu().numpy(), J) < 1e-12
    dev.torch.cuda.synchronize()
    return L, ctx, dev, rve, J, d_J


@pytest.mark.parametrize("cmat", ["ones", "cubic"])
def test_pa_gradient_action_order3(oracle, cmat):
    """ExaNLFIntegratorPATest<cmat_ones>: y_pa = AddMultGradPA(x) against the assembled operator applied to x (EA on the GPU, dense on the oracle)."""
    orc = oracle
    L, ctx, dev, rve, J, d_J = _setup(orc, 3, 0, 0)
    E, Q, n = rve["E"], rve["Q"], rve["n"]; P = E * Q
    Cm = np.ones(36 * P) if cmat == "ones" else _cubic(P)
    x = np.arange(1, 3 * n * E + 1, dtype=np.float64)
    dt = 1.0
    emat = np.zeros(9 * n * n * E)
    orc.lib().orc_assemble_ea(Q, E, n, C.c_double(dt), orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(Cm), orc._p(emat))
    y_fa = np.zeros(3 * n * E); orc.lib().orc_ea_mult(E, n, orc._p(emat), orc._p(x), orc._p(y_fa))
    d_y = dev.zeros(3 * n * E)
    d_C = dev.up(Cm); d_x = dev.up(x)      # (device inputs stay referenced until the results are read back: the launches are asynchronous)
    ctx.check(L.exa_grad_setup(ctx.h, dt, ptr(d_J), ptr(d_C), None))
    ctx.check(L.exa_grad_apply(ctx.h, ptr(d_x), ptr(d_y), None))
    assert rel_l2(d_y.cpu().numpy(), y_fa) < 1e-12
    # the diagonal of the same operator (AssembleGradDiagonalPA)
    d_ref = np.zeros(3 * n * E); orc.lib().orc_ea_diag(E, n, orc._p(emat), orc._p(d_ref))
    d_d = dev.zeros(3 * n * E); ctx.check(L.exa_grad_diagonal(ctx.h, ptr(d_d), None))
    assert rel_l2(d_d.cpu().numpy(), d_ref) < 1e-12
    ctx.close()


def test_pa_residual_order6(oracle):
    """ExaNLFIntegratorPAVecTest: AssemblePA + AddMultPA against the dense AssembleElementVector, sigma = 1."""
    orc = oracle
    L, ctx, dev, rve, J, d_J = _setup(orc, 6, 0, 0)
    E, Q, n = rve["E"], rve["Q"], rve["n"]; P = E * Q
    sig = np.ones(6 * P)
    y_ref = np.zeros(3 * n * E); orc.lib().orc_element_vector(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(sig), orc._p(y_ref))
    d_y = dev.zeros(3 * n * E)
    d_sig = dev.up(sig)
    ctx.check(L.exa_residual_setup(ctx.h, ptr(d_J), ptr(d_sig), None))
    ctx.check(L.exa_residual_apply(ctx.h, ptr(d_y), None))
    assert rel_l2(d_y.cpu().numpy(), y_ref) < 1e-12
    ctx.close()


@pytest.mark.parametrize("integ", [0, 1])
def test_element_assembly_order3(oracle, integ):
    """ExaNLFIntegratorEATest / ICExaNLFIntegratorEATest: the element matrices (exported in the reference's layout), their action on E- and
    L-vectors and their diagonal against the oracle's dense matrices, plain and B-bar, on a distorted mesh for B-bar."""
    orc = oracle
    L, ctx, dev, rve, J, d_J = _setup(orc, 3, 1, integ, distort=0.2 if integ else 0.0)
    E, Q, n, NN = rve["E"], rve["Q"], rve["n"], rve["NN"]; P = E * Q
    Cm = _cubic(P); dt = 0.3 if integ else 1.0
    emat = np.zeros(9 * n * n * E)
    if integ:
        eDS = np.zeros(3 * n * E)
        orc.lib().orc_element_eds(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS))
        orc.lib().orc_assemble_ea_bbar(Q, E, n, C.c_double(dt), orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS), orc._p(Cm), orc._p(emat))
        # B-bar needs the element-average gradients the residual set-up leaves behind (the reference's test calls AssemblePA first, :585-587)
        d_s0 = dev.zeros(6 * P)
        ctx.check(L.exa_residual_setup(ctx.h, ptr(d_J), ptr(d_s0), None))
    else:
        orc.lib().orc_assemble_ea(Q, E, n, C.c_double(dt), orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(Cm), orc._p(emat))
    d_C = dev.up(Cm)
    ctx.check(L.exa_grad_setup(ctx.h, dt, ptr(d_J), ptr(d_C), None))
    d_em = dev.zeros(9 * n * n * E)
    ctx.check(L.exa_grad_get_ea(ctx.h, ptr(d_em), None))
    assert rel_l2(d_em.cpu().numpy(), emat) < 1e-12
    x = np.arange(1, 3 * n * E + 1, dtype=np.float64)
    y_ref = np.zeros(3 * n * E); orc.lib().orc_ea_mult(E, n, orc._p(emat), orc._p(x), orc._p(y_ref))
    d_y = dev.zeros(3 * n * E); d_x = dev.up(x)
    ctx.check(L.exa_grad_apply(ctx.h, ptr(d_x), ptr(d_y), None))
    assert rel_l2(d_y.cpu().numpy(), y_ref) < 1e-12
    d_ref = np.zeros(3 * n * E); orc.lib().orc_ea_diag(E, n, orc._p(emat), orc._p(d_ref))
    d_d = dev.zeros(3 * n * E); ctx.check(L.exa_grad_diagonal(ctx.h, ptr(d_d), None))
    assert rel_l2(d_d.cpu().numpy(), d_ref) < 1e-12
    # fused L-vector element mat-vec
    d_conn = dev.up(rve["conn"])
    ctx.check(L.exa_set_connectivity(ctx.h, ptr(d_conn), NN))
    xL = np.arange(1, 3 * NN + 1, dtype=np.float64)
    ye = np.zeros(3 * n * E); orc.lib().orc_ea_mult(E, n, orc._p(emat), orc._p(hipref.l_to_e(rve, xL)), orc._p(ye))
    yL_ref = hipref.e_to_l(rve, ye)
    d_yL = dev.zeros(3 * NN); d_xL = dev.up(xL)
    ctx.check(L.exa_grad_apply_lvec(ctx.h, ptr(d_xL), ptr(d_yL), None, None))
    assert rel_l2(d_yL.cpu().numpy(), yL_ref) < 1e-12
    ctx.close()


def test_bbar_residual_order6(oracle):
    """ICExaNLFIntegratorPAVecTest: B-bar AssemblePA + AddMultPA against the dense B-bar element vector, sigma = 1."""
    orc = oracle
    L, ctx, dev, rve, J, d_J = _setup(orc, 6, 1, 1)
    E, Q, n = rve["E"], rve["Q"], rve["n"]; P = E * Q
    eDS = np.zeros(3 * n * E)
    orc.lib().orc_element_eds(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS))
    sig = np.ones(6 * P)
    y_ref = np.zeros(3 * n * E)
    orc.lib().orc_element_vector_bbar(Q, E, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(J), orc._p(eDS), orc._p(sig), orc._p(y_ref))
    d_y = dev.zeros(3 * n * E)
    d_sig = dev.up(sig)
    ctx.check(L.exa_residual_setup(ctx.h, ptr(d_J), ptr(d_sig), None))
    ctx.check(L.exa_residual_apply(ctx.h, ptr(d_y), None))
    assert rel_l2(d_y.cpu().numpy(), y_ref) < 1e-12
    ctx.close()
