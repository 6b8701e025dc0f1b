"""The staged AOS form of the constitutive launch (include/exaconstit_hip.h, exa_set_aos_staging; exaconstit_amd/csrc/model_kernel.hpp, PointIO<.., STG>):
a wave moves the contiguous rows of its 64 consecutive points with coalesced 16-byte accesses and transposes them through its LDS stash region.
It is a change of data movement only.  Two statements are tested:
  * against the per-lane (strided) launches of rounds 1-5 (exa_set_aos_staging(ctx, 0)): same evaluation counts at every point, every other output equal
    to round-off (<= 1e-13 relative; tangent 1e-11).  Not the same bits: the staged launch is another instantiation of the kernel (unrolled node loops
    and compile-time kinetics exponents also for the E-vector form, other control flow around the row stores), and the compiler contracts
    multiply-adds per basic block - the same reason tests/test_gpu_parity.py gives for the x^49 instantiation.  The staged launch itself is what every
    AOS test of tests/test_gpu_parity.py / test_gpu_point_fixtures.py / test_adapters.py now compares with the oracle.
  * a tail split of the staged launch is bit-neutral: its dense launches are the same kernel in its per-lane mode, so a listed point gets the bits the
    full launch would have given it.
Cases: both entry points of the reference layout (E-vector + Jacobian field: what HipExaModel::ModelSetup calls, reference
src/mechanics_ecmech.cpp:192-258; L-vector gathers that write the Jacobians), a point count that leaves the last wave partly empty (5^3
elements: 1000 points = 15 waves + 40 points; p = 2: 27^2 = 729 points, an odd number of Jacobian rows), the instantiations with the kinetics'
exponents compiled in and the general ones, and a tail split (main launch staged, dense launches per lane)."""
import numpy as np
import pytest

import hipref
from hipref import ptr

pytestmark = pytest.mark.gpu

# (name, props file, lib model id, property overrides): x^49 compiled in | general power form | Kocks-Mecking p = q = 1 compiled in | general
KINDS = [("fcc_voce", "props_cp_voce.txt", 0, {}), ("bcc_voce_m0p1", "props_cp_voce.txt", 2, {7: 0.1}), ("fcc_voce_nl", "props_cp_vocenl.txt", 1, {}),
         ("bcc_kmdd", "props_cp_mts.txt", 5, {}), ("fcc_kmdd", "props_cp_mts.txt", 4, {}), ("fcc_kmdd_p0p8_q1p4", "props_cp_mts.txt", 4, {10: 0.8, 11: 1.4})]


def _props(orc, fname, overrides):
    props = np.loadtxt(orc.REFDATA + "/" + fname).ravel()
    for i, v in overrides.items():
        props[i] = v
    return props


def _run(orc, model, props, N, order, lvec, staging, cap=0, auto=None):
    """four kinematic steps through exa_model_setup (E-vector form) or exa_model_setup_lvec; returns the final outputs as device tensors"""
    import torch
    import exaconstit_amd.lib as L
    dev = hipref.Dev()
    rve = hipref.make_rve(orc, N, p=order, distort=0.15)
    E, Q, NN, n = rve["E"], rve["Q"], rve["NN"], rve["n"]
    P = E * Q
    ctx = L.Context(model, props, 298.0, order, E)
    ctx.check(L.exa_set_aos_staging(ctx.h, staging))
    assert L.exa_get_aos_staging(ctx.h) == staging and L.exa_get_quadrature_layout(ctx.h) == L.EXA_QLAYOUT_AOS
    if cap:
        ctx.check(L.exa_set_newton_caps(ctx.h, cap, 0, 1))
    if auto:
        ctx.check(L.exa_set_newton_cap_auto(ctx.h, *auto))
    d_conn = torch.from_numpy(rve["conn"].astype(np.int32)).to(dev.dev)
    ctx.check(L.exa_set_connectivity(ctx.h, ptr(d_conn), NN))
    sv = [dev.zeros(28 * P), dev.zeros(28 * P)]; sg = [dev.zeros(6 * P), dev.zeros(6 * P)]; cm = dev.zeros(36 * P); J = dev.zeros(9 * P)
    d_quats = dev.up(hipref.random_quats(E).ravel())
    ctx.check(L.exa_init_state(ctx.h, ptr(sv[0]), ptr(d_quats), None))
    v_nodes = hipref.velocity_field(rve, scale=2.0)
    d_x = dev.up(rve["X"]); d_v = dev.up(v_nodes)
    d_xe = dev.zeros(3 * n * E); d_ve = dev.zeros(3 * n * E)
    ctx.check(L.exa_restrict(ctx.h, ptr(d_v), ptr(d_ve), None))
    tails = 0
    for dt in (0.1, 0.3, 0.5, 0.5):
        d_x += dt * d_v
        if lvec:
            ctx.check(L.exa_model_setup_lvec(ctx.h, dt, ptr(d_x), ptr(d_v), ptr(sg[0]), ptr(sv[0]), ptr(sg[1]), ptr(sv[1]), ptr(cm), ptr(J), None))
        else:
            ctx.check(L.exa_restrict(ctx.h, ptr(d_x), ptr(d_xe), None))
            ctx.check(L.exa_jacobians(ctx.h, ptr(d_xe), ptr(J), None))
            ctx.check(L.exa_model_setup(ctx.h, dt, ptr(J), ptr(d_ve), ptr(sg[0]), ptr(sv[0]), ptr(sg[1]), ptr(sv[1]), ptr(cm), None))
        assert ctx.check(L.exa_model_status(ctx.h, None)) == 0
        tails += L.exa_model_tail_count(ctx.h, None)
        sv.reverse(); sg.reverse()
    out = (sv[0].clone(), sg[0].clone(), cm.clone(), J.clone())
    if auto:
        tails = (tails, L.exa_get_newton_cap(ctx.h))
    ctx.close()
    return out, tails


@pytest.mark.parametrize("name,pfile,model,overrides", KINDS, ids=[k[0] for k in KINDS])
@pytest.mark.parametrize("lvec", [False, True], ids=["evec", "lvec"])
def test_staged_launch_matches_per_lane_launch(oracle, name, pfile, model, overrides, lvec):
    props = _props(oracle, pfile, overrides)
    ref, _ = _run(oracle, model, props, 5, 1, lvec, 0)
    got, _ = _run(oracle, model, props, 5, 1, lvec, 1)
    _compare(ref, got, name)


def _compare(ref, got, name):
    from hipref import rel_l2
    ref = [t.cpu().numpy() for t in ref]; got = [t.cpu().numpy() for t in got]
    sa, sb = ref[0].reshape(-1, 28), got[0].reshape(-1, 28)
    assert sa[:, 3].max() > 4, name                                   # plastic: the local solves iterated
    assert np.array_equal(sa[:, 3], sb[:, 3]), (name, int((sa[:, 3] != sb[:, 3]).sum()))      # same evaluation counts
    keep = np.ones(28, bool); keep[3] = False
    for lo, hi in ((0, 3), (4, 9), (9, 13), (13, 14), (14, 26), (26, 28)):      # slot groups separately: large entries must not hide small ones
        assert rel_l2(sb[:, lo:hi], sa[:, lo:hi]) < 1e-13, (name, "state", lo, hi, rel_l2(sb[:, lo:hi], sa[:, lo:hi]))
    assert rel_l2(got[1], ref[1]) < 1e-13, (name, "stress", rel_l2(got[1], ref[1]))
    assert rel_l2(got[2], ref[2]) < 1e-11, (name, "tangent", rel_l2(got[2], ref[2]))
    assert np.array_equal(got[3], ref[3]), (name, "jacobian")          # pure data movement / the same node loop


@pytest.mark.parametrize("lvec", [False, True], ids=["evec", "lvec"])
def test_staged_launch_order2(oracle, lvec):
    """p = 2 (27 nodes and points per element: run-time node loops, rows of the wave straddle elements, 729 points)"""
    props = _props(oracle, "props_cp_voce.txt", {})
    ref, _ = _run(oracle, 0, props, 3, 2, lvec, 0)
    got, _ = _run(oracle, 0, props, 3, 2, lvec, 1)
    _compare(ref, got, "order2")


@pytest.mark.parametrize("name,pfile,model,cap", [("fcc_voce", "props_cp_voce.txt", 0, 4), ("bcc_kmdd", "props_cp_mts.txt", 5, 4)], ids=["fcc_voce", "bcc_kmdd"])
@pytest.mark.parametrize("lvec", [False, True], ids=["evec", "lvec"])
def test_staged_launch_with_tail_split(oracle, name, pfile, model, cap, lvec):
    """capped staged launch + dense launches (the same kernel, per-lane mode) == uncapped staged launch, bit for bit (a cut-off lane stays in its wave
    for the row stores; what it stores the dense launch overwrites)"""
    import torch
    props = _props(oracle, pfile, {})
    ref, t0 = _run(oracle, model, props, 5, 1, lvec, 1)
    got, t1 = _run(oracle, model, props, 5, 1, lvec, 1, cap=cap)
    assert t0 == 0 and t1 > 0
    for a, b, what in zip(ref, got, ("state", "stress", "tangent", "jacobian")):
        assert torch.equal(a, b), (name, what, float((a - b).abs().max()))


@pytest.mark.parametrize("lvec", [False, True], ids=["evec", "lvec"])
def test_library_chosen_cap_is_bit_neutral(oracle, lvec):
    """exa_set_newton_cap_auto (what the MFEM adapters switch on): the library bins the evaluation counts of its own launches and caps the next ones.
    Same bits as the uncapped launches; mode 1 leaves a Voce context alone, mode 0 clears the cap."""
    import torch
    import exaconstit_amd.lib as L
    props = _props(oracle, "props_cp_mts.txt", {})
    ref, t0 = _run(oracle, 5, props, 5, 1, lvec, 1)
    got, (t1, cap) = _run(oracle, 5, props, 5, 1, lvec, 1, auto=(1, 0.05))      # a cheap dense launch in the cost model: a cap pays on this small RVE too
    assert t0 == 0 and t1 > 0 and cap >= 3, (t1, cap)
    for a, b, what in zip(ref, got, ("state", "stress", "tangent", "jacobian")):
        assert torch.equal(a, b), (what, float((a - b).abs().max()))
    _, (t2, cap2) = _run(oracle, 0, _props(oracle, "props_cp_voce.txt", {}), 5, 1, lvec, 1, auto=(1, 0.05))
    assert t2 == 0 and cap2 == 0
    ctx = L.Context(5, props, 298.0, 1, 8)
    ctx.check(L.exa_set_newton_caps(ctx.h, 4, 0, 1)); assert L.exa_get_newton_cap(ctx.h) == 4
    ctx.check(L.exa_set_newton_cap_auto(ctx.h, 0, 0.0)); assert L.exa_get_newton_cap(ctx.h) == 0
    assert L.exa_set_newton_cap_auto(ctx.h, 3, 0.0) != 0
    ctx.close()
