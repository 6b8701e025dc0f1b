"""Parity at BASELINE.json's full sizes (64^3 and 128^3 elements) through size-independent properties, since the oracle cannot
run 16.8 M quadrature points in test time:
  * a random sample of elements is re-computed by the oracle from the very inputs the GPU pass consumed (stress 1e-9, tangent 1e-7);
  * internal forces are self-equilibrated: the nodal residual sums to zero per component (sum_a grad N_a = 0);
  * the gradient action is linear, and partial assembly == element assembly (on the transposed tangent field, see below).
All calls go through the C ABI; torch is the allocator."""
import ctypes as C
import os

import numpy as np
import pytest

import hipref
from hipref import rel_l2

pytestmark = pytest.mark.gpu


def ptr(t):
    return t.data_ptr()


@pytest.mark.parametrize("N", [64, 128])
def test_full_size_properties(oracle, N):
    import torch
    import exaconstit_amd.lib as L
    orc = oracle
    dev = hipref.Dev()
    rve = hipref.make_rve(orc, N)
    E, Q, n, NN = rve["E"], rve["Q"], rve["n"], rve["NN"]
    P = E * Q
    props = np.loadtxt(os.path.join(orc.REFDATA, "props_cp_voce.txt")).ravel()
    ctx = L.Context(L.EXA_FCC_VOCE, props, 298.0, 1, E, assembly=L.EXA_ASSEMBLY_PA)
    d_conn = torch.from_numpy(rve["conn"].astype(np.int32)).to(dev.dev)
    ctx.check(L.exa_set_connectivity(ctx.h, ptr(d_conn), NN))
    quats = hipref.random_quats(E)
    d_sv = [dev.zeros(28 * P), dev.zeros(28 * P)]
    d_s = [dev.zeros(6 * P), dev.zeros(6 * P)]
    d_cm = dev.zeros(36 * P); d_J = dev.zeros(9 * P)
    d_quats_keep = dev.up(quats.ravel())   # must outlive the asynchronous launch
    ctx.check(L.exa_init_state(ctx.h, ptr(d_sv[0]), ptr(d_quats_keep), None))
    v_nodes = hipref.velocity_field(rve)
    d_v = dev.up(v_nodes); d_x = dev.up(rve["X"])
    dts = [0.005, 0.195, 0.4, 0.4]                      # 0.1 % strain: plastic
    for i, dt in enumerate(dts):
        d_x += dt * d_v
        if i == len(dts) - 1:
            sv_in, s_in = d_sv[0].clone(), d_s[0].clone()
        ctx.check(L.exa_model_setup_lvec(ctx.h, dt, ptr(d_x), ptr(d_v), ptr(d_s[0]), ptr(d_sv[0]), ptr(d_s[1]), ptr(d_sv[1]), ptr(d_cm), ptr(d_J), None))
        assert ctx.check(L.exa_model_status(ctx.h, None)) == 0
        d_sv.reverse(); d_s.reverse()
    sig, sv1 = d_s[0], d_sv[0]                            # end-of-step values of the last pass
    assert float(sv1.view(P, 28)[:, 14:26].abs().sum(dim=1).min()) > 0          # every point is slipping

    # ---- (1) sampled elements against the oracle on identical inputs
    rng = np.random.default_rng(N)
    es = np.sort(rng.choice(E, 512, replace=False))      # 4 096 points: the evaluation-count criterion (>= 99.9 %) allows 4 ties
    qidx = torch.from_numpy((es[:, None] * Q + np.arange(Q)[None, :]).ravel()).to(dev.dev)
    take = lambda t, w: t.view(P, w)[qidx].cpu().numpy().ravel()
    sub = dict(rve, E=len(es))
    conn = rve["conn"].reshape(E, n)[es]
    x_end = d_x.cpu().numpy(); 
    xe = np.stack([x_end[conn + NN * c] for c in range(3)], axis=1).ravel()
    ve = np.stack([v_nodes[conn + NN * c] for c in range(3)], axis=1).ravel()
    Js = np.zeros(9 * len(es) * Q); orc.lib().orc_jacobians(1, len(es), orc._p(xe), orc._p(Js))
    assert rel_l2(take(d_J, 9), Js) < 1e-13
    s1 = np.zeros(6 * len(es) * Q); sv = np.zeros(28 * len(es) * Q); cm = np.zeros(36 * len(es) * Q)
    nf = orc.lib().orc_model_setup(0, 0, orc._p(props), len(props), Q, len(es), n, 28, C.c_double(dts[-1]), C.c_double(298.0), orc._p(Js), orc._p(rve["G"]),
                                   orc._p(ve), orc._p(take(s_in, 6)), orc._p(take(sv_in, 28)), orc._p(s1), orc._p(sv), orc._p(cm), None, 1, 0, 0)
    assert nf == 0
    assert rel_l2(take(sig, 6), s1) < 1e-9
    assert rel_l2(take(d_cm, 36), cm) < 1e-7
    keep = np.ones(28, bool); keep[3] = False
    assert rel_l2(take(sv1, 28).reshape(-1, 28)[:, keep], sv.reshape(-1, 28)[:, keep]) < 1e-8
    nf_gpu, nf_ref = take(sv1, 28).reshape(-1, 28)[:, 3], sv.reshape(-1, 28)[:, 3]      # evaluation counts of the local solver follow the oracle's
    assert np.abs(nf_gpu - nf_ref).max() <= 1 and np.mean(nf_gpu == nf_ref) >= 0.999, (np.abs(nf_gpu - nf_ref).max(), np.mean(nf_gpu == nf_ref))

    # ---- (2) self-equilibrated internal forces
    d_y = dev.zeros(3 * NN)
    ctx.check(L.exa_residual_lvec(ctx.h, ptr(d_J), ptr(sig), ptr(d_y), None))
    y = d_y.view(3, NN)
    assert float(y.sum(dim=1).abs().max()) < 1e-10 * float(y.abs().sum())

    # ---- (3) gradient action: linear, and PA == EA on the same tangent field
    ctx.check(L.exa_grad_setup(ctx.h, dts[-1], ptr(d_J), ptr(d_cm), None))
    g = torch.Generator(device="cpu").manual_seed(5)
    x1 = torch.rand(3 * NN, generator=g, dtype=torch.float64).to(dev.dev) - 0.5
    x2 = torch.rand(3 * NN, generator=g, dtype=torch.float64).to(dev.dev) - 0.5
    mask = torch.zeros(3 * NN, dtype=torch.uint8, device=dev.dev)

    def apply(c, x):
        y = dev.zeros(3 * NN)
        c.check(L.exa_grad_apply_lvec(c.h, ptr(x), ptr(y), ptr(mask), None))
        return y
    y1, y2, y12 = apply(ctx, x1), apply(ctx, x2), apply(ctx, 0.7 * x1 - 1.9 * x2)
    assert float((y12 - (0.7 * y1 - 1.9 * y2)).norm() / y12.norm()) < 1e-12
    ea = L.Context(L.EXA_FCC_VOCE, props, 298.0, 1, E, assembly=L.EXA_ASSEMBLY_EA)
    ea.check(L.exa_set_connectivity(ea.h, ptr(d_conn), NN))
    # The reference's AssembleEA contracts the tangent with the trial/test roles exchanged relative to AssembleGradPA
    # (src/mechanics_integrators.cpp:893-960 vs :425-511,592-620): for a non-symmetric tangent EA(C) == PA(C^T).  Its own
    # equivalence tests use symmetric C only (test/mechanics_test.cpp); both paths here follow the reference, so the identity
    # is checked with the transposed field (and the raw difference is of the size of the tangent's asymmetry, ~1e-4).
    d_cmT = d_cm.view(P, 6, 6).transpose(1, 2).contiguous().view(-1)
    ea.check(L.exa_grad_setup(ea.h, dts[-1], ptr(d_J), ptr(d_cmT), None))
    z1 = apply(ea, x1)
    assert float((z1 - y1).norm() / y1.norm()) < 1e-12
    ea.check(L.exa_grad_setup(ea.h, dts[-1], ptr(d_J), ptr(d_cm), None))
    z_asm = apply(ea, x1)
    assert float((z_asm - y1).norm() / y1.norm()) < 1e-3
    # ---- (4) the byte-saving variants of the driver's default path give the action of the streamed records / assembled matrices:
    #      adj(J) recomputed from the nodal coordinates, tangent in its deviatoric-block + bulk form, element assembly without matrices
    defect = C.c_double(1.0); ctx.check(L.exa_grad_tangent_defect(ctx.h, ptr(d_cm), C.byref(defect), None)); assert defect.value < 1e-13
    ctx.check(L.exa_grad_set_coords(ctx.h, ptr(d_x)))
    assert float((apply(ctx, x1) - y1).norm() / y1.norm()) < 1e-13
    ctx.check(L.exa_set_tangent_form(ctx.h, L.EXA_TANGENT_DEV5_BULK)); ctx.check(L.exa_grad_setup(ctx.h, dts[-1], ptr(d_J), ptr(d_cm), None))
    assert float((apply(ctx, x1) - y1).norm() / y1.norm()) < 1e-13
    ea.check(L.exa_set_ea_matrix_free(ea.h, 1)); ea.check(L.exa_set_tangent_form(ea.h, L.EXA_TANGENT_DEV5_BULK))
    ea.check(L.exa_grad_setup(ea.h, dts[-1], ptr(d_J), ptr(d_cm), None)); ea.check(L.exa_grad_set_coords(ea.h, ptr(d_x)))
    assert float((apply(ea, x1) - z_asm).norm() / z_asm.norm()) < 1e-13
    ea.close(); ctx.close()


def test_full_size_order2_bbar_element_assembly(oracle):
    """BASELINE config 5 shape (64^3 elements, p = 2, B-bar, element assembly): a sample of elements is re-computed by the oracle
    from the inputs of the GPU pass — constitutive update (27 points / element), B-bar residual and B-bar element matrices — and
    the element mat-vec is checked for linearity on the full operator."""
    import torch
    import exaconstit_amd.lib as L
    orc = oracle
    dev = hipref.Dev()
    N, p = 64, 2
    rve = hipref.make_rve(orc, N, p=p)
    E, Q, n, NN = rve["E"], rve["Q"], rve["n"], rve["NN"]
    P = E * Q
    props = np.loadtxt(os.path.join(orc.REFDATA, "props_cp_voce.txt")).ravel()
    ctx = L.Context(L.EXA_FCC_VOCE, props, 298.0, p, E, assembly=L.EXA_ASSEMBLY_EA, integ=L.EXA_INTEG_BBAR)
    d_conn = torch.from_numpy(rve["conn"].astype(np.int32)).to(dev.dev)
    ctx.check(L.exa_set_connectivity(ctx.h, ptr(d_conn), NN))
    d_sv = [dev.zeros(28 * P), dev.zeros(28 * P)]; d_s = [dev.zeros(6 * P), dev.zeros(6 * P)]
    d_cm = dev.zeros(36 * P); d_J = dev.zeros(9 * P)
    d_quats_keep = dev.up(hipref.random_quats(E).ravel())   # must outlive the asynchronous launch
    ctx.check(L.exa_init_state(ctx.h, ptr(d_sv[0]), ptr(d_quats_keep), None))
    v_nodes = hipref.velocity_field(rve)
    d_v = dev.up(v_nodes); d_x = dev.up(rve["X"])
    dts = [0.2, 0.4, 0.4]
    for i, dt in enumerate(dts):
        d_x += dt * d_v
        if i == len(dts) - 1:
            sv_in, s_in = d_sv[0].clone(), d_s[0].clone()
        ctx.check(L.exa_model_setup_lvec(ctx.h, dt, ptr(d_x), ptr(d_v), ptr(d_s[0]), ptr(d_sv[0]), ptr(d_s[1]), ptr(d_sv[1]), ptr(d_cm), ptr(d_J), None))
        assert ctx.check(L.exa_model_status(ctx.h, None)) == 0
        d_sv.reverse(); d_s.reverse()
    sig = d_s[0]
    # ---- sample
    rng = np.random.default_rng(3)
    es = np.sort(rng.choice(E, 16, replace=False)); ns = len(es)
    qidx = torch.from_numpy((es[:, None] * Q + np.arange(Q)[None, :]).ravel()).to(dev.dev)
    take = lambda t, w: t.view(P, w)[qidx].cpu().numpy().ravel()
    conn = rve["conn"].reshape(E, n)[es]
    x_end = d_x.cpu().numpy()
    xe = np.stack([x_end[conn + NN * c] for c in range(3)], axis=1).ravel()
    ve = np.stack([v_nodes[conn + NN * c] for c in range(3)], axis=1).ravel()
    Js = np.zeros(9 * ns * Q); orc.lib().orc_jacobians(p, ns, orc._p(xe), orc._p(Js))
    assert rel_l2(take(d_J, 9), Js) < 1e-13
    s1 = np.zeros(6 * ns * Q); sv = np.zeros(28 * ns * Q); cm = np.zeros(36 * ns * Q)
    nf = orc.lib().orc_model_setup(0, 0, orc._p(props), len(props), Q, ns, n, 28, C.c_double(dts[-1]), C.c_double(298.0), orc._p(Js), orc._p(rve["G"]),
                                   orc._p(ve), orc._p(take(s_in, 6)), orc._p(take(sv_in, 28)), orc._p(s1), orc._p(sv), orc._p(cm), None, 1, 0, 0)
    assert nf == 0
    assert rel_l2(take(sig, 6), s1) < 1e-9 and rel_l2(take(d_cm, 36), cm) < 1e-7
    # ---- B-bar residual of the sampled elements (E-vector) and B-bar element matrices
    eDS = np.zeros(3 * n * ns); orc.lib().orc_element_eds(Q, ns, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(Js), orc._p(eDS))
    y_ref = np.zeros(3 * n * ns)
    orc.lib().orc_add_mult_pa_bbar(Q, ns, n, orc._p(rve["W"]), orc._p(rve["G"]), orc._p(Js), orc._p(eDS), orc._p(s1), orc._p(y_ref))
    d_ye = dev.zeros(3 * n * E)
    ctx.check(L.exa_residual_setup(ctx.h, ptr(d_J), ptr(sig), None))
    ctx.check(L.exa_residual_apply(ctx.h, ptr(d_ye), None))
    eidx = torch.from_numpy(es).to(dev.dev)
    assert rel_l2(d_ye.view(E, 3 * n)[eidx].cpu().numpy(), y_ref) < 1e-9     # sigma itself agrees to 1e-9
    ctx.check(L.exa_grad_setup(ctx.h, dts[-1], ptr(d_J), ptr(d_cm), None))
    emat = np.zeros(9 * n * n * ns)
    orc.lib().orc_assemble_ea_bbar(Q, ns, n, C.c_double(dts[-1]), orc._p(rve["W"]), orc._p(rve["G"]), orc._p(Js), orc._p(eDS), orc._p(take(d_cm, 36)), orc._p(emat))
    d_em = dev.zeros(9 * n * n * E)
    ctx.check(L.exa_grad_get_ea(ctx.h, ptr(d_em), None))
    assert rel_l2(d_em.view(E, 9 * n * n)[eidx].cpu().numpy(), emat) < 1e-11
    del d_em
    # ---- element mat-vec on the whole operator: linear
    g = torch.Generator(device="cpu").manual_seed(5)
    x1 = torch.rand(3 * NN, generator=g, dtype=torch.float64).to(dev.dev) - 0.5
    x2 = torch.rand(3 * NN, generator=g, dtype=torch.float64).to(dev.dev) - 0.5
    mask = torch.zeros(3 * NN, dtype=torch.uint8, device=dev.dev)

    def apply(x):
        y = dev.zeros(3 * NN)
        ctx.check(L.exa_grad_apply_lvec(ctx.h, ptr(x), ptr(y), ptr(mask), None))
        return y
    y1, y2, y12 = apply(x1), apply(x2), apply(0.7 * x1 - 1.9 * x2)
    assert float((y12 - (0.7 * y1 - 1.9 * y2)).norm() / y12.norm()) < 1e-12
    ctx.close()


def test_config3_bcc_kmdd_128_sampled_against_oracle(oracle):
    """BASELINE config 3: 128^3 hex RVE, BCC Kocks-Mecking dislocation-density model on one MI355X.  The constitutive pass in the driver's layout
    (element-blocked, fused L-vector launch, tail split on) through the elastic-plastic transition; 48 sampled elements are recomputed by the
    oracle from the inputs the last GPU pass consumed."""
    import torch
    import exaconstit_amd.lib as L
    orc = oracle
    dev = hipref.Dev()
    N = 128
    rve = hipref.make_rve(orc, N)
    E, Q, n, NN = rve["E"], rve["Q"], rve["n"], rve["NN"]
    P = E * Q
    props = np.loadtxt(os.path.join(orc.REFDATA, "props_cp_mts.txt")).ravel()
    ctx = L.Context(L.EXA_BCC_KMDD, props, 298.0, 1, E, assembly=L.EXA_ASSEMBLY_PA)
    d_conn = torch.from_numpy(rve["conn"].astype(np.int32)).to(dev.dev)
    ctx.check(L.exa_set_connectivity(ctx.h, ptr(d_conn), NN))
    ctx.check(L.exa_set_quadrature_layout(ctx.h, L.EXA_QLAYOUT_EB64))
    ctx.check(L.exa_set_newton_cap(ctx.h, 8))
    nq = lambda w: int(L.exa_qf_size(ctx.h, w))
    quats = hipref.random_quats(E)
    d_sv = [dev.zeros(nq(28)), dev.zeros(nq(28))]; d_s = [dev.zeros(nq(6)), dev.zeros(nq(6))]
    d_cm = dev.zeros(nq(36)); d_J = dev.zeros(nq(9))
    d_quats_keep = dev.up(quats.ravel())
    ctx.check(L.exa_init_state(ctx.h, ptr(d_sv[0]), ptr(d_quats_keep), None))
    v_nodes = hipref.velocity_field(rve)
    d_v = dev.up(v_nodes); d_x = dev.up(rve["X"])
    dts = [0.005, 0.195, 0.1, 0.1, 0.1]
    for i, dt in enumerate(dts):
        d_x += dt * d_v
        if i == len(dts) - 1:
            sv_in, s_in = d_sv[0].clone(), d_s[0].clone()
        ctx.check(L.exa_model_setup_lvec(ctx.h, dt, ptr(d_x), ptr(d_v), ptr(d_s[0]), ptr(d_sv[0]), ptr(d_s[1]), ptr(d_sv[1]), ptr(d_cm), ptr(d_J), None))
        assert ctx.check(L.exa_model_status(ctx.h, None)) == 0
        d_sv.reverse(); d_s.reverse()
    assert ctx.check(L.exa_model_tail_count(ctx.h, None)) > 0            # the last launch did use the tail split

    def rows(t, w):   # element-blocked [block of 64][q][comp][lane] -> (P, w) in the reference's point order
        nb = (E + 63) // 64
        return t.view(nb, Q, w, 64).permute(0, 3, 1, 2).reshape(nb * 64 * Q, w)[:P]
    rng = np.random.default_rng(3)
    es = np.sort(rng.choice(E, 512, replace=False))      # 4 096 points: the evaluation-count criterion (>= 99.9 %) allows 4 ties
    qidx = torch.from_numpy((es[:, None] * Q + np.arange(Q)[None, :]).ravel()).to(dev.dev)
    take = lambda t, w: rows(t, w)[qidx].cpu().numpy().ravel()
    conn = rve["conn"].reshape(E, n)[es]
    x_end = d_x.cpu().numpy()
    xe = np.stack([x_end[conn + NN * c] for c in range(3)], axis=1).ravel()
    ve = np.stack([v_nodes[conn + NN * c] for c in range(3)], axis=1).ravel()
    Js = np.zeros(9 * len(es) * Q); orc.lib().orc_jacobians(1, len(es), orc._p(xe), orc._p(Js))
    assert rel_l2(take(d_J, 9), Js) < 1e-13
    s1 = np.zeros(6 * len(es) * Q); sv = np.zeros(28 * len(es) * Q); cm = np.zeros(36 * len(es) * Q)
    nf = orc.lib().orc_model_setup(1, 2, orc._p(props), len(props), Q, len(es), n, 28, C.c_double(dts[-1]), C.c_double(298.0), orc._p(Js), orc._p(rve["G"]),
                                   orc._p(ve), orc._p(take(s_in, 6)), orc._p(take(sv_in, 28)), orc._p(s1), orc._p(sv), orc._p(cm), None, 1, 0, 0)
    assert nf == 0
    assert rel_l2(take(d_s[0], 6), s1) < 1e-9
    assert rel_l2(take(d_cm, 36), cm) < 1e-7
    keep = np.ones(28, bool); keep[3] = False
    assert rel_l2(take(d_sv[0], 28).reshape(-1, 28)[:, keep], sv.reshape(-1, 28)[:, keep]) < 1e-8
    nf_gpu, nf_ref = take(d_sv[0], 28).reshape(-1, 28)[:, 3], sv.reshape(-1, 28)[:, 3]      # evaluation counts (tail split included) follow the oracle's
    assert nf_ref.max() > 4
    assert np.abs(nf_gpu - nf_ref).max() <= 1 and np.mean(nf_gpu == nf_ref) >= 0.999, (np.abs(nf_gpu - nf_ref).max(), np.mean(nf_gpu == nf_ref))
    assert np.abs(sv.reshape(-1, 28)[:, 14:26]).sum() > 0                  # plastic
    ctx.close()


def test_headline_record_route_128_sampled_against_oracle(oracle):
    """The exact launch of the headline number - k_model_setup<8, true, 8, true, true>: FCC Voce with the exponent compiled in, fused L-vector gathers,
    element-blocked layout, compact gradient records instead of a tangent field, no Jacobian field (exa_model_setup_lvec_records, what bench.py's `value`
    times) - at 128^3 through the elastic-plastic transition.  512 sampled elements (4 096 points) are recomputed by the oracle from the inputs the last
    launch consumed: stress, state, evaluation counts.  The records have no oracle counterpart point by point (the reference streams C4 / D4), so the
    record-based gradient action is compared (a) over the whole RVE with the action built from the tangent FIELD of an AOS context at the same state,
    and (b) that context's element-local action (exa_grad_apply on E-vectors) with the oracle's TransformMatGradTo4D -> AssembleGradPA -> AddMultGradPA
    chain on the oracle's own tangents of the sampled elements (reference src/mechanics_model.cpp:949-1061, src/mechanics_integrators.cpp:331-513, 562-622)."""
    import torch
    import exaconstit_amd.lib as L
    orc = oracle
    dev = hipref.Dev()
    N = 128
    rve = hipref.make_rve(orc, N)
    E, Q, n, NN = rve["E"], rve["Q"], rve["n"], rve["NN"]
    P = E * Q
    props = np.loadtxt(os.path.join(orc.REFDATA, "props_cp_voce.txt")).ravel()
    d_conn = torch.from_numpy(rve["conn"].astype(np.int32)).to(dev.dev)
    ctx = L.Context(L.EXA_FCC_VOCE, props, 298.0, 1, E, assembly=L.EXA_ASSEMBLY_PA)
    ctx.check(L.exa_set_connectivity(ctx.h, ptr(d_conn), NN))
    ctx.check(L.exa_set_quadrature_layout(ctx.h, L.EXA_QLAYOUT_EB64))
    ctx.check(L.exa_set_tangent_form(ctx.h, L.EXA_TANGENT_DEV5_BULK))
    nq = lambda w: int(L.exa_qf_size(ctx.h, w))
    quats = hipref.random_quats(E)
    d_sv = [dev.zeros(nq(28)), dev.zeros(nq(28))]; d_s = [dev.zeros(nq(6)), dev.zeros(nq(6))]
    d_quats_keep = dev.up(quats.ravel())
    ctx.check(L.exa_init_state(ctx.h, ptr(d_sv[0]), ptr(d_quats_keep), None))
    v_nodes = hipref.velocity_field(rve)
    d_v = dev.up(v_nodes); d_x = dev.up(rve["X"])
    dts = [0.005, 0.195, 0.4, 0.4]
    for i, dt in enumerate(dts):
        d_x += dt * d_v
        if i == len(dts) - 1:
            sv_in, s_in = d_sv[0].clone(), d_s[0].clone()
        ctx.check(L.exa_model_setup_lvec_records(ctx.h, dt, ptr(d_x), ptr(d_v), ptr(d_s[0]), ptr(d_sv[0]), ptr(d_s[1]), ptr(d_sv[1]), None, None))   # no Jacobian field
        assert ctx.check(L.exa_model_status(ctx.h, None)) == 0
        d_sv.reverse(); d_s.reverse()

    def rows(t, w):   # element-blocked [block of 64][q][comp][lane] -> (P, w) in the reference's point order
        nb = (E + 63) // 64
        return t.view(nb, Q, w, 64).permute(0, 3, 1, 2).reshape(nb * 64 * Q, w)[:P]
    rng = np.random.default_rng(11)
    es = np.sort(rng.choice(E, 512, replace=False))
    qidx = torch.from_numpy((es[:, None] * Q + np.arange(Q)[None, :]).ravel()).to(dev.dev)
    take = lambda t, w: rows(t, w)[qidx].cpu().numpy().ravel()
    conn = rve["conn"].reshape(E, n)[es]
    x_end = d_x.cpu().numpy()
    xe = np.stack([x_end[conn + NN * c] for c in range(3)], axis=1).ravel()
    ve = np.stack([v_nodes[conn + NN * c] for c in range(3)], axis=1).ravel()
    ns = len(es)
    Js = np.zeros(9 * ns * Q); orc.lib().orc_jacobians(1, ns, orc._p(xe), orc._p(Js))
    s1 = np.zeros(6 * ns * Q); sv = np.zeros(28 * ns * Q); cm = np.zeros(36 * ns * Q)
    nf = orc.lib().orc_model_setup(0, 0, orc._p(props), len(props), Q, ns, n, 28, C.c_double(dts[-1]), C.c_double(298.0), orc._p(Js), orc._p(rve["G"]),
                                   orc._p(ve), orc._p(take(s_in, 6)), orc._p(take(sv_in, 28)), orc._p(s1), orc._p(sv), orc._p(cm), None, 1, 0, 0)
    assert nf == 0
    assert rel_l2(take(d_s[0], 6), s1) < 1e-9
    keep = np.ones(28, bool); keep[3] = False
    assert rel_l2(take(d_sv[0], 28).reshape(-1, 28)[:, keep], sv.reshape(-1, 28)[:, keep]) < 1e-8
    nf_gpu, nf_ref = take(d_sv[0], 28).reshape(-1, 28)[:, 3], sv.reshape(-1, 28)[:, 3]
    assert nf_ref.max() > 4 and np.abs(sv.reshape(-1, 28)[:, 14:26]).sum() > 0          # plastic: the local solves iterated
    assert np.abs(nf_gpu - nf_ref).max() <= 1 and np.mean(nf_gpu == nf_ref) >= 0.999, (np.abs(nf_gpu - nf_ref).max(), np.mean(nf_gpu == nf_ref))

    # ---- the record-based action (geometry from the nodes, compact records of the launch above) over the whole RVE
    ctx.check(L.exa_grad_set_coords(ctx.h, ptr(d_x)))
    g = torch.Generator(device="cpu").manual_seed(5)
    x1 = torch.rand(3 * NN, generator=g, dtype=torch.float64).to(dev.dev) - 0.5
    y_rec = dev.zeros(3 * NN)
    ctx.check(L.exa_grad_apply_lvec(ctx.h, ptr(x1), ptr(y_rec), None, None))
    # the same state through an AOS context with a tangent field and full 46-double records
    aos = L.Context(L.EXA_FCC_VOCE, props, 298.0, 1, E, assembly=L.EXA_ASSEMBLY_PA)
    aos.check(L.exa_set_connectivity(aos.h, ptr(d_conn), NN))
    a_sv0 = rows(sv_in, 28).contiguous().view(-1); a_s0 = rows(s_in, 6).contiguous().view(-1)
    a_s1, a_sv1, a_cm, a_J = dev.zeros(6 * P), dev.zeros(28 * P), dev.zeros(36 * P), dev.zeros(9 * P)
    aos.check(L.exa_model_setup_lvec(aos.h, dts[-1], ptr(d_x), ptr(d_v), ptr(a_s0), ptr(a_sv0), ptr(a_s1), ptr(a_sv1), ptr(a_cm), ptr(a_J), None))
    assert aos.check(L.exa_model_status(aos.h, None)) == 0
    assert float((a_s1.view(P, 6) - rows(d_s[0], 6)).abs().max()) < 1e-12 * float(a_s1.abs().max())      # (another instantiation of the kernel: round-off)
    aos.check(L.exa_grad_setup(aos.h, dts[-1], ptr(a_J), ptr(a_cm), None))
    y_tan = dev.zeros(3 * NN)
    aos.check(L.exa_grad_apply_lvec(aos.h, ptr(x1), ptr(y_tan), None, None))
    assert float((y_rec - y_tan).norm() / y_tan.norm()) < 1e-11
    # ... and that context's element-local action against the oracle's chain on the oracle's tangents of the sampled elements
    d_xe = dev.zeros(3 * n * E); d_ye = dev.zeros(3 * n * E)
    aos.check(L.exa_restrict(aos.h, ptr(x1), ptr(d_xe), None))
    aos.check(L.exa_grad_apply(aos.h, ptr(d_xe), ptr(d_ye), None))
    eidx = torch.from_numpy(es).to(dev.dev)
    ye_gpu = d_ye.view(E, 3 * n)[eidx].cpu().numpy().ravel(); xe_act = d_xe.view(E, 3 * n)[eidx].cpu().numpy().ravel()
    C4 = np.zeros(81 * ns * Q); D4 = np.zeros(81 * ns * Q); ye_ref = np.zeros(3 * n * ns)
    orc.lib().orc_transform_4d(C.c_int64(ns * Q), orc._p(cm), orc._p(C4))
    orc.lib().orc_assemble_grad_pa(Q, ns, C.c_double(dts[-1]), orc._p(rve["W"]), orc._p(Js), orc._p(C4), orc._p(D4))
    orc.lib().orc_add_mult_grad_pa(Q, ns, n, orc._p(rve["G"]), orc._p(D4), orc._p(xe_act), orc._p(ye_ref))
    assert rel_l2(ye_gpu, ye_ref) < 1e-7          # (the tangents agree to 1e-7: the tolerance of the tangent comparison everywhere in this suite)
    aos.close(); ctx.close()
