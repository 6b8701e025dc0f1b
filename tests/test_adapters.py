"""include/exaconstit_mfem_adapters.hpp - the HipExaModel / HipExaNLFIntegrator classes a maintainer adds to ExaConstit - compiled against
tests/mock_mfem/mock_mfem.hpp (MFEM and ExaConstit's own headers are not in this image) and, on the GPU box, run: ModelSetup ->
AssemblePA/AddMultPA -> AssembleGradPA/AddMultGradPA/AssembleGradDiagonalPA -> AssembleEA -> calcDpMat through the base-class seams give
exactly what the same sequence of direct C-ABI calls gives."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_mfem")


def _build(out, src="adapter_run.cpp"):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + MOCK, "-I" + os.path.join(ROOT, "include"), "-o", out,
           os.path.join(MOCK, src), "-L" + os.path.join(ROOT, "exaconstit_amd"), "-lexaconstit_hip", "-Wl,-rpath," + os.path.join(ROOT, "exaconstit_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]


def test_adapters_compile_against_the_mock(tmp_path):
    import exaconstit_amd.lib  # noqa: F401  (builds the library if needed)
    _build(str(tmp_path / "adapter_run"))
    _build(str(tmp_path / "adapter_run_lvec"), "adapter_run_lvec.cpp")


def test_adapter_header_refuses_to_compile_without_mfem(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "exaconstit_mfem_adapters.hpp"\nint main() { return 0; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode != 0 and "needs MFEM" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("model,pfile,ea,order,bbar", [(0, "props_cp_voce.txt", 0, 1, 0), (5, "props_cp_mts.txt", 0, 1, 0), (0, "props_cp_voce.txt", 1, 1, 0),
                                                        (0, "props_cp_voce.txt", 1, 1, 1), (0, "props_cp_voce.txt", 1, 2, 0), (0, "props_cp_voce.txt", 1, 2, 1),
                                                        (0, "props_cp_voce.txt", 0, 2, 0),
                                                        # the orders of the reference's own integrator unit tests (test/mechanics_test.cpp:54,313,471)
                                                        (0, "props_cp_voce.txt", 0, 3, 0), (0, "props_cp_voce.txt", 1, 3, 0), (0, "props_cp_voce.txt", 1, 3, 1)])
def test_adapters_forward_to_the_abi(oracle, tmp_path, model, pfile, ea, order, bbar):
    """order = 2 and bbar = true (ICExaNLFIntegrator users, reference src/mechanics_integrators.hpp:78-124: element assembly only, like the
    reference's B-bar integrator) go through the same adapter classes: HipExaModel(..., order, nelems, assembly, bbar)."""
    import exaconstit_amd.lib as L
    import hipref
    from hipref import ptr
    orc = oracle
    exe = str(tmp_path / "adapter_run")
    _build(exe)
    dev = hipref.Dev()
    rve = hipref.make_rve(orc, 2, p=order, distort=0.2 if order == 1 else 0.1, seed=3)
    E, Q, n = rve["E"], rve["Q"], rve["n"]; P = E * Q
    assert Q == (order + 1) ** 3 and n == Q
    props = np.loadtxt(os.path.join(orc.REFDATA, pfile)).ravel()
    quats = hipref.random_quats(E, seed=9)
    dt = 0.4
    vel_e = hipref.l_to_e(rve, hipref.velocity_field(rve, scale=3.0))
    xe = hipref.l_to_e(rve, rve["X"])
    x_act = np.random.default_rng(1).uniform(-1, 1, 3 * n * E)
    ctx = L.Context(model, props, 298.0, order, E, assembly=L.EXA_ASSEMBLY_EA if ea else L.EXA_ASSEMBLY_PA, integ=L.EXA_INTEG_BBAR if bbar else L.EXA_INTEG_FULL)
    d_J = dev.zeros(9 * P)
    ctx.check(L.exa_jacobians(ctx.h, ptr(dev.up(xe)), ptr(d_J), None))
    J = d_J.cpu().numpy().reshape(E, Q, 9)                       # (3,3,Q,E), first index fastest
    gj = np.ascontiguousarray(J.transpose(0, 2, 1)).ravel()      # (Q,3,3,E): gj[q + Q (c + 9 e)]
    # ---- direct ABI calls
    d_gj = dev.up(gj); d_J2 = dev.zeros(9 * P)
    ctx.check(L.exa_jacobians_from_geom(ctx.h, ptr(d_gj), ptr(d_J2), None))
    assert np.array_equal(d_J2.cpu().numpy(), d_J.cpu().numpy())
    d_sv0 = dev.zeros(28 * P); ctx.check(L.exa_init_state(ctx.h, ptr(d_sv0), ptr(dev.up(quats.ravel())), None))
    d_s0 = dev.zeros(6 * P); o = [dev.zeros(6 * P), dev.zeros(28 * P), dev.zeros(36 * P)]
    ctx.check(L.exa_model_setup(ctx.h, dt, ptr(d_J), ptr(dev.up(vel_e)), ptr(d_s0), ptr(d_sv0), ptr(o[0]), ptr(o[1]), ptr(o[2]), None))
    assert ctx.check(L.exa_model_status(ctx.h, None)) == 0
    yres, ygrad, diag = dev.zeros(3 * n * E), dev.zeros(3 * n * E), dev.zeros(3 * n * E)
    emat = dev.zeros(9 * n * n * E); dp = dev.zeros(9 * P); d_x = dev.up(x_act)
    ctx.check(L.exa_residual_setup(ctx.h, ptr(d_J), ptr(o[0]), None)); ctx.check(L.exa_residual_apply(ctx.h, ptr(yres), None))
    # (what HipExaNLFIntegrator does: p = 1 partial assembly streams the compact tangent + adj(J) after the defect check; other contexts refuse the form)
    if L.exa_set_tangent_form(ctx.h, L.EXA_TANGENT_DEV5_BULK_GEO) == L.EXA_OK:
        assert order == 1 and not ea and not bbar
        defect = C.c_double(1.0); ctx.check(L.exa_grad_tangent_defect(ctx.h, ptr(o[2]), C.byref(defect), None)); assert defect.value < 1e-11
    else:
        assert order != 1 or ea or bbar
    ctx.check(L.exa_grad_setup(ctx.h, dt, ptr(d_J), ptr(o[2]), None))
    if ea:
        ctx.check(L.exa_grad_get_ea(ctx.h, ptr(emat), None))
    else:
        ctx.check(L.exa_grad_apply(ctx.h, ptr(d_x), ptr(ygrad), None)); ctx.check(L.exa_grad_diagonal(ctx.h, ptr(diag), None))
        if order == 1:   # the compact form against the oracle's TransformMatGradTo4D -> AssembleGradPA -> AddMultGradPA chain on the same tangent and Jacobians
            cmh = o[2].cpu().numpy(); Jh = d_J.cpu().numpy(); C4 = np.zeros(81 * P); D4 = np.zeros(81 * P); ye = np.zeros(3 * n * E)
            orc.lib().orc_transform_4d(C.c_int64(P), orc._p(cmh), orc._p(C4))
            orc.lib().orc_assemble_grad_pa(Q, E, C.c_double(dt), orc._p(rve["W"]), orc._p(Jh), orc._p(C4), orc._p(D4))
            orc.lib().orc_add_mult_grad_pa(Q, E, n, orc._p(rve["G"]), orc._p(D4), orc._p(x_act), orc._p(ye))
            assert hipref.rel_l2(ygrad.cpu().numpy(), ye) < 1e-12
    ctx.check(L.exa_calc_dp(ctx.h, ptr(o[1]), ptr(dp), None))
    want = [t.cpu().numpy() for t in (o[0], o[1], o[2], yres, ygrad, diag, emat, dp)]
    ctx.close()
    # ---- the same through the adapter classes
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("iiiiii", E, model, len(props), ea, order, bbar)); f.write(struct.pack("d", dt))
        for a in (props, gj, vel_e, quats.ravel(), x_act):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(fout, dtype=np.float64)
    off = 0
    for w, name in zip(want, ("stress1", "state1", "matGrad", "AddMultPA", "AddMultGradPA", "diagonal", "emat", "DpMat")):
        g = got[off:off + w.size]; off += w.size
        if name in ("stress1", "state1", "matGrad", "AddMultPA") or (ea and name == "emat") or (not ea and name in ("AddMultGradPA", "diagonal")):
            assert np.linalg.norm(w) > 0, name
        assert np.array_equal(g, w), (name, np.abs(g - w).max())
    assert off == got.size
    assert np.abs(want[1].reshape(P, 28)[:, 14:26]).sum() > 0         # the step was plastic


@pytest.mark.gpu
@pytest.mark.parametrize("model,pfile,order,compact", [(0, "props_cp_voce.txt", 1, 1), (0, "props_cp_voce.txt", 1, 0), (5, "props_cp_mts.txt", 1, 1), (0, "props_cp_voce.txt", 2, 1),
                                                        (0, "props_cp_voce.txt", 1, 3), (5, "props_cp_mts.txt", 1, 3)])
def test_lvec_adapters_forward_to_the_abi(oracle, tmp_path, model, pfile, order, compact):
    """The L-vector pair (HipExaModelLVec / HipExaNLFIntegratorLVec): ModelSetup with the velocity L-vector, AddMultPA / AddMultGradPA / the diagonal on
    L-vectors, through base-class pointers of the mock - against the same sequence of direct C-ABI calls (constitutive outputs and Jacobians bit for bit;
    the L-vector sums, which FP64 atomics add in a run-dependent order, to round-off) and, for the gradient action, against the oracle's restatement of the
    reference's TransformMatGradTo4D -> AssembleGradPA -> AddMultGradPA chain on the same tangent and Jacobians."""
    import exaconstit_amd.lib as L
    import hipref
    import torch
    from hipref import ptr, rel_l2
    orc = oracle
    exe = str(tmp_path / "adapter_run_lvec")
    _build(exe, "adapter_run_lvec.cpp")
    dev = hipref.Dev()
    rve = hipref.make_rve(orc, 3 if order == 1 else 2, p=order, distort=0.2 if order == 1 else 0.1, seed=3)
    E, Q, n, NN = rve["E"], rve["Q"], rve["n"], rve["NN"]; P = E * Q
    props = np.loadtxt(os.path.join(orc.REFDATA, pfile)).ravel()
    quats = hipref.random_quats(E, seed=9)
    dt = 0.4
    v_nodes = hipref.velocity_field(rve, scale=3.0)
    xend = rve["X"] + dt * v_nodes
    x_act = np.random.default_rng(1).uniform(-1, 1, 3 * NN)
    # ---- direct ABI calls
    ctx = L.Context(model, props, 298.0, order, E)
    d_conn = torch.from_numpy(rve["conn"].astype(np.int32)).to(dev.dev)
    ctx.check(L.exa_set_connectivity(ctx.h, ptr(d_conn), NN))
    if compact & 1:
        ctx.check(L.exa_set_tangent_form(ctx.h, L.EXA_TANGENT_DEV5_BULK))
    d_sv0 = dev.zeros(28 * P); ctx.check(L.exa_init_state(ctx.h, ptr(d_sv0), ptr(dev.up(quats.ravel())), None))
    d_s0 = dev.zeros(6 * P); o = [dev.zeros(6 * P), dev.zeros(28 * P), dev.zeros(36 * P), dev.zeros(9 * P)]
    d_x = dev.up(xend); d_v = dev.up(v_nodes)
    ctx.check(L.exa_model_setup_lvec(ctx.h, dt, ptr(d_x), ptr(d_v), ptr(d_s0), ptr(d_sv0), ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(o[3]), None))
    assert ctx.check(L.exa_model_status(ctx.h, None)) == 0
    yres, ygrad, diag, ev = dev.zeros(3 * NN), dev.zeros(3 * NN), dev.zeros(3 * NN), dev.zeros(3 * n * E)
    ctx.check(L.exa_residual_lvec(ctx.h, ptr(o[3]), ptr(o[0]), ptr(yres), None))
    ctx.check(L.exa_grad_setup(ctx.h, dt, ptr(o[3]), ptr(o[2]), None)); ctx.check(L.exa_grad_set_coords(ctx.h, ptr(d_x)))
    d_xa = dev.up(x_act)
    ctx.check(L.exa_grad_apply_lvec(ctx.h, ptr(d_xa), ptr(ygrad), None, None))
    ctx.check(L.exa_grad_diagonal(ctx.h, ptr(ev), None)); ctx.check(L.exa_restrict_transpose_add(ctx.h, ptr(ev), ptr(diag), None))
    want = [t.cpu().numpy() for t in (o[0], o[1], o[2], o[3], yres, ygrad, diag)]
    # the action against the oracle's chain (reference src/mechanics_model.cpp:949-1061, src/mechanics_integrators.cpp:331-513, 562-622) between L->E and E->L
    ye = np.zeros(3 * n * E); C4 = np.zeros(81 * P); D4 = np.zeros(81 * P)
    orc.lib().orc_transform_4d(C.c_int64(P), orc._p(want[2]), orc._p(C4))
    orc.lib().orc_assemble_grad_pa(Q, E, C.c_double(dt), orc._p(rve["W"]), orc._p(want[3]), orc._p(C4), orc._p(D4))
    orc.lib().orc_add_mult_grad_pa(Q, E, n, orc._p(rve["G"]), orc._p(D4), orc._p(hipref.l_to_e(rve, x_act)), orc._p(ye))
    assert rel_l2(want[5], hipref.e_to_l(rve, ye)) < 1e-12
    ctx.close()
    # ---- the same through the adapter classes
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("iiiiii", E, model, len(props), order, NN, compact)); f.write(struct.pack("d", dt))
        f.write(np.ascontiguousarray(props, dtype=np.float64).tobytes()); f.write(np.ascontiguousarray(rve["conn"], dtype=np.int32).tobytes())
        for a in (xend, v_nodes, quats.ravel(), x_act):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(fout, dtype=np.float64)
    off = 0
    fused = bool(compact & 2)      # HipExaModelLVec(.., fused_records = true): ModelSetup = exa_model_setup_lvec_records, matGrad and the diagonal stay untouched
    for w, name in zip(want, ("stress1", "state1", "matGrad", "jacobians", "AddMultPA", "AddMultGradPA", "diagonal")):
        g = got[off:off + w.size]; off += w.size
        assert np.linalg.norm(w) > 0, name
        if fused and name in ("matGrad", "diagonal"):
            assert not g.any(), name
        elif fused and name in ("stress1", "state1"):      # another instantiation of the kernel: round-off (evaluation counts equal)
            if name == "state1":
                assert np.array_equal(g.reshape(P, 28)[:, 3], w.reshape(P, 28)[:, 3])
            assert rel_l2(g, w) < 1e-13, (name, rel_l2(g, w))
        elif name in ("AddMultPA", "AddMultGradPA", "diagonal"):
            assert rel_l2(g, w) < (1e-12 if fused else 1e-13), (name, rel_l2(g, w))
        else:
            assert np.array_equal(g, w), (name, np.abs(g - w).max())
    assert off == got.size
    assert np.abs(want[1].reshape(P, 28)[:, 14:26]).sum() > 0         # the step was plastic
