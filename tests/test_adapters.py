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


def _build(out):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + MOCK, "-I" + os.path.join(ROOT, "include"), "-o", out,
           os.path.join(MOCK, "adapter_run.cpp"), "-L" + os.path.join(ROOT, "exaconstit_amd"), "-lexaconstit_hip", "-Wl,-rpath," + os.path.join(ROOT, "exaconstit_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]


def test_adapters_compile_against_the_mock(tmp_path):
    import exaconstit_amd.lib  # noqa: F401  (builds the library if needed)
    _build(str(tmp_path / "adapter_run"))


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
    ctx.check(L.exa_grad_setup(ctx.h, dt, ptr(d_J), ptr(o[2]), None))
    if ea:
        ctx.check(L.exa_grad_get_ea(ctx.h, ptr(emat), None))
    else:
        ctx.check(L.exa_grad_apply(ctx.h, ptr(d_x), ptr(ygrad), None)); ctx.check(L.exa_grad_diagonal(ctx.h, ptr(diag), None))
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
