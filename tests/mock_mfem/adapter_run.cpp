// TEST INFRASTRUCTURE (tests/test_adapters.py).  Drives include/exaconstit_mfem_adapters.hpp, compiled against mock_mfem.hpp, the way
// NonlinearMechOperator does: ModelSetup, then AssemblePA/AddMultPA (residual), AssembleGradPA/AddMultGradPA/AssembleGradDiagonalPA (PA
// gradient) and AssembleEA.  Inputs come from a flat binary file written by the test, outputs go to another one; the test compares them
// with the same calls made directly on the C ABI.
//   in : int32 E, model, nprops, assembly (0 PA, 1 EA), order (1 | 2), bbar (0 | 1); double dt; props[nprops]; geomJ[Q*9*E] (Q,3,3,E); vel[n*3*E]; quats[4*E]; x[n*3*E] (E-vector for the action)
//   out: stress1[6P] state1[28P] matGrad[36P] y_res[3nE] y_grad[3nE] diag[3nE] emat[(3n)^2 E] dp[9P]
#define EXA_ADAPTER_MOCK_MFEM
#include "exaconstit_mfem_adapters.hpp"
#include <cstdio>
#include <vector>

static std::vector<double> rd(FILE* f, size_t n) { std::vector<double> v(n); if (fread(v.data(), 8, n, f) != n) throw std::runtime_error("short input"); return v; }
static void wr(FILE* f, const mfem::Vector& v) { fwrite(v.HostRead(), 8, v.Size(), f); }

int main(int argc, char** argv) {
   if (argc < 3) return 2;
   try {
      FILE* fi = fopen(argv[1], "rb"); if (!fi) return 3;
      int hdr[6]; if (fread(hdr, 4, 6, fi) != 6) return 3;
      const int E = hdr[0], model_id = hdr[1], nprops = hdr[2]; const bool ea = hdr[3] != 0; const int order = hdr[4]; const bool bbar = hdr[5] != 0;
      double dt; if (fread(&dt, 8, 1, fi) != 1) return 3;
      const int n = (order + 1) * (order + 1) * (order + 1), Q = n, P = E * Q;
      std::vector<double> props = rd(fi, nprops), gj = rd(fi, (size_t)Q * 9 * E), vel = rd(fi, (size_t)n * 3 * E), quats = rd(fi, (size_t)4 * E), x = rd(fi, (size_t)n * 3 * E);
      fclose(fi);
      mfem::Vector vprops(nprops); vprops.FromHost(props.data());
      mfem::QuadratureFunction s0(P, 6), s1(P, 6), mg(P, 36), v0(P, 28), v1(P, 28), dp(P, 9);
      mfem::ParGridFunction bc(3 * 27), ec(3 * 27);
      HipExaModel model(&s0, &s1, &mg, &v0, &v1, &bc, &ec, &vprops, nprops, 28, 298.0, model_id, order, E, ea ? Assembly::EA : Assembly::PA, bbar);
      mfem::Vector vq(4 * E); vq.FromHost(quats.data());
      model.InitStateVars(vq);
      model.SetModelDt(dt);
      mfem::Mesh mesh; mesh.factors().J.SetSize(Q * 9 * E); mesh.factors().J.FromHost(gj.data());
      mfem::FiniteElementSpace fes(&mesh, order);
      HipExaNLFIntegrator integ(&model);
      // NonlinearMechOperator::Setup hands ModelSetup the (3,3,Q,E) Jacobians (src/mechanics_operator.cpp:377-391)
      mfem::Vector jac(Q * 9 * E), locgrad(1), vvel(n * 3 * E); vvel.FromHost(vel.data());
      EXA_ADAPTER_VERIFY(exa_jacobians_from_geom(model.ctx(), mesh.factors().J.Read(), jac.Write(), nullptr) == EXA_OK, "relayout");
      ExaModel* base = &model;                               // through the base-class seam, as the reference calls it
      base->ModelSetup(Q, E, 3, n, jac, locgrad, vvel);
      mfem::NonlinearFormIntegrator* nlf = &integ;
      mfem::Vector yres(3 * n * E), ygrad(3 * n * E), diag(3 * n * E), emat(9 * n * n * E), vx(3 * n * E); vx.FromHost(x.data());
      nlf->AssemblePA(fes); nlf->AddMultPA(vx, yres);
      // src/mechanics_operator.cpp:436-443: PA builds the PA gradient (+ diagonal for the smoother), EA assembles the element matrices
      if (!ea) { nlf->AssembleGradPA(fes); nlf->AddMultGradPA(vx, ygrad); nlf->AssembleGradDiagonalPA(diag); }
      else nlf->AssembleEA(fes, emat);
      base->calcDpMat(dp);
      (void)hipDeviceSynchronize();
      FILE* fo = fopen(argv[2], "wb"); if (!fo) return 4;
      wr(fo, s1); wr(fo, v1); wr(fo, mg); wr(fo, yres); wr(fo, ygrad); wr(fo, diag); wr(fo, emat); wr(fo, dp);
      fclose(fo);
      return 0;
   } catch (const std::exception& e) { fprintf(stderr, "adapter_run: %s\n", e.what()); return 1; }
}
