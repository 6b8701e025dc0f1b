// TEST INFRASTRUCTURE (tests/test_adapters.py).  Drives the L-vector pair of include/exaconstit_mfem_adapters.hpp (HipExaModelLVec / HipExaNLFIntegratorLVec),
// compiled against mock_mfem.hpp, the way an operator without element restrictions does (reference src/mechanics_operator.cpp:339-341 hands a model the
// global velocity vector, src/mechanics_operator_ext.cpp:159-165 hands the integrators global vectors): ModelSetup with the velocity L-vector, then
// AssemblePA/AddMultPA, AssembleGradPA/AddMultGradPA/AssembleGradDiagonalPA, all through base-class pointers.
//   in : int32 E, model, nprops, order, nnodes, flags (1 compact tangent form, 2 records fused into ModelSetup); double dt; props[nprops]; int32 conn[n*E]; xend[3*nnodes]; vel[3*nnodes]; quats[4*E]; x[3*nnodes]
//   out: stress1[6P] state1[28P] matGrad[36P] jac[9P] y_res[3*nnodes] y_grad[3*nnodes] diag[3*nnodes]
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
      const int E = hdr[0], model_id = hdr[1], nprops = hdr[2], order = hdr[3], nn = hdr[4]; const bool compact = (hdr[5] & 1) != 0, fused = (hdr[5] & 2) != 0;
      double dt; if (fread(&dt, 8, 1, fi) != 1) return 3;
      const int n = (order + 1) * (order + 1) * (order + 1), Q = n, P = E * Q;
      std::vector<double> props = rd(fi, nprops);
      std::vector<int> conn((size_t)n * E); if (fread(conn.data(), 4, conn.size(), fi) != conn.size()) return 3;
      std::vector<double> xend = rd(fi, (size_t)3 * nn), vel = rd(fi, (size_t)3 * nn), quats = rd(fi, (size_t)4 * E), x = rd(fi, (size_t)3 * nn);
      fclose(fi);
      mfem::Vector vprops(nprops); vprops.FromHost(props.data());
      mfem::QuadratureFunction s0(P, 6), s1(P, 6), mg(P, 36), v0(P, 28), v1(P, 28);
      mfem::ParGridFunction bc(3 * nn), ec(3 * nn); ec.FromHost(xend.data());
      mfem::Mesh mesh; mfem::FiniteElementSpace fes(&mesh, order); fes.SetElementDofs(conn.data(), n, E, nn);
      HipExaModelLVec model(&s0, &s1, &mg, &v0, &v1, &bc, &ec, &vprops, nprops, 28, 298.0, model_id, fes, Assembly::PA, false, true, fused);
      mfem::Vector vq(4 * E); vq.FromHost(quats.data());
      model.InitStateVars(vq);
      model.SetModelDt(dt);
      HipExaNLFIntegratorLVec integ(&model, compact);
      mfem::Vector none(1), vvel(3 * nn); vvel.FromHost(vel.data());
      ExaModel* base = &model;                               // through the base-class seam, as the reference calls it
      base->ModelSetup(Q, E, 3, n, none, none, vvel);
      mfem::NonlinearFormIntegrator* nlf = &integ;
      mfem::Vector yres(3 * nn), ygrad(3 * nn), diag(3 * nn), vx(3 * nn); vx.FromHost(x.data());
      nlf->AssemblePA(fes); nlf->AddMultPA(vx, yres);
      nlf->AssembleGradPA(fes); nlf->AddMultGradPA(vx, ygrad);
      if (!fused) nlf->AssembleGradDiagonalPA(diag);      // (fused records: no tangent field, hence no diagonal - the reference's Jacobi smoother never reads it)
      (void)hipDeviceSynchronize();
      FILE* fo = fopen(argv[2], "wb"); if (!fo) return 4;
      wr(fo, s1); wr(fo, v1); wr(fo, mg); wr(fo, model.Jacobians()); wr(fo, yres); wr(fo, ygrad); wr(fo, diag);
      fclose(fo);
      return 0;
   } catch (const std::exception& e) { fprintf(stderr, "adapter_run_lvec: %s\n", e.what()); return 1; }
}
