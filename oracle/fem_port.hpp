// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under exaconstit_amd/ may include, link or call this.
//
// CPU restatement (serial element loop, inner quadrature loop — the loop structure of the reference's
// rtmodel="CPU" path) of the finite-element side of the hot path:
//   grad_calc                      reference src/mechanics_kernels.cpp:7-78
//   kernel_setup / postprocessing  reference src/mechanics_ecmech.cpp:22-172
//   ExaCMechModel::ModelSetup      reference src/mechanics_ecmech.cpp:192-258
//   ExaNLFIntegrator::AssemblePA / AddMultPA / AssembleGradPA / AddMultGradPA / AssembleGradDiagonalPA /
//   AssembleEA, AssembleElementVector / AssembleElementGrad
//                                  reference src/mechanics_integrators.cpp:31-156,160-314,331-513,518-622,625-748,756-1017
//   ExaModel::TransformMatGradTo4D reference src/mechanics_model.cpp:949-1061
//   ExaModel::GenerateGradMatrix   reference src/mechanics_model.cpp:776-843
// plus the MFEM pieces they rely on (H1 hex basis, Gauss-Legendre rule, Cartesian mesh, L<->E restriction),
// which are not in /root/reference; those are pinned by the reference's self-consistency unit tests
// (test/mechanics_test.cpp, test/grad_test.cpp) restated in tests/test_oracle_fem.py.
//
// Layouts (all column-major = first index fastest):
//   E-vector (node, comp, elem); J (3,3,Q,E) with J(i,j)=dx_i/dxi_j; shape-derivative table G (node, dir, qpt);
//   quadrature data (vdim, Q, E); global vectors byNODES [x0..xN, y0.., z0..].
#pragma once
#include <vector>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "ecmech_port.hpp"

namespace fem {

// ---------------------------------------------------------------------------------------------
// reference element: H1 hexahedron of order p with (p+1)^3 Gauss-Legendre points
// ---------------------------------------------------------------------------------------------
struct RefElem {
   int p, n, Q;
   std::vector<double> G;   // (n,3,Q)
   std::vector<double> W;   // (Q)
   std::vector<int> lex2nat; // lexicographic (i + (p+1)(j + (p+1)k)) -> native node index
};

inline void gauss_legendre_01(int np, double* x, double* w) {
   if (np == 1) { x[0] = 0.5; w[0] = 1.0; }
   else if (np == 2) { const double a = 0.5 / std::sqrt(3.0); x[0] = 0.5 - a; x[1] = 0.5 + a; w[0] = w[1] = 0.5; }
   else if (np == 3) { const double a = 0.5 * std::sqrt(0.6); x[0] = 0.5 - a; x[1] = 0.5; x[2] = 0.5 + a; w[0] = w[2] = 5.0 / 18.0; w[1] = 8.0 / 18.0; }
   else {   // Newton on Legendre polynomials
      for (int i = 0; i < np; i++) {
         double z = std::cos(M_PI * (i + 0.75) / (np + 0.5)), pp = 0;
         for (int it = 0; it < 100; it++) {
            double p1 = 1, p2 = 0;
            for (int j = 1; j <= np; j++) { double p3 = p2; p2 = p1; p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j; }
            pp = np * (z * p1 - p2) / (z * z - 1.0);
            double dz = p1 / pp; z -= dz; if (std::fabs(dz) < 1e-16) break;
         }
         x[np - 1 - i] = 0.5 * (1.0 + z); w[np - 1 - i] = 1.0 / ((1.0 - z * z) * pp * pp);
      }
   }
}

// Gauss-Lobatto-Legendre points on [0, 1] (MFEM's H1 nodes, BasisType::GaussLobatto): closed forms up to four points; beyond, the interior
// points are the roots of P'_n (n = np - 1), bisected between consecutive roots of P_n (which they interlace), with P'_k from the
// recurrence P'_{k+1} = P'_{k-1} + (2k + 1) P_k.
inline double legendre_deriv(int n, double z, double* pn = nullptr) {
   double pkm1 = 1.0, pk = z, dkm1 = 0.0, dk = 1.0;      // P_0, P_1, P'_0, P'_1
   if (n == 0) { if (pn) *pn = 1.0; return 0.0; }
   for (int k = 1; k < n; k++) {
      const double pkp1 = ((2.0 * k + 1.0) * z * pk - k * pkm1) / (k + 1.0);
      const double dkp1 = dkm1 + (2.0 * k + 1.0) * pk;
      pkm1 = pk; pk = pkp1; dkm1 = dk; dk = dkp1;
   }
   if (pn) *pn = pk;
   return dk;
}
inline void gauss_lobatto_01(int np, double* x) {
   if (np == 2) { x[0] = 0; x[1] = 1; return; }
   if (np == 3) { x[0] = 0; x[1] = 0.5; x[2] = 1; return; }
   if (np == 4) { const double a = 0.5 / std::sqrt(5.0); x[0] = 0; x[1] = 0.5 - a; x[2] = 0.5 + a; x[3] = 1; return; }
   const int n = np - 1;
   std::vector<double> xg(n), wg(n); gauss_legendre_01(n, xg.data(), wg.data());      // roots of P_n on [0, 1], ascending
   x[0] = 0.0; x[n] = 1.0;
   for (int i = 1; i < n; i++) {
      double lo = 2.0 * xg[i - 1] - 1.0, hi = 2.0 * xg[i] - 1.0, flo = legendre_deriv(n, lo);
      for (int it = 0; it < 200; it++) {
         const double mid = 0.5 * (lo + hi), fm = legendre_deriv(n, mid);
         if ((fm < 0) == (flo < 0)) { lo = mid; flo = fm; } else hi = mid;
         if (hi - lo < 1e-17) break;
      }
      x[i] = 0.5 * (1.0 + 0.5 * (lo + hi));
   }
   for (int i = 1; 2 * i <= n; i++) { const double s = 0.5 * (x[i] + (1.0 - x[n - i])); x[i] = s; x[n - i] = 1.0 - s; }
}

// 1-D Lagrange basis on nodes xn evaluated at x: value and derivative
inline void lagrange_1d(int np, const double* xn, double x, double* v, double* d) {
   for (int a = 0; a < np; a++) {
      double num = 1, den = 1;
      for (int b = 0; b < np; b++) if (b != a) { num *= (x - xn[b]); den *= (xn[a] - xn[b]); }
      v[a] = num / den;
      double s = 0;
      for (int c = 0; c < np; c++) if (c != a) { double t = 1; for (int b = 0; b < np; b++) if (b != a && b != c) t *= (x - xn[b]); s += t; }
      d[a] = s / den;
   }
}

// native hex ordering: vertices, edge interiors, face interiors, volume interiors (App. D of SURVEY.md)
inline void hex_native_order(int p, std::vector<int>& lex2nat) {
   const int np = p + 1, n = np * np * np;
   lex2nat.assign(n, -1);
   auto L = [&](int i, int j, int k) { return i + np * (j + np * k); };
   int c = 0;
   const int V[8][3] = { { 0, 0, 0 }, { p, 0, 0 }, { p, p, 0 }, { 0, p, 0 }, { 0, 0, p }, { p, 0, p }, { p, p, p }, { 0, p, p } };
   for (int v = 0; v < 8; v++) lex2nat[L(V[v][0], V[v][1], V[v][2])] = c++;
   const int Ed[12][2] = { { 0, 1 }, { 1, 2 }, { 3, 2 }, { 0, 3 }, { 4, 5 }, { 5, 6 }, { 7, 6 }, { 4, 7 }, { 0, 4 }, { 1, 5 }, { 2, 6 }, { 3, 7 } };
   for (int e = 0; e < 12; e++) for (int t = 1; t < p; t++) {
      int a = Ed[e][0], b = Ed[e][1];
      int i = V[a][0] + (V[b][0] - V[a][0]) * t / p, j = V[a][1] + (V[b][1] - V[a][1]) * t / p, k = V[a][2] + (V[b][2] - V[a][2]) * t / p;
      lex2nat[L(i, j, k)] = c++;
   }
   // faces: bottom(z=0), front(y=0), right(x=p), back(y=p), left(x=0), top(z=p)
   for (int f = 0; f < 6; f++) for (int t2 = 1; t2 < p; t2++) for (int t1 = 1; t1 < p; t1++) {
      int i, j, k;
      switch (f) {
         case 0: i = t1; j = t2; k = 0; break;
         case 1: i = t1; j = 0; k = t2; break;
         case 2: i = p; j = t1; k = t2; break;
         case 3: i = t1; j = p; k = t2; break;
         case 4: i = 0; j = t1; k = t2; break;
         default: i = t1; j = t2; k = p; break;
      }
      lex2nat[L(i, j, k)] = c++;
   }
   for (int k = 1; k < p; k++) for (int j = 1; j < p; j++) for (int i = 1; i < p; i++) lex2nat[L(i, j, k)] = c++;
}

inline void ref_elem_init(RefElem& re, int p) {
   re.p = p; const int np = p + 1; re.n = np * np * np; re.Q = re.n;
   std::vector<double> xq(np), wq(np), xn(np);
   gauss_legendre_01(np, xq.data(), wq.data());
   gauss_lobatto_01(np, xn.data());
   hex_native_order(p, re.lex2nat);
   re.G.assign((size_t)re.n * 3 * re.Q, 0.0); re.W.assign(re.Q, 0.0);
   std::vector<double> vx(np), dx(np), vy(np), dy(np), vz(np), dz(np);
   for (int qk = 0; qk < np; qk++) for (int qj = 0; qj < np; qj++) for (int qi = 0; qi < np; qi++) {
      const int q = qi + np * (qj + np * qk);   // x index fastest
      re.W[q] = wq[qi] * wq[qj] * wq[qk];
      lagrange_1d(np, xn.data(), xq[qi], vx.data(), dx.data());
      lagrange_1d(np, xn.data(), xq[qj], vy.data(), dy.data());
      lagrange_1d(np, xn.data(), xq[qk], vz.data(), dz.data());
      for (int k = 0; k < np; k++) for (int j = 0; j < np; j++) for (int i = 0; i < np; i++) {
         const int a = re.lex2nat[i + np * (j + np * k)];
         re.G[a + re.n * (0 + 3 * q)] = dx[i] * vy[j] * vz[k];
         re.G[a + re.n * (1 + 3 * q)] = vx[i] * dy[j] * vz[k];
         re.G[a + re.n * (2 + 3 * q)] = vx[i] * vy[j] * dz[k];
      }
   }
}

// ---------------------------------------------------------------------------------------------
// Cartesian hex mesh (Mesh::MakeCartesian3D(nx,ny,nz,HEX,sx,sy,sz,sfc=false): x fastest)
// ---------------------------------------------------------------------------------------------
struct Mesh {
   int nx, ny, nz, p;
   int nnx, nny, nnz;         // nodes per direction = n*p+1
   int E, NN, n;              // elements, nodes, nodes per element
   std::vector<int> conn;     // (n, E) global node of element-local node
   std::vector<double> X;     // byNODES coordinates (NN*3)
};

inline void mesh_init(Mesh& m, const RefElem& re, int nx, int ny, int nz, double sx, double sy, double sz) {
   const int p = re.p, np = p + 1;
   m.nx = nx; m.ny = ny; m.nz = nz; m.p = p;
   m.nnx = nx * p + 1; m.nny = ny * p + 1; m.nnz = nz * p + 1;
   m.E = nx * ny * nz; m.NN = m.nnx * m.nny * m.nnz; m.n = re.n;
   m.conn.assign((size_t)m.n * m.E, 0); m.X.assign((size_t)m.NN * 3, 0.0);
   std::vector<double> xn(np); gauss_lobatto_01(np, xn.data());
   for (int ez = 0; ez < nz; ez++) for (int ey = 0; ey < ny; ey++) for (int ex = 0; ex < nx; ex++) {
      const int e = ex + nx * (ey + ny * ez);
      for (int k = 0; k < np; k++) for (int j = 0; j < np; j++) for (int i = 0; i < np; i++) {
         const int a = re.lex2nat[i + np * (j + np * k)];
         const int gi = ex * p + i, gj = ey * p + j, gk = ez * p + k;
         const int g = gi + m.nnx * (gj + m.nny * gk);
         m.conn[a + m.n * e] = g;
         m.X[g] = sx * (ex + xn[i]) / nx; m.X[g + m.NN] = sy * (ey + xn[j]) / ny; m.X[g + 2 * m.NN] = sz * (ez + xn[k]) / nz;
      }
   }
}

// Threads of the element / point loops: 1 = serial like the reference's rtmodel=CPU path (MFEM_FORALL on the cpu device); orc_set_threads(n > 1) is
// the stand-in for its rtmodel=OPENMP / mpirun -np n runs in bench.py's cpu_baseline leg.  Every parallel loop below writes disjoint outputs per
// element (the E->L sum stays serial), so the results do not depend on the thread count.
inline int& model_threads() { static int n = 1; return n; }
#define ORC_PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(model_threads()) if (model_threads() > 1)")

// L-vector (byNODES) -> E-vector (node, comp, elem)
inline void restrict_LtoE(const Mesh& m, const double* L, double* Ev) {
   ORC_PAR_FOR
   for (int e = 0; e < m.E; e++) for (int c = 0; c < 3; c++) for (int a = 0; a < m.n; a++)
      Ev[a + m.n * (c + 3 * e)] = L[m.conn[a + m.n * e] + m.NN * c];
}

inline void restrict_EtoL_add(const Mesh& m, const double* Ev, double* L) {
   for (int e = 0; e < m.E; e++) for (int c = 0; c < 3; c++) for (int a = 0; a < m.n; a++)
      L[m.conn[a + m.n * e] + m.NN * c] += Ev[a + m.n * (c + 3 * e)];
}

// J(i,j,q,e) = sum_a x(a,i,e) * G(a,j,q)           (MFEM GeometricFactors + re-layout, mechanics_operator.cpp:350-391)
inline void jacobians(const RefElem& re, int E, const double* xe /*(n,3,E)*/, double* J /*(3,3,Q,E)*/) {
   const int n = re.n, Q = re.Q;
   ORC_PAR_FOR
   for (int e = 0; e < E; e++) for (int q = 0; q < Q; q++) for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) {
      double s = 0;
      for (int a = 0; a < n; a++) s += xe[a + n * (i + 3 * e)] * re.G[a + n * (j + 3 * q)];
      J[i + 3 * (j + 3 * (q + (size_t)Q * e))] = s;
   }
}

inline double det3(const double* J) {   // J col-major (i + 3 j)
   return J[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * J[7] - J[4] * J[6]);
}

// adj[3 r + c] = adj(J)(r,c)  where adj(J) = det(J) * J^-1           (mechanics_integrators.cpp:252-270)
inline void adjugate3(const double* J, double* adj) {
   auto Jf = [&](int i, int j) { return J[i + 3 * j]; };
   for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
      const int r1 = (r + 1) % 3, r2 = (r + 2) % 3, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
      // cofactor of J(c,r)
      adj[3 * r + c] = Jf(c1, r1) * Jf(c2, r2) - Jf(c1, r2) * Jf(c2, r1);
   }
}

// grad_calc: field_grad(q,t,qpt,e) += sum_{r,s} field(r,q,e) G(r,s,qpt) Jinv(s,t)   (output must be pre-zeroed)
inline void grad_calc(int Q, int E, int n, const double* J, const double* G, const double* field, double* fgrad) {
   ORC_PAR_FOR
   for (int e = 0; e < E; e++) {
      for (int q = 0; q < Q; q++) {
         const double* Jq = &J[9 * (q + (size_t)Q * e)];
         const double detJ = det3(Jq);
         double adj[9]; adjugate3(Jq, adj);
         double Jinv[3][3];
         for (int s = 0; s < 3; s++) for (int t = 0; t < 3; t++) Jinv[s][t] = adj[3 * s + t] / detJ;
         double* out = &fgrad[9 * (q + (size_t)Q * e)];
         for (int t = 0; t < 3; t++) for (int s = 0; s < 3; s++) for (int r = 0; r < n; r++) for (int c = 0; c < 3; c++)
            out[c + 3 * t] += field[r + n * (c + 3 * e)] * G[r + n * (s + 3 * q)] * Jinv[s][t];
      }
   }
}

// ---------------------------------------------------------------------------------------------
// ExaCMechModel::ModelSetup  (K2..K9)
// ---------------------------------------------------------------------------------------------
struct ModelOpts { bool transpose_tangent = true; ecm::PointOpts po; };

// returns number of points whose local solve failed
inline int model_setup(const ecm::Model& mdl, int Q, int E, int n, int nstatev, double dt, double temp_k,
                       const double* J, const double* G, const double* vel_e,
                       const double* stress0, const double* state0,
                       double* stress1, double* state1, double* ddsdde, double* vgrad_out /*nullable*/,
                       const ModelOpts& mo = ModelOpts()) {
   const size_t P = (size_t)Q * E;
   std::memcpy(stress1, stress0, sizeof(double) * 6 * P);            // StressSetup     mechanics_model.cpp:158-168
   std::memcpy(state1, state0, sizeof(double) * nstatev * P);        // StateVarsSetup  mechanics_model.cpp:170-180
   std::memset(ddsdde, 0, sizeof(double) * 36 * P);
   std::vector<double> vgrad(9 * P, 0.0);
   grad_calc(Q, E, n, J, G, vel_e, vgrad.data());
   if (vgrad_out) std::memcpy(vgrad_out, vgrad.data(), sizeof(double) * 9 * P);
   const int ind_int_eng = nstatev - 1, ind_vols = ind_int_eng - 1, ind_pl_work = ecm::iHistA_flowStr;
   int nfail = 0;
   // serial like the reference's rtmodel=CPU path unless orc_set_threads(n > 1) asked for the rtmodel=OPENMP analogue
#pragma omp parallel for reduction(+ : nfail) schedule(dynamic, 64) num_threads(model_threads()) if (model_threads() > 1)
   for (size_t ip = 0; ip < P; ip++) {
      const double* L = &vgrad[9 * ip];               // L(i,j) = L[i + 3 j]
      double* sv = &state1[nstatev * ip];
      const double* sv0 = &state0[nstatev * ip];
      double* sig = &stress1[6 * ip];
      auto Lf = [&](int i, int j) { return L[i + 3 * j]; };
      // ---- kernel_setup
      double tempk = temp_k, eng_int[1] = { sv[ind_int_eng] };
      double w_vec[3] = { 0.5 * (Lf(2, 1) - Lf(1, 2)), 0.5 * (Lf(0, 2) - Lf(2, 0)), 0.5 * (Lf(1, 0) - Lf(0, 1)) };
      const double d_mean = -(1.0 / 3.0) * (Lf(0, 0) + Lf(1, 1) + Lf(2, 2));
      double d_svec_p[7] = { Lf(0, 0) + d_mean, Lf(1, 1) + d_mean, Lf(2, 2) + d_mean,
                             0.5 * (Lf(2, 1) + Lf(1, 2)), 0.5 * (Lf(2, 0) + Lf(0, 2)), 0.5 * (Lf(1, 0) + Lf(0, 1)), -3.0 * d_mean };
      double d_vecd[5]; ecm::svec_to_vecd(d_svec_p, d_vecd);
      const double dEff = ecm::vecd_Deff(d_vecd);
      double vr[4];
      vr[0] = sv[ind_vols]; vr[1] = vr[0] * std::exp(d_svec_p[6] * dt); vr[3] = vr[1] - vr[0]; vr[2] = vr[3] / (dt * 0.5 * (vr[0] + vr[1]));
      double s_svec_p[7];
      const double s_mean = -(1.0 / 3.0) * (sig[0] + sig[1] + sig[2]);
      for (int i = 0; i < 6; i++) s_svec_p[i] = sig[i];
      s_svec_p[0] += s_mean; s_svec_p[1] += s_mean; s_svec_p[2] += s_mean; s_svec_p[6] = s_mean;
      // ---- the crystal update
      double sdd[2];
      double* mt = &ddsdde[36 * ip];
      const int rc = ecm::get_response_sngl(mdl, dt, d_svec_p, w_vec, vr, eng_int, s_svec_p, sv, tempk, sdd, mt, mo.po);
      if (rc && std::getenv("ORC_DEBUG") && nfail < 3) {
         std::fprintf(stderr, "ORC_FAIL dt=%.17g\n d=", dt); for (int i = 0; i < 7; i++) std::fprintf(stderr, "%.17g,", d_svec_p[i]);
         std::fprintf(stderr, "\n w="); for (int i = 0; i < 3; i++) std::fprintf(stderr, "%.17g,", w_vec[i]);
         std::fprintf(stderr, "\n vr="); for (int i = 0; i < 4; i++) std::fprintf(stderr, "%.17g,", vr[i]);
         std::fprintf(stderr, "\n sig0="); for (int i = 0; i < 6; i++) std::fprintf(stderr, "%.17g,", stress0[6 * ip + i]);
         std::fprintf(stderr, "\n sv0="); for (int i = 0; i < nstatev; i++) std::fprintf(stderr, "%.17g,", sv0[i]);
         std::fprintf(stderr, "\n");
      }
      nfail += rc;
      // ---- kernel_postprocessing
      sv[ind_vols] = vr[1];
      sv[ind_int_eng] = eng_int[0];
      if (dEff > ecm::idp_tiny_sqrt) sv[ind_pl_work] *= dEff * dt; else sv[ind_pl_work] = 0.0;
      sv[ind_pl_work] += sv0[ind_pl_work];
      for (int i = 0; i < 6; i++) sig[i] = s_svec_p[i];
      sig[0] -= s_svec_p[6]; sig[1] -= s_svec_p[6]; sig[2] -= s_svec_p[6];
      if (mo.transpose_tangent) for (int i = 0; i < 6; i++) for (int j = i + 1; j < 6; j++) std::swap(mt[6 * i + j], mt[6 * j + i]);
   }
   return nfail;
}

// ---------------------------------------------------------------------------------------------
// integrators (full integration)
// ---------------------------------------------------------------------------------------------
// AssemblePA: D(j,k,q,e) = W_q * sum_l sigma(k,l) adj(J)(j,l)          mechanics_integrators.cpp:240-312
inline void assemble_pa(int Q, int E, const double* W, const double* J, const double* stress1, double* dmat) {
   static const int V[3][3] = { { 0, 5, 4 }, { 5, 1, 3 }, { 4, 3, 2 } };
   ORC_PAR_FOR
   for (int e = 0; e < E; e++) for (int q = 0; q < Q; q++) {
      const size_t ip = q + (size_t)Q * e;
      double adj[9]; adjugate3(&J[9 * ip], adj);
      const double* S = &stress1[6 * ip];
      for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) {
         double s = 0; for (int l = 0; l < 3; l++) s += S[V[k][l]] * adj[3 * j + l];
         dmat[j + 3 * (k + 3 * ip)] = s * W[q];
      }
   }
}

// AddMultPA: Y(i,k,e) += sum_q sum_j G(i,j,q) D(j,k,q,e)                mechanics_integrators.cpp:545-555
inline void add_mult_pa(int Q, int E, int n, const double* G, const double* dmat, double* Y) {
   ORC_PAR_FOR
   for (int e = 0; e < E; e++) for (int q = 0; q < Q; q++) {
      const double* D = &dmat[9 * (q + (size_t)Q * e)];
      for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) for (int i = 0; i < n; i++)
         Y[i + n * (k + 3 * e)] += G[i + n * (j + 3 * q)] * D[j + 3 * k];
   }
}

inline int voigt(int i, int j) { static const int V[3][3] = { { 0, 5, 4 }, { 5, 1, 3 }, { 4, 3, 2 } }; return V[i][j]; }

// TransformMatGradTo4D: C4(i,j,k,l,p) = C(voigt(i,j), voigt(k,l), p)    mechanics_model.cpp:970-1060
inline void transform_matgrad_4d(size_t P, const double* C /*(6,6,P)*/, double* C4 /*(3,3,3,3,P)*/) {
   ORC_PAR_FOR
   for (size_t ip = 0; ip < P; ip++)
      for (int l = 0; l < 3; l++) for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++)
         C4[i + 3 * (j + 3 * (k + 3 * (l + 3 * ip)))] = C[voigt(i, j) + 6 * (voigt(k, l) + 6 * ip)];
}

// AssembleGradPA: D4(e,q,i,k,l,n) = dt W/detJ * sum_{j,m} A(j,i) C4(j,k,l,m) A(m,n), A(r,c)=adj[r+3c] (col-major view)
//                                                                       mechanics_integrators.cpp:425-511
inline void assemble_grad_pa(int Q, int E, double dt, const double* W, const double* J, const double* C4, double* D4 /*row-major (E,Q,3,3,3,3)*/) {
   ORC_PAR_FOR
   for (int e = 0; e < E; e++) for (int q = 0; q < Q; q++) {
      const size_t ip = q + (size_t)Q * e;
      const double* Jq = &J[9 * ip];
      double adj[9]; adjugate3(Jq, adj);
      const double c_detJ = 1.0 / det3(Jq) * W[q] * dt;
      auto A = [&](int r, int c) { return adj[r + 3 * c]; };
      const double* C = &C4[81 * ip];
      double* D = &D4[81 * ip];
      for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) for (int nn = 0; nn < 3; nn++) {
         double s = 0;
         for (int j = 0; j < 3; j++) for (int mm = 0; mm < 3; mm++) s += A(j, i) * C[j + 3 * (k + 3 * (l + 3 * mm))] * A(mm, nn);
         D[((i * 3 + k) * 3 + l) * 3 + nn] = s * c_detJ;
      }
   }
}

// AddMultGradPA                                                            mechanics_integrators.cpp:592-620
inline void add_mult_grad_pa(int Q, int E, int n, const double* G, const double* D4, const double* X, double* Y) {
   ORC_PAR_FOR
   for (int e = 0; e < E; e++) for (int q = 0; q < Q; q++) {
      const double* D = &D4[81 * (q + (size_t)Q * e)];
      double gx[3][3];   // gx[i][j] = sum_k G(k,j,q) X(k,i,e)
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < n; k++) s += G[k + n * (j + 3 * q)] * X[k + n * (i + 3 * e)]; gx[i][j] = s; }
      double T[3][3];    // T(a,b) = sum_{i,j} D(a,b,i,j) gx[i][j]
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { double s = 0; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) s += D[((a * 3 + b) * 3 + i) * 3 + j] * gx[i][j]; T[a][b] = s; }
      for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) for (int i = 0; i < n; i++) Y[i + n * (k + 3 * e)] += G[i + n * (j + 3 * q)] * T[j][k];
   }
}

// AssembleGradDiagonalPA                                                   mechanics_integrators.cpp:702-743
inline void assemble_grad_diag_pa(int Q, int E, int n, double dt, const double* W, const double* G, const double* J, const double* K /*(6,6,Q,E)*/, double* Y) {
   // Voigt rows hit by a unit displacement gradient of component c: (xx|xy|xz) etc.
   static const int R[3][3] = { { 0, 5, 4 }, { 5, 1, 3 }, { 4, 3, 2 } };
   ORC_PAR_FOR
   for (int e = 0; e < E; e++) for (int q = 0; q < Q; q++) {
      const size_t ip = q + (size_t)Q * e;
      const double* Jq = &J[9 * ip];
      double adj[9]; adjugate3(Jq, adj);
      const double c_detJ = 1.0 / det3(Jq) * W[q] * dt;
      const double* Kq = &K[36 * ip];
      for (int a = 0; a < n; a++) {
         double b[3];
         for (int t = 0; t < 3; t++) { double s = 0; for (int j = 0; j < 3; j++) s += G[a + n * (j + 3 * q)] * adj[3 * j + t]; b[t] = s; }   // detJ * dN_a/dx_t
         for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int r = 0; r < 3; r++) for (int t = 0; t < 3; t++) s += b[r] * Kq[R[c][r] + 6 * R[c][t]] * b[t];
            Y[a + n * (c + 3 * e)] += c_detJ * s;
         }
      }
   }
}

// B^T rows for node a with physical gradient g: (3 dofs) x 6 Voigt                 mechanics_model.cpp:776-843
inline void b_rows(const double* g, double Bt[3][6]) {
   for (int c = 0; c < 3; c++) for (int v = 0; v < 6; v++) Bt[c][v] = 0.0;
   Bt[0][0] = g[0]; Bt[0][4] = g[2]; Bt[0][5] = g[1];
   Bt[1][1] = g[1]; Bt[1][3] = g[2]; Bt[1][5] = g[0];
   Bt[2][2] = g[2]; Bt[2][3] = g[1]; Bt[2][4] = g[0];
}

// AssembleEA: emat(3n,3n,E) col-major, dof = node + n*comp              mechanics_integrators.cpp:849-1015
inline void assemble_ea(int Q, int E, int n, double dt, const double* W, const double* G, const double* J, const double* K, double* emat) {
   const int nd = 3 * n;
   std::vector<double> Bt((size_t)nd * 6);
   for (int e = 0; e < E; e++) {
      double* M = &emat[(size_t)nd * nd * e];
      for (int i = 0; i < nd * nd; i++) M[i] = 0.0;
      for (int q = 0; q < Q; q++) {
         const size_t ip = q + (size_t)Q * e;
         const double* Jq = &J[9 * ip];
         const double detJ = det3(Jq);
         double adj[9]; adjugate3(Jq, adj);
         const double wt = dt * W[q] * detJ;
         const double* Kq = &K[36 * ip];
         for (int a = 0; a < n; a++) {
            double g[3];
            for (int t = 0; t < 3; t++) { double s = 0; for (int j = 0; j < 3; j++) s += G[a + n * (j + 3 * q)] * adj[3 * j + t] / detJ; g[t] = s; }
            double b[3][6]; b_rows(g, b);
            for (int c = 0; c < 3; c++) for (int v = 0; v < 6; v++) Bt[(a + n * c) * 6 + v] = b[c][v];
         }
         for (int cj = 0; cj < nd; cj++) {
            double CB[6];
            for (int u = 0; u < 6; u++) { double s = 0; for (int v = 0; v < 6; v++) s += Kq[u + 6 * v] * Bt[cj * 6 + v]; CB[u] = s; }
            for (int ri = 0; ri < nd; ri++) { double s = 0; for (int u = 0; u < 6; u++) s += Bt[ri * 6 + u] * CB[u]; M[ri + nd * cj] += wt * s; }
         }
      }
   }
}

// EA mat-vec: Y(j,e) += sum_i A(i,j,e) X(i,e)                            spec mechanics_operator_ext.cpp:303-314
inline void ea_mult(int E, int n, const double* emat, const double* X, double* Y) {
   const int nd = 3 * n;
   for (int e = 0; e < E; e++) for (int j = 0; j < nd; j++) {
      double s = 0; for (int i = 0; i < nd; i++) s += emat[i + nd * (j + (size_t)nd * e)] * X[i + nd * e];
      Y[j + nd * e] += s;
   }
}

inline void ea_diag(int E, int n, const double* emat, double* Y) {
   const int nd = 3 * n;
   for (int e = 0; e < E; e++) for (int j = 0; j < nd; j++) Y[j + nd * e] = emat[j + nd * (j + (size_t)nd * e)];
}

// dense element residual: AssembleElementVector                          mechanics_integrators.cpp:31-94
inline void element_vector(int Q, int E, int n, const double* W, const double* G, const double* J, const double* stress1, double* Y) {
   for (int e = 0; e < E; e++) for (int q = 0; q < Q; q++) {
      const size_t ip = q + (size_t)Q * e;
      const double* Jq = &J[9 * ip];
      const double detJ = det3(Jq);
      double adj[9]; adjugate3(Jq, adj);
      const double* S = &stress1[6 * ip];
      for (int a = 0; a < n; a++) {
         double g[3];
         for (int t = 0; t < 3; t++) { double s = 0; for (int j = 0; j < 3; j++) s += G[a + n * (j + 3 * q)] * adj[3 * j + t] / detJ; g[t] = s; }
         for (int k = 0; k < 3; k++) { double s = 0; for (int l = 0; l < 3; l++) s += g[l] * S[voigt(l, k)]; Y[a + n * (k + 3 * e)] += s * detJ * W[q]; }
      }
   }
}

// ---------------------------------------------------------------------------------------------
// B-bar (Hughes) integrator: ICExaNLFIntegrator                        mechanics_integrators.hpp:78-124
// ---------------------------------------------------------------------------------------------
// element-average shape gradient eDS(a,t,e) = sum_q W G adj(J) / sum_q W detJ      mechanics_integrators.cpp:1895-1953
inline void element_eds(int Q, int E, int n, const double* W, const double* G, const double* J, double* eDS /*(n,3,E)*/) {
   for (int e = 0; e < E; e++) {
      double vol = 0;
      double* ed = &eDS[(size_t)3 * n * e];
      for (int i = 0; i < 3 * n; i++) ed[i] = 0.0;
      for (int q = 0; q < Q; q++) {
         const double* Jq = &J[9 * (q + (size_t)Q * e)];
         double adj[9]; adjugate3(Jq, adj);
         vol += W[q] * det3(Jq);
         for (int a = 0; a < n; a++) for (int t = 0; t < 3; t++) {
            double s = 0; for (int j = 0; j < 3; j++) s += G[a + n * (j + 3 * q)] * adj[3 * j + t];
            ed[a + n * t] += W[q] * s;
         }
      }
      for (int i = 0; i < 3 * n; i++) ed[i] /= vol;
   }
}

// B-bar^T rows of node a: physical gradient g, element-average gradient ge           mechanics_model.cpp:845-877
inline void bbar_rows(const double* g, const double* ge, double Bt[3][6]) {
   b_rows(g, Bt);
   for (int c = 0; c < 3; c++) { const double v = (ge[c] - g[c]) / 3.0; for (int k = 0; k < 3; k++) Bt[c][k] += v; }
}

// PA residual with B-bar, written as the reference writes it (b4..b9)                mechanics_integrators.cpp:2011-2086
inline void add_mult_pa_bbar(int Q, int E, int n, const double* W, const double* G, const double* J, const double* eDS, const double* S, double* Y) {
   const double i3 = 1.0 / 3.0;
   for (int e = 0; e < E; e++) for (int q = 0; q < Q; q++) {
      const size_t ip = q + (size_t)Q * e;
      const double* Jq = &J[9 * ip];
      const double detJ = det3(Jq), idetJ = 1.0 / detJ, c_detJ = detJ * W[q];
      double adj[9]; adjugate3(Jq, adj);
      const double* s = &S[6 * ip];
      for (int a = 0; a < n; a++) {
         double b[3];
         for (int t = 0; t < 3; t++) { double v = 0; for (int j = 0; j < 3; j++) v += G[a + n * (j + 3 * q)] * adj[3 * j + t]; b[t] = idetJ * v; }
         const double b4 = i3 * (eDS[a + n * (0 + 3 * e)] - b[0]), b5 = b4 + b[0];
         const double b6 = i3 * (eDS[a + n * (1 + 3 * e)] - b[1]), b7 = b6 + b[1];
         const double b8 = i3 * (eDS[a + n * (2 + 3 * e)] - b[2]), b9 = b8 + b[2];
         Y[a + n * (0 + 3 * e)] += c_detJ * (b4 * s[1] + b4 * s[2] + b5 * s[0] + b[1] * s[5] + b[2] * s[4]);
         Y[a + n * (1 + 3 * e)] += c_detJ * (b6 * s[0] + b6 * s[2] + b7 * s[1] + b[0] * s[5] + b[2] * s[3]);
         Y[a + n * (2 + 3 * e)] += c_detJ * (b8 * s[0] + b8 * s[1] + b9 * s[2] + b[0] * s[4] + b[1] * s[3]);
      }
   }
}

// element matrices / dense residual with B-bar rows (ICExaNLFIntegrator::AssembleEA, AssembleElementVector)
//                                                                                     mechanics_integrators.cpp:1021-1187,1195-1604
inline void assemble_ea_bbar(int Q, int E, int n, double dt, const double* W, const double* G, const double* J, const double* eDS, const double* K, double* emat) {
   const int nd = 3 * n;
   std::vector<double> Bt((size_t)nd * 6);
   for (int e = 0; e < E; e++) {
      double* M = &emat[(size_t)nd * nd * e];
      for (int i = 0; i < nd * nd; i++) M[i] = 0.0;
      for (int q = 0; q < Q; q++) {
         const size_t ip = q + (size_t)Q * e;
         const double* Jq = &J[9 * ip];
         const double detJ = det3(Jq);
         double adj[9]; adjugate3(Jq, adj);
         const double wt = dt * W[q] * detJ;
         const double* Kq = &K[36 * ip];
         for (int a = 0; a < n; a++) {
            double g[3], ge[3];
            for (int t = 0; t < 3; t++) { double s = 0; for (int j = 0; j < 3; j++) s += G[a + n * (j + 3 * q)] * adj[3 * j + t] / detJ; g[t] = s; ge[t] = eDS[a + n * (t + 3 * e)]; }
            double b[3][6]; bbar_rows(g, ge, b);
            for (int c = 0; c < 3; c++) for (int v = 0; v < 6; v++) Bt[(a + n * c) * 6 + v] = b[c][v];
         }
         for (int cj = 0; cj < nd; cj++) {
            double CB[6];
            for (int u = 0; u < 6; u++) { double s = 0; for (int v = 0; v < 6; v++) s += Kq[u + 6 * v] * Bt[cj * 6 + v]; CB[u] = s; }
            for (int ri = 0; ri < nd; ri++) { double s = 0; for (int u = 0; u < 6; u++) s += Bt[ri * 6 + u] * CB[u]; M[ri + nd * cj] += wt * s; }
         }
      }
   }
}

inline void element_vector_bbar(int Q, int E, int n, const double* W, const double* G, const double* J, const double* eDS, const double* stress1, double* Y) {
   for (int e = 0; e < E; e++) for (int q = 0; q < Q; q++) {
      const size_t ip = q + (size_t)Q * e;
      const double* Jq = &J[9 * ip];
      const double detJ = det3(Jq);
      double adj[9]; adjugate3(Jq, adj);
      const double* S = &stress1[6 * ip];
      for (int a = 0; a < n; a++) {
         double g[3], ge[3];
         for (int t = 0; t < 3; t++) { double s = 0; for (int j = 0; j < 3; j++) s += G[a + n * (j + 3 * q)] * adj[3 * j + t] / detJ; g[t] = s; ge[t] = eDS[a + n * (t + 3 * e)]; }
         double b[3][6]; bbar_rows(g, ge, b);
         for (int c = 0; c < 3; c++) { double s = 0; for (int v = 0; v < 6; v++) s += b[c][v] * S[v]; Y[a + n * (c + 3 * e)] += s * detJ * W[q]; }
      }
   }
}

// volume average: sum_q W detJ val / sum_q W detJ                        mechanics_kernels.hpp:19-134
inline void vol_avg(int Q, int E, int vdim, const double* W, const double* J, const double* qf, double* out, bool normalise) {
   std::vector<double> acc(vdim, 0.0); double vol = 0;
   for (int e = 0; e < E; e++) for (int q = 0; q < Q; q++) {
      const size_t ip = q + (size_t)Q * e;
      const double w = W[q] * det3(&J[9 * ip]);
      vol += w;
      for (int c = 0; c < vdim; c++) acc[c] += w * qf[c + vdim * ip];
   }
   for (int c = 0; c < vdim; c++) out[c] = normalise ? acc[c] / vol : acc[c];
}

// calcDpMat                                                              mechanics_ecmech.hpp:303-357
inline void calc_dp_mat(const ecm::Model& mdl, size_t P, int nstatev, const double* state1, double* dp /*(3,3,P)*/) {
   for (size_t ip = 0; ip < P; ip++) {
      const double* sv = &state1[nstatev * ip];
      double dphat[5] = { 0, 0, 0, 0, 0 };
      for (int k = 0; k < 5; k++) for (int a = 0; a < ecm::NSLIP; a++) dphat[k] += mdl.P[k][a] * sv[ecm::iHistLbGdot + a];
      double C[3][3], Q5[5][5];
      ecm::quat_to_tensor(&sv[ecm::iHistLbQ], C); ecm::rot_mat_vecd(C, Q5);
      double sm[5]; for (int k = 0; k < 5; k++) { sm[k] = 0; for (int l = 0; l < 5; l++) sm[k] += Q5[k][l] * dphat[l]; }
      double T[3][3]; ecm::vecd_to_tensor(sm, T);
      for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) dp[i + 3 * (j + 3 * ip)] = T[i][j];
   }
}

}  // namespace fem
