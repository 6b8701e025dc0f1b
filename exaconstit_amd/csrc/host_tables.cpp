// Host-side set-up: reference-element tables and material-parameter unpacking.
//   reference element table: what the reference obtains from MFEM's H1 hexahedron + IntRules.Get(CUBE, 2p+1)
//     (src/mechanics_operator.cpp:237-261, src/mechanics_integrators.cpp:184-197); native node order = vertices,
//     edge interiors, face interiors, volume interiors; quadrature points x-fastest.
//   material parameters: ECMechXtalModel ctor -> initFromParams (src/mechanics_ecmech.hpp:219-246), parameter order
//     src/mechanics_ecmech.hpp:395-405 (Voce) / :444-458 (KM-DD), scripts/ecmech_prop_file.py:58-122.
#include "exa_internal.hpp"
#include <cmath>

namespace {

void gl_rule(int np, std::vector<double>& x, std::vector<double>& w) {
   x.resize(np); w.resize(np);
   for (int i = 0; i < np; i++) {   // Newton on P_np
      double z = std::cos(M_PI * (i + 0.75) / (np + 0.5)), dp = 1.0;
      for (int it = 0; it < 200; it++) {
         double p0 = 1.0, p1 = z;
         for (int k = 2; k <= np; k++) { const double p2 = ((2.0 * k - 1.0) * z * p1 - (k - 1.0) * p0) / k; p0 = p1; p1 = p2; }
         if (np == 1) { p1 = z; p0 = 1.0; }
         dp = np * (z * p1 - p0) / (z * z - 1.0);
         const double dz = p1 / dp; z -= dz;
         if (std::fabs(dz) < 1e-15) break;
      }
      double p0 = 1.0, p1 = z;
      for (int k = 2; k <= np; k++) { const double p2 = ((2.0 * k - 1.0) * z * p1 - (k - 1.0) * p0) / k; p0 = p1; p1 = p2; }
      dp = np * (z * p1 - p0) / (z * z - 1.0);
      x[np - 1 - i] = 0.5 * (1.0 + z); w[np - 1 - i] = 1.0 / ((1.0 - z * z) * dp * dp);
   }
}

// Gauss-Lobatto-Legendre points on [0, 1]: the end points and the roots of P'_{np-1} (MFEM's H1 default, BasisType::GaussLobatto);
// Newton from the Chebyshev-Lobatto points, (1 - z^2) P'_n = n (P_{n-1} - z P_n), (1 - z^2) P''_n = 2 z P'_n - n (n + 1) P_n.
void gll_nodes(int np, std::vector<double>& x) {
   x.assign(np, 0.0);
   const int n = np - 1;
   x[n] = 1.0;
   for (int i = 1; 2 * i <= n; i++) {
      double z = -std::cos(M_PI * i / n);
      for (int it = 0; it < 100; it++) {
         double p0 = 1.0, p1 = z;
         for (int k = 2; k <= n; k++) { const double p2 = ((2.0 * k - 1.0) * z * p1 - (k - 1.0) * p0) / k; p0 = p1; p1 = p2; }
         const double om = 1.0 - z * z, d1 = n * (p0 - z * p1) / om, d2 = (2.0 * z * d1 - n * (n + 1.0) * p1) / om;
         const double dz = d1 / d2; z -= dz;
         if (std::fabs(dz) < 1e-16) break;
      }
      if (2 * i == n) z = 0.0;
      x[i] = 0.5 * (1.0 + z); x[n - i] = 0.5 * (1.0 - z);
   }
}
void lagrange(const std::vector<double>& xn, double x, std::vector<double>& v, std::vector<double>& d) {
   const int np = (int)xn.size();
   v.assign(np, 0.0); d.assign(np, 0.0);
   for (int a = 0; a < np; a++) {
      double den = 1.0, num = 1.0, ds = 0.0;
      for (int b = 0; b < np; b++) if (b != a) { den *= xn[a] - xn[b]; num *= x - xn[b]; }
      for (int c = 0; c < np; c++) if (c != a) { double t = 1.0; for (int b = 0; b < np; b++) if (b != a && b != c) t *= x - xn[b]; ds += t; }
      v[a] = num / den; d[a] = ds / den;
   }
}

// lexicographic (i,j,k) -> native index of the order-p hexahedron
std::vector<int> native_order(int p) {
   const int np = p + 1;
   std::vector<int> m(np * np * np, -1);
   auto L = [&](int i, int j, int k) { return i + np * (j + np * k); };
   int c = 0;
   const int V[8][3] = { { 0, 0, 0 }, { 1, 0, 0 }, { 1, 1, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 1, 0, 1 }, { 1, 1, 1 }, { 0, 1, 1 } };
   for (auto& v : V) m[L(v[0] * p, v[1] * p, v[2] * p)] = c++;
   const int Ed[12][2] = { { 0, 1 }, { 1, 2 }, { 3, 2 }, { 0, 3 }, { 4, 5 }, { 5, 6 }, { 7, 6 }, { 4, 7 }, { 0, 4 }, { 1, 5 }, { 2, 6 }, { 3, 7 } };
   for (auto& ed : Ed) for (int t = 1; t < p; t++) {
      int ijk[3]; for (int d = 0; d < 3; d++) ijk[d] = V[ed[0]][d] * p + (V[ed[1]][d] - V[ed[0]][d]) * t;
      m[L(ijk[0], ijk[1], ijk[2])] = c++;
   }
   for (int f = 0; f < 6; f++) for (int t2 = 1; t2 < p; t2++) for (int t1 = 1; t1 < p; t1++) {
      int i = t1, j = t2, k = 0;
      if (f == 1) { i = t1; j = 0; k = t2; } else if (f == 2) { i = p; j = t1; k = t2; } else if (f == 3) { i = t1; j = p; k = t2; }
      else if (f == 4) { i = 0; j = t1; k = t2; } else if (f == 5) { i = t1; j = t2; k = p; }
      m[L(i, j, k)] = c++;
   }
   for (int k = 1; k < p; k++) for (int j = 1; j < p; j++) for (int i = 1; i < p; i++) m[L(i, j, k)] = c++;
   return m;
}

}  // namespace

void exa_gll_nodes_01(int np, std::vector<double>& x) { gll_nodes(np, x); }

void exa_build_ref_elem(int p, std::vector<double>& G, std::vector<double>& W) {
   const int np = p + 1, n = np * np * np, Q = n;
   std::vector<double> xq, wq, xn;
   gl_rule(np, xq, wq); gll_nodes(np, xn);
   const std::vector<int> nat = native_order(p);
   G.assign((size_t)n * 3 * Q, 0.0); W.assign(Q, 0.0);
   std::vector<double> vx, dx, vy, dy, vz, dz;
   for (int qk = 0; qk < np; qk++) for (int qj = 0; qj < np; qj++) for (int qi = 0; qi < np; qi++) {
      const int q = qi + np * (qj + np * qk);
      W[q] = wq[qi] * wq[qj] * wq[qk];
      lagrange(xn, xq[qi], vx, dx); lagrange(xn, xq[qj], vy, dy); lagrange(xn, xq[qk], vz, dz);
      for (int k = 0; k < np; k++) for (int j = 0; j < np; j++) for (int i = 0; i < np; i++) {
         const int a = nat[i + np * (j + np * k)];
         G[a + n * (0 + 3 * q)] = dx[i] * vy[j] * vz[k];
         G[a + n * (1 + 3 * q)] = vx[i] * dy[j] * vz[k];
         G[a + n * (2 + 3 * q)] = vx[i] * vy[j] * dz[k];
      }
   }
}

void exa_build_1d_tables(int p, std::vector<double>& T1, std::vector<int>& nat) {
   const int np = p + 1;
   std::vector<double> xq, wq, xn, v, d;
   gl_rule(np, xq, wq); gll_nodes(np, xn);
   T1.assign((size_t)np * 2 * np, 0.0);
   for (int q = 0; q < np; q++) {
      lagrange(xn, xq[q], v, d);
      for (int i = 0; i < np; i++) { T1[2 * np * q + i] = v[i]; T1[2 * np * q + np + i] = d[i]; }
   }
   nat = native_order(p);
}

bool exa_fill_mat_params(const exa_config& cfg, ecmdev::MatParams& mp, double* hist_init, std::string& err) {
   using namespace ecmdev;
   mp = MatParams{};
   bool bcc = false;
   switch (cfg.model) {
      case EXA_FCC_VOCE: mp.kin = KIN_VOCE; break;
      case EXA_FCC_VOCE_NL: mp.kin = KIN_VOCE_NL; break;
      case EXA_BCC_VOCE: mp.kin = KIN_VOCE; bcc = true; break;
      case EXA_BCC_VOCE_NL: mp.kin = KIN_VOCE_NL; bcc = true; break;
      case EXA_FCC_KMDD: mp.kin = KIN_KMBALD; break;
      case EXA_BCC_KMDD: mp.kin = KIN_KMBALD; bcc = true; break;
      default: err = "unknown model id"; return false;
   }
   const int need = mp.kin == KIN_VOCE ? 17 : (mp.kin == KIN_VOCE_NL ? 18 : 24);
   if (cfg.nprops != need || cfg.props == nullptr) { err = "Properties did not contain the expected number of parameters for this model"; return false; }
   mp.qsign = bcc ? -1.0 : 1.0;
   mp.with_g_athermal = bcc ? 1 : 0;
   const double* p = cfg.props;
   int i = 0;
   const double rho0 = p[i++]; (void)rho0;
   const double cvav = p[i++]; mp.tol = p[i++];
   const double c11 = p[i++], c12 = p[i++], c44 = p[i++];
   mp.kd0 = c11 - c12; mp.kd2 = 2.0 * c44; mp.ikd0 = 1.0 / mp.kd0; mp.ikd2 = 1.0 / mp.kd2;
   mp.pk0 = ecmdev::PSC[0] * mp.kd0; mp.pk1 = ecmdev::PSC[1] * mp.kd0; mp.pk2 = ecmdev::PSC[2] * mp.kd2;   // PSC[2] == PSC[3] == PSC[4]
   mp.bulk = (c11 + 2.0 * c12) / 3.0; mp.gmod = (2.0 * mp.kd0 + 3.0 * mp.kd2) / 10.0;
   double hdn_init, xm;
   if (mp.kin != KIN_KMBALD) {
      const double mu = p[i++]; (void)mu;
      xm = p[i++]; mp.gam_w = p[i++]; mp.h0 = p[i++]; mp.tausi = p[i++]; mp.taus0 = p[i++];
      mp.xmprime = 1.0; if (mp.kin == KIN_VOCE_NL) mp.xmprime = p[i++];
      mp.xms = p[i++]; mp.gamss0 = p[i++]; hdn_init = p[i++];
   } else {
      mp.mu_ref = p[i++]; const double tK_ref = p[i++]; mp.c_1 = p[i++]; mp.tau_a = p[i++]; mp.p = p[i++]; mp.q = p[i++];
      mp.gam_wo = p[i++]; mp.gam_ro = p[i++]; mp.wrD = p[i++]; mp.go = p[i++]; mp.s = p[i++];
      mp.k1 = p[i++]; mp.k2o = p[i++]; mp.ninv = p[i++]; mp.gamma_o = p[i++]; hdn_init = p[i++];
      mp.hdn_min = 1.0e-4 * hdn_init;
      xm = 1.0 / (2.0 * ((mp.c_1 / tK_ref) * mp.mu_ref * mp.p * mp.q));
   }
   mp.xnn = 1.0 / xm; mp.xn = mp.xnn - 1.0;
   mp.xn_int = (mp.xn == std::floor(mp.xn) && mp.xn >= 1.0 && mp.xn <= 255.0) ? (int)mp.xn : 0;
   mp.t_min = std::pow(1.0e-60, xm); mp.t_max = std::pow(1.0e45, xm);
   mp.gamma = p[i++]; const double ecold = p[i++];
   mp.dtde = 1.0 / cvav; mp.tK0 = -ecold * mp.dtde;
   for (int k = 0; k < NUM_HIST; k++) hist_init[k] = 0.0;
   hist_init[H_Q] = 1.0; hist_init[H_H] = hdn_init;
   return true;
}
