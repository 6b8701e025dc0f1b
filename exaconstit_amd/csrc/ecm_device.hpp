// Device-side crystal-plasticity point update for gfx950 (MI355X).
//
// Replaces, behind the reference's ExaModel seam, what ExaConstit obtains from
//   ecmech::matModelBase::getResponseECM        (call site reference src/mechanics_ecmech.cpp:176-186)
// for the models of reference src/mechanics_ecmech.hpp:407-414,460-463.  One thread owns one quadrature point; all
// small tensors live in registers with compile-time indexing only (no scratch), the 8x8 point system is solved in
// its 5+3 block form (the rotation block is eliminated analytically, the remaining 5x5 is an un-pivoted LU of a
// column-scaled SPD matrix), and the slip-system tables are compile-time constants so the compiler folds their
// zeros.  BCC {110}<111> and FCC {111}<110> share the symmetric Schmid tensors; only the sign of the plastic spin
// differs, so one table serves both.
//
// This is an independent implementation of the same published algorithm that oracle/ecmech_port.hpp restates on
// the CPU; tests compare the two on identical inputs.  Nothing here includes or links the oracle.
#pragma once
#include <hip/hip_runtime.h>

namespace ecmdev {

#define ECM_DI __device__ __forceinline__

constexpr int NSLIP = 12;
constexpr double SQR2 = 1.4142135623730951, SQR3 = 1.7320508075688772;
constexpr double SQR2I = 0.70710678118654752, SQR6I = 0.40824829046386302, SQR2B3 = 0.81649658092772603;
constexpr double TINY_SQRT = 1.0e-90, EPS_SQRT = 1.0e-8;
constexpr double GAM_RATIO_OVF = 1.0e45, LN_GAM_RATIO_MIN = -138.15510557964274;
constexpr double E_SCALE = 5.0e-4, R_SCALE = 0.01;

// history layout (reference src/mechanics_ecmech.hpp:165-185)
constexpr int H_SHRATE = 0, H_SHR = 1, H_FLOW = 2, H_NFEV = 3, H_E = 4, H_Q = 9, H_H = 13, H_GDOT = 14;
constexpr int NUM_HIST = 26, NSTATEV = 28, IND_VOL = 26, IND_EINT = 27;

enum { KIN_VOCE = 0, KIN_VOCE_NL = 1, KIN_KMBALD = 2 };

// Schmid tensors of the 12 FCC {111}<110> systems: P = vecd(sym(s x m)), Q = axial(skew(s x m)).
// a = sqrt(3)/6, b = sqrt(6)/12.
constexpr double PA = 0.28867513459481288, PB = 0.20412414523193151;
__device__ constexpr double P_TAB[5][NSLIP] = {
   { -PA, -PA, 2 * PA, PA, PA, -2 * PA, -PA, -PA, 2 * PA, PA, PA, -2 * PA },
   { -0.5, 0.5, 0.0, -0.5, 0.5, 0.0, -0.5, 0.5, 0.0, -0.5, 0.5, 0.0 },
   { PA, -PA, 0.0, -PA, PA, 0.0, PA, -PA, 0.0, -PA, PA, 0.0 },
   { -PA, 0.0, PA, 0.0, -PA, PA, PA, 0.0, -PA, 0.0, PA, -PA },
   { 0.0, PA, -PA, -PA, 0.0, PA, 0.0, -PA, PA, PA, 0.0, -PA } };
__device__ constexpr double Q_TAB[3][NSLIP] = {
   { -2 * PB, PB, PB, -PB, 2 * PB, -PB, 2 * PB, -PB, -PB, PB, -2 * PB, PB },
   { PB, -2 * PB, PB, -2 * PB, PB, PB, -PB, 2 * PB, -PB, 2 * PB, -PB, -PB },
   { PB, PB, -2 * PB, PB, PB, -2 * PB, PB, PB, -2 * PB, PB, PB, -2 * PB } };

// Material description, passed by value as a kernel argument (lives in SGPRs / kernarg segment).
struct MatParams {
   int kin;                 // KIN_*
   int with_g_athermal;     // KMBalD: 1 for BCC ("Kin_BCC_A"), 0 for FCC ("Kin_FCC_B")
   double qsign;            // +1 FCC, -1 BCC: sign of the plastic-spin vectors
   double kd0, kd2;         // Kirchhoff' = diag(kd0,kd0,kd2,kd2,kd2) e'
   double bulk, gmod, gamma, tK0, dtde, tol;
   // Voce power law
   double xnn, xn, gam_w, h0, tausi, taus0, xmprime, xms, gamss0, t_min, t_max;
   // Kocks-Mecking balanced dislocation density
   double mu_ref, c_1, tau_a, p, q, gam_wo, gam_ro, wrD, go, s, k1, k2o, ninv, gamma_o, hdn_min;
};

// ------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------
ECM_DI void vecd_to_sym(const double v[5], double& t00, double& t11, double& t22, double& t01, double& t02, double& t12) {
   const double t1 = SQR2I * v[0], t2 = SQR6I * v[1];
   t00 = t1 - t2; t11 = -t1 - t2; t22 = SQR2B3 * v[1]; t01 = SQR2I * v[2]; t02 = SQR2I * v[3]; t12 = SQR2I * v[4];
}

ECM_DI void sym_to_vecd(double t00, double t11, double t22, double t01, double t02, double t12, double v[5]) {
   v[0] = SQR2I * (t00 - t11); v[1] = SQR6I * (2.0 * t22 - t00 - t11);
   v[2] = SQR2 * t01; v[3] = SQR2 * t02; v[4] = SQR2 * t12;
}

// vecd(R^T S R) for symmetric deviatoric S given as vecd and a 3x3 R (row-major r[3*i+j])
ECM_DI void rot_vecd_T(const double R[9], const double s[5], double out[5]) {
   double s00, s11, s22, s01, s02, s12; vecd_to_sym(s, s00, s11, s22, s01, s02, s12);
   const double S[3][3] = { { s00, s01, s02 }, { s01, s11, s12 }, { s02, s12, s22 } };
   double T[3][3];
#pragma unroll
   for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) T[i][j] = S[i][0] * R[j] + S[i][1] * R[3 + j] + S[i][2] * R[6 + j];
   auto U = [&](int i, int j) { return R[i] * T[0][j] + R[3 + i] * T[1][j] + R[6 + i] * T[2][j]; };
   sym_to_vecd(U(0, 0), U(1, 1), U(2, 2), U(0, 1), U(0, 2), U(1, 2), out);
}

// vecd(R S R^T)
ECM_DI void rot_vecd(const double R[9], const double s[5], double out[5]) {
   const double Rt[9] = { R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8] };
   rot_vecd_T(Rt, s, out);
}

ECM_DI void quat_to_mat(const double q[4], double C[9]) {
   const double x0 = q[0], x1 = q[1], x2 = q[2], x3 = q[3];
   C[0] = x0 * x0 + x1 * x1 - x2 * x2 - x3 * x3; C[1] = 2.0 * (x1 * x2 - x0 * x3); C[2] = 2.0 * (x1 * x3 + x0 * x2);
   C[3] = 2.0 * (x1 * x2 + x0 * x3); C[4] = x0 * x0 - x1 * x1 + x2 * x2 - x3 * x3; C[5] = 2.0 * (x2 * x3 - x0 * x1);
   C[6] = 2.0 * (x1 * x3 - x0 * x2); C[7] = 2.0 * (x2 * x3 + x0 * x1); C[8] = x0 * x0 - x1 * x1 - x2 * x2 + x3 * x3;
}

// M35(d): axial w -> vecd(D W - W D);  M[k][j], closed form
ECM_DI void m35(const double d[5], double M[5][3]) {
   M[0][0] = -d[4]; M[1][0] = -SQR3 * d[4]; M[2][0] = d[3]; M[3][0] = -d[2]; M[4][0] = d[0] + SQR3 * d[1];
   M[0][1] = -d[3]; M[1][1] = SQR3 * d[3]; M[2][1] = -d[4]; M[3][1] = d[0] - SQR3 * d[1]; M[4][1] = d[2];
   M[0][2] = 2.0 * d[2]; M[1][2] = 0.0; M[2][2] = -2.0 * d[0]; M[3][2] = d[4]; M[4][2] = -d[3];
}

// exponential map pieces for a rotation vector xi:  A = exp(hat xi) (row-major), Tr = right Jacobian of exp
ECM_DI void exp_map(const double xi[3], double A[9], double Tr[9]) {
   const double th2 = xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2];
   double a, b, c;   // sin(th)/th, (1-cos th)/th^2, (th - sin th)/th^3
   if (th2 < 1.0e-4) {
      a = 1.0 - th2 * (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0)));
      b = 0.5 * (1.0 - th2 * (1.0 / 12.0) * (1.0 - th2 * (1.0 / 30.0) * (1.0 - th2 * (1.0 / 56.0))));
      c = (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0))));
   } else {
      const double th = sqrt(th2); double sn, cs; sincos(th, &sn, &cs);
      a = sn / th; b = (1.0 - cs) / th2; c = (th - sn) / (th2 * th);
   }
   const double x = xi[0], y = xi[1], z = xi[2];
   // hat(xi) = [[0,-z,y],[z,0,-x],[-y,x,0]],  hat^2 = xi xi^T - th2 I
   A[0] = 1.0 + b * (x * x - th2); A[1] = -a * z + b * x * y;        A[2] = a * y + b * x * z;
   A[3] = a * z + b * x * y;        A[4] = 1.0 + b * (y * y - th2); A[5] = -a * x + b * y * z;
   A[6] = -a * y + b * x * z;       A[7] = a * x + b * y * z;        A[8] = 1.0 + b * (z * z - th2);
   Tr[0] = 1.0 + c * (x * x - th2); Tr[1] = b * z + c * x * y;        Tr[2] = -b * y + c * x * z;
   Tr[3] = -b * z + c * x * y;       Tr[4] = 1.0 + c * (y * y - th2); Tr[5] = b * x + c * y * z;
   Tr[6] = b * y + c * x * z;        Tr[7] = -b * x + c * y * z;       Tr[8] = 1.0 + c * (z * z - th2);
}

// ------------------------------------------------------------------------------------------------------------
// slip kinetics
// ------------------------------------------------------------------------------------------------------------
struct KinVals { double g, gam_w, gam_r, c_e; };

ECM_DI void voce_gdot(const MatParams& mp, double g_i, double tau, double& gdot, double& dg) {
   gdot = 0.0; dg = 0.0;
   const double t_frac = tau * g_i, at = fabs(t_frac);
   if (at > mp.t_min) {
      if (at > mp.t_max) {
         gdot = copysign(mp.gam_w * GAM_RATIO_OVF, t_frac);
         dg = fabs(gdot) * mp.xnn / fabs(tau);
      } else {
         const double temp = mp.gam_w * exp(mp.xn * log(at));
         gdot = temp * t_frac; dg = temp * mp.xnn * g_i;
      }
   }
}

ECM_DI void mts_dG(const MatParams& mp, double c_e, double t_frac, double& exp_arg, double& dfac) {
   exp_arg = 0.0; dfac = 0.0;
   if (t_frac >= 1.0) return;
   double p_func, dp_func;
   const double at = fabs(t_frac);
   if (at < TINY_SQRT) { p_func = 0.0; dp_func = (mp.p == 1.0) ? 1.0 : 0.0; }
   else if (mp.p == 1.0) { p_func = t_frac; dp_func = 1.0; }
   else { const double pw = pow(at, mp.p); p_func = copysign(pw, t_frac); dp_func = mp.p * pw / at; }
   const double q_arg = 1.0 - p_func;
   if (q_arg <= TINY_SQRT) return;
   double q_func, dq_func;
   if (mp.q == 1.0) { q_func = q_arg; dq_func = 1.0; }
   else { q_func = pow(q_arg, mp.q); dq_func = mp.q * q_func / q_arg; }
   exp_arg = -c_e * q_func; dfac = c_e * dq_func * dp_func;
}

ECM_DI void kmbald_gdot(const MatParams& mp, const KinVals& kv, double tau, double& gdot, double& dg) {
   gdot = 0.0; dg = 0.0;
   if (tau == 0.0) return;
   const double g_i = mp.with_g_athermal ? 1.0 / mp.tau_a : 1.0 / kv.g;
   const double gAth = mp.with_g_athermal ? kv.g : mp.tau_a;
   const double at = fabs(tau);
   const double at_0 = fmax(0.0, at - gAth) * g_i;
   const double exp_arg_r = (at - gAth) / mp.wrD;
   if (exp_arg_r <= 0.0) return;
   double gdot_r, dgdot_r;
   if (exp_arg_r < EPS_SQRT) { gdot_r = kv.gam_r * exp_arg_r; dgdot_r = kv.gam_r / mp.wrD; }
   else { const double ex = exp(-exp_arg_r); gdot_r = kv.gam_r * (1.0 - ex); dgdot_r = kv.gam_r * ex / mp.wrD; }
   if (at_0 > mp.t_max) { gdot = copysign(gdot_r, tau); dg = dgdot_r; return; }
   double ea_f, df_f, ea_b, df_b;
   mts_dG(mp, kv.c_e, (at - gAth) * g_i, ea_f, df_f);
   if (ea_f < LN_GAM_RATIO_MIN) return;
   mts_dG(mp, kv.c_e, (-at - gAth) * g_i, ea_b, df_b);
   const double ef = exp(ea_f), eb = exp(ea_b);
   double gdot_w = kv.gam_w * (ef - eb);
   double dgdot_w = kv.gam_w * (ef * df_f + eb * df_b) * g_i;
   if (at_0 > mp.t_min) {
      const double temp = (kv.gam_w * 10.0) * exp(mp.xn * log(at_0));
      gdot_w += temp * at_0; dgdot_w += temp * mp.xnn * g_i;
   }
   if (gdot_w <= 0.0) return;
   const double gd = 1.0 / (1.0 / gdot_w + 1.0 / gdot_r);
   dg = gd * gd * (dgdot_w / (gdot_w * gdot_w) + dgdot_r / (gdot_r * gdot_r));
   gdot = copysign(gd, tau);
}

template <int KIN>
ECM_DI void kin_sdot(const MatParams& mp, double h, double shrate, double ev1, double& sdot, double& dsdot) {
   if (KIN == KIN_KMBALD) {
      const double t1 = exp(-0.5 * h);
      sdot = (mp.k1 * t1 - ev1) * shrate; dsdot = (-0.5 * mp.k1 * t1) * shrate;
   } else if (KIN == KIN_VOCE_NL) {
      const double r = (ev1 - h) / (ev1 - mp.tausi);
      const double t1 = (mp.xmprime == 1.0) ? 1.0 : pow(fmax(r, 0.0), mp.xmprime - 1.0);
      sdot = mp.h0 * t1 * r * shrate; dsdot = -mp.h0 * mp.xmprime * t1 / (ev1 - mp.tausi) * shrate;
   } else {
      const double t1 = mp.h0 / (ev1 - mp.tausi);
      sdot = t1 * (ev1 - h) * shrate; dsdot = -t1 * shrate;
   }
}

// backward-Euler hardness update with the begin-of-step slip rates
template <int KIN>
ECM_DI double kin_update_h(const MatParams& mp, double hs_o, double dt, double shrate) {
   double ev1, h_o;
   if (KIN == KIN_KMBALD) {
      ev1 = mp.k2o;
      if (shrate > TINY_SQRT) ev1 = mp.k2o * pow(mp.gamma_o / shrate, mp.ninv);
      h_o = log(fmax(hs_o, mp.hdn_min));
   } else {
      ev1 = mp.taus0;
      if (shrate > TINY_SQRT && mp.xms != 0.0) ev1 = mp.taus0 * pow(shrate / mp.gamss0, mp.xms);
      h_o = hs_o;
   }
   const double x_scale = fmax(fabs(h_o), 1.0), res_scale = 1.0 / x_scale;
   double x = 0.0;
   for (int it = 0; it < 100; it++) {
      double sdot, dsdot; kin_sdot<KIN>(mp, h_o + x * x_scale, shrate, ev1, sdot, dsdot);
      const double r = (x * x_scale - sdot * dt) * res_scale;
      if (fabs(r) < 1.0e-10) break;
      x -= r / ((1.0 - dsdot * dt) * res_scale * x_scale);
   }
   const double h_n = h_o + x * x_scale;
   return (KIN == KIN_KMBALD) ? exp(h_n) : h_n;
}

// ------------------------------------------------------------------------------------------------------------
// the point problem: unknowns x = (delta e' / E_SCALE, xi / R_SCALE)
// ------------------------------------------------------------------------------------------------------------
struct Prob {
   double dt_ri, detV_ri, sc;   // sc = epsdot_scale_inv
   double e_n[5], d_n[5], w_n[3];   // begin-of-step strain; D' and spin vector pulled back with C_n
   KinVals kv;
};

// Jacobian blocks (un-scaled):  Jee = I/dt + A Kd,  Jer = -Mer,  Jre = B Kd,  Jrr = I/dt - Wt
struct Jac { double A[15], B[3][5], Mer[5][3], Wt[3][3]; };

ECM_DI constexpr int sidx(int i, int j) { return i <= j ? (i * (11 - i)) / 2 + (j - i) : (j * (11 - j)) / 2 + (i - j); }   // 5x5 symmetric packing

template <int KIN, bool WITHJ>
ECM_DI bool eval_rj(const MatParams& mp, const Prob& pb, const double x[8], double r[8], Jac& jac,
                    double gdot[NSLIP], double& dis_rate) {
   double e_f[5], xi[3];
#pragma unroll
   for (int i = 0; i < 5; i++) e_f[i] = pb.e_n[i] + x[i] * E_SCALE;
#pragma unroll
   for (int i = 0; i < 3; i++) xi[i] = x[5 + i] * R_SCALE;
   double A[9], Tr[9]; exp_map(xi, A, Tr);
   double d_lat[5]; rot_vecd_T(A, pb.d_n, d_lat);
   double w_lat[3];
#pragma unroll
   for (int i = 0; i < 3; i++) w_lat[i] = A[i] * pb.w_n[0] + A[3 + i] * pb.w_n[1] + A[6 + i] * pb.w_n[2];
   // resolved shear stress from the Kirchhoff stress
   const double k[5] = { mp.kd0 * e_f[0], mp.kd0 * e_f[1], mp.kd2 * e_f[2], mp.kd2 * e_f[3], mp.kd2 * e_f[4] };
   double dgdt[NSLIP];
   const double g_i = 1.0 / pb.kv.g;
   dis_rate = 0.0;
   bool ok = true;
#pragma unroll
   for (int a = 0; a < NSLIP; a++) {
      double tau = 0.0;
#pragma unroll
      for (int c = 0; c < 5; c++) if (P_TAB[c][a] != 0.0) tau += P_TAB[c][a] * k[c];
      if (KIN == KIN_KMBALD) kmbald_gdot(mp, pb.kv, tau, gdot[a], dgdt[a]);
      else voce_gdot(mp, g_i, tau, gdot[a], dgdt[a]);
      dis_rate += tau * gdot[a];
      ok = ok && isfinite(gdot[a]);
   }
   dis_rate *= pb.detV_ri;
   double dp[5] = { 0, 0, 0, 0, 0 }, wp[3] = { 0, 0, 0 };
#pragma unroll
   for (int a = 0; a < NSLIP; a++) {
#pragma unroll
      for (int c = 0; c < 5; c++) if (P_TAB[c][a] != 0.0) dp[c] += P_TAB[c][a] * gdot[a];
#pragma unroll
      for (int c = 0; c < 3; c++) wp[c] += Q_TAB[c][a] * gdot[a];
   }
#pragma unroll
   for (int c = 0; c < 5; c++) r[c] = (x[c] * (E_SCALE * pb.dt_ri) + dp[c] - d_lat[c]) * pb.sc;
#pragma unroll
   for (int c = 0; c < 3; c++) r[5 + c] = (xi[c] * pb.dt_ri + mp.qsign * wp[c] - w_lat[c]) * pb.sc;
   if (WITHJ) {
#pragma unroll
      for (int i = 0; i < 15; i++) jac.A[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
         for (int j = 0; j < 5; j++) jac.B[i][j] = 0.0;
#pragma unroll
      for (int a = 0; a < NSLIP; a++) {
         double gp[5];
#pragma unroll
         for (int c = 0; c < 5; c++) gp[c] = dgdt[a] * P_TAB[c][a];
#pragma unroll
         for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = i; j < 5; j++) if (P_TAB[i][a] != 0.0 && P_TAB[j][a] != 0.0) jac.A[sidx(i, j)] += P_TAB[i][a] * gp[j];
#pragma unroll
         for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 5; j++) if (P_TAB[j][a] != 0.0) jac.B[i][j] += (mp.qsign * Q_TAB[i][a]) * gp[j];
      }
      double M[5][3]; m35(d_lat, M);
#pragma unroll
      for (int c = 0; c < 5; c++)
#pragma unroll
         for (int j = 0; j < 3; j++) jac.Mer[c][j] = M[c][0] * Tr[j] + M[c][1] * Tr[3 + j] + M[c][2] * Tr[6 + j];
      // hat(w_lat) Tr
      const double W[3][3] = { { 0.0, -w_lat[2], w_lat[1] }, { w_lat[2], 0.0, -w_lat[0] }, { -w_lat[1], w_lat[0], 0.0 } };
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
         for (int j = 0; j < 3; j++) jac.Wt[i][j] = W[i][0] * Tr[j] + W[i][1] * Tr[3 + j] + W[i][2] * Tr[6 + j];
   }
   return ok;
}

// y = J v  (un-scaled blocks)
ECM_DI void jac_mult(const MatParams& mp, const Prob& pb, const Jac& J, const double v[8], double y[8]) {
   const double kv[5] = { mp.kd0 * v[0], mp.kd0 * v[1], mp.kd2 * v[2], mp.kd2 * v[3], mp.kd2 * v[4] };
#pragma unroll
   for (int i = 0; i < 5; i++) {
      double s = v[i] * pb.dt_ri;
#pragma unroll
      for (int j = 0; j < 5; j++) s += J.A[sidx(i, j)] * kv[j];
      s -= J.Mer[i][0] * v[5] + J.Mer[i][1] * v[6] + J.Mer[i][2] * v[7];
      y[i] = s;
   }
#pragma unroll
   for (int i = 0; i < 3; i++) {
      double s = v[5 + i] * pb.dt_ri;
#pragma unroll
      for (int j = 0; j < 5; j++) s += J.B[i][j] * kv[j];
      s -= J.Wt[i][0] * v[5] + J.Wt[i][1] * v[6] + J.Wt[i][2] * v[7];
      y[5 + i] = s;
   }
}

// y = J^T u
ECM_DI void jac_mult_T(const MatParams& mp, const Prob& pb, const Jac& J, const double u[8], double y[8]) {
   const double kd[5] = { mp.kd0, mp.kd0, mp.kd2, mp.kd2, mp.kd2 };
#pragma unroll
   for (int j = 0; j < 5; j++) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 5; i++) s += J.A[sidx(i, j)] * u[i];
      s += J.B[0][j] * u[5] + J.B[1][j] * u[6] + J.B[2][j] * u[7];
      y[j] = u[j] * pb.dt_ri + kd[j] * s;
   }
#pragma unroll
   for (int j = 0; j < 3; j++) {
      double s = u[5 + j] * pb.dt_ri;
#pragma unroll
      for (int i = 0; i < 5; i++) s -= J.Mer[i][j] * u[i];
      s -= J.Wt[0][j] * u[5] + J.Wt[1][j] * u[6] + J.Wt[2][j] * u[7];
      y[5 + j] = s;
   }
}

// Factorisation of J in its block form.  Ri = Jrr^-1, Y = Ri Jre, LU = un-pivoted LU of S = Jee - Jer Y.
struct Fact { double Ri[3][3], Y[3][5], LU[5][5]; bool ok; };

ECM_DI void jac_factor(const MatParams& mp, const Prob& pb, const Jac& J, Fact& F) {
   const double kd[5] = { mp.kd0, mp.kd0, mp.kd2, mp.kd2, mp.kd2 };
   double R[3][3];
#pragma unroll
   for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) R[i][j] = (i == j ? pb.dt_ri : 0.0) - J.Wt[i][j];
   const double c00 = R[1][1] * R[2][2] - R[1][2] * R[2][1], c01 = R[1][2] * R[2][0] - R[1][0] * R[2][2], c02 = R[1][0] * R[2][1] - R[1][1] * R[2][0];
   const double det = R[0][0] * c00 + R[0][1] * c01 + R[0][2] * c02;
   const double di = 1.0 / det;
   F.Ri[0][0] = c00 * di; F.Ri[1][0] = c01 * di; F.Ri[2][0] = c02 * di;
   F.Ri[0][1] = (R[0][2] * R[2][1] - R[0][1] * R[2][2]) * di; F.Ri[1][1] = (R[0][0] * R[2][2] - R[0][2] * R[2][0]) * di; F.Ri[2][1] = (R[0][1] * R[2][0] - R[0][0] * R[2][1]) * di;
   F.Ri[0][2] = (R[0][1] * R[1][2] - R[0][2] * R[1][1]) * di; F.Ri[1][2] = (R[0][2] * R[1][0] - R[0][0] * R[1][2]) * di; F.Ri[2][2] = (R[0][0] * R[1][1] - R[0][1] * R[1][0]) * di;
#pragma unroll
   for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 5; j++) F.Y[i][j] = (F.Ri[i][0] * J.B[0][j] + F.Ri[i][1] * J.B[1][j] + F.Ri[i][2] * J.B[2][j]) * kd[j];
#pragma unroll
   for (int i = 0; i < 5; i++)
#pragma unroll
      for (int j = 0; j < 5; j++)
         F.LU[i][j] = (i == j ? pb.dt_ri : 0.0) + J.A[sidx(i, j)] * kd[j] + J.Mer[i][0] * F.Y[0][j] + J.Mer[i][1] * F.Y[1][j] + J.Mer[i][2] * F.Y[2][j];
   bool ok = isfinite(di);
#pragma unroll
   for (int k = 0; k < 5; k++) {
      const double piv = F.LU[k][k];
      ok = ok && (piv > 0.0);
      const double inv = 1.0 / piv;
      F.LU[k][k] = inv;   // store the reciprocal pivot
#pragma unroll
      for (int i = k + 1; i < 5; i++) {
         const double f = F.LU[i][k] * inv; F.LU[i][k] = f;
#pragma unroll
         for (int j = k + 1; j < 5; j++) F.LU[i][j] -= f * F.LU[k][j];
      }
   }
   F.ok = ok;
}

// solve J dx = rhs  (rhs_r may be identically zero: pass ZERO_R = true)
template <bool ZERO_R>
ECM_DI void jac_solve(const Jac& J, const Fact& F, const double rhs[8], double dx[8]) {
   double t[3] = { 0, 0, 0 }, b[5];
   if (!ZERO_R) {
#pragma unroll
      for (int i = 0; i < 3; i++) t[i] = F.Ri[i][0] * rhs[5] + F.Ri[i][1] * rhs[6] + F.Ri[i][2] * rhs[7];
   }
#pragma unroll
   for (int i = 0; i < 5; i++) b[i] = rhs[i] + (ZERO_R ? 0.0 : (J.Mer[i][0] * t[0] + J.Mer[i][1] * t[1] + J.Mer[i][2] * t[2]));
#pragma unroll
   for (int k = 0; k < 5; k++)
#pragma unroll
      for (int i = k + 1; i < 5; i++) b[i] -= F.LU[i][k] * b[k];
#pragma unroll
   for (int k = 4; k >= 0; k--) {
#pragma unroll
      for (int j = k + 1; j < 5; j++) b[k] -= F.LU[k][j] * b[j];
      b[k] *= F.LU[k][k];
   }
#pragma unroll
   for (int i = 0; i < 5; i++) dx[i] = b[i];
#pragma unroll
   for (int i = 0; i < 3; i++) {
      double s = t[i];
#pragma unroll
      for (int j = 0; j < 5; j++) s -= F.Y[i][j] * b[j];
      dx[5 + i] = s;
   }
}

ECM_DI double norm8(const double v[8]) { double s = 0; for (int i = 0; i < 8; i++) s += v[i] * v[i]; return sqrt(s); }

// ------------------------------------------------------------------------------------------------------------
// one quadrature point: reference kernel_setup -> getResponseECM -> kernel_postprocessing, fused
//   vgrad  : velocity gradient L(i,j) = dv_i/dx_j as L[i + 3 j]
//   sv0/s0 : begin-of-step state (28) / Voigt stress (6)
//   sv1/s1 : end-of-step outputs;  cmat: 6x6 tangent d sigma / d eps (engineering shear), column-major
// returns 0 on success, 1 if the local solve failed to converge
// ------------------------------------------------------------------------------------------------------------
template <int KIN>
ECM_DI int point_update(const MatParams& mp, double dt, const double L[9], const double sv0[NSTATEV], const double s0[6],
                        double sv1[NSTATEV], double s1[6], double cmat[36]) {
   // ---- kernel_setup (reference src/mechanics_ecmech.cpp:42-99)
   const double w_sm[3] = { 0.5 * (L[2 + 3 * 1] - L[1 + 3 * 2]), 0.5 * (L[0 + 3 * 2] - L[2 + 3 * 0]), 0.5 * (L[1 + 3 * 0] - L[0 + 3 * 1]) };
   const double dkk = L[0] + L[4] + L[8];
   const double d_mean = -(1.0 / 3.0) * dkk;
   double d_sm[5];
   sym_to_vecd(L[0] + d_mean, L[4] + d_mean, L[8] + d_mean, 0.5 * (L[1 + 3 * 0] + L[0 + 3 * 1]), 0.5 * (L[2 + 3 * 0] + L[0 + 3 * 2]),
               0.5 * (L[2 + 3 * 1] + L[1 + 3 * 2]), d_sm);
   double dnorm2 = 0; for (int i = 0; i < 5; i++) dnorm2 += d_sm[i] * d_sm[i];
   const double dnorm = sqrt(dnorm2), dEff = SQR2B3 * dnorm;
   const double vOld = sv0[IND_VOL], vNew = vOld * exp(dkk * dt), delv = vNew - vOld;
   const double pOld = -(1.0 / 3.0) * (s0[0] + s0[1] + s0[2]);
   double s_old[5]; sym_to_vecd(s0[0] + pOld, s0[1] + pOld, s0[2] + pOld, s0[5], s0[4], s0[3], s_old);
   // ---- EOS ("updateSimple", EosModelConst<false>): p = K (1/v - 1) + Gamma e
   double eNew = sv0[IND_EINT] - delv * pOld;
   const double tK = mp.tK0 + eNew * mp.dtde;
   const double bulkNew = mp.bulk * vNew + mp.gamma * pOld * vNew;
   // ---- hardness to end of step with begin-of-step slip rates
   double shrate_o = 0; for (int a = 0; a < NSLIP; a++) shrate_o += fabs(sv0[H_GDOT + a]);
   const double h_u = kin_update_h<KIN>(mp, sv0[H_H], dt, shrate_o);
   // ---- point problem set-up
   Prob pb;
   pb.dt_ri = 1.0 / dt; pb.detV_ri = 1.0 / vNew;
   double qn[4]; { double n2 = 0; for (int i = 0; i < 4; i++) n2 += sv0[H_Q + i] * sv0[H_Q + i]; const double ni = 1.0 / sqrt(n2); for (int i = 0; i < 4; i++) qn[i] = sv0[H_Q + i] * ni; }
   {
      double Cn[9]; quat_to_mat(qn, Cn);
      rot_vecd_T(Cn, d_sm, pb.d_n);
      for (int i = 0; i < 3; i++) pb.w_n[i] = Cn[i] * w_sm[0] + Cn[3 + i] * w_sm[1] + Cn[6 + i] * w_sm[2];
   }
   for (int i = 0; i < 5; i++) pb.e_n[i] = sv0[H_E + i];
   double adots_ref;
   if (KIN == KIN_KMBALD) {
      const double sq = sqrt(h_u);
      pb.kv.g = mp.go + mp.s * sq; pb.kv.gam_w = mp.gam_wo / sq; pb.kv.gam_r = mp.gam_ro * sq * sq; pb.kv.c_e = (mp.c_1 / tK) * mp.mu_ref;
      adots_ref = pb.kv.gam_w;
   } else { pb.kv.g = h_u; pb.kv.gam_w = mp.gam_w; pb.kv.gam_r = 0; pb.kv.c_e = 0; adots_ref = mp.gam_w; }
   pb.sc = (dnorm < EPS_SQRT * adots_ref) ? 1.0 / adots_ref : fmin(1.0 / dnorm, 1.0e6 * dt);

   // ---- trust-region dog-leg Newton (SNLS "TrDlDenseG" defaults)
   double x[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
   double r[8], gdot[NSLIP], dis_rate;
   Jac J; Fact F;
   int nfev = 1; bool conv = false;
   bool ok = eval_rj<KIN, true>(mp, pb, x, r, J, gdot, dis_rate);
   double res = norm8(r), res_0 = res;
   if (ok && res < mp.tol) conv = true;
   if (ok && !conv) {
      const double cs[8] = { E_SCALE, E_SCALE, E_SCALE, E_SCALE, E_SCALE, R_SCALE, R_SCALE, R_SCALE };
      double delta = 1.0;
      double nr[8], grad[8];
      double nr2norm = 0, Jg_2 = 0, norm_grad = 0, norm2_grad = 0, norm_s_sd_opt = 0, res_cauchy = 0;
      bool reject_prev = false;
      for (int it = 0; it < 200; it++) {
         if (!reject_prev) {
            // grad = Js^T r, Jg = Js grad, Newton step; only vectors and scalars survive the next evaluation
            double t[8], u[8];
            jac_mult_T(mp, pb, J, r, t);
            for (int i = 0; i < 8; i++) { grad[i] = pb.sc * cs[i] * t[i]; u[i] = cs[i] * grad[i]; }
            jac_mult(mp, pb, J, u, t);
            Jg_2 = 0; norm2_grad = 0;
            for (int i = 0; i < 8; i++) { const double jg = pb.sc * t[i]; Jg_2 += jg * jg; norm2_grad += grad[i] * grad[i]; }
            norm_grad = sqrt(norm2_grad);
            const double fac = (Jg_2 > 0) ? norm2_grad / Jg_2 : 0.0;
            norm_s_sd_opt = (Jg_2 > 0) ? fac * norm_grad : 1e300;
            // |r + Js sd|, sd = -fac grad  (needed for the dog-leg prediction)
            { double s2 = 0; for (int i = 0; i < 8; i++) { const double v = r[i] - fac * pb.sc * t[i]; s2 += v * v; } res_cauchy = sqrt(s2); }
            jac_factor(mp, pb, J, F);
            if (F.ok) {
               double rhs[8]; for (int i = 0; i < 8; i++) rhs[i] = -r[i] / pb.sc;
               jac_solve<false>(J, F, rhs, t);
               for (int i = 0; i < 8; i++) nr[i] = t[i] / cs[i];
               nr2norm = norm8(nr);
            } else { nr2norm = 1e300; for (int i = 0; i < 8; i++) nr[i] = 0; }
         }
         double delx[8], pred_resid; bool use_nr = false;
         if (nr2norm <= delta) { use_nr = true; for (int i = 0; i < 8; i++) delx[i] = nr[i]; pred_resid = 0.0; }
         else if (norm_s_sd_opt >= delta) {
            const double f = delta / norm_grad;
            for (int i = 0; i < 8; i++) delx[i] = -grad[i] * f;
            pred_resid = sqrt(fmax(res_0 * res_0 - 2.0 * delta * norm_grad + delta * delta * Jg_2 / norm2_grad, 0.0));
         } else {
            const double fac = norm2_grad / Jg_2;
            double qa = 0, qb = 0;
            for (int i = 0; i < 8; i++) { const double sd = -grad[i] * fac, p = nr[i] - sd; qa += p * p; qb += p * sd; }
            const double qc = norm_s_sd_opt * norm_s_sd_opt - delta * delta;
            const double beta = (-qb + sqrt(fmax(qb * qb - qa * qc, 0.0))) / qa;
            for (int i = 0; i < 8; i++) { const double sd = -grad[i] * fac; delx[i] = sd + beta * (nr[i] - sd); }
            pred_resid = (1.0 - beta) * res_cauchy;   // the Newton point zeroes the linear model
         }
         for (int i = 0; i < 8; i++) x[i] += delx[i];
         // evaluate straight into (r, J): after a rejection only nr/grad/scalars of the accepted point are needed
         ok = eval_rj<KIN, true>(mp, pb, x, r, J, gdot, dis_rate); nfev++;
         bool reject;
         if (!ok) { reject = true; delta = fmax(delta * 0.25, 1e-12); }
         else {
            res = norm8(r);
            if (res < mp.tol) { conv = true; break; }
            const double actual = res - res_0, pred = pred_resid - res_0;
            if (pred == 0.0) delta = fmin(delta * 1.5, 1e4);
            else {
               const double rho = actual / pred;
               if (rho > 0.75 && actual < 0.0 && !use_nr) delta = fmin(delta * 1.5, 1e4);
               else if (rho < 0.35) delta = fmax(delta * 0.25, 1e-12);
            }
            reject = (actual > 0.0);
         }
         if (reject) { for (int i = 0; i < 8; i++) x[i] -= delx[i]; res = res_0; reject_prev = true; if (delta <= 1e-12) break; }
         else { res_0 = res; reject_prev = false; }
      }
   }
   // ---- converged state, by-products and tangent at the solution
   ok = eval_rj<KIN, true>(mp, pb, x, r, J, gdot, dis_rate);
   double e_f[5], xi[3];
   for (int i = 0; i < 5; i++) e_f[i] = pb.e_n[i] + x[i] * E_SCALE;
   for (int i = 0; i < 3; i++) xi[i] = x[5 + i] * R_SCALE;
   double qf[4];
   {
      const double th2 = xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2];
      double cq, sq;   // cos(th/2), sin(th/2)/th
      if (th2 < 1.0e-4) { const double h2 = 0.25 * th2; cq = 1.0 - 0.5 * h2 * (1.0 - h2 * (1.0 / 12.0) * (1.0 - h2 * (1.0 / 30.0))); sq = 0.5 * (1.0 - h2 * (1.0 / 6.0) * (1.0 - h2 * (1.0 / 20.0) * (1.0 - h2 * (1.0 / 42.0)))); }
      else { const double th = sqrt(th2); double sn, cs; sincos(0.5 * th, &sn, &cs); cq = cs; sq = sn / th; }
      const double a[4] = { cq, sq * xi[0], sq * xi[1], sq * xi[2] };
      qf[0] = qn[0] * a[0] - qn[1] * a[1] - qn[2] * a[2] - qn[3] * a[3];
      qf[1] = qn[0] * a[1] + qn[1] * a[0] + qn[2] * a[3] - qn[3] * a[2];
      qf[2] = qn[0] * a[2] - qn[1] * a[3] + qn[2] * a[0] + qn[3] * a[1];
      qf[3] = qn[0] * a[3] + qn[1] * a[2] - qn[2] * a[1] + qn[3] * a[0];
   }
   double Cf[9]; quat_to_mat(qf, Cf);
   const double kdj[5] = { mp.kd0 * pb.detV_ri, mp.kd0 * pb.detV_ri, mp.kd2 * pb.detV_ri, mp.kd2 * pb.detV_ri, mp.kd2 * pb.detV_ri };
   double s_lat[5], s_sm[5];
   for (int i = 0; i < 5; i++) s_lat[i] = kdj[i] * e_f[i];
   rot_vecd(Cf, s_lat, s_sm);
   // ---- tangent: lattice-frame d sigma'/d D' by implicit differentiation, rotated to the sample frame, then to Voigt
   {
      jac_factor(mp, pb, J, F);
      double A[9], Tr[9]; exp_map(xi, A, Tr);
      double Ms[5][3]; m35(s_lat, Ms);
      double Llat[5][5];
#pragma unroll
      for (int c = 0; c < 5; c++) {
         double rhs[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, dx[8];
         rhs[c] = 1.0;
         jac_solve<true>(J, F, rhs, dx);
         double dth[3];
         for (int i = 0; i < 3; i++) dth[i] = Tr[3 * i] * dx[5] + Tr[3 * i + 1] * dx[6] + Tr[3 * i + 2] * dx[7];
         for (int k = 0; k < 5; k++) Llat[k][c] = kdj[k] * dx[k] - (Ms[k][0] * dth[0] + Ms[k][1] * dth[1] + Ms[k][2] * dth[2]);
      }
      // D55 = Q5 Llat Q5^T : rotate columns then rows with vecd(C . C^T)
      double T1[5][5];
#pragma unroll
      for (int c = 0; c < 5; c++) { double col[5], out[5]; for (int k = 0; k < 5; k++) col[k] = Llat[k][c]; rot_vecd(Cf, col, out); for (int k = 0; k < 5; k++) T1[k][c] = out[k]; }
      double D55[5][5];
#pragma unroll
      for (int k = 0; k < 5; k++) { double row[5], out[5]; for (int c = 0; c < 5; c++) row[c] = T1[k][c]; rot_vecd(Cf, row, out); for (int c = 0; c < 5; c++) D55[k][c] = out[c]; }
      // Voigt: sigma_svec = V65 sigma_vecd ; d_vecd = S56 eps_svec(eng. shear) / dt
      const double dti = pb.dt_ri * (F.ok ? 1.0 : 0.0);
      double T2[5][6];
#pragma unroll
      for (int k = 0; k < 5; k++) {
         T2[k][0] = (SQR2I * D55[k][0] - SQR6I * D55[k][1]) * dti;
         T2[k][1] = (-SQR2I * D55[k][0] - SQR6I * D55[k][1]) * dti;
         T2[k][2] = (2.0 * SQR6I * D55[k][1]) * dti;
         T2[k][3] = (SQR2I * D55[k][4]) * dti;
         T2[k][4] = (SQR2I * D55[k][3]) * dti;
         T2[k][5] = (SQR2I * D55[k][2]) * dti;
      }
#pragma unroll
      for (int j = 0; j < 6; j++) {
         const double t1 = SQR2I * T2[0][j], t2 = SQR6I * T2[1][j];
         const double bk = (j < 3) ? bulkNew : 0.0;
         // column-major C(i,j) at cmat[i + 6 j]
         cmat[0 + 6 * j] = t1 - t2 + bk; cmat[1 + 6 * j] = -t1 - t2 + bk; cmat[2 + 6 * j] = SQR2B3 * T2[1][j] + bk;
         cmat[3 + 6 * j] = SQR2I * T2[4][j]; cmat[4 + 6 * j] = SQR2I * T2[3][j]; cmat[5 + 6 * j] = SQR2I * T2[2][j];
      }
   }
   // ---- energy, history, outputs (getResponseSngl tail + reference kernel_postprocessing src/mechanics_ecmech.cpp:116-152)
   { double wrk = 0; for (int k = 0; k < 5; k++) wrk += (s_old[k] + s_sm[k]) * d_sm[k]; eNew += 0.25 * (vOld + vNew) * dt * wrk; }
   double shrate = 0; for (int a = 0; a < NSLIP; a++) shrate += fabs(gdot[a]);
   sv1[H_SHRATE] = shrate;
   sv1[H_SHR] = sv0[H_SHR] + shrate * dt;
   sv1[H_FLOW] = ((dEff > TINY_SQRT) ? dis_rate * dt : 0.0) + sv0[H_FLOW];   // accumulated plastic work
   sv1[H_NFEV] = (double)nfev;
   for (int i = 0; i < 5; i++) sv1[H_E + i] = e_f[i];
   { double dq = 0; for (int i = 0; i < 4; i++) dq += qf[i] * qn[i]; const double sg = dq < 0 ? -1.0 : 1.0; for (int i = 0; i < 4; i++) sv1[H_Q + i] = sg * qf[i]; }
   sv1[H_H] = h_u;
   for (int a = 0; a < NSLIP; a++) sv1[H_GDOT + a] = gdot[a];
   sv1[IND_VOL] = vNew; sv1[IND_EINT] = eNew;
   const double pNew = mp.bulk * (1.0 / vNew - 1.0) + mp.gamma * eNew;
   {
      const double t1 = SQR2I * s_sm[0], t2 = SQR6I * s_sm[1];
      s1[0] = t1 - t2 - pNew; s1[1] = -t1 - t2 - pNew; s1[2] = SQR2B3 * s_sm[1] - pNew;
      s1[3] = SQR2I * s_sm[4]; s1[4] = SQR2I * s_sm[3]; s1[5] = SQR2I * s_sm[2];
   }
   return (conv && ok) ? 0 : 1;
}

}  // namespace ecmdev
