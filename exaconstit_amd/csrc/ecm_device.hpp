// Device-side crystal-plasticity point update for gfx950 (MI355X).
//
// Replaces, behind the reference's ExaModel seam, what ExaConstit obtains from
//   ecmech::matModelBase::getResponseECM        (call site reference src/mechanics_ecmech.cpp:176-186)
// for the models of reference src/mechanics_ecmech.hpp:407-414,460-463.  One thread owns one quadrature point; small
// tensors are indexed at compile time only, the 8x8 point system is kept in its 5+3 block form (LDL^T of the symmetric
// 5x5 block, closed-form inverse of the 3x3 rotation block, block Gauss-Seidel for a Newton step, exact Schur
// complement for the tangent), and the slip-system tables are small-integer compile-time constants so that the 12
// systems unroll into adds / FMAs with inline constants.  What does not fit the 256 VGPRs of two waves per SIMD lives in
// a per-lane LDS stash (see "Register budget").  BCC {110}<111> and FCC {111}<110> share the symmetric Schmid tensors;
// only the sign of the plastic spin differs, so one table serves both.
//
// This is an independent implementation of the same published algorithm that oracle/ecmech_port.hpp restates on
// the CPU; tests compare the two on identical inputs.  Nothing here includes or links the oracle.
#pragma once
#include <hip/hip_runtime.h>

namespace ecmdev {

#define ECM_DI __device__ __forceinline__
#ifndef ECM_SLIP_UNROLL
#define ECM_SLIP_UNROLL 1   // slip-system loop unroll factor of the point problem (ILP vs registers; tuned on MI355X)
#endif

constexpr int NSLIP = 12;
constexpr int kSlipUnroll = ECM_SLIP_UNROLL;
constexpr double SQR2 = 1.4142135623730951, SQR3 = 1.7320508075688772;
constexpr double SQR2I = 0.70710678118654752, SQR6I = 0.40824829046386302, SQR2B3 = 0.81649658092772603;
constexpr double TINY_SQRT = 1.0e-90, EPS_SQRT = 1.0e-8;
constexpr double GAM_RATIO_OVF = 1.0e45, LN_GAM_RATIO_MIN = -138.15510557964274;
constexpr double E_SCALE = 5.0e-4, R_SCALE = 0.01;
// Streaming accesses of the per-point records (read once / written once per launch) carry the non-temporal hint: the state / stress / record rows a
// launch writes are next read by ANOTHER launch after gigabytes of other traffic (3.5 GB of records, 3.8 GB of state at 128^3), so keeping them out
// of L2 leaves it to the node rows the gathers of neighbouring waves share.  History: round 2 measured the hint as a loss (plastic pass 7.28 -> 7.56 ms)
// when the kernel still spilled - its scratch lines are what the hint displaced; round 5 (no scratch): stores 4.21 -> 4.18 ms, loads alone +-0, both
// 4.21 / 4.23 -> 4.17 / 4.13 ms, elastic pass 2.76 -> 2.70 ms (profiles/r05_kernel_experiments.txt).  ECM_NT=0 restores plain accesses (A/B switch).
#ifndef ECM_NT
#define ECM_NT 1
#endif
#ifndef ECM_NT_LD
#define ECM_NT_LD ECM_NT
#endif
#ifndef ECM_NT_ST
#define ECM_NT_ST ECM_NT
#endif
#ifndef ECM_NT_REC
#define ECM_NT_REC ECM_NT_ST   // the 13 16-byte pairs of the compact gradient record (read by the NEXT kernel, 3.5 GB at 128^3: never from cache)
#endif
__device__ __forceinline__ double ldg(const double* p) {
#if ECM_NT_LD
   return __builtin_nontemporal_load(p);
#else
   return *p;
#endif
}
__device__ __forceinline__ void stg(double* p, double v) {
#if ECM_NT_ST
   __builtin_nontemporal_store(v, p);
#else
   *p = v;
#endif
}
// output store of the point update: global memory (non-temporal), or - LO, the staged AOS launch (model_kernel.hpp, PointIO<.., STG>) - the lane's row of
// the wave's LDS stage, from where the wave stores whole rows of 64 points coalesced
template <bool LO>
__device__ __forceinline__ void ost(double* p, double v) { if constexpr (LO) *p = v; else stg(p, v); }
__device__ __forceinline__ void stg2(double2* p, double a, double b) {
#if ECM_NT_REC
   typedef double vd2 __attribute__((ext_vector_type(2)));
   vd2 v; v.x = a; v.y = b;
   __builtin_nontemporal_store(v, reinterpret_cast<vd2*>(p));
#else
   *p = make_double2(a, b);
#endif
}

// history layout (reference src/mechanics_ecmech.hpp:165-185)
constexpr int H_SHRATE = 0, H_SHR = 1, H_FLOW = 2, H_NFEV = 3, H_E = 4, H_Q = 9, H_H = 13, H_GDOT = 14;
constexpr int RS_N = 10;   // doubles of a cut-off point's solver state (point_update, TailIO)
// tail split (model_kernels.hip, launch_levels): where a cut-off point goes, and the saved state a listed point resumes from (already offset to its slot)
// defer_reject (capped full launch with state buffers): a point whose trial is rejected leaves for the dense launch right away (point_update)
struct TailIO { int* list_out = nullptr; double* rs_out = nullptr; const double* rs_in = nullptr; int64_t stride = 0; bool defer_reject = false; };
constexpr int NUM_HIST = 26, NSTATEV = 28, IND_VOL = 26, IND_EINT = 27;

enum { KIN_VOCE = 0, KIN_VOCE_NL = 1, KIN_KMBALD = 2,
       KIN_KMBALD_GA = 3,     // compile-time variant of KIN_KMBALD for with_g_athermal (BCC): same arithmetic, window systems deferred (eval_rj)
       KIN_PQ1 = 4,           // flag on the two Kocks-Mecking kinds: thermal-activation exponents p == q == 1 known at compile time (no pow()
                              // code in the kinetics: 7-11 % fewer cycles at 128^3 through lower register pressure; same arithmetic)
       KIN_XN49 = 8 };        // flag on the two Voce kinds: power-law exponent 1/m - 1 == 49 (m = 0.02, the shipped sets) known at compile time.  The
                              // run-time choice among the x^9 / x^19 / x^49 / x^99 / rolled / exp-log forms sits in every evaluation: each form leaves
                              // its 12 powers in other registers, and the merge costs ~100 v_mov / v_and per evaluation (9 % of the launch's VALU work)
constexpr int kin_base(int k) { return k & 3; }
constexpr bool kin_pq1(int k) { return (k & KIN_PQ1) != 0; }
constexpr bool kin_sc_exp(int k) { return kin_base(k) == KIN_KMBALD_GA; }   // exp_n coefficients through scalar registers (see exp_n)
constexpr int kin_xn_ct(int k) { return (k & KIN_XN49) ? 49 : 0; }   // compile-time power-law exponent (0: run-time choice)
constexpr bool kin_is_km(int k) { return kin_base(k) == KIN_KMBALD || kin_base(k) == KIN_KMBALD_GA; }

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
// The same tables as small integers: P_TAB[c][a] = PSC[c] * SP[c][a], Q_TAB[c][a] = PB * SQ[c][a].  With the slip loop fully
// unrolled every product with +-1 / +-2 is an add or an FMA with an inline constant and the zeros disappear at compile time.
__device__ constexpr int SP[5][NSLIP] = {
   { -1, -1, 2, 1, 1, -2, -1, -1, 2, 1, 1, -2 },
   { -1, 1, 0, -1, 1, 0, -1, 1, 0, -1, 1, 0 },
   { 1, -1, 0, -1, 1, 0, 1, -1, 0, -1, 1, 0 },
   { -1, 0, 1, 0, -1, 1, 1, 0, -1, 0, 1, -1 },
   { 0, 1, -1, -1, 0, 1, 0, -1, 1, 1, 0, -1 } };
__device__ constexpr int SQ[3][NSLIP] = {
   { -2, 1, 1, -1, 2, -1, 2, -1, -1, 1, -2, 1 },
   { 1, -2, 1, -2, 1, 1, -1, 2, -1, 2, -1, -1 },
   { 1, 1, -2, 1, 1, -2, 1, 1, -2, 1, 1, -2 } };
__device__ constexpr double PSC[5] = { PA, 0.5, PA, PA, PA };
constexpr bool slip_tables_agree() {
   for (int a = 0; a < NSLIP; a++) {
      for (int c = 0; c < 5; c++) if (P_TAB[c][a] != PSC[c] * SP[c][a]) return false;
      for (int c = 0; c < 3; c++) if (Q_TAB[c][a] != PB * SQ[c][a]) return false;
   }
   return true;
}
static_assert(slip_tables_agree(), "integer slip tables must reproduce P_TAB / Q_TAB");
// the same numbers, one row of 8 per slip system (P0..P4, Q0..Q2): the slip-system loop of the point problem is kept ROLLED
// (uniform index -> scalar loads), which more than halves the register footprint of the fused kernel
__device__ const double PQ_TAB[NSLIP][8] = {
   { -PA, -0.5, PA, -PA, 0.0, -2 * PB, PB, PB },       { -PA, 0.5, -PA, 0.0, PA, PB, -2 * PB, PB },
   { 2 * PA, 0.0, 0.0, PA, -PA, PB, PB, -2 * PB },     { PA, -0.5, -PA, 0.0, -PA, -PB, -2 * PB, PB },
   { PA, 0.5, PA, -PA, 0.0, 2 * PB, PB, PB },          { -2 * PA, 0.0, 0.0, PA, PA, -PB, PB, -2 * PB },
   { -PA, -0.5, PA, PA, 0.0, 2 * PB, -PB, PB },        { -PA, 0.5, -PA, 0.0, -PA, -PB, 2 * PB, PB },
   { 2 * PA, 0.0, 0.0, -PA, PA, -PB, -PB, -2 * PB },   { PA, -0.5, -PA, 0.0, PA, PB, 2 * PB, PB },
   { PA, 0.5, PA, PA, 0.0, -2 * PB, -PB, PB },         { -2 * PA, 0.0, 0.0, -PA, -PA, PB, -PB, -2 * PB } };

// Material description, passed by value as a kernel argument (lives in SGPRs / kernarg segment).
struct MatParams {
   int kin;                 // KIN_*
   int with_g_athermal;     // KMBalD: 1 for BCC ("Kin_BCC_A"), 0 for FCC ("Kin_FCC_B")
   int xn_int;              // 1/m - 1 when it is a small integer (m = 0.02 -> 49): power by repeated multiplication, else 0
   double qsign;            // +1 FCC, -1 BCC: sign of the plastic-spin vectors
   double kd0, kd2;         // Kirchhoff' = diag(kd0,kd0,kd2,kd2,kd2) e'
   double pk0, pk1, pk2;    // PSC[0] kd0, PSC[1] kd0, PSC[2] kd2: Kirchhoff stress in the scaled integer slip basis straight from the strain
   double ikd0, ikd2;       // their reciprocals (host-computed: the Newton step would otherwise divide by them in every sweep)
   double bulk, gmod, gamma, tK0, dtde, tol;
   // Voce power law
   double xnn, xn, gam_w, h0, tausi, taus0, xmprime, xms, gamss0, t_min, t_max;
   // Kocks-Mecking balanced dislocation density
   double mu_ref, c_1, tau_a, p, q, gam_wo, gam_ro, wrD, go, s, k1, k2o, ninv, gamma_o, hdn_min;
};

// ------------------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------------------
// reciprocal of a well-scaled, non-zero double: v_rcp_f64 seed + two Newton steps (5 instructions instead of the ~15 of an IEEE
// division; relative error ~1e-16).  Only used for pivots / determinants that are checked for sign or finiteness afterwards.
ECM_DI double frcp(double x) {
   double y = __builtin_amdgcn_rcp(x);
   y = fma(fma(-x, y, 1.0), y, y);
   y = fma(fma(-x, y, 1.0), y, y);
   return y;
}

// exp for N arguments (Kocks-Mecking kinetics: three exponentials per slip system and evaluation).  Same algorithm as the library routine -
// n = rint(x log2 e), r = x - n ln2 (two-term), polynomial, ldexp - without its overflow / underflow selects: arguments are clamped to
// [-800, 720], where ldexp itself produces 0 / inf; Taylor degree 13 on |r| <= ln2 / 2 (truncation 4e-18), |error| <= 1.5 ulp.  22 FP64
// instructions per value instead of 38.  The values are computed one after the other (ECM_KM_EXP == 1: the empty asm keeps the compiler
// from interleaving them): at 128^3 the interleaved form costs FCC 24.1 ms against 21.2 - the kernel has no registers left for six
// polynomial chains in flight, and two waves per SIMD hide the FMA latency of a single chain.
#ifndef ECM_KM_EXP
#define ECM_KM_EXP 1   // A/B switch: 0 = library exp(), 1 = this routine, 2 = this routine, interleaving left to the compiler
#endif
// a double constant that is materialised in a scalar register pair where it is used.  Left alone, the compiler hoists the polynomial
// coefficients of exp_n / log_near1 out of the Newton loop INTO VECTOR REGISTERS (v_fmac wants its addend in the destination register) - 28
// VGPRs for the exp coefficients alone, the logarithm's went to scratch and were re-loaded, one dependent round trip each, inside the
// slip-system loop of the Kocks-Mecking kernels.  The volatile asm pins the materialisation (two s_mov, no VALU work) to the point of use.
#ifndef ECM_SCONST
#define ECM_SCONST 1
#endif
ECM_DI double sconst(double c) {
#if ECM_SCONST
   asm volatile("" : "+s"(c));
#endif
   return c;
}
#ifndef ECM_SCONST_EXP
#define ECM_SCONST_EXP 1   // ... in exp_n (A/B switch)
#endif
template <bool SC> ECM_DI double sconst_e(double c) { return (ECM_SCONST_EXP && SC) ? sconst(c) : c; }
// sin and cos of a moderate angle (rotation increments: |x| below ~1e5): Cody-Waite reduction by pi/2 in two parts and the fdlibm kernel
// polynomials on |r| <= pi/4, |error| ~ 1 ulp.  Takes the place of the library sincos in the large-angle branches of the exponential map,
// which are rarely taken but whose 12 polynomial coefficients the compiler kept in vector registers - or in scratch - across the whole
// Newton loop (see sconst)
ECM_DI void sincos_s(const double x, double& sn, double& cs) {
   const double k = rint(x * sconst(6.36619772367581382433e-01));
   double r = fma(-k, sconst(1.57079632673412561417e+00), x);
   r = fma(-k, sconst(6.07710050650619224932e-11), r);
   const double z = r * r;
   double ps = sconst(1.58969099521155010221e-10);
   ps = fma(ps, z, sconst(-2.50507602534068634195e-08)); ps = fma(ps, z, sconst(2.75573137070700676789e-06));
   ps = fma(ps, z, sconst(-1.98412698298579493134e-04)); ps = fma(ps, z, sconst(8.33333333332248946124e-03));
   ps = fma(ps, z, sconst(-1.66666666666666324348e-01));
   const double s0 = fma(r * z, ps, r);
   double pc = sconst(-1.13596475577881948265e-11);
   pc = fma(pc, z, sconst(2.08757232129817482790e-09)); pc = fma(pc, z, sconst(-2.75573143513906633035e-07));
   pc = fma(pc, z, sconst(2.48015872894767294178e-05)); pc = fma(pc, z, sconst(-1.38888888888741095749e-03));
   pc = fma(pc, z, sconst(4.16666666666666019037e-02));
   const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
   const int q = (int)k & 3;
   const double sa = (q & 1) ? c0 : s0, ca = (q & 1) ? s0 : c0;
   sn = (q & 2) ? -sa : sa;
   cs = (q == 1 || q == 2) ? -ca : ca;
}

// SC: coefficients through sconst.  Measured at 128^3: athermal-threshold (BCC) kernel 7.95 -> 7.61 ms with, batched FCC kernel 16.8 -> 18.2 ms (36
// exponentials per evaluation there: the 28 scalar moves per value get in the way), so the kernel kind decides (kin_sc_exp)
template <int N, bool FAST, bool SC = true>   // FAST = false: library routine (the general instantiations, whose register allocation the routine upsets)
ECM_DI void exp_n(double v[N]) {
   if constexpr (ECM_KM_EXP != 0 && FAST) {
   constexpr double tab[14] = { 1.4426950408889634074, -6.93147180369123816490e-01, -1.90821492927058770002e-10,
      1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0 };
#pragma unroll
   for (int a = 0; a < N; a++) {
      const double x = fmin(fmax(v[a], -800.0), 720.0); const double n = rint(x * sconst_e<SC>(tab[0])); double r = fma(n, sconst_e<SC>(tab[1]), x); r = fma(n, sconst_e<SC>(tab[2]), r);
      double p = fma(sconst_e<SC>(tab[3]), r, sconst_e<SC>(tab[4]));
#pragma unroll
      for (int k = 5; k < 14; k++) p = fma(p, r, sconst_e<SC>(tab[k]));
      p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0); v[a] = ldexp(p, (int)n);
#if ECM_KM_EXP == 1
      asm volatile("" : "+v"(v[a]));
#endif
   }
   } else {
#pragma unroll
   for (int a = 0; a < N; a++) v[a] = exp(v[a]);
   }
}

// log(x) for x in [0.75, 1.25] (|error| ~ 1 ulp): 2 atanh(s), s = (x - 1) / (x + 1) <= 0.112 with the quotient corrected by one residual step;
// 25 FP64 instructions instead of the ~60 of the general routine (its table / double-double range reduction is not needed this close to 1)
ECM_DI double log_near1(double x) {
   const double w = x - 1.0, d = 2.0 + w;       // w exact for x in [0.5, 2]
   const double di = frcp(d);
   double s = w * di;
   s = fma(fma(-s, d, w), di, s);
   const double s2 = s * s;
   double P = sconst(1.0 / 17.0);
   P = fma(P, s2, sconst(1.0 / 15.0)); P = fma(P, s2, sconst(1.0 / 13.0)); P = fma(P, s2, sconst(1.0 / 11.0)); P = fma(P, s2, sconst(1.0 / 9.0));
   P = fma(P, s2, sconst(1.0 / 7.0)); P = fma(P, s2, sconst(1.0 / 5.0)); P = fma(P, s2, sconst(1.0 / 3.0));
   const double s_2 = s + s;
   return fma(s_2 * s2, P, s_2);
}

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
      const double th = sqrt(th2); double sn, cs; sincos_s(th, sn, cs);
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

ECM_DI double pow_xn(const MatParams& mp, double at) {   // at^(1/m - 1)
   if (mp.xn_int > 0) {
      double r = 1.0, b = at;
      for (int e = mp.xn_int; e; e >>= 1) { if (e & 1) r *= b; b *= b; }   // uniform trip count
      return r;
   }
   return exp(mp.xn * log(at));
}

ECM_DI void voce_gdot(const MatParams& mp, double g_i, double tau, double& gdot, double& dg) {
   gdot = 0.0; dg = 0.0;
   const double t_frac = tau * g_i, at = fabs(t_frac);
   if (at > mp.t_min) {
      if (at > mp.t_max) {
         gdot = copysign(mp.gam_w * GAM_RATIO_OVF, t_frac);
         dg = fabs(gdot) * mp.xnn / fabs(tau);
      } else {
         const double temp = mp.gam_w * pow_xn(mp, at);
         gdot = temp * t_frac; dg = temp * mp.xnn * g_i;
      }
   }
}

// |t|^E for the 12 systems with the exponent known at compile time: the square-and-multiply chain fully unrolled (x^49 = 7 multiplications
// per system, no loop, no exponent bits in registers).  The usual rate sensitivities m = 0.1, 0.05, 0.02, 0.01 give E = 1/m - 1 = 9, 19, 49, 99;
// any other integer exponent takes the rolled loop below, non-integers exp(xn log|t|).  Same multiplication order as the loop: same bits.
template <int E>
ECM_DI void pow12_ct(const double tf[NSLIP], double pw[NSLIP]) {
   static_assert(E >= 1, "positive exponent");
   double b[NSLIP];
#pragma unroll
   for (int a = 0; a < NSLIP; a++) { b[a] = fabs(tf[a]); pw[a] = 1.0; }
#pragma unroll
   for (int e = E; e; e >>= 1) {
      if (e & 1) {
#pragma unroll
         for (int a = 0; a < NSLIP; a++) pw[a] *= b[a];
      }
      if (e >> 1) {
#pragma unroll
         for (int a = 0; a < NSLIP; a++) b[a] *= b[a];
      }
   }
}

// Voce power law for all 12 systems at once (independent chains -> the FP64 pipeline stays full).  WITHD: also d gdot / d tau.
template <bool WITHD, bool CUT, int XNCT = 0>
ECM_DI void voce_gdot12(const MatParams& mp, double g_i, const double tau[NSLIP], double gd[NSLIP], double dg[NSLIP]) {
   double tf[NSLIP], pw[NSLIP];
#pragma unroll
   for (int a = 0; a < NSLIP; a++) tf[a] = tau[a] * g_i;
   if constexpr (XNCT > 0) pow12_ct<XNCT>(tf, pw);
   else if (mp.xn_int == 49) pow12_ct<49>(tf, pw);
   else if (mp.xn_int == 99) pow12_ct<99>(tf, pw);
   else if (mp.xn_int == 19) pow12_ct<19>(tf, pw);
   else if (mp.xn_int == 9) pow12_ct<9>(tf, pw);
   else if (mp.xn_int > 0) {
      double b[NSLIP];
#pragma unroll
      for (int a = 0; a < NSLIP; a++) { pw[a] = 1.0; b[a] = fabs(tf[a]); }
      for (int e = mp.xn_int;;) {   // uniform trip count
         if (e & 1) {
#pragma unroll
            for (int a = 0; a < NSLIP; a++) pw[a] *= b[a];
         }
         e >>= 1;
         if (!e) break;
#pragma unroll
         for (int a = 0; a < NSLIP; a++) b[a] *= b[a];
      }
   } else {
#pragma unroll
      for (int a = 0; a < NSLIP; a++) pw[a] = exp(mp.xn * log(fabs(tf[a])));
   }
   const double dfac = mp.xnn * g_i;
   double amax = 0.0;
#pragma unroll
   for (int a = 0; a < NSLIP; a++) {
      const double at = fabs(tf[a]);
      amax = fmax(amax, at);
      const double temp = mp.gam_w * pw[a];
      // below t_min = (1e-60)^m the reference returns exactly 0; the power law itself is < 1e-60 there, so the cut only matters
      // for the slip rates that are written to the state (CUT), not for the sums of an evaluation
      const bool on = !CUT || at > mp.t_min;
      gd[a] = on ? temp * tf[a] : 0.0;
      if (WITHD) dg[a] = on ? temp * dfac : 0.0;
   }
   if (amax > mp.t_max) {   // rare: rate overflow guard of the reference, one branch for all systems
#pragma unroll
      for (int a = 0; a < NSLIP; a++) if (fabs(tf[a]) > mp.t_max) {
         gd[a] = copysign(mp.gam_w * GAM_RATIO_OVF, tf[a]);
         if (WITHD) dg[a] = fabs(gd[a]) * mp.xnn / fabs(tau[a]);
      }
   }
}

// keeps a uniform branch a branch: without it the compiler speculates the (pure) pow() of the general case and selects afterwards,
// i.e. every call pays two full pow() expansions (~400 instructions) even when p == q == 1
#define ECM_NO_SPECULATE() asm volatile("" ::: "memory")
#ifndef ECM_KM_LOG_NEAR1
#define ECM_KM_LOG_NEAR1 1   // Kocks-Mecking power-law tail: short-series logarithm when (t_min, t_max] lies in [0.75, 1.25] (A/B switch)
#endif
#ifndef ECM_KM_ONE_RCP
#define ECM_KM_ONE_RCP 1     // Kocks-Mecking: series combination of the thermal and drag branches with one reciprocal instead of three (A/B switch)
#endif
#ifndef ECM_EXP_PQ1
#define ECM_EXP_PQ1 0   // timing experiment: p == q == 1 known at compile time (no pow() code in the kinetics at all)
#endif
#ifndef ECM_KM_SKIP_BACK
#define ECM_KM_SKIP_BACK 1   // p == q == 1: backward-jump term dropped where it is below half an ulp of the forward term (A/B switch)
#endif
template <bool PQ1>
ECM_DI void mts_dG(const MatParams& mp, double c_e, double t_frac, double& exp_arg, double& dfac) {
   exp_arg = 0.0; dfac = 0.0;
   if (t_frac >= 1.0) return;
   double p_func, dp_func;
   const double at = fabs(t_frac);
   if (PQ1 || ECM_EXP_PQ1 || mp.p == 1.0) { p_func = t_frac; dp_func = 1.0; }
   else if (at < TINY_SQRT) { p_func = 0.0; dp_func = 0.0; }
   else { ECM_NO_SPECULATE(); const double pw = pow(at, mp.p); p_func = copysign(pw, t_frac); dp_func = mp.p * pw / at; }
   const double q_arg = 1.0 - p_func;
   if (q_arg <= TINY_SQRT) return;
   double q_func, dq_func;
   if (PQ1 || ECM_EXP_PQ1 || mp.q == 1.0) { q_func = q_arg; dq_func = 1.0; }
   else { ECM_NO_SPECULATE(); q_func = pow(q_arg, mp.q); dq_func = mp.q * q_func / q_arg; }
   exp_arg = -c_e * q_func; dfac = c_e * dq_func * dp_func;
}

// Kocks-Mecking balanced kinetics (thermally activated forward - backward slip with a power-law tail, in series with a drag-limited
// branch) for KW slip systems at once: the exp / log chains of the systems are independent instruction streams the FP64 pipeline can
// overlap (one system at a time leaves it half idle at two waves per SIMD).  The early returns of a one-system formulation - dormant
// below the athermal threshold, drag-limited beyond t_max, forward exponent below ln(1e-60), power-law tail only above t_min - are kept
// as nested lane conditions, so a wave still skips every phase none of its lanes needs (with the athermal-threshold BCC variant most
// systems of a wave are dormant or saturated).  Measured at 128^3 against the one-system form: FCC 72 -> 55 ms, BCC 16.2 -> 14.3 ms.
#ifndef ECM_KW
#define ECM_KW 2
#endif
#ifndef ECM_KD
#define ECM_KD 2
#endif
constexpr int KD = ECM_KD;   // systems per group of the cheap classes in the deferred form (only one exp each: wider groups for ILP)
#ifndef ECM_KM_DEFER
#define ECM_KM_DEFER 1   // athermal-threshold (BCC) variant: window systems are treated one per lane after the group loop (eval_rj)
#endif
constexpr int KW = ECM_KW;   // slip systems evaluated together by the Kocks-Mecking kinetics (ILP vs registers; tuned on MI355X)
template <bool WITHD, int KW = ECM_KW, bool PQ1 = false, bool SC = true>
ECM_DI void kmbald_gdot4(const MatParams& mp, const KinVals& kv, const double tau[KW], double gdot[KW], double dg[KW]) {
   const double g_i = mp.with_g_athermal ? 1.0 / mp.tau_a : 1.0 / kv.g;
   const double gAth = mp.with_g_athermal ? kv.g : mp.tau_a;
   const double wi = 1.0 / mp.wrD;
   double at[KW], xr[KW], at0[KW];
   bool live[KW], over[KW], any_live = false;
#pragma unroll
   for (int a = 0; a < KW; a++) {
      at[a] = fabs(tau[a]); xr[a] = (at[a] - gAth) * wi; at0[a] = fmax(0.0, at[a] - gAth) * g_i;
      live[a] = (tau[a] != 0.0) && (xr[a] > 0.0); over[a] = at0[a] > mp.t_max;
      any_live = any_live || live[a];
      gdot[a] = 0.0; if (WITHD) dg[a] = 0.0;
   }
   // the nested ifs are lane conditions: a wave skips a phase when none of its lanes needs it, like the early returns of the scalar form
   if (any_live) {
      double ex[KW], gr[KW], dgr[KW];
#pragma unroll
      for (int a = 0; a < KW; a++) ex[a] = -fmax(xr[a], 0.0);
      exp_n<KW, PQ1, SC>(ex);
#pragma unroll
      for (int a = 0; a < KW; a++) {
         const bool small = xr[a] < EPS_SQRT;
         gr[a] = small ? kv.gam_r * xr[a] : kv.gam_r * (1.0 - ex[a]);
         dgr[a] = (small ? kv.gam_r : kv.gam_r * ex[a]) * wi;
         if (live[a] && over[a]) { gdot[a] = copysign(gr[a], tau[a]); if (WITHD) dg[a] = dgr[a]; }   // drag-limited beyond t_max
      }
      // thermally activated forward / backward terms: exp_arg = -c_e q_func(1 - p_func(t))  (mts_dG), p == 1 and q == 1 are uniform cases
      double eaf[KW], dff[KW];
      bool inwin[KW], any_win = false, any_tail = false;
#pragma unroll
      for (int a = 0; a < KW; a++) {
         mts_dG<PQ1>(mp, kv.c_e, (at[a] - gAth) * g_i, eaf[a], dff[a]);
         inwin[a] = live[a] && !over[a] && !(eaf[a] < LN_GAM_RATIO_MIN);
         any_win = any_win || inwin[a];
         any_tail = any_tail || (inwin[a] && at0[a] > mp.t_min);
      }
      if (any_win) {
         double eab[KW], dfb[KW], ef[KW], eb[KW], pw[KW];
         // backward jumps.  With p == q == 1 the exponents are eaf = -c_e (1 - t), eab = -c_e (1 + t_b), t_b >= 0, and dff == dfb == c_e; an
         // in-window system has eaf >= ln(1e-60) = -138.2, so for c_e > 180 the backward term is below e^-41.8 = 7e-19 < 2^-54 of the forward
         // one in both gw = gam_w (ef - eb) and its derivative: dropping it leaves every bit as it was (the shipped material sets have
         // c_e ~ 300).  In general (and always for exp(x) == 0, x < -745.2) the call is skipped when no lane of the wave needs it.
         const bool pq1 = ECM_KM_SKIP_BACK && (PQ1 || ECM_EXP_PQ1 || (mp.p == 1.0 && mp.q == 1.0));
         bool any_b = false;
         if (pq1) {
#pragma unroll
            for (int a = 0; a < KW; a++) { eab[a] = -800.0; dfb[a] = 0.0; any_b = any_b || (inwin[a] && !(kv.c_e > 180.0)); }
         } else any_b = true;
         if (any_b) {
#pragma unroll
            for (int a = 0; a < KW; a++) mts_dG<PQ1>(mp, kv.c_e, (-at[a] - gAth) * g_i, eab[a], dfb[a]);
            any_b = (KW != 1);     // (the grouped form keeps the unconditional call: its exp chains interleave with the forward ones)
#pragma unroll
            for (int a = 0; a < KW; a++) any_b = any_b || (inwin[a] && eab[a] > -746.0);
         }
#pragma unroll
         for (int a = 0; a < KW; a++) ef[a] = eaf[a];
         exp_n<KW, PQ1, SC>(ef);
         if (any_b) {
#pragma unroll
            for (int a = 0; a < KW; a++) eb[a] = eab[a];
            exp_n<KW, PQ1, SC>(eb);
         } else {
#pragma unroll
            for (int a = 0; a < KW; a++) eb[a] = 0.0;
         }
#pragma unroll
         for (int a = 0; a < KW; a++) pw[a] = 0.0;
         if (any_tail) {   // power-law tail: only above t_min = (1e-60)^m (rare for large 1/m)
            if constexpr (PQ1 && ECM_KM_LOG_NEAR1) {
               // p == q == 1 instantiation: the host selects it only when the tail's window (t_min, t_max] lies in [0.75, 1.25] and 1/m is not an
               // integer (model_kernels.hip, km_pq1), so the short-series logarithm is the only form compiled in - the general log() brought
               // eight more coefficients into the register file of the whole Newton loop
#pragma unroll
               for (int a = 0; a < KW; a++) pw[a] = mp.xn * log_near1(at0[a]);
               exp_n<KW, PQ1, SC>(pw);
            } else if (mp.xn_int > 0) {
               double b[KW];
#pragma unroll
               for (int a = 0; a < KW; a++) { pw[a] = 1.0; b[a] = at0[a]; }
               for (int e = mp.xn_int;;) {
                  if (e & 1) {
#pragma unroll
                     for (int a = 0; a < KW; a++) pw[a] *= b[a];
                  }
                  e >>= 1;
                  if (!e) break;
#pragma unroll
                  for (int a = 0; a < KW; a++) b[a] *= b[a];
               }
            } else if (ECM_KM_LOG_NEAR1 && mp.t_min >= 0.75 && mp.t_max <= 1.25) {
               // the tail is only used for t_min < at0 <= t_max, and with 1/m = 2 c_e p q of a few hundred both bounds are close to 1
               // (1e-60^m, 1e45^m): logarithm by the short series (other lanes compute a finite value that is not used)
#pragma unroll
               for (int a = 0; a < KW; a++) pw[a] = mp.xn * log_near1(at0[a]);
               exp_n<KW, PQ1, SC>(pw);
            } else {
#pragma unroll
               for (int a = 0; a < KW; a++) pw[a] = mp.xn * log(fmax(at0[a], 1.0e-300));
               exp_n<KW, PQ1, SC>(pw);
            }
         }
#pragma unroll
         for (int a = 0; a < KW; a++) {
            double gw = kv.gam_w * (ef[a] - eb[a]);
            double dgw = kv.gam_w * (ef[a] * dff[a] + eb[a] * dfb[a]) * g_i;
            const bool tail = at0[a] > mp.t_min;
            const double temp = (kv.gam_w * 10.0) * pw[a];
            gw += tail ? temp * at0[a] : 0.0;
            dgw += tail ? temp * mp.xnn * g_i : 0.0;
            const bool valid = inwin[a] && (gw > 0.0);
#if ECM_KM_ONE_RCP
            // series combination of the two branches 1 / (1/gw + 1/gr) = gw gr / (gw + gr) with one reciprocal (lanes that are not `valid` may hold junk here)
            const double R = frcp(gw + gr[a]);
            const double gd = (gw * gr[a]) * R;
            if (valid) { gdot[a] = copysign(gd, tau[a]); if (WITHD) dg[a] = (dgw * (gr[a] * gr[a]) + dgr[a] * (gw * gw)) * (R * R); }
#else
            const double r1 = frcp(gw), r2 = frcp(gr[a]);   // series combination of the two branches; lanes that are not `valid` may hold junk here
            const double gd = frcp(r1 + r2);
            if (valid) { gdot[a] = copysign(gd, tau[a]); if (WITHD) dg[a] = gd * gd * (dgw * r1 * r1 + dgr[a] * r2 * r2); }
#endif
         }
      }
   }
}

template <int KIN>
ECM_DI void kin_sdot(const MatParams& mp, double h, double shrate, double ev1, double& sdot, double& dsdot) {
   if (kin_is_km(KIN)) {
      const double t1 = exp(-0.5 * h);
      sdot = (mp.k1 * t1 - ev1) * shrate; dsdot = (-0.5 * mp.k1 * t1) * shrate;
   } else if (kin_base(KIN) == KIN_VOCE_NL) {
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
   if (kin_is_km(KIN)) {
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
   return kin_is_km(KIN) ? exp(h_n) : h_n;
}

// Register budget.  Two waves per SIMD need <= 256 VGPRs; the naive point update wants ~370.  What is not touched inside the
// slip-system loop is therefore parked outside the register file:
//   * per-thread LDS stash (slot s of thread t at stash[s * ECM_STASH_STRIDE + t], conflict-free), 38 doubles so that two 256-thread
//     blocks fit the 160 KB of a CU: the vectors an evaluation reads once (e_n, d_n, w_n), the restore copy of x, and 17 of the 20
//     values only needed after the local solve (D', old stress, quaternion, volumes, energy);
//   * the last 3 of those (dEff, bulk modulus, hardness) sit in the point's own tangent slot in global memory, which is written last.
// The rotation data of an evaluation (Tr, d_lat, w_lat) is never live across one and stays in registers (struct Jac).  Rule measured
// on MI355X: inside a thread's lifetime nothing written to global memory is still in L2 when it is read back, and a reload waits for
// every earlier store of the wave (vmcnt), so anything that must survive the Newton loop belongs in LDS, not in global memory.
#ifndef ECM_KM_GDOT_AT_END
#define ECM_KM_GDOT_AT_END 1   // Kocks-Mecking: slip rates written once from the converged point (A/B switch)
#endif
#ifndef ECM_DEFER_REJECT
#define ECM_DEFER_REJECT 1   // capped launch with resumed tail points: a rejected trial hands the point over instead of re-evaluating (A/B switch;
                             // 2 = also in instantiations that keep the dog-leg data: FCC 20.1 ms against 17.2 / 18.0)
#endif
#ifndef ECM_KEEP_DOGLEG
#define ECM_KEEP_DOGLEG 0   // Kocks-Mecking without athermal threshold: dog-leg data kept across a trial evaluation instead of a re-evaluation after a
                            // rejection.  Worth 14 % (FCC 29.7 -> 25.6 ms) until ECM_DEFER_REJECT made the capped launch free of re-evaluations
                            // without the 20 carried values (18.0 -> 17.2 ms); still the better choice for a launch WITHOUT tail split (A/B switch)
#endif
#ifndef ECM_KM_BATCH
#define ECM_KM_BATCH 1   // p == q == 1 FCC Kocks-Mecking instantiation: batched straight-line slip loop (A/B switch)
#endif
#ifndef ECM_KB
#define ECM_KB 4         // systems per batch of that form (3, 4, 6 within 1.5 % of each other; 12: +70 %)
#endif
#ifndef ECM_KM_FORMS_CSE
#define ECM_KM_FORMS_CSE 2   // athermal-threshold Kocks-Mecking kernel: cheap classes through the factored slip forms: 0 never, 1 always, 2 in the
                             // p == q == 1 instantiation only.  Measured at 128^3: general instantiation 12.7 ms against 12.3 (448 B of scratch
                             // instead of 320: spill latency), p == q == 1 instantiation 9.6 ms against 10.4 (profiles/r03_kernel_experiments.txt)
#endif
#ifndef ECM_KM_PEND_INSERT
#define ECM_KM_PEND_INSERT 1   // athermal-threshold Kocks-Mecking kernel, factored-forms path: window systems inserted into the per-system arrays (eval_rj; A/B switch)
#endif
#ifndef ECM_TANGENT_WX
#define ECM_TANGENT_WX 1   // tangent block as (Q5 Kt)(S^-1 Q5^T) with the rotated coupling through the equivariance of M35 (point_update; A/B switch)
#endif
#ifndef ECM_TANGENT_FIRST
#define ECM_TANGENT_FIRST 1   // epilogue order: tangent before the state / stress outputs (see point_update)
#endif
#ifndef ECM_DEFER_DIS
#define ECM_DEFER_DIS 1   // Voce: dissipation / effective shear rate from the converged point only (voce_slip_rates)
#endif
#ifndef ECM_SWEEP_UNROLL
#define ECM_SWEEP_UNROLL 0   // block Gauss-Seidel sweeps of the Newton step rolled (1: unrolled; A/B on MI355X)
#endif
#ifndef ECM_STASH_STRIDE
#define ECM_STASH_STRIDE 64    // lanes per stash region: every WAVE owns ST_SLOTS x 64 contiguous doubles (slot s of lane l at region[s * 64 + l]), the regions of a
                               // block's waves one behind the other (model_kernel.hpp, PointIO::stash).  All slots are within the reach of a ds_read / ds_write
                               // immediate offset from ONE address register (rounds 3-5: stride = block size 128, slots interleaved across the two waves; with
                               // 256 the slots from 32 up needed their own address registers, which the allocator spilled).  Per-wave regions are what lets the
                               // staged AOS launch use the wave's region as a transposition buffer for its 64 contiguous point rows (round 6); 128 = the
                               // interleaved form of rounds 3-5 (A/B switch; no staged launch then)
#endif
#ifndef ECM_EPI_NO_LOADS
#define ECM_EPI_NO_LOADS 1   // no global or scratch load behind the first output store: on gfx9 loads and stores share the in-order vmcnt counter, so a load
                             // issued after the record / state stores waits until every one of them has reached memory (a store-queue drain of a few
                             // microseconds, twice per wave in the round-3 kernel).  The two begin-of-step values the outputs need are read with the other
                             // inputs and parked in the stash, and the output addresses are re-derived from the thread index after the local solve
                             // (PointIO::refresh) instead of being carried - and spilled - through it (A/B switch)
#endif
constexpr int ST_EN = 0, ST_DN = 5, ST_WN = 10, ST_XS = 13, ST_CD = 21, ST_PB = 33, ST_SLOTS = 38;   // ST_XS: restore copy of the unknowns; ST_CD: parking slots; ST_PB: rarely used scalars of the point problem
constexpr int PB_SCI = 0, PB_ESCI = 1, PB_DETVRI = 2;   // 1/sc, 1/esc, 1/detV: read once per Newton step / in the epilogue only
constexpr int PB_SHR0 = 3, PB_FLOW0 = 4;               // begin-of-step accumulated shear / plastic work: only the outputs need them (see ECM_EPI_NO_LOADS)
static_assert(ST_PB + PB_FLOW0 < ST_SLOTS, "stash slots");
constexpr int ST_NCD = ST_PB - ST_CD;
// Staged AOS launch (point_update<.., STG>): its tangent rows only cover slots 0 .. 17 of the wave's region, so what the state / stress outputs need after the
// tangent waits in slots that are dead by then - the tail of the restore copy of x, the begin-of-step quaternion (read), the record scale (REC only), 1/sc
// (Newton loop only), the bulk modulus (read) - instead of in registers across the tangent arithmetic (the allocator spilled 13 of them to scratch, and a
// scratch reload behind the row stores waits for the store queue of the wave)
// (1/detV is in a register by then as well.  Slot 18 stays free: padded tangent rows - 19 doubles, bank-conflict free - reach into it)
constexpr int ST_EPI_E = ST_XS + 6, ST_EPI_Q0 = ST_EPI_E + 5, ST_EPI_Q1 = 31, ST_EPI_Q2 = 33, ST_EPI_Q3 = 35, ST_EPI_WRK = 29;
static_assert(ST_EPI_E == 19 && ST_EPI_Q0 == ST_CD + 3, "epilogue parking slots: rows of the tangent halves end in slot 18; e_f in 19..23 (x tail, q_n[0..2]), q[0] in 24 (q_n[3])");
// (the deviatoric stress work of the step needs D' and the old stress only through  sum (s_old + s_new) . D' = s_old . D' + s_lat . d_lat:
//  the first scalar is parked, the second uses the lattice-frame D' of the converged evaluation - 10 slots fewer than parking both vectors)
constexpr int CD_QN = 0, CD_VOLD = 4, CD_VNEW = 5, CD_ENEW = 6, CD_DEFF = 7, CD_BULK = 8, CD_HU = 9, CD_TSC = 10, CD_WRKOLD = 11;
static_assert(CD_WRKOLD < ST_NCD, "every parked value lives in the LDS stash");
#define ECM_ST(p, slot) (p)[(slot) * ECM_STASH_STRIDE]
// compiler-only barrier: what was parked must be re-loaded later instead of being kept alive in registers
#define ECM_PARK_BARRIER() asm volatile("" ::: "memory")
// parking slot c (compile-time) of the LDS stash (every parked value lives there: static_assert above)
#define ECM_CD(c) ECM_ST(st, ST_CD + (c))

// ------------------------------------------------------------------------------------------------------------
// the point problem: unknowns x = (delta e / E_SCALE, xi / R_SCALE).  e is the library's strain STATE = a_V * E with a_V = detV^(1/3)
// (end of step) and E the lattice-frame deviatoric elastic strain of the elastic law; the stash holds E_n = e_n / a_V, so that inside the
// solve everything is in terms of E and x only needs the factor esc = E_SCALE / a_V (oracle/ecmech_port.hpp, struct Problem)
// ------------------------------------------------------------------------------------------------------------
struct Prob {
   double dt_ri, sc, g_i;                  // sc = epsdot_scale_inv, g_i = 1/g   (1/sc, 1/esc, 1/detV: stash slots ST_PB + PB_*)
   double esc;                             // E_SCALE / a_V
   double* st;                             // per-thread stash
   const double* pqt;                      // Kocks-Mecking: slip table in LDS (12 rows of 8), nullptr -> PQ_TAB in global memory
   int gs;                                 // stride of the slip-rate outputs (1 or 64, see point_update's QS)
   KinVals kv;
};

// Jacobian (un-scaled):
//   Jee = I/dt + A Kd      A = P G P^T (packed symmetric 15); after jac_factor the same 15 slots hold the LDL^T factor of
//                          M = diag(1/(kd dt)) + A  (Jee = M Kd)
//   Jre = B Kd             B = Q G P^T
//   Jer = -M35(d_lat) Tr,  Jrr = I/dt - hat(w_lat) Tr      with Tr, d_lat, w_lat of the last evaluation in the stash
struct Jac { double A[15], B[3][5]; double Tr[9], dl[5], wl[3]; };   // + rotation data of the evaluation (never live across one)

ECM_DI constexpr int sidx(int i, int j) { return i <= j ? (i * (11 - i)) / 2 + (j - i) : (j * (11 - j)) / 2 + (i - j); }   // 5x5 symmetric packing

// tau = P^T k, D^p / W^p sums and the Jacobian blocks A = P G P^T, B = Q G P^T over the integer slip tables, with the partial sums the
// 12 systems share factored out: 104 instead of 320 additions per evaluation (generated: scripts/gen_slip_forms.py)
#ifndef ECM_SLIP_FORMS_CSE
#define ECM_SLIP_FORMS_CSE 1
#endif
#include "slip_forms_gen.hpp"

// One evaluation of residual (+ Jacobian).  gdot_out: nullable pointer (global memory) receiving the 12 slip rates.
template <int KIN, bool WITHJ>
ECM_DI bool eval_rj(const MatParams& mp, const Prob& pb, const double x[8], double r[8], Jac& jac,
                    double* __restrict__ gdot_out, double& dis_rate, double& shrate) {
   // resolved shear stress from the Kirchhoff stress K e'
   double k[5], e_f[5];
#pragma unroll
   for (int i = 0; i < 5; i++) e_f[i] = ECM_ST(pb.st, ST_EN + i) + x[i] * pb.esc;
   k[0] = mp.kd0 * e_f[0]; k[1] = mp.kd0 * e_f[1]; k[2] = mp.kd2 * e_f[2]; k[3] = mp.kd2 * e_f[3]; k[4] = mp.kd2 * e_f[4];   // (Voce: unused, see ks)
   const double g_i = pb.g_i;
   double dis = 0.0, shr = 0.0;
   double dp[5] = { 0, 0, 0, 0, 0 }, wp[3] = { 0, 0, 0 };
   bool ok = true;
   if constexpr (!kin_is_km(KIN)) {
      // fully unrolled, integer-coefficient form (see SP/SQ): resolved shear stresses, batched kinetics, then the Jacobian blocks
      const double ks[5] = { mp.pk0 * e_f[0], mp.pk1 * e_f[1], mp.pk2 * e_f[2], mp.pk2 * e_f[3], mp.pk2 * e_f[4] };   // PSC[c] * Kirchhoff component c
      double tau[NSLIP], gd[NSLIP], dg[NSLIP];
      if (ECM_SLIP_FORMS_CSE) slip_tau12(ks, tau);
      else {
#pragma unroll
      for (int a = 0; a < NSLIP; a++) {
         double t = 0.0;
#pragma unroll
         for (int c = 0; c < 5; c++) if (SP[c][a] != 0) t += (double)SP[c][a] * ks[c];
         tau[a] = t;
      }
      }
      voce_gdot12<WITHJ, false, kin_xn_ct(KIN)>(mp, g_i, tau, gd, dg);
      // the dissipation rate is an output of the converged point only: voce_slip_rates computes it there (12 FMAs fewer per evaluation)
      if (!ECM_DEFER_DIS) {
#pragma unroll
         for (int a = 0; a < NSLIP; a++) { dis += tau[a] * gd[a]; shr += fabs(gd[a]); }
         ok = isfinite(shr);   // any non-finite rate poisons the sum
      }   // deferred: the caller tests the residual norm for finiteness (a non-finite rate poisons D^p and with it the residual)
      if (ECM_SLIP_FORMS_CSE) {
         double dps[5], wps[3];
         slip_dpwp(gd, dps, wps);
#pragma unroll
         for (int c = 0; c < 5; c++) dp[c] = PSC[c] * dps[c];
#pragma unroll
         for (int c = 0; c < 3; c++) wp[c] = PB * wps[c];
         if (WITHJ) slip_jac_blocks(dg, jac.A, jac.B);
      } else {
#pragma unroll
      for (int c = 0; c < 5; c++) {
         double t = 0.0;
#pragma unroll
         for (int a = 0; a < NSLIP; a++) if (SP[c][a] != 0) t += (double)SP[c][a] * gd[a];
         dp[c] = PSC[c] * t;
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
         double t = 0.0;
#pragma unroll
         for (int a = 0; a < NSLIP; a++) t += (double)SQ[c][a] * gd[a];
         wp[c] = PB * t;
      }
      if (WITHJ) {
#pragma unroll
         for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = i; j < 5; j++) {
               double t = 0.0;
#pragma unroll
               for (int a = 0; a < NSLIP; a++) if (SP[i][a] * SP[j][a] != 0) t += (double)(SP[i][a] * SP[j][a]) * dg[a];
               jac.A[sidx(i, j)] = (PSC[i] * PSC[j]) * t;
            }
#pragma unroll
         for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 5; j++) {
               double t = 0.0;
#pragma unroll
               for (int a = 0; a < NSLIP; a++) if (SQ[i][a] * SP[j][a] != 0) t += (double)(SQ[i][a] * SP[j][a]) * dg[a];
               jac.B[i][j] = (PB * PSC[j]) * t;
            }
      }
      }
   } else {
   if (WITHJ) {
#pragma unroll
      for (int i = 0; i < 15; i++) jac.A[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
         for (int j = 0; j < 5; j++) jac.B[i][j] = 0.0;
   }
   // (the runtime flag is always set in the KIN_KMBALD_GA instantiation; with the test compiled away the register allocator of ROCm 7.2
   //  spills 850 B/lane instead of 330 and the kernel runs 2.5x slower, so the never-taken alternative stays in that instantiation)
   if (ECM_KM_DEFER && kin_base(KIN) == KIN_KMBALD_GA && mp.with_g_athermal) {
      // Athermal-threshold variant (BCC): a system is dormant (|tau| <= g), drag-limited (at_0 > t_max: one exp) or - for |tau| - g inside
      // the narrow thermally activated window (0, t_max tau_a] - needs the full balanced kinetics (four more exp/log).  About 1 % of
      // the (point, system) pairs are in the window, but a wave of 64 points x KW systems nearly always holds one, so the grouped form
      // below pays the window phases in almost every group.  Here the groups only do the cheap classes and note the window systems in a
      // per-lane bit mask; a second loop then treats ONE pending system per lane and pass (table row by lane-varying index), which
      // ends after max-over-lanes(pending) ~ 1-2 passes.  Same arithmetic per system; only the order of the 12 contributions to the
      // sums changes (round-off).
      const double g_ia = 1.0 / mp.tau_a, gAth = pb.kv.g, wi = 1.0 / mp.wrD;
      unsigned pend = 0;
      if constexpr (ECM_KM_FORMS_CSE == 1 || (ECM_KM_FORMS_CSE == 2 && kin_pq1(KIN))) {
      // cheap classes of all 12 systems with static indices: resolved shear stresses through the shared partial sums (slip_tau12), one exp
      // per drag-limited system, then D^p / W^p and the Jacobian blocks of these systems through the factored forms (slip_dpwp,
      // slip_jac_blocks: 114 instead of ~540 multiply-adds per evaluation); the window systems are added by the pending loop below
      {
         double ks[5];
#pragma unroll
         for (int c = 0; c < 5; c++) ks[c] = PSC[c] * k[c];
         double tau[NSLIP], gd[NSLIP], dg[NSLIP], xr[NSLIP];
         slip_tau12(ks, tau);
         bool any_drag = false;
#pragma unroll
         for (int a = 0; a < NSLIP; a++) {
            const double at = fabs(tau[a]);
            xr[a] = (at - gAth) * wi;
            const bool live = (tau[a] != 0.0) && (xr[a] > 0.0), over = fmax(0.0, at - gAth) * g_ia > mp.t_max;
            gd[a] = 0.0; dg[a] = 0.0;
            if (live && !over) pend |= 1u << a;
            // xr <- -1 marks "not drag-limited" for the second pass
            if (!(live && over)) xr[a] = -1.0; else any_drag = true;
         }
         if (any_drag) {
            double exv[NSLIP];
#pragma unroll
            for (int a = 0; a < NSLIP; a++) exv[a] = -fmax(xr[a], 0.0);
            exp_n<NSLIP, kin_pq1(KIN), kin_sc_exp(KIN)>(exv);
#pragma unroll
            for (int a = 0; a < NSLIP; a++) {
               const double ex = exv[a];
               const bool small = xr[a] < EPS_SQRT;
               const double gr = small ? pb.kv.gam_r * xr[a] : pb.kv.gam_r * (1.0 - ex);
               const double dgr = (small ? pb.kv.gam_r : pb.kv.gam_r * ex) * wi;
               if (xr[a] >= 0.0) { gd[a] = copysign(gr, tau[a]); dg[a] = dgr; }
            }
         }
         if (ECM_KM_PEND_INSERT) {
            // window systems, one per lane and pass: the rate and its derivative are INSERTED into gd[] / dg[] (compare-and-select over the 12
            // static slots) and the factored forms below then run once over all 12 systems.  The loop no longer carries the 38 accumulators of
            // D^p, W^p and the two Jacobian blocks plus a table row (they were what the allocator spilled inside the Newton loop), only the 24
            // per-system values
            while (__ballot(pend != 0) != 0ull) {
               if (pend != 0) {
                  const int a = __ffs((int)pend) - 1; pend &= pend - 1;
                  double t1 = 0.0;
#pragma unroll
                  for (int s2 = 0; s2 < NSLIP; s2++) t1 = (a == s2) ? tau[s2] : t1;
                  double tau1[1] = { t1 }, gd1[1], dg1[1] = { 0.0 };
                  kmbald_gdot4<WITHJ, 1, kin_pq1(KIN), kin_sc_exp(KIN)>(mp, pb.kv, tau1, gd1, dg1);
#pragma unroll
                  for (int s2 = 0; s2 < NSLIP; s2++) { gd[s2] = (a == s2) ? gd1[0] : gd[s2]; if (WITHJ) dg[s2] = (a == s2) ? dg1[0] : dg[s2]; }
               }
            }
         }
#pragma unroll
         for (int a = 0; a < NSLIP; a++) {
            if (gdot_out) gdot_out[a * pb.gs] = gd[a];
            dis += tau[a] * gd[a]; shr += fabs(gd[a]);
         }
         double dps[5], wps[3];
         slip_dpwp(gd, dps, wps);
#pragma unroll
         for (int c = 0; c < 5; c++) dp[c] = PSC[c] * dps[c];
#pragma unroll
         for (int c = 0; c < 3; c++) wp[c] = PB * wps[c];
         if (WITHJ) slip_jac_blocks(dg, jac.A, jac.B);
      }
      } else {
#pragma unroll 1
      for (int a0 = 0; a0 < NSLIP; a0 += KD) {
         double pq[KD][8], tau[KD], gd[KD], dg[KD], xr[KD];
         bool live[KD], over[KD], any_live = false;
#pragma unroll
         for (int a = 0; a < KD; a++) {
#pragma unroll
            for (int c = 0; c < 8; c++) pq[a][c] = PQ_TAB[a0 + a][c];
            tau[a] = pq[a][0] * k[0] + pq[a][1] * k[1] + pq[a][2] * k[2] + pq[a][3] * k[3] + pq[a][4] * k[4];
            const double at = fabs(tau[a]);
            xr[a] = (at - gAth) * wi;
            live[a] = (tau[a] != 0.0) && (xr[a] > 0.0); over[a] = fmax(0.0, at - gAth) * g_ia > mp.t_max;
            any_live = any_live || live[a];
            gd[a] = 0.0; dg[a] = 0.0;
            if (live[a] && !over[a]) pend |= 1u << (a0 + a);
         }
         if (any_live) {
            double ex[KD];
#pragma unroll
            for (int a = 0; a < KD; a++) ex[a] = -fmax(xr[a], 0.0);
            exp_n<KD, kin_pq1(KIN), kin_sc_exp(KIN)>(ex);
#pragma unroll
            for (int a = 0; a < KD; a++) {
               const bool small = xr[a] < EPS_SQRT;
               const double gr = small ? pb.kv.gam_r * xr[a] : pb.kv.gam_r * (1.0 - ex[a]);
               const double dgr = (small ? pb.kv.gam_r : pb.kv.gam_r * ex[a]) * wi;
               if (live[a] && over[a]) { gd[a] = copysign(gr, tau[a]); dg[a] = dgr; }
            }
         }
#pragma unroll
         for (int a = 0; a < KD; a++) {
            if (gdot_out) gdot_out[(a0 + a) * pb.gs] = gd[a];
            dis += tau[a] * gd[a]; shr += fabs(gd[a]);
#pragma unroll
            for (int c = 0; c < 5; c++) dp[c] += pq[a][c] * gd[a];
#pragma unroll
            for (int c = 0; c < 3; c++) wp[c] += pq[a][5 + c] * gd[a];
            if (WITHJ) {
               double gp[5];
#pragma unroll
               for (int c = 0; c < 5; c++) gp[c] = dg[a] * pq[a][c];
#pragma unroll
               for (int i = 0; i < 5; i++)
#pragma unroll
                  for (int j = i; j < 5; j++) jac.A[sidx(i, j)] += pq[a][i] * gp[j];
#pragma unroll
               for (int i = 0; i < 3; i++)
#pragma unroll
                  for (int j = 0; j < 5; j++) jac.B[i][j] += pq[a][5 + i] * gp[j];
            }
         }
      }
      }
      constexpr bool PEND_DONE = ECM_KM_PEND_INSERT && (ECM_KM_FORMS_CSE == 1 || (ECM_KM_FORMS_CSE == 2 && kin_pq1(KIN)));   // handled above
      if constexpr (!PEND_DONE)
      while (__ballot(pend != 0) != 0ull) {      // one pending window system per lane and pass
         if (pend != 0) {
            const int a = __ffs((int)pend) - 1; pend &= pend - 1;
            double pq[8];
#pragma unroll
            for (int c = 0; c < 8; c++) pq[c] = pb.pqt ? pb.pqt[8 * a + c] : PQ_TAB[a][c];      // lane-varying row: LDS copy of the 768-byte table
            double tau1[1] = { pq[0] * k[0] + pq[1] * k[1] + pq[2] * k[2] + pq[3] * k[3] + pq[4] * k[4] }, gd1[1], dg1[1];
            kmbald_gdot4<WITHJ, 1, kin_pq1(KIN), kin_sc_exp(KIN)>(mp, pb.kv, tau1, gd1, dg1);
            if (gdot_out) gdot_out[a * pb.gs] = gd1[0];
            dis += tau1[0] * gd1[0]; shr += fabs(gd1[0]);
#pragma unroll
            for (int c = 0; c < 5; c++) dp[c] += pq[c] * gd1[0];
#pragma unroll
            for (int c = 0; c < 3; c++) wp[c] += pq[5 + c] * gd1[0];
            if (WITHJ) {
               double gp[5];
#pragma unroll
               for (int c = 0; c < 5; c++) gp[c] = dg1[0] * pq[c];
#pragma unroll
               for (int i = 0; i < 5; i++)
#pragma unroll
                  for (int j = i; j < 5; j++) jac.A[sidx(i, j)] += pq[i] * gp[j];
#pragma unroll
               for (int i = 0; i < 3; i++)
#pragma unroll
                  for (int j = 0; j < 5; j++) jac.B[i][j] += pq[5 + i] * gp[j];
            }
         }
      }
      ok = isfinite(shr);
   } else if constexpr (ECM_KM_BATCH && kin_pq1(KIN) && kin_base(KIN) == KIN_KMBALD) {
      // p == q == 1 instantiation without an athermal threshold (FCC): with the pow() alternatives compiled out the kinetics of a system is
      // a short straight-line sequence (three exp), so the 12 systems go through it in batches of ECM_KB with static indices like the Voce
      // form: resolved shear stresses, D^p / W^p and the Jacobian blocks through the factored slip forms (slip_forms_gen.hpp: 104 instead of
      // 620 multiply-adds per evaluation), ECM_KB independent exp chains in flight per phase.  Same arithmetic per system as the rolled
      // group loop below; the order of the 12 contributions to the sums changes (round-off).
      double ks[5];
#pragma unroll
      for (int c = 0; c < 5; c++) ks[c] = PSC[c] * k[c];
      double tau[NSLIP], gd[NSLIP], dg[NSLIP];
      slip_tau12(ks, tau);
#pragma unroll
      for (int a0 = 0; a0 < NSLIP; a0 += ECM_KB) kmbald_gdot4<WITHJ, ECM_KB, true, kin_sc_exp(KIN)>(mp, pb.kv, tau + a0, gd + a0, dg + a0);
#pragma unroll
      for (int a = 0; a < NSLIP; a++) {
         if (gdot_out) gdot_out[a * pb.gs] = gd[a];
         dis += tau[a] * gd[a]; shr += fabs(gd[a]);
      }
      double dps[5], wps[3];
      slip_dpwp(gd, dps, wps);
#pragma unroll
      for (int c = 0; c < 5; c++) dp[c] = PSC[c] * dps[c];
#pragma unroll
      for (int c = 0; c < 3; c++) wp[c] = PB * wps[c];
      if (WITHJ) slip_jac_blocks(dg, jac.A, jac.B);
      ok = isfinite(shr);
   } else {
#pragma unroll 1
      for (int a0 = 0; a0 < NSLIP; a0 += KW) {   // rolled over groups: the table rows of a group come in through scalar loads
         double pq[KW][8], tau[KW], gd[KW], dg[KW];
#pragma unroll
         for (int a = 0; a < KW; a++) {
#pragma unroll
            for (int c = 0; c < 8; c++) pq[a][c] = PQ_TAB[a0 + a][c];
            tau[a] = pq[a][0] * k[0] + pq[a][1] * k[1] + pq[a][2] * k[2] + pq[a][3] * k[3] + pq[a][4] * k[4];
         }
         kmbald_gdot4<WITHJ, ECM_KW, kin_pq1(KIN), kin_sc_exp(KIN)>(mp, pb.kv, tau, gd, dg);
#pragma unroll
         for (int a = 0; a < KW; a++) {
            if (gdot_out) gdot_out[(a0 + a) * pb.gs] = gd[a];
            dis += tau[a] * gd[a]; shr += fabs(gd[a]);
#pragma unroll
            for (int c = 0; c < 5; c++) dp[c] += pq[a][c] * gd[a];
#pragma unroll
            for (int c = 0; c < 3; c++) wp[c] += pq[a][5 + c] * gd[a];
            if (WITHJ) {
               double gp[5];
#pragma unroll
               for (int c = 0; c < 5; c++) gp[c] = dg[a] * pq[a][c];
#pragma unroll
               for (int i = 0; i < 5; i++)
#pragma unroll
                  for (int j = i; j < 5; j++) jac.A[sidx(i, j)] += pq[a][i] * gp[j];
#pragma unroll
               for (int i = 0; i < 3; i++)
#pragma unroll
                  for (int j = 0; j < 5; j++) jac.B[i][j] += pq[a][5 + i] * gp[j];
            }
         }
      }
      ok = isfinite(shr);
   }
   }
   dis_rate = dis; shrate = shr;   // (un-scaled: the caller divides the converged value by detV)
   if (WITHJ && mp.qsign < 0.0) {
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
         for (int j = 0; j < 5; j++) jac.B[i][j] = -jac.B[i][j];
   }
   // rotation part after the slip loop so that nothing of it is live across the loop
   double xi[3];
#pragma unroll
   for (int i = 0; i < 3; i++) xi[i] = x[5 + i] * R_SCALE;
   double A[9], Tr[9]; exp_map(xi, A, Tr);
   if (WITHJ) {
#pragma unroll
      for (int c = 0; c < 9; c++) jac.Tr[c] = Tr[c];
   }
   {
      double dn[5], d_lat[5];
#pragma unroll
      for (int i = 0; i < 5; i++) dn[i] = ECM_ST(pb.st, ST_DN + i);
      rot_vecd_T(A, dn, d_lat);
#pragma unroll
      for (int c = 0; c < 5; c++) {
         r[c] = x[c] * (pb.esc * pb.dt_ri) + dp[c] - d_lat[c];   // UN-scaled residual: the caller carries the scale sc (SNLS scales by epsdot_scale_inv)
         if (WITHJ) jac.dl[c] = d_lat[c];
      }
   }
   {
      const double w0 = ECM_ST(pb.st, ST_WN), w1 = ECM_ST(pb.st, ST_WN + 1), w2 = ECM_ST(pb.st, ST_WN + 2);
#pragma unroll
      for (int c = 0; c < 3; c++) {
         const double w_lat = A[c] * w0 + A[3 + c] * w1 + A[6 + c] * w2;
         r[5 + c] = xi[c] * pb.dt_ri + mp.qsign * wp[c] - w_lat;
         if (WITHJ) jac.wl[c] = w_lat;
      }
   }
   ECM_PARK_BARRIER();
   return ok;
}

// slip rates at the converged point (Voce family): written once, instead of one global store per system and evaluation
// LO: gdot_out is the lane's row in the LDS stage (see ost); the stash may then no longer be read, 1/detV comes in through detv_lo
template <int XNCT, bool LO = false>
ECM_DI void voce_slip_rates(const MatParams& mp, const Prob& pb, const double e_f[5], double* __restrict__ gdot_out, double& dis_rate, double& shrate, const double detv_lo = 0.0) {
   const double ks[5] = { mp.pk0 * e_f[0], mp.pk1 * e_f[1], mp.pk2 * e_f[2], mp.pk2 * e_f[3], mp.pk2 * e_f[4] };
   double tau[NSLIP], gd[NSLIP];
   if (ECM_SLIP_FORMS_CSE) slip_tau12(ks, tau);
   else {
#pragma unroll
   for (int a = 0; a < NSLIP; a++) {
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < 5; c++) if (SP[c][a] != 0) t += (double)SP[c][a] * ks[c];
      tau[a] = t;
   }
   }
   voce_gdot12<false, true, XNCT>(mp, pb.g_i, tau, gd, nullptr);
   double dis = 0.0, shr = 0.0;
#pragma unroll
   for (int a = 0; a < NSLIP; a++) { ost<LO>(&gdot_out[a * pb.gs], gd[a]); dis += tau[a] * gd[a]; shr += fabs(gd[a]); }
   if (ECM_DEFER_DIS) { dis_rate = dis * (LO ? detv_lo : ECM_ST(pb.st, ST_PB + PB_DETVRI)); shrate = shr; }   // (rates below t_min = (1e-60)^m count as 0 here: below 1e-60 of the reference rate)
}

// slip rates at the converged point (Kocks-Mecking family, ECM_KM_GDOT_AT_END): one more pass through the kinetics (no derivatives) instead
// of 12 global stores per evaluation - on gfx9 every scratch reload of the Newton loop otherwise waits for those stores (vmcnt)
template <bool PQ1, bool SC, bool LO = false>
ECM_DI void km_slip_rates(const MatParams& mp, const Prob& pb, const double e_f[5], double* __restrict__ gdot_out) {
   const double k[5] = { mp.kd0 * e_f[0], mp.kd0 * e_f[1], mp.kd2 * e_f[2], mp.kd2 * e_f[3], mp.kd2 * e_f[4] };
#pragma unroll 1
   for (int a0 = 0; a0 < NSLIP; a0 += KW) {
      double tau[KW], gd[KW];
#pragma unroll
      for (int a = 0; a < KW; a++) tau[a] = PQ_TAB[a0 + a][0] * k[0] + PQ_TAB[a0 + a][1] * k[1] + PQ_TAB[a0 + a][2] * k[2] + PQ_TAB[a0 + a][3] * k[3] + PQ_TAB[a0 + a][4] * k[4];
      kmbald_gdot4<false, ECM_KW, PQ1, SC>(mp, pb.kv, tau, gd, nullptr);
#pragma unroll
      for (int a = 0; a < KW; a++) ost<LO>(&gdot_out[(a0 + a) * pb.gs], gd[a]);
   }
}

// the same for the athermal-threshold (BCC) kernel, organised like its evaluation: cheap classes of all 12 systems with static indices (one
// exponential per drag-limited system), then one window system per lane and pass, inserted into the 12 slots (ECM_KM_GDOT_GA; the grouped
// form above pays the window phases in nearly every group of this variant)
#ifndef ECM_KM_GDOT_GA
#define ECM_KM_GDOT_GA 1
#endif
template <bool PQ1, bool SC, bool LO = false>
ECM_DI void km_slip_rates_ga(const MatParams& mp, const Prob& pb, const double e_f[5], double* __restrict__ gdot_out) {
   const double ks[5] = { mp.pk0 * e_f[0], mp.pk1 * e_f[1], mp.pk2 * e_f[2], mp.pk2 * e_f[3], mp.pk2 * e_f[4] };
   const double g_ia = 1.0 / mp.tau_a, gAth = pb.kv.g, wi = 1.0 / mp.wrD;
   double tau[NSLIP], gd[NSLIP], xr[NSLIP];
   slip_tau12(ks, tau);
   unsigned pend = 0; bool any_drag = false;
#pragma unroll
   for (int a = 0; a < NSLIP; a++) {
      const double at = fabs(tau[a]);
      xr[a] = (at - gAth) * wi;
      const bool live = (tau[a] != 0.0) && (xr[a] > 0.0), over = fmax(0.0, at - gAth) * g_ia > mp.t_max;
      gd[a] = 0.0;
      if (live && !over) pend |= 1u << a;
      if (!(live && over)) xr[a] = -1.0; else any_drag = true;
   }
   if (any_drag) {
      double exv[NSLIP];
#pragma unroll
      for (int a = 0; a < NSLIP; a++) exv[a] = -fmax(xr[a], 0.0);
      exp_n<NSLIP, PQ1, SC>(exv);
#pragma unroll
      for (int a = 0; a < NSLIP; a++) {
         const bool small = xr[a] < EPS_SQRT;
         const double gr = small ? pb.kv.gam_r * xr[a] : pb.kv.gam_r * (1.0 - exv[a]);
         if (xr[a] >= 0.0) gd[a] = copysign(gr, tau[a]);
      }
   }
   while (__ballot(pend != 0) != 0ull) {
      if (pend != 0) {
         const int a = __ffs((int)pend) - 1; pend &= pend - 1;
         double t1 = 0.0;
#pragma unroll
         for (int s2 = 0; s2 < NSLIP; s2++) t1 = (a == s2) ? tau[s2] : t1;
         double tau1[1] = { t1 }, gd1[1];
         kmbald_gdot4<false, 1, PQ1, SC>(mp, pb.kv, tau1, gd1, nullptr);
#pragma unroll
         for (int s2 = 0; s2 < NSLIP; s2++) gd[s2] = (a == s2) ? gd1[0] : gd[s2];
      }
   }
#pragma unroll
   for (int a = 0; a < NSLIP; a++) ost<LO>(&gdot_out[a * pb.gs], gd[a]);
}

// ---- pieces of the Jacobian action (rotation data from the stash) ---------------------------------------------------
ECM_DI void load_tr(const Jac& J, double Tr[9]) { for (int c = 0; c < 9; c++) Tr[c] = J.Tr[c]; }
ECM_DI void load_dl(const Jac& J, double d[5]) { for (int c = 0; c < 5; c++) d[c] = J.dl[c]; }

// Jer v_r = -M35(d_lat) (Tr v_r)
ECM_DI void jer_mult(const Jac& J, const double vr[3], double out[5]) {
   double Tr[9]; load_tr(J, Tr);
   double th[3];
#pragma unroll
   for (int i = 0; i < 3; i++) th[i] = Tr[3 * i] * vr[0] + Tr[3 * i + 1] * vr[1] + Tr[3 * i + 2] * vr[2];
   double d[5]; load_dl(J, d);
   double M[5][3]; m35(d, M);
#pragma unroll
   for (int k = 0; k < 5; k++) out[k] = -(M[k][0] * th[0] + M[k][1] * th[1] + M[k][2] * th[2]);
}
// Jer^T u_e
ECM_DI void jer_mult_T(const Jac& J, const double ue[5], double out[3]) {
   double d[5]; load_dl(J, d);
   double M[5][3]; m35(d, M);
   double th[3];
#pragma unroll
   for (int j = 0; j < 3; j++) { double v = 0; for (int k = 0; k < 5; k++) v += M[k][j] * ue[k]; th[j] = -v; }
   double Tr[9]; load_tr(J, Tr);
#pragma unroll
   for (int j = 0; j < 3; j++) out[j] = Tr[j] * th[0] + Tr[3 + j] * th[1] + Tr[6 + j] * th[2];
}
// hat(w_lat) Tr as a 3x3 (row-major)
ECM_DI void wt_matrix(const Jac& J, double Wt[9]) {
   double Tr[9]; load_tr(J, Tr);
   const double w0 = J.wl[0], w1 = J.wl[1], w2 = J.wl[2];
   const double W[3][3] = { { 0.0, -w2, w1 }, { w2, 0.0, -w0 }, { -w1, w0, 0.0 } };
#pragma unroll
   for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) Wt[3 * i + j] = W[i][0] * Tr[j] + W[i][1] * Tr[3 + j] + W[i][2] * Tr[6 + j];
}

// Factorisation.  In place: J.A <- LDL^T of M = diag(1/(kd dt)) + A (unit lower L stored in the strict upper slots, 1/D on the
// diagonal); Ri = Jrr^-1.  The coupling blocks are O(|D| dt) ~ 1e-4 relative, so J dx = rhs is solved by block Gauss-Seidel
// sweeps on (e, r) with these two exact diagonal-block inverses: contraction ~1e-4 per sweep; the Newton step uses 2 sweeps
// (step error ~1e-8 relative, immaterial for the iteration), the tangent eliminates the rotation block exactly instead.
struct Fact { double Ri[9]; bool ok; };

// Ri = Jrr^-1 = (I/dt - hat(w_lat) Tr)^-1, row-major; returns false if singular
ECM_DI bool rot_block_inverse(const Prob& pb, const Jac& J, double Ri[9]) {
   double Wt[9]; wt_matrix(J, Wt);
   double R[3][3];
#pragma unroll
   for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) R[i][j] = (i == j ? pb.dt_ri : 0.0) - Wt[3 * i + j];
   const double c00 = R[1][1] * R[2][2] - R[1][2] * R[2][1], c01 = R[1][2] * R[2][0] - R[1][0] * R[2][2], c02 = R[1][0] * R[2][1] - R[1][1] * R[2][0];
   const double det = R[0][0] * c00 + R[0][1] * c01 + R[0][2] * c02;
   const double di = frcp(det);
   Ri[0] = c00 * di; Ri[3] = c01 * di; Ri[6] = c02 * di;
   Ri[1] = (R[0][2] * R[2][1] - R[0][1] * R[2][2]) * di; Ri[4] = (R[0][0] * R[2][2] - R[0][2] * R[2][0]) * di; Ri[7] = (R[0][1] * R[2][0] - R[0][0] * R[2][1]) * di;
   Ri[2] = (R[0][1] * R[1][2] - R[0][2] * R[1][1]) * di; Ri[5] = (R[0][2] * R[1][0] - R[0][0] * R[1][2]) * di; Ri[8] = (R[0][0] * R[1][1] - R[0][1] * R[1][0]) * di;
   return isfinite(di);
}

ECM_DI void jac_factor(const MatParams& mp, const Prob& pb, Jac& J, Fact& F) {
   const double kdi0 = pb.dt_ri * mp.ikd0, kdi2 = pb.dt_ri * mp.ikd2;
   J.A[sidx(0, 0)] += kdi0; J.A[sidx(1, 1)] += kdi0; J.A[sidx(2, 2)] += kdi2; J.A[sidx(3, 3)] += kdi2; J.A[sidx(4, 4)] += kdi2;
   bool ok = true;
#pragma unroll
   for (int k = 0; k < 5; k++) {
      const double d = J.A[sidx(k, k)];
      ok = ok && (d > 0.0);
      const double inv = frcp(d);
      double li[5];
#pragma unroll
      for (int i = k + 1; i < 5; i++) li[i] = J.A[sidx(k, i)] * inv;
#pragma unroll
      for (int i = k + 1; i < 5; i++)
#pragma unroll
         for (int j = i; j < 5; j++) J.A[sidx(i, j)] -= li[i] * J.A[sidx(k, j)];
#pragma unroll
      for (int i = k + 1; i < 5; i++) J.A[sidx(k, i)] = li[i];
      J.A[sidx(k, k)] = inv;
   }
   const bool okr = rot_block_inverse(pb, J, F.Ri);
   F.ok = ok && okr;
}

// b <- Jee^-1 b  using the in-place factor:  Jee = M Kd  =>  x = Kd^-1 M^-1 b
ECM_DI void jee_solve(const MatParams& mp, const Jac& J, double b[5]) {
#pragma unroll
   for (int k = 0; k < 5; k++)
#pragma unroll
      for (int i = k + 1; i < 5; i++) b[i] -= J.A[sidx(k, i)] * b[k];
#pragma unroll
   for (int k = 0; k < 5; k++) b[k] *= J.A[sidx(k, k)];
#pragma unroll
   for (int k = 4; k >= 0; k--)
#pragma unroll
      for (int j = k + 1; j < 5; j++) b[k] -= J.A[sidx(k, j)] * b[j];
   const double i0 = mp.ikd0, i2 = mp.ikd2;
   b[0] *= i0; b[1] *= i0; b[2] *= i2; b[3] *= i2; b[4] *= i2;
}

// w <- M w from the factor (M = L D L^T)
ECM_DI void m_mult_factored(const Jac& J, double w[5]) {
#pragma unroll
   for (int k = 0; k < 5; k++) {
      double t = w[k];
#pragma unroll
      for (int j = k + 1; j < 5; j++) t += J.A[sidx(k, j)] * w[j];
      w[k] = t / J.A[sidx(k, k)];
   }
#pragma unroll
   for (int i = 4; i >= 0; i--) {
      double t = w[i];
#pragma unroll
      for (int k = 0; k < i; k++) t += J.A[sidx(k, i)] * w[k];
      w[i] = t;
   }
}

ECM_DI void jre_mult(const MatParams& mp, const Jac& J, const double ve[5], double out[3]) {
   const double kv[5] = { mp.kd0 * ve[0], mp.kd0 * ve[1], mp.kd2 * ve[2], mp.kd2 * ve[3], mp.kd2 * ve[4] };
#pragma unroll
   for (int i = 0; i < 3; i++) { double t = 0; for (int j = 0; j < 5; j++) t += J.B[i][j] * kv[j]; out[i] = t; }
}

// y = J v and y = J^T u with the FACTORED J: only needed when the Newton step leaves the trust region
ECM_DI void jac_mult(const MatParams& mp, const Prob& pb, const Jac& J, const double v[8], double y[8]) {
   double a[5] = { mp.kd0 * v[0], mp.kd0 * v[1], mp.kd2 * v[2], mp.kd2 * v[3], mp.kd2 * v[4] };
   m_mult_factored(J, a);                       // Jee v = M (Kd v)
   double b[5]; jer_mult(J, v + 5, b);
   double c[3]; jre_mult(mp, J, v, c);
   double Wt[9]; wt_matrix(J, Wt);
#pragma unroll
   for (int i = 0; i < 5; i++) y[i] = a[i] + b[i];
#pragma unroll
   for (int i = 0; i < 3; i++) y[5 + i] = c[i] + v[5 + i] * pb.dt_ri - (Wt[3 * i] * v[5] + Wt[3 * i + 1] * v[6] + Wt[3 * i + 2] * v[7]);
}

ECM_DI void jac_mult_T(const MatParams& mp, const Prob& pb, const Jac& J, const double u[8], double y[8]) {
   double mu[5] = { u[0], u[1], u[2], u[3], u[4] };
   m_mult_factored(J, mu);                      // Jee^T u = Kd (M u)
   const double kd[5] = { mp.kd0, mp.kd0, mp.kd2, mp.kd2, mp.kd2 };
#pragma unroll
   for (int j = 0; j < 5; j++) y[j] = kd[j] * (mu[j] + J.B[0][j] * u[5] + J.B[1][j] * u[6] + J.B[2][j] * u[7]);
   double c[3]; jer_mult_T(J, u, c);
   double Wt[9]; wt_matrix(J, Wt);
#pragma unroll
   for (int j = 0; j < 3; j++) y[5 + j] = c[j] + u[5 + j] * pb.dt_ri - (Wt[j] * u[5] + Wt[3 + j] * u[6] + Wt[6 + j] * u[7]);
}

// solve J dx = rhs by block Gauss-Seidel on the exact diagonal-block inverses (rhs_r may be identically zero: ZERO_R)
// ECM_GS_START0: the sweeps start from x_r = 0 instead of x_r = Jrr^-1 rhs_r.  The first sweep then has no coupling term in its strain
// half (x_e = Jee^-1 rhs_e) and the start-up product disappears: 39 FP64 operations fewer per Newton step.  Both starts leave an error
// of (contraction)^2 ~ 1e-8 of an O(1) quantity after two sweeps (x_r* itself here, Jrr^-1 Jre x_e* there).
#ifndef ECM_GS_START0
#define ECM_GS_START0 1   // measured at 128^3 (profiles/r04_kernel_experiments.txt): nothing while the launch waited on scratch and stores (4.85 ms either way), 4.43 -> 4.35 ms once it was issue-bound again
#endif
#ifndef ECM_EXP_NSWEEP
#define ECM_EXP_NSWEEP 0   // timing experiment: number of sweeps of the Newton step (0 = the product's two)
#endif
template <bool ZERO_R, int NSWEEP>
ECM_DI void jac_solve(const MatParams& mp, const Prob& pb, const Jac& J, const Fact& F, const double rhs[8], double dx[8]) {
   double xr[3] = { 0, 0, 0 };
   double xe[5];
   if (ECM_GS_START0) {
#pragma unroll
      for (int i = 0; i < 5; i++) xe[i] = rhs[i];
      jee_solve(mp, J, xe);
      double c[3]; jre_mult(mp, J, xe, c);
      double br[3];
#pragma unroll
      for (int i = 0; i < 3; i++) br[i] = (ZERO_R ? 0.0 : rhs[5 + i]) - c[i];
#pragma unroll
      for (int i = 0; i < 3; i++) xr[i] = F.Ri[3 * i] * br[0] + F.Ri[3 * i + 1] * br[1] + F.Ri[3 * i + 2] * br[2];
   } else if (!ZERO_R) {
#pragma unroll
      for (int i = 0; i < 3; i++) xr[i] = F.Ri[3 * i] * rhs[5] + F.Ri[3 * i + 1] * rhs[6] + F.Ri[3 * i + 2] * rhs[7];
   }
#if ECM_SWEEP_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
   for (int sweep = (ECM_GS_START0 ? 1 : 0); sweep < NSWEEP; sweep++) {
      double t[5]; jer_mult(J, xr, t);
#pragma unroll
      for (int i = 0; i < 5; i++) xe[i] = rhs[i] - t[i];
      jee_solve(mp, J, xe);
      double c[3]; jre_mult(mp, J, xe, c);
      double br[3];
#pragma unroll
      for (int i = 0; i < 3; i++) br[i] = (ZERO_R ? 0.0 : rhs[5 + i]) - c[i];
#pragma unroll
      for (int i = 0; i < 3; i++) xr[i] = F.Ri[3 * i] * br[0] + F.Ri[3 * i + 1] * br[1] + F.Ri[3 * i + 2] * br[2];
   }
#pragma unroll
   for (int i = 0; i < 5; i++) dx[i] = xe[i];
#pragma unroll
   for (int i = 0; i < 3; i++) dx[5 + i] = xr[i];
}

ECM_DI double norm8(const double v[8]) { double s = 0; for (int i = 0; i < 8; i++) s += v[i] * v[i]; return sqrt(s); }
ECM_DI double norm8sq(const double v[8]) { double s = 0; for (int i = 0; i < 8; i++) s += v[i] * v[i]; return s; }

// ------------------------------------------------------------------------------------------------------------
// one quadrature point: reference kernel_setup -> getResponseECM -> kernel_postprocessing, fused
//   vgrad  : velocity gradient L(i,j) = dv_i/dx_j as L[i + 3 j]
//   sv0/s0 : begin-of-step state (28) / Voigt stress (6)
//   sv1/s1 : end-of-step outputs;  cmat: 6x6 tangent d sigma / d eps (engineering shear), column-major (used as a parking
//            area for cold values until it is written at the very end)
//   st     : per-thread stash (LDS), ST_SLOTS slots of stride ECM_STASH_STRIDE
// returns 0 on success, 1 if the local solve failed to converge, 2 if it was cut off after kcap evaluations (nothing written)
// ------------------------------------------------------------------------------------------------------------
// QS: distance (in doubles) between consecutive values of one point in the state / stress / tangent arrays: 1 for the reference's
// AoS quadrature functions, 64 for the element-blocked layout (exa_internal.hpp, QView)
// REC: instead of the 36 tangent entries the point's COMPACT GRADIENT RECORD is written (exa_internal.hpp, PAC_PAIRS): the 5 x 5 block the
// tangent is built from, D = D55 / dt, and the bulk term K, both times tsc = dt W_q / detJ - what AssembleGradPA + the projection of
// k_grad_setup_pa<.., CMP> would produce from the 36 entries (reference src/mechanics_integrators.cpp:331-414), without the round trip.
// cmat then points at the lane's first 16-byte pair of the record ([13 pairs][64 lanes][2]); trd stores D^T (element-assembly contexts).
// IO: where the point's data lives (model_kernels.hip, PointIO): sv0() / s0() begin-of-step state / stress, sv1() / s1() / cm() outputs, stash()
// the lane's LDS stash, ipt() the point id, refresh() re-derives all of them from the thread index (see ECM_EPI_NO_LOADS)
// The begin-of-step values a point reads of its old state and stress (21 of the 34: slip rates and evaluation count are not inputs, see
// ECM_SHRATE_FROM_STATE).  The caller requests them BEFORE it gathers nodes and forms the velocity gradient, so that the state rows and the node
// rows travel together: as loads inside point_update they were issued behind the gathers' wait - one more memory round trip at the start of every
// wave (round 5: the prologue of the fused launch went from five dependent round trips to two).
struct PointIn { double shrate, shr, flow, e[5], q[4], h, vol, eint, s[6]; };
// LI: the rows are in the wave's LDS stage (staged AOS launch): plain reads
template <int QS, bool LI = false>
ECM_DI void load_point_in(const double* __restrict__ sv0, const double* __restrict__ s0, PointIn& p) {
   auto ld = [](const double* a) { if constexpr (LI) return *a; else return ldg(a); };
   p.shrate = ld(&sv0[(H_SHRATE) * QS]); p.shr = ld(&sv0[(H_SHR) * QS]); p.flow = ld(&sv0[(H_FLOW) * QS]);
#pragma unroll
   for (int i = 0; i < 5; i++) p.e[i] = ld(&sv0[(H_E + i) * QS]);
#pragma unroll
   for (int i = 0; i < 4; i++) p.q[i] = ld(&sv0[(H_Q + i) * QS]);
   p.h = ld(&sv0[(H_H) * QS]); p.vol = ld(&sv0[(IND_VOL) * QS]); p.eint = ld(&sv0[(IND_EINT) * QS]);
#pragma unroll
   for (int i = 0; i < 6; i++) p.s[i] = ld(&s0[i * QS]);
}

// Tangent output of the staged AOS launch: two rounds of 32 whole rows.  The rows of 32 points are 9 216 contiguous bytes of memory - 72 whole 128-byte
// lines, each written by one wave in one round (rounds of 3 columns of all 64 rows, 144-byte pieces, left every line to be completed a microsecond later:
// 4.96 against 4.80 ms at 128^3 depending on how long L2 kept the first piece).  Lanes 0-31 put their 36 entries into the stage (cmat = the lane's row there),
// the wave stores them, then lanes 32-63; emit(put) produces the 36 entries from values both rounds keep in registers.
template <class IO, class Emit>
ECM_DI void staged_tangent(const IO& io, double* cmat, Emit&& emit) {
#pragma unroll
   for (int h = 0; h < 2; h++) {
      if (io.half() == h) emit([&](const int i, const double v) { cmat[i] = v; });
      ECM_PARK_BARRIER(); __builtin_amdgcn_wave_barrier();
      io.flush_tangent(h);
      ECM_PARK_BARRIER(); __builtin_amdgcn_wave_barrier();
   }
}

// STG (staged AOS launch): the outputs go to the lane's rows of the wave's LDS stage - which is the wave's stash region - and the wave stores them
// coalesced in three rounds (staged_tangent: io.flush_tangent(0 | 1); io.flush_state()): every lane must stay in the wave until the last round (a point that is cut off by the
// tail split only marks itself), and nothing may be read from the stash once the first row of a round has been written.
template <int KIN, int QS, bool REC = false, bool STG = false, class IO>
ECM_DI int point_update(const MatParams& mp, double dt, const double L[9], IO io, const int kcap, const PointIn& pin,
                        const double* pq_lds = nullptr, const double tsc = 0.0, const bool trd = false, const TailIO tio = TailIO()) {
   // (STG with REC: state and stress rows staged, the compact record written by each lane straight to its slot of the element-blocked record array)
   static_assert(!STG || (QS == 1 && ECM_STASH_STRIDE == 64 && ECM_EPI_NO_LOADS && ECM_TANGENT_FIRST && ECM_KM_GDOT_AT_END), "staged outputs: AOS rows, per-wave stash regions, tangent first");
   bool cut = false;   // STG: handed over to the dense launch (the lane stays for the wave's stores; what it writes the dense launch overwrites)
   const bool resume = tio.rs_in != nullptr;
   const double* __restrict__ sv0 = io.sv0();
   double* st = io.stash();
   Prob pb; pb.st = st; pb.gs = QS; pb.pqt = pq_lds;
   pb.dt_ri = 1.0 / dt;
   {
      // ---- kernel_setup (reference src/mechanics_ecmech.cpp:42-99)
      const double w_sm[3] = { 0.5 * (L[2 + 3 * 1] - L[1 + 3 * 2]), 0.5 * (L[0 + 3 * 2] - L[2 + 3 * 0]), 0.5 * (L[1 + 3 * 0] - L[0 + 3 * 1]) };
      const double dkk = L[0] + L[4] + L[8];
      const double d_mean = -(1.0 / 3.0) * dkk;
      double d_sm[5];
      sym_to_vecd(L[0] + d_mean, L[4] + d_mean, L[8] + d_mean, 0.5 * (L[1 + 3 * 0] + L[0 + 3 * 1]), 0.5 * (L[2 + 3 * 0] + L[0 + 3 * 2]),
                  0.5 * (L[2 + 3 * 1] + L[1 + 3 * 2]), d_sm);
      double dnorm2 = 0; for (int i = 0; i < 5; i++) dnorm2 += d_sm[i] * d_sm[i];
      const double dnorm = sqrt(dnorm2), dEff = SQR2B3 * dnorm;
      const double vOld = pin.vol, vNew = vOld * exp(dkk * dt), delv = vNew - vOld;
      const double pOld = -(1.0 / 3.0) * (pin.s[0] + pin.s[1] + pin.s[2]);
      double s_old[5]; sym_to_vecd(pin.s[0] + pOld, pin.s[1] + pOld, pin.s[2] + pOld, pin.s[5], pin.s[4], pin.s[3], s_old);
      // ---- EOS ("updateSimple", EosModelConst<false>): p = K (1/v - 1) + Gamma e
      const double eNew = pin.eint - delv * pOld;
      const double tK = mp.tK0 + eNew * mp.dtde;
      const double bulkNew = mp.bulk * vNew + mp.gamma * pOld * vNew;
      // ---- hardness to end of step with begin-of-step slip rates
#ifndef ECM_SHRATE_FROM_STATE
#define ECM_SHRATE_FROM_STATE 1   // measured at 128^3: 4.83 -> 4.65 ms (profiles/r04_kernel_experiments.txt)
#endif
      // begin-of-step effective shear rate sum_a |gdot_a|: state slot 0 holds exactly this sum (written below from the same 12 values), so one
      // load can replace twelve (88 B of the 208 B a point reads of its old state)
      double shrate_o = 0;
      if (ECM_SHRATE_FROM_STATE) shrate_o = pin.shrate;
      else for (int a = 0; a < NSLIP; a++) shrate_o += fabs(ldg(&sv0[(H_GDOT + a) * QS]));
      const double h_u = kin_update_h<KIN>(mp, pin.h, dt, shrate_o);
      // ---- point problem set-up
      // (reciprocals of well-scaled positive numbers through frcp / rsqrt: 5 instructions instead of the ~13 of an IEEE division)
      const double detV_ri = frcp(vNew); ECM_ST(st, ST_PB + PB_DETVRI) = detV_ri;
      const double a_V = cbrt(vNew), a_V_ri = frcp(a_V);
      pb.esc = E_SCALE * a_V_ri; ECM_ST(st, ST_PB + PB_ESCI) = a_V * (1.0 / E_SCALE);
      double qn[4]; { double n2 = 0; for (int i = 0; i < 4; i++) n2 += pin.q[i] * pin.q[i]; const double ni = rsqrt(n2); for (int i = 0; i < 4; i++) qn[i] = pin.q[i] * ni; }
      double Cn[9]; quat_to_mat(qn, Cn);
      double dn[5]; rot_vecd_T(Cn, d_sm, dn);
      double wrk_old = 0.0;
      for (int i = 0; i < 5; i++) { ECM_ST(st, ST_DN + i) = dn[i]; ECM_ST(st, ST_EN + i) = pin.e[i] * a_V_ri; wrk_old += s_old[i] * d_sm[i]; }
      ECM_CD(CD_WRKOLD) = wrk_old;
      for (int i = 0; i < 3; i++) ECM_ST(st, ST_WN + i) = Cn[i] * w_sm[0] + Cn[3 + i] * w_sm[1] + Cn[6 + i] * w_sm[2];
      for (int i = 0; i < 4; i++) ECM_CD(CD_QN + i) = qn[i];
      ECM_CD(CD_VOLD) = vOld; ECM_CD(CD_VNEW) = vNew; ECM_CD(CD_ENEW) = eNew; ECM_CD(CD_DEFF) = dEff; ECM_CD(CD_BULK) = bulkNew; ECM_CD(CD_HU) = h_u;
      if (REC) ECM_CD(CD_TSC) = tsc;
      if (ECM_EPI_NO_LOADS) { ECM_ST(st, ST_PB + PB_SHR0) = pin.shr; ECM_ST(st, ST_PB + PB_FLOW0) = pin.flow; }
      double adots_ref;
      if (kin_is_km(KIN)) {
         const double sq = sqrt(h_u);
         pb.kv.g = mp.go + mp.s * sq; pb.kv.gam_w = mp.gam_wo / sq; pb.kv.gam_r = mp.gam_ro * sq * sq; pb.kv.c_e = (mp.c_1 / tK) * mp.mu_ref;
         adots_ref = pb.kv.gam_w;
      } else { pb.kv.g = h_u; pb.kv.gam_w = mp.gam_w; pb.kv.gam_r = 0; pb.kv.c_e = 0; adots_ref = mp.gam_w; }
      if (dnorm < EPS_SQRT * adots_ref) { ECM_ST(st, ST_PB + PB_SCI) = adots_ref; pb.sc = 1.0 / adots_ref; }
      else {
         const double s1 = frcp(dnorm), cap = 1.0e6 * dt;
         if (s1 <= cap) { pb.sc = s1; ECM_ST(st, ST_PB + PB_SCI) = dnorm; } else { pb.sc = cap; ECM_ST(st, ST_PB + PB_SCI) = 1.0 / cap; }
      }
      pb.g_i = frcp(pb.kv.g);
   }
   ECM_PARK_BARRIER();

#ifdef ECM_EXP_IO_ONLY   // timing experiment only: the launch's memory traffic without the constitutive arithmetic
   {
      double* sv1 = io.sv1(); double* s1 = io.s1(); double* cmat = io.cm();
      double acc = 0.0;
      for (int i = 0; i < NSTATEV; i++) { const double v = ldg(&sv0[i * QS]); acc += v; stg(&sv1[i * QS], v + L[i % 9]); }
      for (int i = 0; i < 6; i++) stg(&s1[i * QS], pin.s[i] + acc);
      if (REC) { double2* rc = reinterpret_cast<double2*>(cmat); for (int pr = 0; pr < 13; pr++) rc[pr * 64] = make_double2(acc + pr, tsc); }
      else for (int i = 0; i < 36; i++) stg(&cmat[i * QS], acc + i);
      return 0;
   }
#endif
   // ---- trust-region dog-leg Newton (SNLS "TrDlDenseG" defaults).  Every evaluation leaves (r, J, slip rates, dissipation) of
   // the point it was asked for; a rejected trial is followed by a re-evaluation at the restored point (rare), so nothing but x
   // and a few scalars has to survive an evaluation and the converged evaluation doubles as the one the tangent needs.
   // TailIO: a point that is cut off after kcap evaluations appends itself to list_out and leaves its solver state - the accepted iterate x, the
   // trust radius and the evaluation count - in rs_out at its list slot.  A launch that is handed such a state (rs_in) starts from it: one evaluation at x restores (r, J) (not counted:
   // it repeats one the cut-off launch has made, bit for bit), then the iteration goes on as if it had never stopped.
   double x[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
   if (resume) { for (int i = 0; i < 8; i++) x[i] = ldg(&tio.rs_in[i * tio.stride]); }
   double r[8], dis_rate, shrate;
   Jac J; Fact F;
   double* gdot_out = (kin_is_km(KIN) && !ECM_KM_GDOT_AT_END) ? io.sv1() + H_GDOT * QS : nullptr;
   int nfev = 1; bool conv = false;
   bool ok = eval_rj<KIN, true>(mp, pb, x, r, J, gdot_out, dis_rate, shrate);
   // Norms are carried SQUARED: the common iteration (full Newton step inside the trust region) only compares them - |r| < tol, |dx| <= delta,
   // |r_new| > 0.65 |r_old| (SNLS: rho = actual / predicted < 0.35 with predicted = -|r_old|), |r_new| > |r_old| - and the two square roots per
   // iteration (~20 instructions each in FP64) are only taken on the dog-leg path, which needs the values.
   // eval_rj returns the UN-scaled residual; SNLS works with r * sc (sc = epsdot_scale_inv): the squared norm carries sc^2, the Newton right-hand
   // side needs no scale at all (16 multiplications per evaluation fewer), the dog-leg path scales its own copy.
   const double tol2 = mp.tol * mp.tol, sc2 = pb.sc * pb.sc;
   double res2_0 = sc2 * norm8sq(r);
   ok = ok && isfinite(res2_0);
   if (ok && res2_0 < tol2 && !resume) conv = true;
#ifdef ECM_EXP_SKIP_SOLVE   // timing experiment only (scripts/tune_model.sh): no Newton iterations, the rest of the launch unchanged
   conv = true;
#endif
   if (ok && !conv) {
      double delta = 1.0;
      if (resume) { delta = ldg(&tio.rs_in[8 * tio.stride]); nfev = (int)ldg(&tio.rs_in[9 * tio.stride]); }
      // Kocks-Mecking without athermal threshold (ECM_KEEP_DOGLEG): the dog-leg ingredients of the accepted point - Newton step, steepest-descent direction and its
      // three scalars - are computed with every accepted evaluation and kept across the trial, like SNLS does (reject_prev), so a rejected
      // trial only restores x and shrinks the trust region: no second evaluation at the old point.  With these kinetics nearly every wave
      // holds a rejecting lane in every iteration, i.e. the re-evaluation of the Voce form below (rare per lane) was paid by all of them.
      // (not the athermal-threshold variant: the 20 values carried through the evaluation cost 7 % there).  Off by default since the capped
      // launch hands rejecting points over to the dense launch (ECM_DEFER_REJECT below), which removes the re-evaluations at no cost.
      constexpr bool KEEP = ECM_KEEP_DOGLEG && kin_base(KIN) == KIN_KMBALD;
      double nr[8], grad[8], nr2sq = 0.0, norm2_grad = 0.0, Jg_2 = 0.0, s2 = 0.0;
      bool reject_prev = false;
      // A rejected trial needs (r, J) of the accepted point again: one more evaluation, rare per lane but paid by the whole wave.  In a capped
      // launch whose listed points resume from their saved state (ECM_DEFER_REJECT) such a point is handed over instead - x restored, trust
      // radius shrunk: exactly the state a resumed point starts from, and the evaluation that restores (r, J) there is the one saved here.
      // The points that reject are the ones with long iteration histories, i.e. the ones the cap would list a few evaluations later anyway.
      // Kocks-Mecking instantiations only: the Voce launches run uncapped (their controller never finds a paying cap), and the mere presence
      // of the extra exit costs their register allocation 4.7 % (5.13 against 4.90 ms at 128^3).
      constexpr bool DEFER = ECM_DEFER_REJECT != 0 && kin_is_km(KIN);
      bool hand_over = false;
      auto dogleg_data = [&]() {   // grad = Js^T r, Jg = Js grad, s2 = |r + Js sd_opt|^2 (all in SNLS's scaled variables)
         double u[8], rs[8], tt[8];
         for (int i = 0; i < 8; i++) rs[i] = r[i] * pb.sc;
         jac_mult_T(mp, pb, J, rs, tt);
         for (int i = 0; i < 8; i++) { const double cs = (i < 5) ? pb.esc : R_SCALE; grad[i] = pb.sc * cs * tt[i]; u[i] = cs * grad[i]; }
         jac_mult(mp, pb, J, u, tt);
         Jg_2 = 0; norm2_grad = 0;
         for (int i = 0; i < 8; i++) { const double jg = pb.sc * tt[i]; Jg_2 += jg * jg; norm2_grad += grad[i] * grad[i]; }
         const double fac = (Jg_2 > 0) ? norm2_grad / Jg_2 : 0.0;
         s2 = 0; for (int i = 0; i < 8; i++) { const double v = rs[i] - fac * pb.sc * tt[i]; s2 += v * v; }
      };
      for (int it = 0; it < 200; it++) {
         // tail split: a point that needs more than kcap evaluations is handed to the dense tail launch (which starts over, so the
         // result is the one of an uncapped solve); the wave stops waiting for its slowest lanes
         if (nfev >= kcap || (DEFER && hand_over)) {
            const int slot = atomicAdd(&tio.list_out[0], 1); tio.list_out[1 + slot] = io.ipt();
            if (tio.rs_out) {
               double* o = tio.rs_out + slot;
               for (int i = 0; i < 8; i++) stg(&o[i * tio.stride], x[i]);
               stg(&o[8 * tio.stride], delta); stg(&o[9 * tio.stride], (double)nfev);
            }
            if constexpr (STG) { cut = true; break; } else return 2;
         }
         // Newton step first; the steepest-descent data (grad = Js^T r, Jg = Js grad) only when the step leaves the trust region
         if (!KEEP || !reject_prev) {
            double t[8];
            jac_factor(mp, pb, J, F);
            if (F.ok) {
               double rhs[8]; for (int i = 0; i < 8; i++) rhs[i] = -r[i];
               jac_solve<false, (ECM_EXP_NSWEEP ? ECM_EXP_NSWEEP : 2)>(mp, pb, J, F, rhs, t);
               const double esc_i = ECM_ST(st, ST_PB + PB_ESCI);
               for (int i = 0; i < 8; i++) nr[i] = t[i] * ((i < 5) ? esc_i : (1.0 / R_SCALE));
               nr2sq = norm8sq(nr);
            } else { nr2sq = 1e300; for (int i = 0; i < 8; i++) nr[i] = 0; }
            if (KEEP) dogleg_data();
         }
         double delx[8], pred_resid; bool use_nr = false;
         if (nr2sq <= delta * delta) { use_nr = true; for (int i = 0; i < 8; i++) delx[i] = nr[i]; pred_resid = 0.0; }
         else {
            if (!KEEP) dogleg_data();
            const double res_0 = sqrt(res2_0);
            const double norm_grad = sqrt(norm2_grad);
            const double fac = (Jg_2 > 0) ? norm2_grad / Jg_2 : 0.0;
            const double norm_s_sd_opt = (Jg_2 > 0) ? fac * norm_grad : 1e300;
            if (norm_s_sd_opt >= delta) {
               const double f = delta / norm_grad;
               for (int i = 0; i < 8; i++) delx[i] = -grad[i] * f;
               pred_resid = sqrt(fmax(res_0 * res_0 - 2.0 * delta * norm_grad + delta * delta * Jg_2 / norm2_grad, 0.0));
            } else {
               // |r + Js sd|, sd = -fac grad: the Newton point zeroes the linear model, so the dog-leg point predicts (1-beta) of it
               double qa = 0, qb = 0;
               for (int i = 0; i < 8; i++) { const double sd = -grad[i] * fac, p = nr[i] - sd; qa += p * p; qb += p * sd; }
               const double qc = norm_s_sd_opt * norm_s_sd_opt - delta * delta;
               const double beta = (-qb + sqrt(fmax(qb * qb - qa * qc, 0.0))) / qa;
               for (int i = 0; i < 8; i++) { const double sd = -grad[i] * fac; delx[i] = sd + beta * (nr[i] - sd); }
               pred_resid = (1.0 - beta) * sqrt(s2);
            }
         }
         for (int i = 0; i < 8; i++) { ECM_ST(st, ST_XS + i) = x[i]; x[i] += delx[i]; }
         ECM_PARK_BARRIER();
         ok = eval_rj<KIN, true>(mp, pb, x, r, J, gdot_out, dis_rate, shrate); nfev++;
         const double res2 = sc2 * norm8sq(r);
         ok = ok && isfinite(res2);
         bool reject;
         if (!ok) { reject = true; delta = fmax(delta * 0.25, 1e-12); }
         else {
            if (res2 < tol2) { conv = true; break; }
#ifdef ECM_EXP_MAXEVAL   // timing experiment only: what the launch would cost if no lane needed more than ECM_EXP_MAXEVAL evaluations
            if (nfev >= ECM_EXP_MAXEVAL) { conv = true; break; }
#endif
            if (use_nr) {      // predicted residual 0: rho = 1 - |r| / |r_old| (never > 0.75 with a smaller residual AND a dog-leg step: no growth)
               if (res2_0 == 0.0) delta = fmin(delta * 1.5, 1e4);
               else if (res2 > (0.65 * 0.65) * res2_0) delta = fmax(delta * 0.25, 1e-12);
            } else {
               const double res = sqrt(res2), res_0 = sqrt(res2_0);
               const double actual = res - res_0, pred = pred_resid - res_0;
               if (pred == 0.0) delta = fmin(delta * 1.5, 1e4);
               else {
                  const double rho = actual / pred;
                  if (rho > 0.75 && actual < 0.0) delta = fmin(delta * 1.5, 1e4);
                  else if (rho < 0.35) delta = fmax(delta * 0.25, 1e-12);
               }
            }
            reject = (res2 > res2_0);
            if (!reject) res2_0 = res2;
         }
         reject_prev = reject;
         if (reject) {
            for (int i = 0; i < 8; i++) x[i] = ECM_ST(st, ST_XS + i);
            if (DEFER && (!KEEP || ECM_DEFER_REJECT == 2) && tio.defer_reject && delta > 1e-12) { hand_over = true; continue; }
            if (!KEEP) ok = eval_rj<KIN, true>(mp, pb, x, r, J, gdot_out, dis_rate, shrate);   // restore (r, J) of the accepted point
            else ok = true;   // (the accepted point evaluated fine; r and J hold the rejected trial until the next accepted evaluation)
            if (!ok || delta <= 1e-12) break;
         }
      }
   }
   // ---- converged state: stress, energy, history (getResponseSngl tail + reference kernel_postprocessing src/mechanics_ecmech.cpp:116-152)
   if (ECM_EPI_NO_LOADS) { io.refresh(); st = io.stash(); pb.st = st; }
   double* __restrict__ sv1 = io.sv1(); double* __restrict__ s1 = io.s1(); double* __restrict__ cmat = io.cm();
   double e_f[5], xi[3];
   for (int i = 0; i < 5; i++) e_f[i] = ECM_ST(st, ST_EN + i) + x[i] * pb.esc;
   for (int i = 0; i < 3; i++) xi[i] = x[5 + i] * R_SCALE;
   double Cf[9], qout[4];
   {
      const double qn[4] = { ECM_CD(CD_QN), ECM_CD(CD_QN + 1), ECM_CD(CD_QN + 2), ECM_CD(CD_QN + 3) };
      double qf[4];
      const double th2 = xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2];
      double cq, sq;   // cos(th/2), sin(th/2)/th
      if (th2 < 1.0e-4) { const double h2 = 0.25 * th2; cq = 1.0 - 0.5 * h2 * (1.0 - h2 * (1.0 / 12.0) * (1.0 - h2 * (1.0 / 30.0))); sq = 0.5 * (1.0 - h2 * (1.0 / 6.0) * (1.0 - h2 * (1.0 / 20.0) * (1.0 - h2 * (1.0 / 42.0)))); }
      else { const double th = sqrt(th2); double sn, cs; sincos_s(0.5 * th, sn, cs); cq = cs; sq = sn / th; }
      const double a[4] = { cq, sq * xi[0], sq * xi[1], sq * xi[2] };
      qf[0] = qn[0] * a[0] - qn[1] * a[1] - qn[2] * a[2] - qn[3] * a[3];
      qf[1] = qn[0] * a[1] + qn[1] * a[0] + qn[2] * a[3] - qn[3] * a[2];
      qf[2] = qn[0] * a[2] - qn[1] * a[3] + qn[2] * a[0] + qn[3] * a[1];
      qf[3] = qn[0] * a[3] + qn[1] * a[2] - qn[2] * a[1] + qn[3] * a[0];
      quat_to_mat(qf, Cf);
      double dq = 0; for (int i = 0; i < 4; i++) dq += qf[i] * qn[i];
      const double sg = dq < 0 ? -1.0 : 1.0;
      for (int i = 0; i < 4; i++) qout[i] = sg * qf[i];
   }
   const double detV_ri = ECM_ST(st, ST_PB + PB_DETVRI);
   dis_rate *= detV_ri;   // per current volume
   const double kdj[5] = { mp.kd0 * detV_ri, mp.kd0 * detV_ri, mp.kd2 * detV_ri, mp.kd2 * detV_ri, mp.kd2 * detV_ri };
   double s_lat[5];
   for (int i = 0; i < 5; i++) s_lat[i] = kdj[i] * e_f[i];
   const double bulkNew = ECM_CD(CD_BULK);
   // The tangent goes FIRST (ECM_TANGENT_FIRST): it is the only consumer of the 47 Jacobian values of the converged evaluation, so they die
   // before the state / stress outputs are computed instead of being carried (and spilled) through them; and no scratch reload of the
   // tangent arithmetic has to wait behind the 34 output stores (on gfx9 a reload waits for every earlier store of the wave).  The two
   // parked values the outputs need from the slot the tangent overwrites are read before.
   const double hu_keep = ECM_CD(CD_HU), deff_keep = ECM_CD(CD_DEFF);      // (STG: read again in write_state_staged, not carried)
   double wrk_new = 0.0;   // s_new . D' in the lattice frame of the converged evaluation (the inner product of the 5-vectors is frame-invariant)
   for (int k = 0; k < 5; k++) wrk_new += s_lat[k] * J.dl[k];
   [[maybe_unused]] double rsc_keep = 0.0;
   if constexpr (STG) {   // see ST_EPI_*
      static_assert(ST_EPI_Q1 == ST_CD + CD_TSC && ST_EPI_Q2 == ST_PB + PB_SCI && ST_EPI_Q3 == ST_PB + PB_DETVRI && ST_EPI_WRK == ST_CD + CD_BULK, "dead slots");
      if constexpr (REC) rsc_keep = ECM_CD(CD_TSC);      // (the record scale sits where q[1] is about to be parked)
      for (int i = 0; i < 5; i++) ECM_ST(st, ST_EPI_E + i) = e_f[i];
      ECM_ST(st, ST_EPI_Q0) = qout[0]; ECM_ST(st, ST_EPI_Q1) = qout[1]; ECM_ST(st, ST_EPI_Q2) = qout[2]; ECM_ST(st, ST_EPI_Q3) = qout[3];
      ECM_ST(st, ST_EPI_WRK) = wrk_new;
      ECM_PARK_BARRIER();
   }
   auto write_state = [&]() {
      for (int i = 0; i < 4; i++) stg(&sv1[(H_Q + i) * QS], qout[i]);
      double s_sm[5]; rot_vecd(Cf, s_lat, s_sm);
      const double vNew = ECM_CD(CD_VNEW);
      double eNew = ECM_CD(CD_ENEW);
      eNew += 0.25 * (ECM_CD(CD_VOLD) + vNew) * dt * (ECM_CD(CD_WRKOLD) + wrk_new);
      if constexpr (!kin_is_km(KIN)) voce_slip_rates<kin_xn_ct(KIN)>(mp, pb, e_f, sv1 + H_GDOT * QS, dis_rate, shrate);
      else if (ECM_KM_GDOT_AT_END) {
         // (the runtime flag is always set in this instantiation, see eval_rj; only the p == q == 1 kernel has the factored forms of the evaluation)
         if (ECM_KM_GDOT_GA && kin_base(KIN) == KIN_KMBALD_GA && kin_pq1(KIN) && mp.with_g_athermal) km_slip_rates_ga<kin_pq1(KIN), kin_sc_exp(KIN)>(mp, pb, e_f, sv1 + H_GDOT * QS);
         else km_slip_rates<kin_pq1(KIN), kin_sc_exp(KIN)>(mp, pb, e_f, sv1 + H_GDOT * QS);
      }
      stg(&sv1[(H_SHRATE) * QS], shrate);
      stg(&sv1[(H_SHR) * QS], (ECM_EPI_NO_LOADS ? ECM_ST(st, ST_PB + PB_SHR0) : ldg(&sv0[(H_SHR) * QS])) + shrate * dt);
      stg(&sv1[(H_FLOW) * QS], ((deff_keep > TINY_SQRT) ? dis_rate * dt : 0.0) + (ECM_EPI_NO_LOADS ? ECM_ST(st, ST_PB + PB_FLOW0) : ldg(&sv0[(H_FLOW) * QS])));   // accumulated plastic work
      stg(&sv1[(H_NFEV) * QS], (double)nfev);
      { const double a_V = E_SCALE * ECM_ST(st, ST_PB + PB_ESCI); for (int i = 0; i < 5; i++) stg(&sv1[(H_E + i) * QS], e_f[i] * a_V); }   // state e = a_V E
      stg(&sv1[(H_H) * QS], hu_keep);
      stg(&sv1[(IND_VOL) * QS], vNew); stg(&sv1[(IND_EINT) * QS], eNew);
      const double pNew = mp.bulk * (1.0 / vNew - 1.0) + mp.gamma * eNew;
      const double t1 = SQR2I * s_sm[0], t2 = SQR6I * s_sm[1];
      stg(&s1[(0) * QS], t1 - t2 - pNew); stg(&s1[(1) * QS], -t1 - t2 - pNew); stg(&s1[(2) * QS], SQR2B3 * s_sm[1] - pNew);
      stg(&s1[(3) * QS], SQR2I * s_sm[4]); stg(&s1[(4) * QS], SQR2I * s_sm[3]); stg(&s1[(5) * QS], SQR2I * s_sm[2]);
   };
   // the same with every stash value read BEFORE the first row store (the rows of the wave's 64 points cover the stash slots of all its lanes)
   auto write_state_staged = [&]() {
      const double vNew = ECM_CD(CD_VNEW), vOld = ECM_CD(CD_VOLD), wrkOld = ECM_CD(CD_WRKOLD), shr0 = ECM_ST(st, ST_PB + PB_SHR0), flow0 = ECM_ST(st, ST_PB + PB_FLOW0);
      const double a_V = E_SCALE * ECM_ST(st, ST_PB + PB_ESCI), hu = ECM_CD(CD_HU), deff = ECM_CD(CD_DEFF), wrkn = ECM_ST(st, ST_EPI_WRK);
      double eNew = ECM_CD(CD_ENEW);
      double ef[5], qo[4];
      for (int i = 0; i < 5; i++) ef[i] = ECM_ST(st, ST_EPI_E + i);
      qo[0] = ECM_ST(st, ST_EPI_Q0); qo[1] = ECM_ST(st, ST_EPI_Q1); qo[2] = ECM_ST(st, ST_EPI_Q2); qo[3] = ECM_ST(st, ST_EPI_Q3);
      ECM_PARK_BARRIER(); __builtin_amdgcn_wave_barrier();
      for (int i = 0; i < 4; i++) ost<true>(&sv1[(H_Q + i) * QS], qo[i]);
      double Cq[9]; quat_to_mat(qo, Cq);      // = Cf: the rotation matrix is even in the quaternion
      // (the products are pinned to registers: as single-use values they would be contracted into the rotation's multiply-adds, which the per-lane form -
      //  where s_lat also feeds the tangent - does not do, and the staged form is held to the per-lane form's bits)
      double sl[5]; for (int i = 0; i < 5; i++) { sl[i] = kdj[i] * ef[i]; asm volatile("" : "+v"(sl[i])); }
      double s_sm[5]; rot_vecd(Cq, sl, s_sm);
      eNew += 0.25 * (vOld + vNew) * dt * (wrkOld + wrkn);
      if constexpr (!kin_is_km(KIN)) voce_slip_rates<kin_xn_ct(KIN), true>(mp, pb, ef, sv1 + H_GDOT * QS, dis_rate, shrate, detV_ri);
      else {
         if (ECM_KM_GDOT_GA && kin_base(KIN) == KIN_KMBALD_GA && kin_pq1(KIN) && mp.with_g_athermal) km_slip_rates_ga<kin_pq1(KIN), kin_sc_exp(KIN), true>(mp, pb, ef, sv1 + H_GDOT * QS);
         else km_slip_rates<kin_pq1(KIN), kin_sc_exp(KIN), true>(mp, pb, ef, sv1 + H_GDOT * QS);
      }
      ost<true>(&sv1[(H_SHRATE) * QS], shrate);
      ost<true>(&sv1[(H_SHR) * QS], shr0 + shrate * dt);
      ost<true>(&sv1[(H_FLOW) * QS], ((deff > TINY_SQRT) ? dis_rate * dt : 0.0) + flow0);
      ost<true>(&sv1[(H_NFEV) * QS], (double)nfev);
      for (int i = 0; i < 5; i++) ost<true>(&sv1[(H_E + i) * QS], ef[i] * a_V);
      ost<true>(&sv1[(H_H) * QS], hu);
      ost<true>(&sv1[(IND_VOL) * QS], vNew); ost<true>(&sv1[(IND_EINT) * QS], eNew);
      const double pNew = mp.bulk * (1.0 / vNew - 1.0) + mp.gamma * eNew;
      const double t1 = SQR2I * s_sm[0], t2 = SQR6I * s_sm[1];
      ost<true>(&s1[(0) * QS], t1 - t2 - pNew); ost<true>(&s1[(1) * QS], -t1 - t2 - pNew); ost<true>(&s1[(2) * QS], SQR2B3 * s_sm[1] - pNew);
      ost<true>(&s1[(3) * QS], SQR2I * s_sm[4]); ost<true>(&s1[(4) * QS], SQR2I * s_sm[3]); ost<true>(&s1[(5) * QS], SQR2I * s_sm[2]);
      ECM_PARK_BARRIER(); __builtin_amdgcn_wave_barrier();
      io.flush_state();
   };
   if (!ECM_TANGENT_FIRST) write_state();
#ifdef ECM_EXP_TAN_FAKE   // timing experiment: the 13 record stores without the tangent arithmetic
   if constexpr (REC) { double2* rc = reinterpret_cast<double2*>(cmat); for (int pr = 0; pr < 13; pr++) rc[pr * 64] = make_double2(J.A[pr] * detV_ri, bulkNew + J.B[pr % 3][pr % 5]); }
#define ECM_NO_TANGENT 1
#endif
#ifndef ECM_NO_TANGENT
   // ---- tangent (last: it overwrites the parking area): lattice-frame d sigma'/d D' by implicit differentiation on the converged
   // factorisation, rotated to the sample frame, then to Voigt (engineering shear) + bulk term, column-major
   // Two forms of the same arithmetic: the W X form (ECM_TANGENT_WX) for the Voce kinds - 170 multiply-adds fewer, 4.62 -> 4.30 ms at 128^3 - and the
   // round-3 chain for the Kocks-Mecking kinds, whose register allocation the W X form upsets (BCC p = q = 1: 4 -> 37 spills)
   if constexpr (ECM_TANGENT_WX != 0 && !kin_is_km(KIN))
   {
      // J [xe; xr] = [e_c; 0] with Jee = M Kd, Jre = B Kd, Jer = -E (E = M35(d_lat) Tr), xr eliminated exactly:
      //   y = Kd xe,  (M + E G) y = e_c,  G = Jrr^-1 B,  xr = -G y   =>   Llat = (I/J + M35(s') Tr G) (M + E G)^-1
      // explicit 5x5 rotation of deviatoric 5-vectors (columns = images of the unit vectors)
      double Q5[5][5];
#pragma unroll
      for (int l = 0; l < 5; l++) {
         double e[5] = { 0, 0, 0, 0, 0 }, out[5]; e[l] = 1.0;
         rot_vecd(Cf, e, out);
#pragma unroll
         for (int k = 0; k < 5; k++) Q5[k][l] = out[k];
      }
      double D55[5][5];   // sample-frame deviatoric tangent block (times dt)
      // (Voce kinds only: in the Kocks-Mecking kernels the W X form costs registers they do not have - BCC 4 -> 37 spills, elastic pass 3.4 -> 4.1 ms)
      constexpr bool WX = true;
      double Llat[WX ? 1 : 5][5];
      bool okT;
      {
         double Ri[9]; okT = rot_block_inverse(pb, J, Ri);
         // G <- Tr Jrr^-1 B: both couplings enter as M35(.) Tr G, so Tr is folded into G once (27 + 45 multiply-adds) instead of into each M35
         double G[3][5];
         {
            double Tr[9]; load_tr(J, Tr);
            double TRi[9];
#pragma unroll
            for (int m = 0; m < 3; m++)
#pragma unroll
               for (int i = 0; i < 3; i++) TRi[3 * m + i] = Tr[3 * m] * Ri[i] + Tr[3 * m + 1] * Ri[3 + i] + Tr[3 * m + 2] * Ri[6 + i];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
               for (int j = 0; j < 5; j++) G[i][j] = TRi[3 * i] * J.B[0][j] + TRi[3 * i + 1] * J.B[1][j] + TRi[3 * i + 2] * J.B[2][j];
         }
         double S[5][5];
         {
            double dl[5]; load_dl(J, dl);
            double Md[5][3]; m35(dl, Md);
            const double kdi0 = pb.dt_ri * mp.ikd0, kdi2 = pb.dt_ri * mp.ikd2;
#pragma unroll
            for (int k = 0; k < 5; k++) {
#pragma unroll
               for (int j = 0; j < 5; j++) S[k][j] = J.A[sidx(k, j)] + ((k == j) ? (k < 2 ? kdi0 : kdi2) : 0.0) + Md[k][0] * G[0][j] + Md[k][1] * G[1][j] + Md[k][2] * G[2][j];
            }
         }
         // un-pivoted LU of S (M is SPD and dominates the O(|D| dt) coupling), then Y = S^-1 column by column
#pragma unroll
         for (int k = 0; k < 5; k++) {
            okT = okT && (S[k][k] > 0.0);
            const double inv = frcp(S[k][k]);
            S[k][k] = inv;
#pragma unroll
            for (int i = k + 1; i < 5; i++) {
               const double l = S[i][k] * inv; S[i][k] = l;
#pragma unroll
               for (int j = k + 1; j < 5; j++) S[i][j] -= l * S[k][j];
            }
         }
         if constexpr (WX) {
         // D55 = Q5 Llat Q5^T = (Q5 Kt) (S^-1 Q5^T) =: W X.  W = detV_ri Q5 + M35(s_sm) (Cf G): the map w -> vecd(S W - W S) is equivariant under the
         // rotation (Q5 M35(s) w = M35(Q5 s) (Cf w)), so the rotated coupling costs 45 + 65 multiply-adds instead of 75 + 125; X is five
         // forward / backward substitutions with the columns of Q5^T as right-hand sides.  375 + 110 multiply-adds where the chain
         // S^-1 -> Kt S^-1 -> Q5 (.) -> (.) Q5^T took 555 (same numbers in another order of summation)
         {
            double CG[3][5];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
               for (int j = 0; j < 5; j++) CG[i][j] = Cf[3 * i] * G[0][j] + Cf[3 * i + 1] * G[1][j] + Cf[3 * i + 2] * G[2][j];
            double s_sm5[5]; rot_vecd(Cf, s_lat, s_sm5);
            double Mr[5][3]; m35(s_sm5, Mr);
            double Wm[5][5], X[5][5];
#pragma unroll
            for (int k = 0; k < 5; k++)
#pragma unroll
               for (int j = 0; j < 5; j++) Wm[k][j] = fma(detV_ri, Q5[k][j], Mr[k][0] * CG[0][j] + Mr[k][1] * CG[1][j] + Mr[k][2] * CG[2][j]);
#pragma unroll
            for (int c = 0; c < 5; c++) {       // column c of X = S^-1 (row c of Q5)
               double y[5];
#pragma unroll
               for (int i = 0; i < 5; i++) { double t = Q5[c][i]; for (int j = 0; j < i; j++) t -= S[i][j] * y[j]; y[i] = t; }
#pragma unroll
               for (int i = 4; i >= 0; i--) { double t = y[i]; for (int j = i + 1; j < 5; j++) t -= S[i][j] * y[j]; y[i] = t * S[i][i]; }
#pragma unroll
               for (int l = 0; l < 5; l++) X[l][c] = y[l];
            }
#pragma unroll
            for (int k = 0; k < 5; k++)
#pragma unroll
               for (int c = 0; c < 5; c++) D55[k][c] = Wm[k][0] * X[0][c] + Wm[k][1] * X[1][c] + Wm[k][2] * X[2][c] + Wm[k][3] * X[3][c] + Wm[k][4] * X[4][c];
         }
         } else {
         double Kt[5][5];   // detV_ri I + M35(s_lat) (Tr G)
         {
            double Ms[5][3]; m35(s_lat, Ms);
#pragma unroll
            for (int k = 0; k < 5; k++) {
#pragma unroll
               for (int j = 0; j < 5; j++) Kt[k][j] = ((k == j) ? detV_ri : 0.0) + Ms[k][0] * G[0][j] + Ms[k][1] * G[1][j] + Ms[k][2] * G[2][j];
            }
         }
#pragma unroll
         for (int c = 0; c < 5; c++) {
            double y[5];
#pragma unroll
            for (int i = 0; i < 5; i++) {       // forward: L y = e_c (entries above c stay zero)
               double t = (i == c) ? 1.0 : 0.0;
#pragma unroll
               for (int j = c; j < i; j++) t -= S[i][j] * y[j];
               y[i] = (i < c) ? 0.0 : t;
            }
#pragma unroll
            for (int i = 4; i >= 0; i--) {      // backward: U y = y
               double t = y[i];
#pragma unroll
               for (int j = i + 1; j < 5; j++) t -= S[i][j] * y[j];
               y[i] = t * S[i][i];
            }
#pragma unroll
            for (int k = 0; k < 5; k++) Llat[k][c] = Kt[k][0] * y[0] + Kt[k][1] * y[1] + Kt[k][2] * y[2] + Kt[k][3] * y[3] + Kt[k][4] * y[4];
         }
               }
      }
      if constexpr (!WX) {  // D55 = Q5 Llat Q5^T
         double T1[5][5];
#pragma unroll
         for (int k = 0; k < 5; k++)
#pragma unroll
            for (int c = 0; c < 5; c++) { double v = 0; for (int l = 0; l < 5; l++) v += Q5[k][l] * Llat[l][c]; T1[k][c] = v; }
#pragma unroll
         for (int k = 0; k < 5; k++)
#pragma unroll
            for (int c = 0; c < 5; c++) { double v = 0; for (int l = 0; l < 5; l++) v += T1[k][l] * Q5[c][l]; D55[k][c] = v; }
      }
      if constexpr (REC) {
         const double rsc = STG ? rsc_keep : ECM_CD(CD_TSC);
         const double dsc = rsc * pb.dt_ri * (okT ? 1.0 : 0.0);
         double Dm[26];
#pragma unroll
         for (int k = 0; k < 5; k++)
#pragma unroll
            for (int c = 0; c < 5; c++) Dm[k + 5 * c] = D55[k][c] * dsc;
         if (trd) {
#pragma unroll
            for (int k = 0; k < 5; k++)
#pragma unroll
               for (int c = k + 1; c < 5; c++) { const double v = Dm[k + 5 * c]; Dm[k + 5 * c] = Dm[c + 5 * k]; Dm[c + 5 * k] = v; }
         }
         Dm[25] = bulkNew * rsc;
         double2* rc = reinterpret_cast<double2*>(cmat);
#ifdef ECM_EXP_TAN_NOSTORE   // timing experiment: the tangent arithmetic without its 13 record stores (one store of a checksum keeps it alive)
         { double a = 0.0, b = 0.0; for (int pr = 0; pr < 13; pr++) { a += Dm[2 * pr]; b += Dm[2 * pr + 1]; } rc[0] = make_double2(a, b); }
#else
#pragma unroll
         for (int pr = 0; pr < 13; pr++) stg2(&rc[pr * 64], Dm[2 * pr], Dm[2 * pr + 1]);
#endif
      } else {
      const double dti = pb.dt_ri * (okT ? 1.0 : 0.0);
      double T2a[5][6];   // D55 S56 / dt  (d_vecd = S56 eps_svec(eng. shear) / dt)
#pragma unroll
      for (int k = 0; k < 5; k++) {
         const double* D = D55[k];
         T2a[k][0] = (SQR2I * D[0] - SQR6I * D[1]) * dti;
         T2a[k][1] = (-SQR2I * D[0] - SQR6I * D[1]) * dti;
         T2a[k][2] = (2.0 * SQR6I * D[1]) * dti;
         T2a[k][3] = (SQR2I * D[4]) * dti;
         T2a[k][4] = (SQR2I * D[3]) * dti;
         T2a[k][5] = (SQR2I * D[2]) * dti;
      }
      // sigma_svec = V65 sigma_vecd; column-major C(i,j) at cmat[(i + 6 j) * QS]
      auto emit = [&](auto&& put) {
#pragma unroll
         for (int j = 0; j < 6; j++) {
            double T2[5];
#pragma unroll
            for (int k = 0; k < 5; k++) T2[k] = T2a[k][j];
            const double t1 = SQR2I * T2[0], t2 = SQR6I * T2[1];
            const double bk = (j < 3) ? bulkNew : 0.0;
            put(0 + 6 * j, t1 - t2 + bk); put(1 + 6 * j, -t1 - t2 + bk); put(2 + 6 * j, SQR2B3 * T2[1] + bk);
            put(3 + 6 * j, SQR2I * T2[4]); put(4 + 6 * j, SQR2I * T2[3]); put(5 + 6 * j, SQR2I * T2[2]);
         }
      };
      if constexpr (STG) staged_tangent(io, cmat, emit);
      else emit([&](const int i, const double v) { stg(&cmat[i * QS], v); });
      }
   }
   else
   {
      // J [xe; xr] = [e_c; 0] with Jee = M Kd, Jre = B Kd, Jer = -E (E = M35(d_lat) Tr), xr eliminated exactly:
      //   y = Kd xe,  (M + E G) y = e_c,  G = Jrr^-1 B,  xr = -G y   =>   Llat = (I/J + M35(s') Tr G) (M + E G)^-1
      double Llat[5][5];
      bool okT;
      {
         double Ri[9]; okT = rot_block_inverse(pb, J, Ri);
         // G <- Tr Jrr^-1 B: both couplings enter as M35(.) Tr G, so Tr is folded into G once (27 + 45 multiply-adds) instead of into each M35
         double G[3][5];
         {
            double Tr[9]; load_tr(J, Tr);
            double TRi[9];
#pragma unroll
            for (int m = 0; m < 3; m++)
#pragma unroll
               for (int i = 0; i < 3; i++) TRi[3 * m + i] = Tr[3 * m] * Ri[i] + Tr[3 * m + 1] * Ri[3 + i] + Tr[3 * m + 2] * Ri[6 + i];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
               for (int j = 0; j < 5; j++) G[i][j] = TRi[3 * i] * J.B[0][j] + TRi[3 * i + 1] * J.B[1][j] + TRi[3 * i + 2] * J.B[2][j];
         }
         double S[5][5];
         {
            double dl[5]; load_dl(J, dl);
            double Md[5][3]; m35(dl, Md);
            const double kdi0 = pb.dt_ri * mp.ikd0, kdi2 = pb.dt_ri * mp.ikd2;
#pragma unroll
            for (int k = 0; k < 5; k++) {
#pragma unroll
               for (int j = 0; j < 5; j++) S[k][j] = J.A[sidx(k, j)] + ((k == j) ? (k < 2 ? kdi0 : kdi2) : 0.0) + Md[k][0] * G[0][j] + Md[k][1] * G[1][j] + Md[k][2] * G[2][j];
            }
         }
         // un-pivoted LU of S (M is SPD and dominates the O(|D| dt) coupling), then Y = S^-1 column by column
#pragma unroll
         for (int k = 0; k < 5; k++) {
            okT = okT && (S[k][k] > 0.0);
            const double inv = frcp(S[k][k]);
            S[k][k] = inv;
#pragma unroll
            for (int i = k + 1; i < 5; i++) {
               const double l = S[i][k] * inv; S[i][k] = l;
#pragma unroll
               for (int j = k + 1; j < 5; j++) S[i][j] -= l * S[k][j];
            }
         }
         double Kt[5][5];   // detV_ri I + M35(s_lat) (Tr G)
         {
            double Ms[5][3]; m35(s_lat, Ms);
#pragma unroll
            for (int k = 0; k < 5; k++) {
#pragma unroll
               for (int j = 0; j < 5; j++) Kt[k][j] = ((k == j) ? detV_ri : 0.0) + Ms[k][0] * G[0][j] + Ms[k][1] * G[1][j] + Ms[k][2] * G[2][j];
            }
         }
#pragma unroll
         for (int c = 0; c < 5; c++) {
            double y[5];
#pragma unroll
            for (int i = 0; i < 5; i++) {       // forward: L y = e_c (entries above c stay zero)
               double t = (i == c) ? 1.0 : 0.0;
#pragma unroll
               for (int j = c; j < i; j++) t -= S[i][j] * y[j];
               y[i] = (i < c) ? 0.0 : t;
            }
#pragma unroll
            for (int i = 4; i >= 0; i--) {      // backward: U y = y
               double t = y[i];
#pragma unroll
               for (int j = i + 1; j < 5; j++) t -= S[i][j] * y[j];
               y[i] = t * S[i][i];
            }
#pragma unroll
            for (int k = 0; k < 5; k++) Llat[k][c] = Kt[k][0] * y[0] + Kt[k][1] * y[1] + Kt[k][2] * y[2] + Kt[k][3] * y[3] + Kt[k][4] * y[4];
         }
      }
      // D55 = Q5 Llat Q5^T with the explicit 5x5 rotation of deviatoric 5-vectors (columns = images of the unit vectors)
      double Q5[5][5];
#pragma unroll
      for (int l = 0; l < 5; l++) {
         double e[5] = { 0, 0, 0, 0, 0 }, out[5]; e[l] = 1.0;
         rot_vecd(Cf, e, out);
#pragma unroll
         for (int k = 0; k < 5; k++) Q5[k][l] = out[k];
      }
      double T1[5][5];
#pragma unroll
      for (int k = 0; k < 5; k++)
#pragma unroll
         for (int c = 0; c < 5; c++) { double v = 0; for (int l = 0; l < 5; l++) v += Q5[k][l] * Llat[l][c]; T1[k][c] = v; }
      if constexpr (REC) {
         const double rsc = STG ? rsc_keep : ECM_CD(CD_TSC);
         const double dsc = rsc * pb.dt_ri * (okT ? 1.0 : 0.0);
         double Dm[26];
#pragma unroll
         for (int k = 0; k < 5; k++)
#pragma unroll
            for (int c = 0; c < 5; c++) { double v = 0; for (int l = 0; l < 5; l++) v += T1[k][l] * Q5[c][l]; Dm[k + 5 * c] = v * dsc; }
         if (trd) {
#pragma unroll
            for (int k = 0; k < 5; k++)
#pragma unroll
               for (int c = k + 1; c < 5; c++) { const double v = Dm[k + 5 * c]; Dm[k + 5 * c] = Dm[c + 5 * k]; Dm[c + 5 * k] = v; }
         }
         Dm[25] = bulkNew * rsc;
         double2* rc = reinterpret_cast<double2*>(cmat);
#ifdef ECM_EXP_TAN_NOSTORE   // timing experiment: the tangent arithmetic without its 13 record stores (one store of a checksum keeps it alive)
         { double a = 0.0, b = 0.0; for (int pr = 0; pr < 13; pr++) { a += Dm[2 * pr]; b += Dm[2 * pr + 1]; } rc[0] = make_double2(a, b); }
#else
#pragma unroll
         for (int pr = 0; pr < 13; pr++) stg2(&rc[pr * 64], Dm[2 * pr], Dm[2 * pr + 1]);
#endif
      } else {
      const double dti = pb.dt_ri * (okT ? 1.0 : 0.0);
#pragma unroll
      for (int k = 0; k < 5; k++) {
         double D[5];   // row k of D55 = T1 Q5^T
#pragma unroll
         for (int c = 0; c < 5; c++) { double v = 0; for (int l = 0; l < 5; l++) v += T1[k][l] * Q5[c][l]; D[c] = v; }
         // row k of T2 = D55 S56 / dt  (d_vecd = S56 eps_svec(eng. shear) / dt), overwriting T1's row
         T1[k][0] = (SQR2I * D[0] - SQR6I * D[1]) * dti;
         T1[k][1] = (-SQR2I * D[0] - SQR6I * D[1]) * dti;
         T1[k][2] = (2.0 * SQR6I * D[1]) * dti;
         T1[k][3] = (SQR2I * D[4]) * dti;
         T1[k][4] = (SQR2I * D[3]) * dti;
         Llat[k][0] = (SQR2I * D[2]) * dti;   // sixth column parked in Llat's first column
      }
      // sigma_svec = V65 sigma_vecd; column-major C(i,j) at cmat[(i + 6 j) * QS]
      auto emit = [&](auto&& put) {
#pragma unroll
         for (int j = 0; j < 6; j++) {
            double T2[5];
#pragma unroll
            for (int k = 0; k < 5; k++) T2[k] = (j < 5) ? T1[k][j] : Llat[k][0];
            const double t1 = SQR2I * T2[0], t2 = SQR6I * T2[1];
            const double bk = (j < 3) ? bulkNew : 0.0;
            put(0 + 6 * j, t1 - t2 + bk); put(1 + 6 * j, -t1 - t2 + bk); put(2 + 6 * j, SQR2B3 * T2[1] + bk);
            put(3 + 6 * j, SQR2I * T2[4]); put(4 + 6 * j, SQR2I * T2[3]); put(5 + 6 * j, SQR2I * T2[2]);
         }
      };
      if constexpr (STG) staged_tangent(io, cmat, emit);
      else emit([&](const int i, const double v) { stg(&cmat[i * QS], v); });
      }
   }
#endif
   if constexpr (STG) { write_state_staged(); return cut ? 2 : ((conv && ok) ? 0 : 1); }
   if (ECM_TANGENT_FIRST) write_state();
   return (conv && ok) ? 0 : 1;
}

}  // namespace ecmdev
