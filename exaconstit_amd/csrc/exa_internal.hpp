// Internal definitions of libexaconstit_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstdint>
#include <cstdlib>
#include "../../include/exaconstit_hip.h"
#include "ecm_device.hpp"

#define EXA_HIP_CHECK(ctx, call)                                                                 \
   do {                                                                                          \
      hipError_t e_ = (call);                                                                    \
      if (e_ != hipSuccess) { (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return EXA_ERR_HIP; } \
   } while (0)

// Internal layout of the partial-assembly gradient data ("pa data"), one record per quadrature point:
//   36 doubles  Ct(i,j) = C(i,j) * dt * W_q / detJ      (column-major 6x6, Voigt, engineering shear)
//    9 doubles  adj(J)(r,c) at slot 36 + 3 r + c
//    1 double   W_q * detJ (quadrature weight in the current configuration)
// stored as [block of 64 elements][qpt][23 slot-pairs][64 lanes][2] so that lane = element reads 16 B and a wave reads
// 1 KiB contiguous per instruction.  (The reference stores 81 + 81 doubles per point: src/mechanics_integrators.cpp:395-414.)
constexpr int PA_SLOTS = 46;
constexpr int PA_PAIRS = 23;
constexpr int PA_BLK = 64;

struct exa_ctx {
   exa_config cfg;
   ecmdev::MatParams mp;
   int p, n, Q, E;
   int64_t P;               // E * Q
   int nstatev;
   int device;
   std::string err;
   std::vector<double> G_host, W_host;      // (n,3,Q), (Q)
   double* G_dev = nullptr; double* W_dev = nullptr;
   // residual PA data (D of AssemblePA), gradient PA data / EA matrices
   double* dmat = nullptr;                  // (3,3,Q,E)
   double* pa = nullptr;                    // see layout above
   double* emat = nullptr;                  // EA: p=1 full integration [block][24][12 pairs][64][2]; otherwise [block][3n][3n][64]
   double* T1_dev = nullptr;                // p = 2 matrix-free action: one-dimensional basis tables (3 x 6)
   double* eDS = nullptr;                   // B-bar: element-average shape gradient [block][n x 3][64 lanes]
   double* tbuf = nullptr;                  // generic PA action: per-point T (3,3,Q,E)
   double* vgrad_ref = nullptr;             // p = 2 element-blocked constitutive launch: reference-space velocity gradients of its geometry pre-pass (3,3,Q,E)
   const double* resid_J = nullptr; const double* resid_S = nullptr;   // B-bar residual reads J and sigma at apply time, like the reference
   bool ea_generic = false;
   bool ea_matfree = false, emat_valid = false;   // EA, p = 2: L-vector action computed from the point records, matrices assembled on demand
   bool qblk = false;                       // quadrature functions in the element-blocked layout (see QView below)
   bool aos_stage = true;                   // AOS layout: constitutive launches move a wave's contiguous point rows coalesced and transpose them through LDS (exa_set_aos_staging)
   bool have_resid = false, have_grad = false;
   bool pa_lazy = false; const double* lazy_J = nullptr; const double* lazy_C = nullptr; double lazy_dt = 0.0;   // 46-double records not built yet (pa_kernels.hip, pa_full_on_demand)
   bool grad_records_only = false;          // gradient data = compact records written by the constitutive launch: no 46-double records, no element matrices
   // L-vector support
   const int32_t* conn = nullptr; int nnodes = 0;
   double* pa_c = nullptr;                  // compact tangent records (25 + 1 per point) of the geometry-recomputing p = 1 action; allocated when the form is selected
   int tangent_form = 0;                    // EXA_TANGENT_*
   int pac_pairs = 0;                       // 16-byte pairs per point of the compact record: 13 (D, K) at p = 1, 18 (+ geometry) at p = 2
   // deterministic E->L (exa_set_deterministic): node -> (element, local node) table in element order + per-element output scratch
   int det = 0; int32_t* n2e_off = nullptr; int32_t* n2e_idx = nullptr; double* ev_det = nullptr; int det_nnodes = 0; const int32_t* det_conn = nullptr;
   const double* coords_lvec = nullptr;     // optional: nodal coordinates the Jacobians of exa_grad_setup came from (geometry recomputed in the apply)
   // status
   int* fail_count_dev = nullptr;
   int newton_cap = 0; int* tail_dev = nullptr;   // tail split of the constitutive launch: [0] = count, [1..] = deferred point ids
   int newton_cap2 = 0; int* tail2_dev = nullptr; // second level (exa_set_newton_caps): list of the points the first tail launch cuts off
   int tail_resume = 1; double* resume_dev[2] = { nullptr, nullptr };   // solver states of the listed points, [RS_N][P] by list slot
   int cap_auto = 0; double cap_tail_cost = 0.0; long cap_calls = 0;   // exa_set_newton_cap_auto: the library picks newton_cap from the evaluation counts of its own launches
   double* scratch_dev = nullptr; size_t scratch_bytes = 0;   // reductions
   double hist_init[ecmdev::NUM_HIST];
};

// host-side reference element (H1 hex of order p at (p+1)^3 Gauss-Legendre points), src/mechanics_operator.cpp:237-261
int exa_ensure_p2_tables(struct exa_ctx* ctx);   // gen_kernels.hip
void exa_build_ref_elem(int p, std::vector<double>& G, std::vector<double>& W);
// one-dimensional tables [1D Gauss point][B_0..B_p, D_0..D_p] of the order-p nodal basis and the lexicographic -> native node map
void exa_build_1d_tables(int p, std::vector<double>& T1, std::vector<int>& nat);
bool exa_fill_mat_params(const exa_config& cfg, ecmdev::MatParams& mp, double* hist_init, std::string& err);

// Quadrature-function addressing.  AOS is the reference's QuadratureFunction layout (vdim values of a point contiguous, points of an
// element consecutive).  EB64 is the internal layout of the stand-alone driver: [block of 64 elements][point q][value k][lane = element],
// the same blocking as the PA record, so that a wave whose lanes are 64 consecutive elements reads or writes one contiguous 512-byte
// row per value — the per-lane 224/288-byte strides of AOS cost the constitutive kernel 5 of its 7.5 ms at 128^3.
struct QView { int64_t base; int stride; };
template <bool QB>
__device__ __forceinline__ QView qview(int W, int Q, int64_t e, int q) {
   if (QB) return { ((((e >> 6) * Q + q) * (int64_t)W) << 6) + (e & 63), 64 };
   return { (int64_t)W * (q + (int64_t)Q * e), 1 };
}
// Shape-derivative table (n,3,Q) of the reference element: the kernels that walk it stage it in LDS when it is small (p = 1: 1.5 KB, p = 2:
// 17 KB) and read it from global memory at the orders the reference's unit tests also run (p = 3: 96 KB ... p = 6: 2.8 MB); one rule for
// the kernel (stage_shape) and for the launch (exa_g_lds_bytes)
constexpr int64_t EXA_G_LDS_MAX_DOUBLES = 6144;
static inline __host__ __device__ bool exa_g_in_lds(int n, int Q) { return (int64_t)n * 3 * Q <= EXA_G_LDS_MAX_DOUBLES; }
static inline size_t exa_g_lds_bytes(int n, int Q) { return exa_g_in_lds(n, Q) ? sizeof(double) * (size_t)n * 3 * Q : 0; }
#ifdef __HIPCC__
__device__ __forceinline__ const double* stage_shape(double* lds, const double* __restrict__ G, int n, int Q) {
   if (!exa_g_in_lds(n, Q)) return G;      // kernel-uniform
   for (int i = threadIdx.x; i < n * 3 * Q; i += blockDim.x) lds[i] = G[i];
   __syncthreads();
   return lds;
}
#endif
static inline size_t exa_qf_doubles(const exa_ctx* ctx, int vdim) {
   return ctx->qblk ? (size_t)vdim * 64 * ctx->Q * ((ctx->E + 63) / 64) : (size_t)vdim * ctx->P;
}

#ifdef __HIPCC__
// Point records and element matrices are streams: a launch of an operator action reads each 16-byte pair exactly once, gigabytes per launch (3.5 GB of
// compact records at 128^3), and the next launch reads them again from HBM whatever the caches held.  With the non-temporal hint they pass through
// without displacing the node rows (x, coordinates, the atomically updated y) that neighbouring waves share in L2: p = 1 action 0.712 -> 0.664 ms,
// PCG 1 169 -> 1 232 it/s at 128^3 in one call (round 5).  EXA_APPLY_NT=0: plain loads (A/B switch).
#ifndef EXA_APPLY_NT
#define EXA_APPLY_NT 1
#endif
template <bool NT = (EXA_APPLY_NT != 0)>
__device__ __forceinline__ double2 ld_rec(const double2* p) {
   if constexpr (NT) {
      typedef double vd2 __attribute__((ext_vector_type(2)));
      const vd2 v = __builtin_nontemporal_load(reinterpret_cast<const vd2*>(p));
      return make_double2(v.x, v.y);
   } else return *p;
}
// ... when the stream is larger than the caches can hold from one launch to the next.  Round 6, same call, PCG iterations/s with the hints | without: 16^3 (7 MB of
// compact records) 40 226 | 42 137, 32^3 (54 MB) 30 469 | 31 393, 64^3 (436 MB) 8 494 | 7 738: a small partition's records and vectors live in L2 / MALL across
// iterations and the hint throws that away.  The p = 1 action and the PCG vector kernels therefore carry it only above this size (EXA_NT_MIN_MB overrides; 0 = always)
static inline bool exa_stream_nt(size_t bytes_per_launch) {
   static const double min_mb = [] { const char* e = std::getenv("EXA_NT_MIN_MB"); return e ? std::atof(e) : 128.0; }();
   return EXA_APPLY_NT != 0 && (double)bytes_per_launch >= min_mb * 1048576.0;
}
// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Remapped, XCD x works on one contiguous eighth of the element blocks, so the node rows
// neighbouring blocks share (gathers of x / coordinates, scatter atomics) stay within one L2 (p = 1 action since round 4; p = 2 action / residual / pre-pass: round 6)
#ifndef EXA_XCD_REMAP
#define EXA_XCD_REMAP 1
#endif
__device__ __forceinline__ int64_t xcd_block(const unsigned b, const unsigned nb) {
   if (!EXA_XCD_REMAP) return b;
   const unsigned q = nb >> 3, r = nb & 7u, x = b & 7u, i = b >> 3;
   return x < r ? (int64_t)x * (q + 1) + i : (int64_t)r * (q + 1) + (int64_t)(x - r) * q + i;
}
// ---- compact tangent form (include/exaconstit_hip.h, EXA_TANGENT_DEV5_BULK) ----------------------------------------------------
// d sigma / d eps = V65 D V65^T + K m m^T: a 5 x 5 block in ExaCMech's deviatoric vector basis plus the bulk term, m = (1,1,1,0,0,0).
constexpr int PAC_PAIRS = 13;       // compact record of the p = 1 action: 25 D entries + K, scaled by dt W / detJ
constexpr int PAC_PAIRS_GEO = 18;   // compact record of the p = 2 action: the same + adj(J) (9) + W detJ
template <int NP>
__device__ __forceinline__ int64_t pac_off(int64_t blk, int Q, int q, int pair) { return (((blk * Q + q) * NP + pair) * PA_BLK) * 2; }
constexpr double C_SQR2I = 0.70710678118654752440, C_SQR6I = 0.40824829046386301637;
__device__ __forceinline__ void v65t(const double c0, const double c1, const double c2, const double c3, const double c4, const double c5, double o[5]) {
   o[0] = C_SQR2I * (c0 - c1); o[1] = C_SQR6I * (2.0 * c2 - c0 - c1); o[2] = C_SQR2I * c5; o[3] = C_SQR2I * c4; o[4] = C_SQR2I * c3;
}
// D = L^-1 V65^T C V65 L^-1 with L = V65^T V65 = diag(1,1,1/2,1/2,1/2), K = m^T C m / 9 (exact when C has the form above)
__device__ __forceinline__ void tangent_to_d55(const double* c, const int64_t st, double D[25], double& K) {
   double T[5][6];
#pragma unroll
   for (int j = 0; j < 6; j++) {
      double o[5]; v65t(c[(0 + 6 * j) * st], c[(1 + 6 * j) * st], c[(2 + 6 * j) * st], c[(3 + 6 * j) * st], c[(4 + 6 * j) * st], c[(5 + 6 * j) * st], o);
#pragma unroll
      for (int k = 0; k < 5; k++) T[k][j] = o[k];
   }
#pragma unroll
   for (int k = 0; k < 5; k++) {
      double o[5]; v65t(T[k][0], T[k][1], T[k][2], T[k][3], T[k][4], T[k][5], o);
#pragma unroll
      for (int l = 0; l < 5; l++) D[k + 5 * l] = o[l] * ((k < 2 ? 1.0 : 2.0) * (l < 2 ? 1.0 : 2.0));
   }
   double t = 0;
#pragma unroll
   for (int j = 0; j < 3; j++)
#pragma unroll
      for (int i = 0; i < 3; i++) t += c[(i + 6 * j) * st];
   K = t * (1.0 / 9.0);
}
// s = (V65 D V65^T + K m m^T) eps
__device__ __forceinline__ void d55_apply(const double D[25], const double K, const double eps[6], double sg[6]) {
   double e5[5], s5[5];
   v65t(eps[0], eps[1], eps[2], eps[3], eps[4], eps[5], e5);
#pragma unroll
   for (int k = 0; k < 5; k++) s5[k] = D[k] * e5[0] + D[k + 5] * e5[1] + D[k + 10] * e5[2] + D[k + 15] * e5[3] + D[k + 20] * e5[4];
   const double t1 = C_SQR2I * s5[0], t2 = C_SQR6I * s5[1], bk = K * (eps[0] + eps[1] + eps[2]);
   sg[0] = t1 - t2 + bk; sg[1] = -t1 - t2 + bk; sg[2] = 2.0 * C_SQR6I * s5[1] + bk; sg[3] = C_SQR2I * s5[4]; sg[4] = C_SQR2I * s5[3]; sg[5] = C_SQR2I * s5[2];
}
#endif

static inline size_t pa_bytes(int E, int Q) { return (size_t)((E + PA_BLK - 1) / PA_BLK) * Q * PA_SLOTS * PA_BLK * sizeof(double); }
