// Fused constitutive kernels (gfx950): one launch replaces the seven passes of
// ExaCMechModel::ModelSetup (reference src/mechanics_ecmech.cpp:192-258):
//   StressSetup/StateVarsSetup copies, matGrad/vel_grad zero fills, grad_calc (src/mechanics_kernels.cpp:36-77),
//   kernel_setup (src/mechanics_ecmech.cpp:42-99), getResponseECM, kernel_postprocessing (:116-171).
// One thread per quadrature point; the reference shape-derivative table is staged in LDS; per-point state is read
// and written once (928 B per point algorithmic traffic).  Variants (templates): E-vector + Jacobian inputs (the reference's
// ModelSetup signature) or fused L-vector gathers that also write the Jacobians; reference (AOS) or element-blocked quadrature-function
// layout; run-time tail split of long local solves into a dense second launch.
#include "exa_internal.hpp"
#include "p2_basis.hpp"
#include <cstdlib>
#include <cstring>

using namespace ecmdev;

#ifndef EXA_MODEL_OCC
#define EXA_MODEL_OCC 2   // waves per SIMD the register allocator is asked to fit (tuned on MI355X)
#endif
#ifndef EXA_MODEL_BS
#define EXA_MODEL_BS ECM_STASH_STRIDE   // threads per block of the constitutive launch = stride of the per-lane LDS stash
#endif
static_assert(EXA_MODEL_BS == ECM_STASH_STRIDE && EXA_MODEL_BS % 64 == 0, "stash stride must equal the block size");

// Where one thread's quadrature point lives.  Everything is a function of (block index, thread index) and kernel-uniform data, so nothing
// per-lane has to survive the local Newton solve: locate() derives the point from the thread index, refresh() does it again behind a compiler
// barrier after the solve, and the accessors form the addresses where they are used.  Before, five 64-bit row pointers and two LDS addresses
// were live across the solve; the register allocator spilled some of them and re-loaded them from scratch BEHIND the first output stores,
// where a load waits for the whole store queue of the wave (ecm_device.hpp, ECM_EPI_NO_LOADS).
template <bool QB, bool REC>
struct PointIO {
   const double* state0; const double* stress0; double* state1; double* stress1; double* cmat;   // kernel-uniform array bases
   double* stash0;                  // LDS stash of thread 0 of the block
   const int* tail; int tail_mode;  // dense tail launch: thread t owns point tail[1 + t]
   int Q; int64_t bidx; int wpb;    // points per element, (remapped) block index, waves per block
   int q; int64_t e; int tid;       // this thread's point and its index in the block
   __device__ __forceinline__ void locate(const int tid_) {
      tid = tid_;
      if (tail_mode) { const int64_t ipt = tail[1 + bidx * (int64_t)(wpb * 64) + tid]; q = (int)(ipt % Q); e = ipt / Q; }
      else if (QB) { const int64_t gw = bidx * wpb + (tid >> 6); q = (int)(gw % Q); e = (gw / Q) * 64 + (tid & 63); }   // wave = (block of 64 elements, q); lane = element
      else { const int64_t ip = bidx * (int64_t)(wpb * 64) + tid; q = (int)(ip % Q); e = ip / Q; }
   }
   __device__ __forceinline__ void refresh() { int t = threadIdx.x; asm volatile("" : "+v"(t)); locate(t); }
   __device__ __forceinline__ const double* sv0() const { return state0 + qview<QB>(ecmdev::NSTATEV, Q, e, q).base; }
   __device__ __forceinline__ const double* s0() const { return stress0 + qview<QB>(6, Q, e, q).base; }
   __device__ __forceinline__ double* sv1() const { return state1 + qview<QB>(ecmdev::NSTATEV, Q, e, q).base; }
   __device__ __forceinline__ double* s1() const { return stress1 + qview<QB>(6, Q, e, q).base; }
   // REC: the lane's first 16-byte pair of its compact record ([block][q][13 pairs][64 lanes][2]); else the tangent slot
   __device__ __forceinline__ double* cm() const { return REC ? cmat + pac_off<PAC_PAIRS>(e >> 6, Q, q, 0) + 2 * (e & 63) : cmat + qview<QB>(36, Q, e, q).base; }
   __device__ __forceinline__ double* stash() const { return stash0 + tid; }
   __device__ __forceinline__ int ipt() const { return (int)(e * Q + q); }
};

// LVEC = false: J (3,3,Q,E) and the velocity E-vector (n,3,E) are inputs (the reference's ModelSetup signature).
// LVEC = true : the kernel gathers nodal coordinates and velocities from the L-vectors through the connectivity, computes J itself
//               and WRITES it to Jio for the integrator kernels: NonlinearMechOperator::Setup's L->E restrictions and
//               SetupJacobianTerms (reference src/mechanics_operator.cpp:310-391) ride inside this VALU-bound kernel for free.
// NFIX = 8: trilinear elements, node loops fully unrolled so that all gathers of a point are in flight together (two memory round
// trips instead of one dependent index->value chain per node); NFIX = 0: run-time n.
// QB: quadrature functions (J, stress, state, tangent) in the element-blocked layout; then a wave is (64 consecutive elements, one
// point index q) so that every per-value access of the wave is one contiguous 512-byte row.
// REC (fused p = 1 launch of the stand-alone driver): cmat is the buffer of compact gradient records and the launch writes, instead of the
// 36 tangent entries, the record the gradient action streams (ecm_device.hpp, point_update<.., REC>): AssembleGradPA rides in the launch.
template <int KIN, bool LVEC, int NFIX, bool QB, bool REC = false>
__global__ __launch_bounds__(EXA_MODEL_BS, EXA_MODEL_OCC) void k_model_setup(const MatParams mp, const int Q, const int n_rt, const int64_t P, const double dt,
                                                     double* __restrict__ Jio, const double* __restrict__ G,
                                                     const double* __restrict__ vel, const double* __restrict__ xl, const int32_t* __restrict__ conn, const int nnodes,
                                                     const double* __restrict__ stress0,
                                                     const double* __restrict__ state0, double* __restrict__ stress1,
                                                     double* __restrict__ state1, double* __restrict__ cmat, int* __restrict__ fail,
                                                     const int kcap, int* __restrict__ tail, const int tail_mode,
                                                     const double* __restrict__ Wq = nullptr, const int trd = 0,
                                                     int* __restrict__ tail_out = nullptr, const double* __restrict__ rs_in = nullptr, double* __restrict__ rs_out = nullptr) {
   // tail: list this launch works on (tail_mode) or appends to; tail_out: list a tail launch appends the points to that it cuts off itself
   // (second level); rs_in / rs_out: solver states of the listed points, [RS_N][P] by list slot (nullptr: a listed point starts over)
   if (!tail_out) tail_out = tail;
   static_assert(!REC || (LVEC && QB && NFIX == 8), "record output is built for the fused element-blocked p = 1 launch");
   if (tail_mode && (int64_t)blockIdx.x * blockDim.x >= tail[0]) return;   // tail launch: its grid covers the worst case, blocks beyond the list leave before the table fill
   const int n = NFIX ? NFIX : n_rt;
   constexpr bool P2F = (NFIX == 27);   // triquadratic fused path: G holds the 3 x 6 one-dimensional tables, read through scalar loads
#ifndef EXA_MODEL_XCD_REMAP
#define EXA_MODEL_XCD_REMAP 0
#endif
   // workgroups are dealt round-robin to the 8 XCDs; remapped, an XCD works on one contiguous eighth of the element blocks (node gathers of
   // neighbouring blocks then hit the same L2)
   int64_t bidx = blockIdx.x;
   if (EXA_MODEL_XCD_REMAP && !tail_mode) {
      const unsigned nb = gridDim.x, qq = nb >> 3, r = nb & 7u, x = blockIdx.x & 7u, i = blockIdx.x >> 3;
      bidx = x < r ? (int64_t)x * (qq + 1) + i : (int64_t)r * (qq + 1) + (int64_t)(x - r) * qq + i;
   }
   // LDS: shape-derivative rows (not for P2F), the per-thread stash, the slip table (Kocks-Mecking).  A wave of the element-blocked launch works
   // on ONE point index q, so only the rows of the block's waves are staged (row w = the (n,3) table of wave w's q: 192 B per wave instead of
   // the 1.5 KB table - what lets four 128-thread blocks of the Kocks-Mecking kernels fit the 160 KB of a CU); the dense tail launch and the
   // reference layout have lane-varying q and stage the whole (n,3,Q) table (model_lds_bytes gives the launch the matching size)
   extern __shared__ double sG[];
   const bool rows_by_wave = QB && !tail_mode;
   const int wpb = (int)(blockDim.x >> 6);
   // fused trilinear launch, element-blocked: q is wave-uniform, so the 24 shape derivatives of the wave's point come through scalar loads
   // (no LDS staging, no barrier, no round trip in front of everything else); the dense tail launch (lane-varying q) stages the whole table
   constexpr bool SROW = LVEC && NFIX == 8 && QB;
   const bool g_lds = !P2F && !(SROW && rows_by_wave) && exa_g_in_lds(n, rows_by_wave ? wpb : Q);   // orders above 2: the table stays in global memory (exa_internal.hpp)
   const int tab = g_lds ? n * 3 * (rows_by_wave ? wpb : Q) : 0;
   // Kocks-Mecking: the slip table (12 rows of 8) behind the stash, for the rows that are read by lane-varying index (ecm_device.hpp, eval_rj)
   const int pqo = tab + ecmdev::ST_SLOTS * ECM_STASH_STRIDE;
   if (g_lds) for (int i = threadIdx.x; i < tab; i += blockDim.x) {
      const int row = i / (3 * n), k = i - row * (3 * n);
      sG[i] = G[3 * n * (rows_by_wave ? (int)((bidx * wpb + row) % Q) : row) + k];
   }
   if (ecmdev::kin_is_km(KIN)) for (int i = threadIdx.x; i < 8 * ecmdev::NSLIP; i += blockDim.x) sG[pqo + i] = (&ecmdev::PQ_TAB[0][0])[i];
   if (g_lds || ecmdev::kin_is_km(KIN)) __syncthreads();
   if (tail_mode) {   // dense pass over the points the capped launch handed over: thread t owns point tail[1 + t]
      const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
      if (t >= tail[0]) return;
      if (rs_in) rs_in += t;   // this point's slot
   }
   PointIO<QB, REC> io{ state0, stress0, state1, stress1, cmat, sG + tab, tail, tail_mode, Q, bidx, wpb, 0, 0, 0 };
   io.locate(threadIdx.x);
   const int q = io.q; const int64_t e = io.e;
   if (!tail_mode && e * Q >= P) return;
   constexpr int QS = QB ? 64 : 1;
   // the point's begin-of-step state and stress are requested first: they depend on nothing but (e, q) and travel while the nodes are gathered
   ecmdev::PointIn pin; ecmdev::load_point_in<QS>(io.sv0(), io.s0(), pin);
   const QView vJ = qview<QB>(9, Q, e, q);
   const double* Gq = g_lds ? sG + 3 * n * (rows_by_wave ? (int)(threadIdx.x >> 6) : q) : G + 3 * n * q;
   double J11, J21, J31, J12, J22, J32, J13, J23, J33;
   double tsc = 0.0;   // REC: dt W_q / detJ
   double L[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
   if constexpr (P2F) {
      static_assert(!P2F || LVEC, "the triquadratic fused path gathers from L-vectors");
      // both node contractions as three one-dimensional passes (p2_basis.hpp); q is wave-uniform except in the tail launch
      const int qu = (QB && !tail_mode) ? __builtin_amdgcn_readfirstlane(q) : q;
      p2::Rows r; p2::load_rows(p2::as_const(G), qu, r);
      const int32_t* ce = conn + (int64_t)27 * e;
      int gn[27];
#pragma unroll
      for (int a = 0; a < 27; a++) gn[a] = ce[a];
      double gx[3][3];
      p2::gather(r, [&](int a, int c) { return xl[gn[a] + (int64_t)nnodes * c]; }, gx);
      J11 = gx[0][0]; J21 = gx[1][0]; J31 = gx[2][0]; J12 = gx[0][1]; J22 = gx[1][1]; J32 = gx[2][1]; J13 = gx[0][2]; J23 = gx[1][2]; J33 = gx[2][2];
      double* Jo = Jio + vJ.base;
      ecmdev::stg(Jo, J11); ecmdev::stg(Jo + QS, J21); ecmdev::stg(Jo + 2 * QS, J31); ecmdev::stg(Jo + 3 * QS, J12); ecmdev::stg(Jo + 4 * QS, J22); ecmdev::stg(Jo + 5 * QS, J32);
      ecmdev::stg(Jo + 6 * QS, J13); ecmdev::stg(Jo + 7 * QS, J23); ecmdev::stg(Jo + 8 * QS, J33);
      const double detJ = J11 * (J22 * J33 - J32 * J23) - J21 * (J12 * J33 - J32 * J13) + J31 * (J12 * J23 - J22 * J13);
      const double di = 1.0 / detJ;
      const double Ji[3][3] = { { di * (J22 * J33 - J23 * J32), di * (J32 * J13 - J12 * J33), di * (J12 * J23 - J22 * J13) },
                                { di * (J31 * J23 - J21 * J33), di * (J11 * J33 - J13 * J31), di * (J21 * J13 - J11 * J23) },
                                { di * (J21 * J32 - J31 * J22), di * (J31 * J12 - J11 * J32), di * (J11 * J22 - J12 * J21) } };
      p2::gather(r, [&](int a, int c) { return vel[gn[a] + (int64_t)nnodes * c]; }, gx);   // dv_c / dxi_s
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int t = 0; t < 3; t++) L[c + 3 * t] = gx[c][0] * Ji[0][t] + gx[c][1] * Ji[1][t] + gx[c][2] * Ji[2][t];
   } else if constexpr (LVEC && NFIX == 8) {
      // Trilinear fused launch: connectivity once, then the 24 coordinates AND the 24 velocities of the element in one round trip, then the
      // arithmetic.  (Written as two node loops - Jacobian, velocity gradient - the compiler re-read the connectivity for the second one and
      // issued the velocity gathers behind the Jacobian stores: five dependent memory round trips before the first state value was used.)
      const int32_t* ce = conn + (int64_t)8 * e;
      int gi[8];
#pragma unroll
      for (int r = 0; r < 8; r++) gi[r] = ce[r];
      double xs[3][8], vs[3][8];
#pragma unroll
      for (int r = 0; r < 8; r++) {
         xs[0][r] = xl[gi[r]]; xs[1][r] = xl[gi[r] + nnodes]; xs[2][r] = xl[gi[r] + 2 * (int64_t)nnodes];
         vs[0][r] = vel[gi[r]]; vs[1][r] = vel[gi[r] + nnodes]; vs[2][r] = vel[gi[r] + 2 * (int64_t)nnodes];
      }
      double Jc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, Lx[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };   // Jc[i + 3 j] = dx_i/dxi_j, Lx[c + 3 s] = dv_c/dxi_s
      double wq = 0.0;
      auto contract = [&](auto&& Gv) {
#pragma unroll
         for (int r = 0; r < 8; r++) {
            const double g0 = Gv(r), g1 = Gv(r + 8), g2 = Gv(r + 16);
#pragma unroll
            for (int c = 0; c < 3; c++) {
               Jc[c] += xs[c][r] * g0; Jc[c + 3] += xs[c][r] * g1; Jc[c + 6] += xs[c][r] * g2;
               Lx[c] += vs[c][r] * g0; Lx[c + 3] += vs[c][r] * g1; Lx[c + 6] += vs[c][r] * g2;
            }
         }
      };
      if (SROW && rows_by_wave) {
         const int qu = __builtin_amdgcn_readfirstlane(q);
         const p2::cptr Gc = p2::as_const(G) + 24 * qu;
         if (REC) wq = p2::as_const(Wq)[qu];
         contract([&](int i) { return Gc[i]; });
      } else {
         if (REC) wq = Wq[q];
         contract([&](int i) { return Gq[i]; });
      }
      J11 = Jc[0]; J21 = Jc[1]; J31 = Jc[2]; J12 = Jc[3]; J22 = Jc[4]; J32 = Jc[5]; J13 = Jc[6]; J23 = Jc[7]; J33 = Jc[8];
      if (Jio) {   // optional (uniform): the driver's p = 1 record route needs no Jacobian field - its integrator kernels take the geometry from the nodes
         double* Jo = Jio + vJ.base;
         ecmdev::stg(Jo, J11); ecmdev::stg(Jo + QS, J21); ecmdev::stg(Jo + 2 * QS, J31); ecmdev::stg(Jo + 3 * QS, J12); ecmdev::stg(Jo + 4 * QS, J22); ecmdev::stg(Jo + 5 * QS, J32);
         ecmdev::stg(Jo + 6 * QS, J13); ecmdev::stg(Jo + 7 * QS, J23); ecmdev::stg(Jo + 8 * QS, J33);
      }
      const double detJ = J11 * (J22 * J33 - J32 * J23) - J21 * (J12 * J33 - J32 * J13) + J31 * (J12 * J23 - J22 * J13);
      const double di = 1.0 / detJ;
      if (REC) tsc = dt * wq * di;
      const double Ji[3][3] = { { di * (J22 * J33 - J23 * J32), di * (J32 * J13 - J12 * J33), di * (J12 * J23 - J22 * J13) },
                                { di * (J31 * J23 - J21 * J33), di * (J11 * J33 - J13 * J31), di * (J21 * J13 - J11 * J23) },
                                { di * (J21 * J32 - J31 * J22), di * (J31 * J12 - J11 * J32), di * (J11 * J22 - J12 * J21) } };
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int tt = 0; tt < 3; tt++) L[c + 3 * tt] = Lx[c] * Ji[0][tt] + Lx[c + 3] * Ji[1][tt] + Lx[c + 6] * Ji[2][tt];
   } else {
   if (LVEC) {
      // J(i,j) = sum_r x_r,i dN_r/dxi_j   (column-major 3x3 per point, like MFEM's geometric factors after the re-layout)
      J11 = J21 = J31 = J12 = J22 = J32 = J13 = J23 = J33 = 0.0;
      const int32_t* ce = conn + (int64_t)n * e;
#pragma unroll
      for (int r = 0; r < n; r++) {
         const int g = ce[r];
         const double x0 = xl[g], x1 = xl[g + nnodes], x2 = xl[g + 2 * (int64_t)nnodes];
         const double g0 = Gq[r], g1 = Gq[r + n], g2 = Gq[r + 2 * n];
         J11 += x0 * g0; J21 += x1 * g0; J31 += x2 * g0;
         J12 += x0 * g1; J22 += x1 * g1; J32 += x2 * g1;
         J13 += x0 * g2; J23 += x1 * g2; J33 += x2 * g2;
      }
      if (Jio) {   // optional (uniform): the driver's p = 1 record route needs no Jacobian field - its integrator kernels take the geometry from the nodes
         double* Jo = Jio + vJ.base;
         ecmdev::stg(Jo, J11); ecmdev::stg(Jo + QS, J21); ecmdev::stg(Jo + 2 * QS, J31); ecmdev::stg(Jo + 3 * QS, J12); ecmdev::stg(Jo + 4 * QS, J22); ecmdev::stg(Jo + 5 * QS, J32);
         ecmdev::stg(Jo + 6 * QS, J13); ecmdev::stg(Jo + 7 * QS, J23); ecmdev::stg(Jo + 8 * QS, J33);
      }
   } else {
      const double* Jq = Jio + vJ.base;
      J11 = Jq[0]; J21 = Jq[QS]; J31 = Jq[2 * QS]; J12 = Jq[3 * QS]; J22 = Jq[4 * QS]; J32 = Jq[5 * QS]; J13 = Jq[6 * QS]; J23 = Jq[7 * QS]; J33 = Jq[8 * QS];
   }
   // inverse Jacobian (reference src/mechanics_kernels.cpp:38-61)
   const double detJ = J11 * (J22 * J33 - J32 * J23) - J21 * (J12 * J33 - J32 * J13) + J31 * (J12 * J23 - J22 * J13);
   const double di = 1.0 / detJ;
   if (REC) tsc = dt * Wq[q] * di;
   // Ji[s][t] = dxi_s/dx_t
   const double Ji[3][3] = { { di * (J22 * J33 - J23 * J32), di * (J32 * J13 - J12 * J33), di * (J12 * J23 - J22 * J13) },
                             { di * (J31 * J23 - J21 * J33), di * (J11 * J33 - J13 * J31), di * (J21 * J13 - J11 * J23) },
                             { di * (J21 * J32 - J31 * J22), di * (J31 * J12 - J11 * J32), di * (J11 * J22 - J12 * J21) } };
   // velocity gradient L(c,t) = sum_r v(r,c) dN_r/dx_t = (sum_r v(r,c) dN_r/dxi_s) dxi_s/dx_t: the reference-space gradient first
   // (9 multiply-adds per node), then one 3 x 3 product - instead of pushing every node's shape gradient through J^-1 (18 per node)
   const double* ve = vel + (int64_t)3 * n * e;
   const int32_t* ce = conn + (int64_t)n * e;
   double Lx[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };   // Lx[c + 3 s] = d v_c / d xi_s
#pragma unroll
   for (int r = 0; r < n; r++) {
      const double g0 = Gq[r], g1 = Gq[r + n], g2 = Gq[r + 2 * n];
      double v0, v1, v2;
      if (LVEC) { const int g = ce[r]; v0 = vel[g]; v1 = vel[g + nnodes]; v2 = vel[g + 2 * (int64_t)nnodes]; }
      else { v0 = ve[r]; v1 = ve[r + n]; v2 = ve[r + 2 * n]; }
      Lx[0] += v0 * g0; Lx[1] += v1 * g0; Lx[2] += v2 * g0;
      Lx[3] += v0 * g1; Lx[4] += v1 * g1; Lx[5] += v2 * g1;
      Lx[6] += v0 * g2; Lx[7] += v1 * g2; Lx[8] += v2 * g2;
   }
#pragma unroll
   for (int c = 0; c < 3; c++)
#pragma unroll
      for (int tt = 0; tt < 3; tt++) L[c + 3 * tt] = Lx[c] * Ji[0][tt] + Lx[c + 3] * Ji[1][tt] + Lx[c + 6] * Ji[2][tt];
   }
   // per-thread stash behind the shape table in LDS: slot s of this thread at stash[s * ECM_STASH_STRIDE + threadIdx.x] (PointIO::stash)
   const int rc = point_update<KIN, QS, REC>(mp, dt, L, io, kcap, pin, sG + pqo, tsc, trd != 0,
                                             TailIO{ tail_out, rs_out, tail_mode ? rs_in : nullptr, P, !tail_mode && rs_out != nullptr });
   if (rc == 1) atomicAdd(fail, 1);
}

template <bool QB>
__global__ void k_init_state(const int Q, const int64_t P, const double* __restrict__ hist, const double* __restrict__ quats, double* __restrict__ state0) {
   const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (ip >= P) return;
   const int64_t e = ip / Q;
   const QView v = qview<QB>(NSTATEV, Q, e, (int)(ip % Q));
   double* sv = state0 + v.base; const int64_t st = v.stride;
   for (int i = 0; i < NUM_HIST; i++) sv[i * st] = hist[i];
   for (int i = 0; i < 4; i++) sv[(H_Q + i) * st] = quats[4 * e + i];
   sv[IND_VOL * st] = 1.0; sv[IND_EINT * st] = 0.0;
}

// state slot 0 (effective shear rate) rebuilt from the stored slip rates: the invariant the hardness update relies on when it reads the
// begin-of-step rate from slot 0 (include/exaconstit_hip.h, "State layout"); for states that were not written by this library
template <bool QB>
__global__ void k_state_normalize(const int Q, const int64_t P, double* __restrict__ state) {
   const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (ip >= P) return;
   const QView v = qview<QB>(NSTATEV, Q, ip / Q, (int)(ip % Q));
   double* sv = state + v.base; const int64_t st = v.stride;
   double sum = 0.0;
#pragma unroll
   for (int a = 0; a < NSLIP; a++) sum += fabs(sv[(H_GDOT + a) * st]);
   sv[H_SHRATE * st] = sum;
}

// calcDpMat (reference src/mechanics_ecmech.hpp:315-356)
template <bool QB>
__global__ void k_calc_dp(const double qsign, const int Q, const int64_t P, const double* __restrict__ state, double* __restrict__ dp) {
   const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (ip >= P) return;
   const QView vs = qview<QB>(NSTATEV, Q, ip / Q, (int)(ip % Q)), vd = qview<QB>(9, Q, ip / Q, (int)(ip % Q));
   const double* sv = state + vs.base; const int64_t ss = vs.stride, ds = vd.stride;
   double dphat[5] = { 0, 0, 0, 0, 0 };
#pragma unroll
   for (int a = 0; a < NSLIP; a++) {
      const double g = sv[(H_GDOT + a) * ss];
#pragma unroll
      for (int c = 0; c < 5; c++) dphat[c] += P_TAB[c][a] * g;
   }
   (void)qsign;
   const double q[4] = { sv[H_Q * ss], sv[(H_Q + 1) * ss], sv[(H_Q + 2) * ss], sv[(H_Q + 3) * ss] };
   double C[9]; quat_to_mat(q, C);
   double sm[5]; rot_vecd(C, dphat, sm);
   double t00, t11, t22, t01, t02, t12; vecd_to_sym(sm, t00, t11, t22, t01, t02, t12);
   double* o = dp + vd.base;
   o[0] = t00; o[ds] = t01; o[2 * ds] = t02; o[3 * ds] = t01; o[4 * ds] = t11; o[5 * ds] = t12; o[6 * ds] = t02; o[7 * ds] = t12; o[8 * ds] = t22;
}

// histogram of the local-solver evaluation counts (state slot 3) of a state array: input of the tail-split controller
template <bool QB>
__global__ void k_nfev_hist(const int Q, const int64_t P, const double* __restrict__ state, int* __restrict__ hist /*64*/) {
   __shared__ int sh[64];
   if (threadIdx.x < 64) sh[threadIdx.x] = 0;
   __syncthreads();
   for (int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ip < P; ip += (int64_t)gridDim.x * blockDim.x) {
      const QView v = qview<QB>(NSTATEV, Q, ip / Q, (int)(ip % Q));
      const double nf = state[v.base + (int64_t)H_NFEV * v.stride];
      const int b = nf < 0.0 ? 0 : (nf > 63.0 ? 63 : (int)nf);
      atomicAdd(&sh[b], 1);
   }
   __syncthreads();
   if (threadIdx.x < 64 && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

// self-test hook of the kinetics' own elementary functions (ecm_device.hpp: exp_n, log_near1): out[i] = exp_n(x[i]), out[n + i] = log_near1(x[i])
__global__ void k_selftest_km_math(const double* __restrict__ x, double* __restrict__ out, const int n) {
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   double v[1] = { x[i] };
   ecmdev::exp_n<1, true>(v);
   out[i] = v[0]; out[n + i] = ecmdev::log_near1(x[i]);
}
int exa_launch_selftest_km_math(const double* x, double* out, int n, hipStream_t s) {
   hipLaunchKernelGGL(k_selftest_km_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, n);
   return hipGetLastError() == hipSuccess ? EXA_OK : EXA_ERR_HIP;
}

int exa_launch_nfev_hist(exa_ctx* ctx, const double* state, int* hist_dev, hipStream_t s) {
   EXA_HIP_CHECK(ctx, hipMemsetAsync(hist_dev, 0, sizeof(int) * 64, s));
   if (ctx->qblk) hipLaunchKernelGGL(k_nfev_hist<true>, dim3(1024), dim3(256), 0, s, ctx->Q, ctx->P, state, hist_dev);
   else hipLaunchKernelGGL(k_nfev_hist<false>, dim3(1024), dim3(256), 0, s, ctx->Q, ctx->P, state, hist_dev);
   EXA_HIP_CHECK(ctx, hipGetLastError());
   return EXA_OK;
}

// Launch sequence of the tail split (include/exaconstit_hip.h, exa_set_newton_caps): the full launch stops a point after newton_cap evaluations
// and lists it; a dense launch of the same kernel (thread = listed point; its grid covers the worst case, blocks beyond the list leave at
// once) takes the list up - from the saved solver state when the context holds the state buffers, from scratch otherwise - and, with a
// second cap, lists what it cuts off itself for a third launch.
// dynamic LDS of a launch of k_model_setup: shape rows (per wave for the element-blocked full launch, the whole table otherwise) + stash + slip table
// The largest shape table that passes exa_g_in_lds is the p = 2 one (27 * 3 * 27 doubles; p = 3 has 64 * 3 * 64 > EXA_G_LDS_MAX_DOUBLES and stays
// in global memory): with the stash and the slip table it must fit the 64 KB a launch gets without an attribute change.
static_assert(64 * 3 * 64 > EXA_G_LDS_MAX_DOUBLES, "order-3 shape tables must not be staged in LDS by the constitutive launch");
static_assert(sizeof(double) * ((size_t)27 * 3 * 27 + (size_t)ecmdev::ST_SLOTS * ECM_STASH_STRIDE + 8 * ecmdev::NSLIP) <= 65536, "dynamic LDS of k_model_setup exceeds 64 KB");
static size_t model_lds_bytes(const exa_ctx* ctx, bool km, bool p2f, bool qb, int tail_mode, bool lvec8) {
   const int nrow = (qb && !tail_mode) ? EXA_MODEL_BS / 64 : ctx->Q;
   const bool srow = lvec8 && qb && !tail_mode;      // k_model_setup, SROW: the wave's shape row comes through scalar loads
   const size_t rows = (p2f || srow || !exa_g_in_lds(ctx->n, nrow)) ? 0 : (size_t)ctx->n * 3 * nrow;
   return sizeof(double) * (rows + (size_t)ecmdev::ST_SLOTS * ECM_STASH_STRIDE + (km ? (size_t)8 * ecmdev::NSLIP : (size_t)0));
}

template <typename Go>
static void launch_levels(exa_ctx* ctx, int64_t nb, Go&& go) {
   const int NOCAP = 1 << 30;
   const bool split = ctx->newton_cap > 0 && ctx->tail_dev != nullptr;
   if (!split) { go(nb, NOCAP, ctx->tail_dev, 0, (int*)nullptr, (const double*)nullptr, (double*)nullptr); return; }
   const bool two = ctx->newton_cap2 > ctx->newton_cap && ctx->tail2_dev != nullptr && ctx->resume_dev[0] && ctx->resume_dev[1];
   const int64_t nbt = (ctx->P + EXA_MODEL_BS - 1) / EXA_MODEL_BS;
   go(nb, ctx->newton_cap, ctx->tail_dev, 0, ctx->tail_dev, (const double*)nullptr, ctx->resume_dev[0]);
   go(nbt, two ? ctx->newton_cap2 : NOCAP, ctx->tail_dev, 1, two ? ctx->tail2_dev : ctx->tail_dev, (const double*)ctx->resume_dev[0], two ? ctx->resume_dev[1] : (double*)nullptr);
   if (two) go(nbt, NOCAP, ctx->tail2_dev, 1, ctx->tail2_dev, (const double*)ctx->resume_dev[1], (double*)nullptr);
}

template <int KIN, bool LVEC, int NFIX, bool QB>
static void launch_model_q(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                         double* stress1, double* state1, double* cmat, hipStream_t s) {
   const int bs = EXA_MODEL_BS;
   // QB: one wave per (64-element block, q)
   const int64_t nb = QB ? (((int64_t)((ctx->E + 63) / 64) * ctx->Q) + (bs / 64) - 1) / (bs / 64) : (ctx->P + bs - 1) / bs;
   const double* G = NFIX == 27 ? ctx->T1_dev : ctx->G_dev;
   launch_levels(ctx, nb, [&](int64_t blocks, int kcap, int* list, int mode, int* list_out, const double* rs_in, double* rs_out) {
      hipLaunchKernelGGL((k_model_setup<KIN, LVEC, NFIX, QB>), dim3((unsigned)blocks), dim3(bs), model_lds_bytes(ctx, ecmdev::kin_is_km(KIN), NFIX == 27, QB, mode, LVEC && NFIX == 8), s, ctx->mp, ctx->Q, ctx->n, ctx->P, dt, J, G, vel, xl, ctx->conn,
                         ctx->nnodes, stress0, state0, stress1, state1, cmat, ctx->fail_count_dev, kcap, list, mode, (const double*)nullptr, 0, list_out, rs_in, rs_out);
   });
}

// fused p = 1 launch that writes the compact gradient records (exa_model_setup_lvec_records)
template <int KIN>
static void launch_model_rec(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                             double* stress1, double* state1, hipStream_t s) {
   const int bs = EXA_MODEL_BS;
   const int64_t nb = (((int64_t)((ctx->E + 63) / 64) * ctx->Q) + (bs / 64) - 1) / (bs / 64);
   const int trd = ctx->cfg.assembly == EXA_ASSEMBLY_EA;
   launch_levels(ctx, nb, [&](int64_t blocks, int kcap, int* list, int mode, int* list_out, const double* rs_in, double* rs_out) {
      hipLaunchKernelGGL((k_model_setup<KIN, true, 8, true, true>), dim3((unsigned)blocks), dim3(bs), model_lds_bytes(ctx, ecmdev::kin_is_km(KIN), false, true, mode, true), s, ctx->mp, ctx->Q, ctx->n, ctx->P, dt, J, ctx->G_dev, vel, xl, ctx->conn,
                         ctx->nnodes, stress0, state0, stress1, state1, ctx->pa_c, ctx->fail_count_dev, kcap, list, mode, ctx->W_dev, trd, list_out, rs_in, rs_out);
   });
}

// Kocks-Mecking sets with thermal-activation exponents p == q == 1 (the shipped sets) run the instantiation that has the two exponents
// compiled in (ecmdev::KIN_PQ1: same arithmetic, no pow() code); EXA_KM_PQ1=off keeps the general instantiation for A/B runs
static bool km_pq1(const exa_ctx* ctx) {
   const char* e = std::getenv("EXA_KM_PQ1");   // read per launch: the tests flip it inside one process
   // (the instantiation has the short-series logarithm of the power-law tail compiled in: ecm_device.hpp, kmbald_gdot4)
   return !(e && std::strcmp(e, "off") == 0) && ctx->mp.p == 1.0 && ctx->mp.q == 1.0 && ctx->mp.xn_int == 0 && ctx->mp.t_min >= 0.75 && ctx->mp.t_max <= 1.25;
}

// Voce sets whose power-law exponent 1/m - 1 is 49 (m = 0.02: the shipped sets) run the instantiation with the exponent compiled in
// (ecmdev::KIN_XN49: same multiplication chain, no run-time choice among the power forms inside every evaluation); EXA_VOCE_XN_CT=off keeps the
// general instantiation for A/B runs.  Element-blocked fused launches only (the driver's routes).
static bool voce_xn49(const exa_ctx* ctx) {
   const char* e = std::getenv("EXA_VOCE_XN_CT");
   return !(e && std::strcmp(e, "off") == 0) && ctx->mp.xn_int == 49;
}

// lists and solver-state buffers of the tail split, allocated on first use; the list counters are cleared for the coming launch sequence
static int exa_prepare_tail_lists(exa_ctx* ctx, hipStream_t s) {
   if (!ctx->tail_dev) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->tail_dev, sizeof(int) * ((size_t)ctx->P + 1)));
   EXA_HIP_CHECK(ctx, hipMemsetAsync(ctx->tail_dev, 0, sizeof(int), s));
   if (ctx->tail_resume) {
      if (!ctx->resume_dev[0]) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->resume_dev[0], sizeof(double) * ecmdev::RS_N * (size_t)ctx->P));
      if (ctx->newton_cap2 > ctx->newton_cap) {
         if (!ctx->tail2_dev) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->tail2_dev, sizeof(int) * ((size_t)ctx->P + 1)));
         if (!ctx->resume_dev[1]) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->resume_dev[1], sizeof(double) * ecmdev::RS_N * (size_t)ctx->P));
         EXA_HIP_CHECK(ctx, hipMemsetAsync(ctx->tail2_dev, 0, sizeof(int), s));
      }
   } else if (ctx->resume_dev[0]) {   // switched off after use: the launches test the pointers
      (void)hipFree(ctx->resume_dev[0]); (void)hipFree(ctx->resume_dev[1]); ctx->resume_dev[0] = ctx->resume_dev[1] = nullptr;
   }
   return EXA_OK;
}

int exa_launch_model_setup_rec(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                               double* stress1, double* state1, hipStream_t s) {
   EXA_HIP_CHECK(ctx, hipMemsetAsync(ctx->fail_count_dev, 0, sizeof(int), s));
   if (ctx->newton_cap > 0) {
      if (int rc = exa_prepare_tail_lists(ctx, s)) return rc;
   }
   switch (ctx->mp.kin) {
      case KIN_VOCE:
         if (voce_xn49(ctx)) launch_model_rec<KIN_VOCE | KIN_XN49>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, s);
         else launch_model_rec<KIN_VOCE>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, s);
         break;
      case KIN_VOCE_NL:
         if (voce_xn49(ctx)) launch_model_rec<KIN_VOCE_NL | KIN_XN49>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, s);
         else launch_model_rec<KIN_VOCE_NL>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, s);
         break;
#ifdef EXA_VARIANT_VOCE_ONLY   // timing builds (make variant): the Kocks-Mecking instantiations are left out
      default: ctx->err = "variant build without Kocks-Mecking kernels"; return EXA_ERR_UNSUPPORTED;
#else
      default:
         if (km_pq1(ctx)) {
            if (ECM_KM_DEFER && ctx->mp.with_g_athermal) launch_model_rec<KIN_KMBALD_GA | KIN_PQ1>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, s);
            else launch_model_rec<KIN_KMBALD | KIN_PQ1>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, s);
         } else if (ECM_KM_DEFER && ctx->mp.with_g_athermal) launch_model_rec<KIN_KMBALD_GA>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, s);
         else launch_model_rec<KIN_KMBALD>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, s);
         break;
#endif
   }
   EXA_HIP_CHECK(ctx, hipGetLastError());
   return EXA_OK;
}

template <int KIN, bool LVEC, int NFIX>
static void launch_model_n(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                           double* stress1, double* state1, double* cmat, hipStream_t s) {
   if (ctx->qblk) launch_model_q<KIN, LVEC, NFIX, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
   else launch_model_q<KIN, LVEC, NFIX, false>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
}

template <int KIN, bool LVEC>
static void launch_model(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                         double* stress1, double* state1, double* cmat, hipStream_t s) {
   if (LVEC && ctx->n == 8) launch_model_n<KIN, LVEC, 8>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
   else if (LVEC && ctx->n == 27 && ctx->qblk) launch_model_q<KIN, true, 27, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);   // T1 uploaded by the caller
   else launch_model_n<KIN, LVEC, 0>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
}

// xl == nullptr: J is an input and vel an E-vector; otherwise xl / vel are L-vectors (byNODES) and J is written
int exa_launch_model_setup(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                           double* stress1, double* state1, double* cmat, hipStream_t s) {
   EXA_HIP_CHECK(ctx, hipMemsetAsync(ctx->fail_count_dev, 0, sizeof(int), s));
   if (ctx->newton_cap > 0) {
      if (int rc = exa_prepare_tail_lists(ctx, s)) return rc;
   }
   const bool lv = xl != nullptr;
   if (lv && ctx->n == 27 && ctx->qblk) { if (int rc = exa_ensure_p2_tables(ctx)) return rc; }
   switch (ctx->mp.kin) {
      case KIN_VOCE:
         if (lv && ctx->qblk && (ctx->n == 8 || ctx->n == 27) && voce_xn49(ctx)) {   // element-blocked fused launches with the exponent compiled in
            if (ctx->n == 8) launch_model_q<KIN_VOCE | KIN_XN49, true, 8, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
            else launch_model_q<KIN_VOCE | KIN_XN49, true, 27, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         } else if (lv) launch_model<KIN_VOCE, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         else launch_model<KIN_VOCE, false>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         break;
      case KIN_VOCE_NL:
         if (lv && ctx->qblk && (ctx->n == 8 || ctx->n == 27) && voce_xn49(ctx)) {
            if (ctx->n == 8) launch_model_q<KIN_VOCE_NL | KIN_XN49, true, 8, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
            else launch_model_q<KIN_VOCE_NL | KIN_XN49, true, 27, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         } else if (lv) launch_model<KIN_VOCE_NL, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         else launch_model<KIN_VOCE_NL, false>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         break;
#ifdef EXA_VARIANT_VOCE_ONLY
      default: ctx->err = "variant build without Kocks-Mecking kernels"; return EXA_ERR_UNSUPPORTED;
#else
      default:
         if (lv && ctx->n == 8 && ctx->qblk && km_pq1(ctx)) {   // p = 1 element-blocked route with the tangent field (Jacobi / element-assembly set-ups)
            if (ECM_KM_DEFER && ctx->mp.with_g_athermal) launch_model_q<KIN_KMBALD_GA | KIN_PQ1, true, 8, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
            else launch_model_q<KIN_KMBALD | KIN_PQ1, true, 8, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         } else if (ECM_KM_DEFER && ctx->mp.with_g_athermal) {   // athermal-threshold variant (BCC): instantiation with the deferred window systems
            if (lv) launch_model<KIN_KMBALD_GA, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
            else launch_model<KIN_KMBALD_GA, false>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         } else {
            if (lv) launch_model<KIN_KMBALD, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
            else launch_model<KIN_KMBALD, false>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         }
         break;
#endif
   }
   EXA_HIP_CHECK(ctx, hipGetLastError());
   return EXA_OK;
}

int exa_launch_init_state(exa_ctx* ctx, double* state0, const double* quats, const double* hist_dev, hipStream_t s) {
   const int bs = 256;
   if (ctx->qblk) hipLaunchKernelGGL(k_init_state<true>, dim3((unsigned)((ctx->P + bs - 1) / bs)), dim3(bs), 0, s, ctx->Q, ctx->P, hist_dev, quats, state0);
   else hipLaunchKernelGGL(k_init_state<false>, dim3((unsigned)((ctx->P + bs - 1) / bs)), dim3(bs), 0, s, ctx->Q, ctx->P, hist_dev, quats, state0);
   EXA_HIP_CHECK(ctx, hipGetLastError());
   return EXA_OK;
}

int exa_launch_state_normalize(exa_ctx* ctx, double* state, hipStream_t s) {
   const int bs = 256;
   if (ctx->qblk) hipLaunchKernelGGL(k_state_normalize<true>, dim3((unsigned)((ctx->P + bs - 1) / bs)), dim3(bs), 0, s, ctx->Q, ctx->P, state);
   else hipLaunchKernelGGL(k_state_normalize<false>, dim3((unsigned)((ctx->P + bs - 1) / bs)), dim3(bs), 0, s, ctx->Q, ctx->P, state);
   EXA_HIP_CHECK(ctx, hipGetLastError());
   return EXA_OK;
}

int exa_launch_calc_dp(exa_ctx* ctx, const double* state, double* dp, hipStream_t s) {
   const int bs = 256;
   if (ctx->qblk) hipLaunchKernelGGL(k_calc_dp<true>, dim3((unsigned)((ctx->P + bs - 1) / bs)), dim3(bs), 0, s, ctx->mp.qsign, ctx->Q, ctx->P, state, dp);
   else hipLaunchKernelGGL(k_calc_dp<false>, dim3((unsigned)((ctx->P + bs - 1) / bs)), dim3(bs), 0, s, ctx->mp.qsign, ctx->Q, ctx->P, state, dp);
   EXA_HIP_CHECK(ctx, hipGetLastError());
   return EXA_OK;
}
