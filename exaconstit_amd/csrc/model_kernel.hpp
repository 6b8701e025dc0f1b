// The fused constitutive kernel and its launch helpers, shared by the two translation units that instantiate it (model_kernels.hip: the element-blocked
// and per-lane AOS launches; model_kernels_aos.hip: the staged AOS launches).  See model_kernels.hip for the reference functions the launch replaces.
#pragma once
#include "exa_internal.hpp"
#include "p2_basis.hpp"
#include <cstdlib>
#include <cstring>

using namespace ecmdev;

#ifndef EXA_MODEL_OCC
#define EXA_MODEL_OCC 2   // waves per SIMD the register allocator is asked to fit (tuned on MI355X)
#endif
#ifndef EXA_MODEL_BS
#define EXA_MODEL_BS 128   // threads per block of the constitutive launch (4 blocks of 2 waves per CU: 38.9 KB of stash each; a block waits for 2 waves, not 4)
#endif
static_assert(EXA_MODEL_BS % ECM_STASH_STRIDE == 0 && ECM_STASH_STRIDE % 64 == 0, "a block is a whole number of stash regions, a region a whole number of waves");
constexpr int EXA_STASH_DOUBLES = ecmdev::ST_SLOTS * EXA_MODEL_BS;   // LDS stash of a block

#ifndef EXA_STG_NT_LD
#define EXA_STG_NT_LD 1   // row pieces of the staged launch: non-temporal loads / stores like the per-lane accesses of the other launches (A/B switches)
#endif
#ifndef EXA_STG_NT_ST
#define EXA_STG_NT_ST 1   // (every round stores whole 128-byte lines: with the tangent in 144-byte pieces - rounds of 3 columns - plain stores were the faster ones)
#endif
__device__ __forceinline__ double2 ldg2(const double* p) {
#if EXA_STG_NT_LD
   typedef double vd2 __attribute__((ext_vector_type(2)));
   const vd2 v = __builtin_nontemporal_load(reinterpret_cast<const vd2*>(p));
   return make_double2(v.x, v.y);
#else
   return *reinterpret_cast<const double2*>(p);
#endif
}
// Row strides (doubles) in the stage: the rows lie there as in memory.  (Measured: rows one double apart - odd strides 29 / 7, no LDS bank conflicts of the lanes'
// 8-byte row accesses, two 8-byte LDS accesses per 16-byte piece - are slower, 4.87 against 4.83 ms at 128^3: the row accesses are not what the launch waits for.)
constexpr int RS_SV = ecmdev::NSTATEV, RS_S = 6, RS_T = 36;
__device__ __forceinline__ void stg2r(double* p, const double a, const double b) {
#if EXA_STG_NT_ST
   ecmdev::stg2(reinterpret_cast<double2*>(p), a, b);
#else
   *reinterpret_cast<double2*>(p) = make_double2(a, b);
#endif
}
// Where one thread's quadrature point lives.  Everything is a function of (block index, thread index) and kernel-uniform data, so nothing
// per-lane has to survive the local Newton solve: locate() derives the point from the thread index, refresh() does it again behind a compiler
// barrier after the solve, and the accessors form the addresses where they are used.  Before, five 64-bit row pointers and two LDS addresses
// were live across the solve; the register allocator spilled some of them and re-loaded them from scratch BEHIND the first output stores,
// where a load waits for the whole store queue of the wave (ecm_device.hpp, ECM_EPI_NO_LOADS).
// STG (staged AOS launch, QB = false): a wave owns the 64 CONSECUTIVE points pt0() ... pt0() + 63, whose rows are contiguous in every (vdim, Q, E) array.
// The wave's stash region (ST_SLOTS x 64 doubles of LDS) doubles as its transposition buffer: rows come in through coalesced 16-byte loads and are
// read back by their lanes (model_kernel.hpp, stage_in); outputs are written by their lanes as rows - sv1() / s1() / cm() point into the region - and
// leave through coalesced 16-byte stores (flush_tangent, flush_state).  A lane beyond the last point repeats the last point (live = false) and
// stores nothing: all 64 lanes take part in every round.  The dense launches of a tail split run the SAME kernel (tail_mode, so that a listed point is
// finished with the bits the full launch would have given it): their lanes are scattered points, which read their inputs per lane, write their
// output rows into the stage like everybody and copy them out per lane.
template <bool QB, bool REC, bool STG = false, int NPAIR = PAC_PAIRS>
struct PointIO {
   const double* state0; const double* stress0; double* state1; double* stress1; double* cmat;   // kernel-uniform array bases
   double* stash0;                  // LDS stash of the block
   const int* tail; int tail_mode;  // dense tail launch: thread t owns point tail[1 + t]
   int Q; int64_t bidx; int wpb;    // points per element, (remapped) block index, waves per block
   int q; int64_t e; int tid;       // this thread's point and its index in the block
   int64_t P; bool live;            // STG: points of the launch; this lane's point exists
   static_assert(!STG || (!QB && ECM_STASH_STRIDE == 64), "staged rows: AOS layout, per-wave stash regions");
   __device__ __forceinline__ int wave() const { return STG ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6); }   // (STG: wave-uniform row addresses in scalar registers)
   __device__ __forceinline__ int64_t pt0() const { return (bidx * wpb + wave()) * 64; }
   __device__ __forceinline__ int nvalid() const { const int64_t r = P - pt0(); return r < 64 ? (int)r : 64; }   // points of this wave that exist (may be <= 0)
   __device__ __forceinline__ void locate(const int tid_) {
      tid = tid_;
      if (STG && !tail_mode) { const int64_t ip = pt0() + (tid & 63); live = ip < P; const int64_t ie = live ? ip : P - 1; q = (int)(ie % Q); e = ie / Q; }
      else if (tail_mode) { const int64_t ipt = tail[1 + bidx * (int64_t)(wpb * 64) + tid]; q = (int)(ipt % Q); e = ipt / Q; }
      else if (QB) { const int64_t gw = bidx * wpb + (tid >> 6); q = (int)(gw % Q); e = (gw / Q) * 64 + (tid & 63); }   // wave = (block of 64 elements, q); lane = element
      else { const int64_t ip = bidx * (int64_t)(wpb * 64) + tid; q = (int)(ip % Q); e = ip / Q; }
   }
   // the wave's region of the stash (STG: its stage); row of the point this lane computes on input (the last existing one for a lane beyond the end)
   __device__ __forceinline__ double* wreg() const { return stash0 + wave() * (ecmdev::ST_SLOTS * 64); }
   __device__ __forceinline__ int row_in() const { return (int)(e * Q + q - pt0()); }
   // one round of coalesced row stores: n doubles of the stage at lds -> g (16-byte aligned), 16 bytes per lane and pass
   // (rows of W doubles RS apart in the stage; RS == W: the stage holds them as they lie in memory)
   template <int W, int RS, int NROWS = 64>
   __device__ __forceinline__ void flush_rows(const double* lds, double* g, const int n) const {
      const int lane = tid & 63;
#pragma unroll
      for (int m = 0; m < (NROWS * W / 2 + 63) / 64; m++) {
         const int u = m * 64 + lane;
         const int i0 = (RS == W) ? 2 * u : ((2 * u) / W) * RS + (2 * u) % W;      // (W even when RS != W: both doubles of a piece lie in one row)
         if (2 * u + 1 < n) stg2r(g + 2 * u, lds[i0], lds[i0 + 1]);
         else if (2 * u < n) ecmdev::stg(g + 2 * u, lds[i0]);      // (odd row counts: Jacobian rows of a wave that ends the array at an odd-order element)
      }
   }
   // the 36-double tangent rows of points 32 h ... 32 h + 31 of the wave (staged_tangent, ecm_device.hpp): 9 216 contiguous bytes
   __device__ __forceinline__ int half() const { return (tid >> 5) & 1; }
   __device__ __forceinline__ void flush_tangent(const int h) const {
      if (tail_mode) {   // (uniform) this lane's row to its own point
         if (half() != h) return;
         const double* row = wreg() + (tid & 31) * RS_T; double* g = cmat + (e * Q + q) * 36;
#pragma unroll
         for (int c = 0; c < 18; c++) stg2r(g + 2 * c, row[2 * c], row[2 * c + 1]);
         return;
      }
      const int nv = nvalid() - 32 * h;
      if (nv > 0) flush_rows<36, 36, 32>(wreg(), cmat + (pt0() + 32 * h) * 36, (nv < 32 ? nv : 32) * 36);
   }
   __device__ __forceinline__ void flush_state() const {
      if (tail_mode) {
         const double* row = wreg() + (tid & 63) * RS_SV; double* g = state1 + (e * Q + q) * ecmdev::NSTATEV;
#pragma unroll
         for (int c = 0; c < ecmdev::NSTATEV / 2; c++) ecmdev::stg2(reinterpret_cast<double2*>(g + 2 * c), row[2 * c], row[2 * c + 1]);
         const double* rs = wreg() + 64 * RS_SV + (tid & 63) * RS_S; double* gs = stress1 + (e * Q + q) * 6;
#pragma unroll
         for (int c = 0; c < 3; c++) ecmdev::stg2(reinterpret_cast<double2*>(gs + 2 * c), rs[2 * c], rs[2 * c + 1]);
         return;
      }
      const int nv = nvalid();
      flush_rows<ecmdev::NSTATEV, RS_SV>(wreg(), state1 + pt0() * ecmdev::NSTATEV, nv * ecmdev::NSTATEV);
      flush_rows<6, RS_S>(wreg() + 64 * RS_SV, stress1 + pt0() * 6, nv * 6);
   }
   __device__ __forceinline__ void refresh() { int t = threadIdx.x; asm volatile("" : "+v"(t)); locate(t); }
   __device__ __forceinline__ const double* sv0() const { return state0 + qview<QB>(ecmdev::NSTATEV, Q, e, q).base; }
   __device__ __forceinline__ const double* s0() const { return stress0 + qview<QB>(6, Q, e, q).base; }
   // (STG: the lane's rows of the stage - state 28, stress 6 behind the 64 state rows, tangent half 18)
   __device__ __forceinline__ double* sv1() const { return STG ? wreg() + (tid & 63) * RS_SV : state1 + qview<QB>(ecmdev::NSTATEV, Q, e, q).base; }
   __device__ __forceinline__ double* s1() const { return STG ? wreg() + 64 * RS_SV + (tid & 63) * RS_S : stress1 + qview<QB>(6, Q, e, q).base; }
   // REC: the lane's first 16-byte pair of its compact record ([block][q][13 pairs][64 lanes][2]); else the tangent slot
   __device__ __forceinline__ double* cm() const {
      if (STG && !REC) return wreg() + (tid & 31) * RS_T;      // (lanes l and l + 32 use the row one after the other)
      return REC ? cmat + pac_off<NPAIR>(e >> 6, Q, q, 0) + 2 * (e & 63) : cmat + qview<QB>(36, Q, e, q).base;
   }
   // slot s of this thread at stash()[s * ECM_STASH_STRIDE]: regions of ECM_STASH_STRIDE lanes, one behind the other
   __device__ __forceinline__ double* stash() const { return stash0 + (tid / ECM_STASH_STRIDE) * (ecmdev::ST_SLOTS * ECM_STASH_STRIDE) + (tid % ECM_STASH_STRIDE); }
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
// STG (reference layout only): the staged form - see PointIO.  Input rows (state 28, stress 6; without LVEC also the Jacobians 9, and for trilinear elements
// the velocity E-vector rows of the wave's 8 elements and the shape table) are requested as 16-byte pieces, contiguous across the wave, go through the
// wave's stage and are read back by the lanes that own them; the Jacobians of an LVEC launch and all outputs leave the same way.  No block barrier: a
// wave only ever touches its own region.  Never a tail launch (scattered points: the per-lane form takes those).
// n doubles at g (16-byte aligned) -> registers, 16 bytes per lane and pass, contiguous across the wave; NU = passes (2 * 64 * NU >= total doubles)
template <int NU>
__device__ __forceinline__ void rows_load(const double* __restrict__ g, const int n, const int lane, double2 (&r)[NU]) {
#pragma unroll
   for (int m = 0; m < NU; m++) {
      const int u = m * 64 + lane;
      r[m] = make_double2(0.0, 0.0);
      if (2 * u + 1 < n) r[m] = ldg2(g + 2 * u);
      else if (2 * u < n) r[m].x = ecmdev::ldg(g + 2 * u);
   }
}
// ... -> the stage (TOT = doubles the rows occupy there; pieces beyond are not written)
// (W, RS: rows of W doubles go RS apart; 0, 0: as they lie in memory)
template <int NU, int TOT, int W = 0, int RS = 0>
__device__ __forceinline__ void rows_to_stage(double* lds, const int lane, const double2 (&r)[NU]) {
#pragma unroll
   for (int m = 0; m < NU; m++) {
      const int u = m * 64 + lane;
      if (2 * NU * 64 <= TOT || 2 * u < TOT) {
         if constexpr (RS == W) *reinterpret_cast<double2*>(lds + 2 * u) = r[m];
         else { const int i0 = ((2 * u) / W) * RS + (2 * u) % W; lds[i0] = r[m].x; lds[i0 + 1] = r[m].y; }
      }
   }
}
constexpr int STG_J = 0, STG_V = 576, STG_G = 768;   // stage offsets (doubles) of the prologue: 64 Jacobian rows, 8 velocity E-vector rows, the trilinear shape table

template <int KIN, bool LVEC, int NFIX, bool QB, bool REC = false, bool STG = false>
__global__ __launch_bounds__(EXA_MODEL_BS, EXA_MODEL_OCC) void k_model_setup(const MatParams mp, const int Q, const int n_rt, const int64_t P, const double dt,
                                                     double* __restrict__ Jio, const double* __restrict__ G,
                                                     const double* __restrict__ vel, const double* __restrict__ xl, const int32_t* __restrict__ conn, const int nnodes,
                                                     const double* __restrict__ stress0,
                                                     const double* __restrict__ state0, double* __restrict__ stress1,
                                                     double* __restrict__ state1, double* __restrict__ cmat, int* __restrict__ fail,
                                                     const int kcap, int* __restrict__ tail, const int tail_mode_rt,
                                                     const double* __restrict__ Wq = nullptr, const int trd = 0,
                                                     int* __restrict__ tail_out = nullptr, const double* __restrict__ rs_in = nullptr, double* __restrict__ rs_out = nullptr) {
   // tail: list this launch works on (tail_mode) or appends to; tail_out: list a tail launch appends the points to that it cuts off itself
   // (second level); rs_in / rs_out: solver states of the listed points, [RS_N][P] by list slot (nullptr: a listed point starts over)
   if (!tail_out) tail_out = tail;
   // VG (p = 2, element-blocked): the geometry comes from the pre-pass (gen_kernels.hip, k_geom_p2) - Jio holds the Jacobians (input), `vel` the reference-space
   // velocity gradients dv_c/dxi_d, both (3,3,Q,E) - instead of 189 scattered loads per point.  With REC the launch writes the p = 2 record of the matrix-free
   // action (18 pairs: D, K, then adj(J) and W detJ - the latter five pairs right here in the prologue, where J is at hand).
   constexpr bool VG = !LVEC && NFIX == 27;
   static_assert(!VG || QB, "velocity-gradient input: element-blocked layout");
   // REC with STG (round 6): the fused p = 1 launch on the REFERENCE layout - state and stress rows staged, the record written per lane: for one pair the 8 points x
   // 8 consecutive elements of a wave store 8 whole 128-byte lines of the record array ([block][q][pair][lane = element])
   static_assert(!REC || (LVEC && NFIX == 8 && (QB || STG)) || (QB && VG), "record output: the fused p = 1 launches (element-blocked, or staged reference layout) and the p = 2 launch behind its geometry pre-pass");
   static_assert(!STG || (!QB && NFIX != 27), "staged rows: reference layout, generic or trilinear node loops");
   const int tail_mode = tail_mode_rt;
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
   constexpr bool GSTG = STG && NFIX == 8;   // staged trilinear launch: every wave keeps its own copy of the 1.5 KB table in its stage while it forms L (no block barrier, no LDS beyond the stash)
   const bool g_lds = !P2F && !GSTG && !(SROW && rows_by_wave) && exa_g_in_lds(n, rows_by_wave ? wpb : Q);   // orders above 2: the table stays in global memory (exa_internal.hpp)
   const int tab = g_lds ? n * 3 * (rows_by_wave ? wpb : Q) : 0;
   // Kocks-Mecking: the slip table (12 rows of 8) behind the stash, for the rows that are read by lane-varying index (ecm_device.hpp, eval_rj)
   const int pqo = tab + EXA_STASH_DOUBLES;
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
   PointIO<QB, REC, STG, (NFIX == 27) ? PAC_PAIRS_GEO : PAC_PAIRS> io{ state0, stress0, state1, stress1, cmat, sG + tab, tail, tail_mode, Q, bidx, wpb, 0, 0, 0, P, true };
   io.locate(threadIdx.x);
   const int q = io.q; const int64_t e = io.e;
   if (STG && !tail_mode) { if (io.nvalid() <= 0) return; }   // (wave-uniform: none of the wave's points exists)
   else if (!tail_mode && e * Q >= P) return;
   constexpr int QS = QB ? 64 : 1;
   // the point's begin-of-step state and stress are requested first: they depend on nothing but (e, q) and travel while the nodes are gathered
   ecmdev::PointIn pin;
   [[maybe_unused]] double2 rws[STG ? 14 : 1], rwt[STG ? 3 : 1];
   [[maybe_unused]] const int lane = (int)(threadIdx.x & 63);
   const double* Jq_in = nullptr; const double* ve_in = nullptr;   // !LVEC: this point's Jacobian row and its element's velocity E-vector row
   const bool staged = STG && !tail_mode;   // (kernel-uniform)
   if (STG && tail_mode) {   // dense launch of the staged kernel: scattered points, every input per lane (the 1.5 KB shape table from global memory: lanes
      // beyond the list have left, nothing here may need the whole wave; state and stress are requested behind the geometry - see below)
      if constexpr (!LVEC) { Jq_in = Jio + qview<QB>(9, Q, e, q).base; ve_in = vel + (int64_t)3 * n * e; }
   } else if constexpr (STG) {
      // (requested in the order of use - loads return in order: the geometry rows first, so that L is formed while the state rows are still on their way)
      const int nv = io.nvalid(); const int64_t p0 = io.pt0(); double* wr = io.wreg();
      [[maybe_unused]] double2 rg[2], rj[5], rv[2];
      if constexpr (GSTG) rows_load<2>(G, 192, lane, rg);
      if constexpr (!LVEC) {
         rows_load<5>(Jio + p0 * 9, nv * 9, lane, rj);
         if constexpr (NFIX == 8) rows_load<2>(vel + (p0 >> 3) * 24, (nv >> 3) * 24, lane, rv);   // Q = 8: the wave's 64 points are 8 whole elements, whose E-vector rows are contiguous
      }
      rows_load<14>(state0 + p0 * ecmdev::NSTATEV, nv * ecmdev::NSTATEV, lane, rws);
      rows_load<3>(stress0 + p0 * 6, nv * 6, lane, rwt);
      if constexpr (GSTG) rows_to_stage<2, 192>(wr + STG_G, lane, rg);
      if constexpr (!LVEC) {
         rows_to_stage<5, 576>(wr + STG_J, lane, rj);
         Jq_in = wr + STG_J + io.row_in() * 9;
         if constexpr (NFIX == 8) { rows_to_stage<2, 192>(wr + STG_V, lane, rv); ve_in = wr + STG_V + (io.row_in() >> 3) * 24; }
         else ve_in = vel + (int64_t)3 * n * e;
      }
      ECM_PARK_BARRIER(); __builtin_amdgcn_wave_barrier();
   } else {
      ecmdev::load_point_in<QS>(io.sv0(), io.s0(), pin);
      if constexpr (!LVEC) { Jq_in = Jio + qview<QB>(9, Q, e, q).base; ve_in = vel + (int64_t)3 * n * e; }
   }
   const QView vJ = qview<QB>(9, Q, e, q);
   const double* Gq = (GSTG && staged) ? io.wreg() + STG_G + 24 * q : (g_lds ? sG + 3 * n * (rows_by_wave ? (int)(threadIdx.x >> 6) : q) : G + 3 * n * q);
   // Jacobian output of an LVEC launch: the point's slot, or (STG) its row of the stage, stored by the wave once all 64 rows are there
   auto store_J = [&](const double a11, const double a21, const double a31, const double a12, const double a22, const double a32, const double a13, const double a23, const double a33) {
      if (STG && staged) {
         double* Jo = io.wreg() + STG_J + (int)(threadIdx.x & 63) * 9;
         Jo[0] = a11; Jo[1] = a21; Jo[2] = a31; Jo[3] = a12; Jo[4] = a22; Jo[5] = a32; Jo[6] = a13; Jo[7] = a23; Jo[8] = a33;
         ECM_PARK_BARRIER(); __builtin_amdgcn_wave_barrier();
         io.template flush_rows<9, 9>(io.wreg() + STG_J, Jio + io.pt0() * 9, io.nvalid() * 9);
      } else {
         double* Jo = Jio + vJ.base;
         ecmdev::stg(Jo, a11); ecmdev::stg(Jo + QS, a21); ecmdev::stg(Jo + 2 * QS, a31); ecmdev::stg(Jo + 3 * QS, a12); ecmdev::stg(Jo + 4 * QS, a22); ecmdev::stg(Jo + 5 * QS, a32);
         ecmdev::stg(Jo + 6 * QS, a13); ecmdev::stg(Jo + 7 * QS, a23); ecmdev::stg(Jo + 8 * QS, a33);
      }
   };
   double J11, J21, J31, J12, J22, J32, J13, J23, J33;
   double tsc = 0.0;   // REC: dt W_q / detJ
   double L[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
   if constexpr (VG) {
      const double* Jq = Jio + vJ.base; const double* Lq = vel + vJ.base;
      J11 = ecmdev::ldg(Jq); J21 = ecmdev::ldg(Jq + QS); J31 = ecmdev::ldg(Jq + 2 * QS); J12 = ecmdev::ldg(Jq + 3 * QS); J22 = ecmdev::ldg(Jq + 4 * QS); J32 = ecmdev::ldg(Jq + 5 * QS);
      J13 = ecmdev::ldg(Jq + 6 * QS); J23 = ecmdev::ldg(Jq + 7 * QS); J33 = ecmdev::ldg(Jq + 8 * QS);
      double gv[9];   // gv[c + 3 d] = dv_c/dxi_d
#pragma unroll
      for (int i = 0; i < 9; i++) gv[i] = ecmdev::ldg(Lq + i * QS);
      // adj(J) exactly as AssembleGradPA stores it (pa_kernels.hip, adj_det)
      const double a0 = J22 * J33 - J23 * J32, a1 = J32 * J13 - J12 * J33, a2 = J12 * J23 - J22 * J13;
      const double a3 = J31 * J23 - J21 * J33, a4 = J11 * J33 - J13 * J31, a5 = J21 * J13 - J11 * J23;
      const double a6 = J21 * J32 - J31 * J22, a7 = J31 * J12 - J11 * J32, a8 = J11 * J22 - J12 * J21;
      const double detJ = J11 * (J22 * J33 - J32 * J23) - J21 * (J12 * J33 - J32 * J13) + J31 * (J12 * J23 - J22 * J13);
      const double di = 1.0 / detJ;
      if constexpr (REC) {
         const int qu = tail_mode ? q : __builtin_amdgcn_readfirstlane(q);
         const double wq = tail_mode ? Wq[q] : p2::as_const(Wq)[qu];
         tsc = dt * wq * di;
         double2* rc = reinterpret_cast<double2*>(io.cm());
         ecmdev::stg2(&rc[13 * 64], a0, a1); ecmdev::stg2(&rc[14 * 64], a2, a3); ecmdev::stg2(&rc[15 * 64], a4, a5); ecmdev::stg2(&rc[16 * 64], a6, a7);
         ecmdev::stg2(&rc[17 * 64], a8, wq * (J11 * a0 + J21 * a1 + J31 * a2));      // W detJ with detJ as adj_det forms it
      }
      const double Ji[3][3] = { { di * (J22 * J33 - J23 * J32), di * (J32 * J13 - J12 * J33), di * (J12 * J23 - J22 * J13) },
                                { di * (J31 * J23 - J21 * J33), di * (J11 * J33 - J13 * J31), di * (J21 * J13 - J11 * J23) },
                                { di * (J21 * J32 - J31 * J22), di * (J31 * J12 - J11 * J32), di * (J11 * J22 - J12 * J21) } };
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int t = 0; t < 3; t++) L[c + 3 * t] = gv[c] * Ji[0][t] + gv[c + 3] * Ji[1][t] + gv[c + 6] * Ji[2][t];
   } else if constexpr (P2F) {
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
      store_J(J11, J21, J31, J12, J22, J32, J13, J23, J33);
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
         store_J(J11, J21, J31, J12, J22, J32, J13, J23, J33);
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
         store_J(J11, J21, J31, J12, J22, J32, J13, J23, J33);
      }
   } else {
      const double* Jq = Jq_in;
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
   const double* ve = ve_in;
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
   if constexpr (STG) { if (staged) {   // geometry done (J, velocity and table rows are in registers): the state and stress rows go through the stage
      ECM_PARK_BARRIER(); __builtin_amdgcn_wave_barrier();
      double* wr = io.wreg();
      rows_to_stage<14, 64 * ecmdev::NSTATEV, ecmdev::NSTATEV, RS_SV>(wr, lane, rws);
      rows_to_stage<3, 64 * 6, 6, RS_S>(wr + 64 * RS_SV, lane, rwt);
      ECM_PARK_BARRIER(); __builtin_amdgcn_wave_barrier();
      ecmdev::load_point_in<1, true>(wr + io.row_in() * RS_SV, wr + 64 * RS_SV + io.row_in() * RS_S, pin);
      ECM_PARK_BARRIER(); __builtin_amdgcn_wave_barrier();   // (the stash slots of point_update overwrite the rows)
   } else ecmdev::load_point_in<1>(io.sv0(), io.s0(), pin); }   // (dense launch: not across the gathers - their registers are what the staged form's row pieces need)
   // per-thread stash behind the shape table in LDS (PointIO::stash)
   // (STG: a lane beyond the last point repeats the last point to the end - it is neither listed nor counted)
   const bool mine = !STG || io.live;
   const int rc = point_update<KIN, QS, REC, STG>(mp, dt, L, io, mine ? kcap : (1 << 30), pin, sG + pqo, tsc, trd != 0,
                                                  TailIO{ tail_out, rs_out, tail_mode ? rs_in : nullptr, P, !tail_mode && rs_out != nullptr && mine });
   if (rc == 1 && mine) atomicAdd(fail, 1);
}

// Launch sequence of the tail split (include/exaconstit_hip.h, exa_set_newton_caps): the full launch stops a point after newton_cap evaluations
// and lists it; a dense launch of the same kernel (thread = listed point; its grid covers the worst case, blocks beyond the list leave at
// once) takes the list up - from the saved solver state when the context holds the state buffers, from scratch otherwise - and, with a
// second cap, lists what it cuts off itself for a third launch.
// dynamic LDS of a launch of k_model_setup: shape rows (per wave for the element-blocked full launch, the whole table otherwise) + stash + slip table
// The largest shape table that passes exa_g_in_lds is the p = 2 one (27 * 3 * 27 doubles; p = 3 has 64 * 3 * 64 > EXA_G_LDS_MAX_DOUBLES and stays
// in global memory): with the stash and the slip table it must fit the 64 KB a launch gets without an attribute change.
static_assert(64 * 3 * 64 > EXA_G_LDS_MAX_DOUBLES, "order-3 shape tables must not be staged in LDS by the constitutive launch");
static_assert(sizeof(double) * ((size_t)27 * 3 * 27 + (size_t)EXA_STASH_DOUBLES + 8 * ecmdev::NSLIP) <= 65536, "dynamic LDS of k_model_setup exceeds 64 KB");
static_assert(STG_G + 192 <= ecmdev::ST_SLOTS * 64 && 64 * (RS_SV + RS_S) <= ecmdev::ST_SLOTS * 64 && 32 * RS_T <= 64 * ecmdev::ST_EPI_E, "the staged rows fit a wave's stash region (tangent rows below the parking slots)");
// stg8: staged trilinear launch (k_model_setup, GSTG): the table lives in the waves' stages
static size_t model_lds_bytes(const exa_ctx* ctx, bool km, bool p2f, bool qb, int tail_mode, bool lvec8, bool stg8 = false) {
   const int nrow = (qb && !tail_mode) ? EXA_MODEL_BS / 64 : ctx->Q;
   const bool srow = lvec8 && qb && !tail_mode;      // k_model_setup, SROW: the wave's shape row comes through scalar loads
   const size_t rows = (p2f || srow || stg8 || !exa_g_in_lds(ctx->n, nrow)) ? 0 : (size_t)ctx->n * 3 * nrow;
   return sizeof(double) * (rows + (size_t)EXA_STASH_DOUBLES + (km ? (size_t)8 * ecmdev::NSLIP : (size_t)0));
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

