// Generic-order and B-bar integrator kernels (gfx950):
//   ICExaNLFIntegrator::AssemblePA (element-average gradient)   reference src/mechanics_integrators.cpp:1895-1953
//   ICExaNLFIntegrator::AddMultPA                               reference src/mechanics_integrators.cpp:2011-2086
//   ICExaNLFIntegrator::AssembleEA / ExaNLFIntegrator::AssembleEA for any order     :756-1017, 1195-1604
//   element mat-vec / diagonal                                   spec src/mechanics_operator_ext.cpp:246-252,303-314
//   AddMultGradPA / AssembleGradDiagonalPA for orders other than 1                   :562-622, 625-748
// Element matrices are kept in an element-blocked layout [block of 64 elements][column j][row i][64 lanes] so that one lane per
// element streams them with loads that are contiguous across the wave; p = 2 (81 x 81 per element, 52 KB) is HBM-bound on exactly
// this stream.  The p = 1 full-integration fast paths live in pa_kernels.hip.
#include "exa_internal.hpp"
#include "p2_basis.hpp"
#include <type_traits>

namespace {

__device__ __forceinline__ int64_t pa_off(int64_t blk, int Q, int q, int pair) { return (((blk * Q + q) * PA_PAIRS + pair) * PA_BLK) * 2; }
__device__ __forceinline__ int64_t eag_off(int64_t blk, int nd, int j, int i) { return ((blk * nd + j) * nd + i) * PA_BLK; }
// element-average gradients, element-blocked like everything else a lane-per-element kernel streams: [block][dof a + n c][64 lanes]
__device__ __forceinline__ int64_t eds_off(int64_t e, int n, int a, int c) { return ((e / PA_BLK) * (3 * n) + a + n * c) * PA_BLK + (e % PA_BLK); }

__device__ __forceinline__ void adj_det(const double* Jq, double adj[9], double& detJ, const int64_t st = 1) {
   const double J11 = Jq[0], J21 = Jq[st], J31 = Jq[2 * st], J12 = Jq[3 * st], J22 = Jq[4 * st], J32 = Jq[5 * st], J13 = Jq[6 * st], J23 = Jq[7 * st], J33 = Jq[8 * st];
   adj[0] = J22 * J33 - J23 * J32; adj[1] = J32 * J13 - J12 * J33; adj[2] = J12 * J23 - J22 * J13;
   adj[3] = J31 * J23 - J21 * J33; adj[4] = J11 * J33 - J13 * J31; adj[5] = J21 * J13 - J11 * J23;
   adj[6] = J21 * J32 - J31 * J22; adj[7] = J31 * J12 - J11 * J32; adj[8] = J11 * J22 - J12 * J21;
   detJ = J11 * adj[0] + J21 * adj[1] + J31 * adj[2];
}

// Voigt row of B (or B-bar) for a dof of component c with scaled gradient b and volumetric correction v (0 for plain B)
__device__ __forceinline__ void b_row(int c, const double b[3], double v, double row[6]) {
   row[0] = v; row[1] = v; row[2] = v; row[3] = 0.0; row[4] = 0.0; row[5] = 0.0;
   if (c == 0) { row[0] += b[0]; row[4] = b[2]; row[5] = b[1]; }
   else if (c == 1) { row[1] += b[1]; row[3] = b[2]; row[5] = b[0]; }
   else { row[2] += b[2]; row[3] = b[1]; row[4] = b[0]; }
}

// eDS(a,t,e) = sum_q W_q (G adj)(a,t) / sum_q W_q detJ ; one thread per (node, element)
template <bool QB>
__global__ void k_eds(const int Q, const int n, const int E, const double* __restrict__ W, const double* __restrict__ G,
                      const double* __restrict__ J, double* __restrict__ eDS) {
   extern __shared__ double sG_lds[];
   const double* sG = stage_shape(sG_lds, G, n, Q);
   const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= (int64_t)n * E) return;
   const int a = (int)(t % n); const int64_t e = t / n;
   double acc[3] = { 0, 0, 0 }, vol = 0;
   for (int q = 0; q < Q; q++) {
      const QView vJ = qview<QB>(9, Q, e, q);
      double adj[9], detJ; adj_det(J + vJ.base, adj, detJ, vJ.stride);
      const double w = W[q]; vol += w * detJ;
      const double g0 = sG[a + n * (3 * q)], g1 = sG[a + n * (3 * q + 1)], g2 = sG[a + n * (3 * q + 2)];
      for (int c = 0; c < 3; c++) acc[c] += w * (g0 * adj[c] + g1 * adj[3 + c] + g2 * adj[6 + c]);
   }
   for (int c = 0; c < 3; c++) eDS[eds_off(e, n, a, c)] = acc[c] / vol;
}

// Y(a,c,e) += sum_q detJ W B-bar(a,c,:) . sigma ; one thread per (node, element)
__global__ void k_residual_bbar(const int Q, const int n, const int E, const double* __restrict__ W, const double* __restrict__ G,
                                const double* __restrict__ J, const double* __restrict__ S, const double* __restrict__ eDS, double* __restrict__ Y) {
   extern __shared__ double sG_lds[];
   const double* sG = stage_shape(sG_lds, G, n, Q);
   const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= (int64_t)n * E) return;
   const int a = (int)(t % n); const int64_t e = t / n;
   const double ge[3] = { eDS[eds_off(e, n, a, 0)], eDS[eds_off(e, n, a, 1)], eDS[eds_off(e, n, a, 2)] };
   double y[3] = { 0, 0, 0 };
   for (int q = 0; q < Q; q++) {
      const int64_t ip = q + (int64_t)Q * e;
      double adj[9], detJ; adj_det(J + 9 * ip, adj, detJ);
      const double idet = 1.0 / detJ, cw = detJ * W[q];
      const double g0 = sG[a + n * (3 * q)], g1 = sG[a + n * (3 * q + 1)], g2 = sG[a + n * (3 * q + 2)];
      double b[3];
      for (int c = 0; c < 3; c++) b[c] = idet * (g0 * adj[c] + g1 * adj[3 + c] + g2 * adj[6 + c]);
      const double* s = S + 6 * ip;
      for (int c = 0; c < 3; c++) {
         double row[6]; b_row(c, b, (ge[c] - b[c]) * (1.0 / 3.0), row);
         double v = 0; for (int k = 0; k < 6; k++) v += row[k] * s[k];
         y[c] += cw * v;
      }
   }
   for (int c = 0; c < 3; c++) Y[a + n * (c + 3 * e)] += y[c];
}

// element matrices, any order, optional B-bar; grid (blocks of 64 elements, column dof)
template <int N, bool BBAR>
__global__ __launch_bounds__(PA_BLK) void k_assemble_ea_gen(const int E, const double* __restrict__ pa, const double* __restrict__ G, const double* __restrict__ W,
                                                            const double* __restrict__ eDS, double* __restrict__ emat) {
   constexpr int Q = N, ND = 3 * N;
   extern __shared__ double sG_lds[];
   const double* sG = stage_shape(sG_lds, G, N, Q);
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   const int cj = blockIdx.y, aj = cj % N, kj = cj / N;
   if (e >= E) return;
   double M[ND];
#pragma unroll
   for (int i = 0; i < ND; i++) M[i] = 0.0;
   double gej[3] = { 0, 0, 0 };
   if (BBAR) for (int c = 0; c < 3; c++) gej[c] = eDS[eds_off(e, N, aj, c)];
   for (int q = 0; q < Q; q++) {
      const double2* rec = reinterpret_cast<const double2*>(pa + pa_off(blk, Q, q, 0)) + lane;
      double v[PA_SLOTS];
#pragma unroll
      for (int pr = 0; pr < PA_PAIRS; pr++) { const double2 t = ld_rec(&rec[pr * PA_BLK]); v[2 * pr] = t.x; v[2 * pr + 1] = t.y; }
      const double* Ct = v; const double* adj = v + 36;
      const double detJ = v[45] / W[q];
      const double* Gq = sG + 3 * N * q;
      double bj[3];
      { const double g0 = Gq[aj], g1 = Gq[aj + N], g2 = Gq[aj + 2 * N];
        for (int t = 0; t < 3; t++) bj[t] = g0 * adj[t] + g1 * adj[3 + t] + g2 * adj[6 + t]; }
      double epsj[6]; b_row(kj, bj, BBAR ? (detJ * gej[kj] - bj[kj]) * (1.0 / 3.0) : 0.0, epsj);
      double cb[6];
#pragma unroll
      for (int u = 0; u < 6; u++) { double s = 0; for (int w = 0; w < 6; w++) s += Ct[u + 6 * w] * epsj[w]; cb[u] = s; }
      const double cb012 = cb[0] + cb[1] + cb[2];
#pragma unroll
      for (int a = 0; a < N; a++) {
         const double g0 = Gq[a], g1 = Gq[a + N], g2 = Gq[a + 2 * N];
         const double b0 = g0 * adj[0] + g1 * adj[3] + g2 * adj[6], b1 = g0 * adj[1] + g1 * adj[4] + g2 * adj[7], b2 = g0 * adj[2] + g1 * adj[5] + g2 * adj[8];
         double r0 = b0 * cb[0] + b2 * cb[4] + b1 * cb[5];
         double r1 = b1 * cb[1] + b2 * cb[3] + b0 * cb[5];
         double r2 = b2 * cb[2] + b1 * cb[3] + b0 * cb[4];
         if (BBAR) {
            r0 += (detJ * eDS[eds_off(e, N, a, 0)] - b0) * (1.0 / 3.0) * cb012;
            r1 += (detJ * eDS[eds_off(e, N, a, 1)] - b1) * (1.0 / 3.0) * cb012;
            r2 += (detJ * eDS[eds_off(e, N, a, 2)] - b2) * (1.0 / 3.0) * cb012;
         }
         M[a] += r0; M[a + N] += r1; M[a + 2 * N] += r2;
      }
   }
#pragma unroll
   for (int i = 0; i < ND; i++) emat[eag_off(blk, ND, cj, i) + lane] = M[i];
}

// y(j,e) += sum_i A(i,j,e) x(i,e)
template <int N, bool LVEC>
__global__ __launch_bounds__(PA_BLK) void k_ea_apply_gen(const int E, const double* __restrict__ emat, const double* __restrict__ x, double* __restrict__ y,
                                                         const int32_t* __restrict__ conn, const int nnodes, const uint8_t* __restrict__ mask,
                                                         const double* __restrict__ gate) {
   constexpr int ND = 3 * N;
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   if (gate != nullptr && gate[0] != 0.0) return;
   double X[ND];
   if (LVEC) {
#pragma unroll
      for (int a = 0; a < N; a++) {
         const int g = conn[a + N * e];
#pragma unroll
         for (int c = 0; c < 3; c++) X[a + N * c] = x[g + (int64_t)nnodes * c];
      }
      if (mask != nullptr) {   // second pass: no x load waits for its mask byte (pa_kernels.hip, k_grad_apply_p1)
#pragma unroll
         for (int a = 0; a < N; a++) {
            const int g = conn[a + N * e];
#pragma unroll
            for (int c = 0; c < 3; c++) { const uint8_t m = mask[g + (int64_t)nnodes * c]; X[a + N * c] = m ? 0.0 : X[a + N * c]; }
         }
      }
   } else {
#pragma unroll
      for (int i = 0; i < ND; i++) X[i] = x[i + (int64_t)ND * e];
   }
   for (int j = 0; j < ND; j++) {
      const double* col = emat + eag_off(blk, ND, j, 0) + lane;
      double s = 0;
#pragma unroll
      for (int i = 0; i < ND; i++) s += col[(int64_t)i * PA_BLK] * X[i];
      if (LVEC) atomicAdd(&y[conn[(j % N) + N * e] + (int64_t)nnodes * (j / N)], s);
      else y[j + (int64_t)ND * e] += s;
   }
}

// ---- any order (run-time n): the element-assembly kernels for the orders above 2, which the reference's own unit tests use (order 3:
// test/mechanics_test.cpp:313,471).  Nothing is held per lane but a 3-entry accumulator: grid (block of 64 elements, column dof, row node),
// the record of a point is re-read by every (column, row node) pair - slow by design, these orders are outside every BASELINE config.
template <bool BBAR>
__global__ __launch_bounds__(PA_BLK) void k_assemble_ea_rt(const int n, const int Q, const int E, const double* __restrict__ pa, const double* __restrict__ G,
                                                           const double* __restrict__ W, const double* __restrict__ eDS, double* __restrict__ emat) {
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   const int cj = blockIdx.y, aj = cj % n, kj = cj / n, a = blockIdx.z;
   if (e >= E) return;
   const int ND = 3 * n;
   double r[3] = { 0, 0, 0 };
   double gej[3] = { 0, 0, 0 }, gea[3] = { 0, 0, 0 };
   if (BBAR) for (int c = 0; c < 3; c++) { gej[c] = eDS[eds_off(e, n, aj, c)]; gea[c] = eDS[eds_off(e, n, a, c)]; }
   for (int q = 0; q < Q; q++) {
      const double2* rec = reinterpret_cast<const double2*>(pa + pa_off(blk, Q, q, 0)) + lane;
      double v[PA_SLOTS];
#pragma unroll
      for (int pr = 0; pr < PA_PAIRS; pr++) { const double2 t = ld_rec(&rec[pr * PA_BLK]); v[2 * pr] = t.x; v[2 * pr + 1] = t.y; }
      const double* Ct = v; const double* adj = v + 36;
      const double detJ = v[45] / W[q];
      const double* Gq = G + (int64_t)3 * n * q;
      double bj[3];
      { const double g0 = Gq[aj], g1 = Gq[aj + n], g2 = Gq[aj + 2 * n];
        for (int t = 0; t < 3; t++) bj[t] = g0 * adj[t] + g1 * adj[3 + t] + g2 * adj[6 + t]; }
      double epsj[6]; b_row(kj, bj, BBAR ? (detJ * gej[kj] - bj[kj]) * (1.0 / 3.0) : 0.0, epsj);
      double cb[6];
#pragma unroll
      for (int u = 0; u < 6; u++) { double s = 0; for (int w = 0; w < 6; w++) s += Ct[u + 6 * w] * epsj[w]; cb[u] = s; }
      const double cb012 = cb[0] + cb[1] + cb[2];
      const double g0 = Gq[a], g1 = Gq[a + n], g2 = Gq[a + 2 * n];
      const double b0 = g0 * adj[0] + g1 * adj[3] + g2 * adj[6], b1 = g0 * adj[1] + g1 * adj[4] + g2 * adj[7], b2 = g0 * adj[2] + g1 * adj[5] + g2 * adj[8];
      double r0 = b0 * cb[0] + b2 * cb[4] + b1 * cb[5];
      double r1 = b1 * cb[1] + b2 * cb[3] + b0 * cb[5];
      double r2 = b2 * cb[2] + b1 * cb[3] + b0 * cb[4];
      if (BBAR) {
         r0 += (detJ * gea[0] - b0) * (1.0 / 3.0) * cb012;
         r1 += (detJ * gea[1] - b1) * (1.0 / 3.0) * cb012;
         r2 += (detJ * gea[2] - b2) * (1.0 / 3.0) * cb012;
      }
      r[0] += r0; r[1] += r1; r[2] += r2;
   }
   for (int c = 0; c < 3; c++) emat[eag_off(blk, ND, cj, a + n * c) + lane] = r[c];
}

// y(j,e) += sum_i A(i,j,e) x(i,e): grid (block of 64 elements, column dof j)
template <bool LVEC>
__global__ __launch_bounds__(PA_BLK) void k_ea_apply_rt(const int n, const int E, const double* __restrict__ emat, const double* __restrict__ x, double* __restrict__ y,
                                                        const int32_t* __restrict__ conn, const int nnodes, const uint8_t* __restrict__ mask,
                                                        const double* __restrict__ gate) {
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   const int j = blockIdx.y, ND = 3 * n;
   if (e >= E) return;
   if (gate != nullptr && gate[0] != 0.0) return;
   const double* col = emat + eag_off(blk, ND, j, 0) + lane;
   double s = 0;
   for (int i = 0; i < ND; i++) {
      double xi;
      if (LVEC) {      // value and mask byte requested together, selected afterwards (no load waits for its mask byte)
         const int64_t idx = conn[(i % n) + (int64_t)n * e] + (int64_t)nnodes * (i / n);
         const double xv = x[idx]; const uint8_t m = mask != nullptr ? mask[idx] : (uint8_t)0;
         xi = m ? 0.0 : xv;
      }
      else xi = x[i + (int64_t)ND * e];
      s += col[(int64_t)i * PA_BLK] * xi;
   }
   if (LVEC) atomicAdd(&y[conn[(j % n) + (int64_t)n * e] + (int64_t)nnodes * (j / n)], s);
   else y[j + (int64_t)ND * e] += s;
}

// ---- matrix-free action for p = 2 (27 nodes, 27 points), plain or B-bar, on L-vectors ---------------------------------------------
// The element matrices of a triquadratic hex are 81 x 81 doubles (52 KB per element); the operator they represent is determined by the
// 27 x 46-double point records (10 KB per element) [+ 81 element-average gradients for B-bar], so the action is computed from those:
//   eps = B(-bar) x_e,  s = Ct eps (TRANS: Ct^T eps, the operator the assembled matrices apply, see k_ea_apply_gen),  y_e += B(-bar)^T s.
// One lane per element, one wave per 64-element block, points in sequence.  y_e (81 doubles) lives in registers, x_e in LDS used as a
// lane-private register-file extension ([dof][lane], no barriers, no conflicts); 76 of its 81 values, because 4 x 76 x 512 B fits
// the CU's 160 KB with one wave per SIMD (and leaves the allocator 8 KB of slack).
// Shape gradients: dN_a/dxi(q) = D[qi][i] B[qj][j] B[qk][k] etc. with the 3 x 3 one-dimensional tables B, D.  Both contractions over
// the 27 nodes are done as three one-dimensional passes whose coefficients are wave-uniform SGPR operands (18 doubles per point)
// instead of an 81-double table per point (which does not fit the SGPR file and stalls a lone wave on scalar loads).
// Software pipeline: the record of point q+1 is requested as soon as the record of point q is consumed and lands behind the two
// long register-only passes (scatter of q, gather of q+1: 2 x 270 FMAs) - at one wave per SIMD nothing else hides it.
constexpr int P2N = p2::N, P2ND = p2::ND, P2XL = 76;
#ifndef EXA_P2_XCD
#define EXA_P2_XCD 0   // p = 2 action / residual / geometry pre-pass with the element blocks dealt to the XCDs in contiguous eighths (exa_internal.hpp, xcd_block): measured at
                       // 64^3 B-bar (round 6, same call, twice): action 0.589 | 0.589 ms with, 0.579 | 0.580 ms without - a block of 64 triquadratic elements shares few
                       // nodes with the next one along x, and the round-robin order spreads the 7.8 KB record rows of a block over more channels.  Off (A/B switch)
#endif
using p2::cptr; using p2::as_const;
typedef p2::Rows Rows1D;

#define P2X(i_) ((i_) < P2XL ? sX[(i_) * PA_BLK] : XR[(i_) < P2XL ? 0 : (i_) - P2XL])

// CMP: the record is the compact one (D 25, K, adj 9, W detJ: 18 pairs instead of 23)
template <bool BBAR, bool TRANS, bool CMP>
__device__ __forceinline__ void mf_point_p2(const double2 (&rec)[CMP ? PAC_PAIRS_GEO : PA_PAIRS], const double (&gx)[3][3], const double wq, const double dbar, double& sbar, double (&T)[3][3]) {
   constexpr int NPR = CMP ? PAC_PAIRS_GEO : PA_PAIRS;
   double v[2 * NPR];
#pragma unroll
   for (int pr = 0; pr < NPR; pr++) { v[2 * pr] = rec[pr].x; v[2 * pr + 1] = rec[pr].y; }
   const double* Ct = v; const double* adj = v + (CMP ? 26 : 36);
   const double wdet = v[CMP ? 35 : 45];
   double h[3][3];
#pragma unroll
   for (int c = 0; c < 3; c++)
#pragma unroll
      for (int t = 0; t < 3; t++) h[c][t] = gx[c][0] * adj[t] + gx[c][1] * adj[3 + t] + gx[c][2] * adj[6 + t];
   double eps[6] = { h[0][0], h[1][1], h[2][2], h[1][2] + h[2][1], h[0][2] + h[2][0], h[0][1] + h[1][0] };
   double detJ = 0.0;
   if (BBAR) {   // volumetric part replaced by the element average: sum_dofs (detJ gbar - b)/3 x
      detJ = wdet / wq;
      const double vol = (detJ * dbar - (h[0][0] + h[1][1] + h[2][2])) * (1.0 / 3.0);
      eps[0] += vol; eps[1] += vol; eps[2] += vol;
   }
   double sg[6];
   if (CMP) d55_apply(v, v[25], eps, sg);   // the compact record holds D or D^T as the context needs (k_grad_setup_pa<.., TRD>)
   else {
#pragma unroll
      for (int i = 0; i < 6; i++) { double t = 0;
#pragma unroll
         for (int j = 0; j < 6; j++) t += (TRANS ? Ct[j + 6 * i] : Ct[i + 6 * j]) * eps[j];
         sg[i] = t; }
   }
   const double Sm[3][3] = { { sg[0], sg[5], sg[4] }, { sg[5], sg[1], sg[3] }, { sg[4], sg[3], sg[2] } };
#pragma unroll
   for (int j = 0; j < 3; j++)
#pragma unroll
      for (int c = 0; c < 3; c++) T[j][c] = adj[3 * j] * Sm[0][c] + adj[3 * j + 1] * Sm[1][c] + adj[3 * j + 2] * Sm[2][c];
   if (BBAR) {   // B-bar^T s = B^T s + (detJ gbar - b) tr(s)/3: the -b part folds into T, the gbar part is element-constant
      const double tr3 = (sg[0] + sg[1] + sg[2]) * (1.0 / 3.0);
      sbar += detJ * tr3;
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
         for (int c = 0; c < 3; c++) T[j][c] -= adj[3 * j + c] * tr3;
   }
}

// T1 = one-dimensional tables [1D point][B0 B1 B2 D0 D1 D2]
template <bool BBAR, bool TRANS, bool CMP>
__global__ __launch_bounds__(PA_BLK) void k_mf_apply_p2(const int E, const double* __restrict__ pa, const double* __restrict__ T1, const double* __restrict__ W,
                                                        const double* __restrict__ eDS, const double* __restrict__ x, double* __restrict__ y,
                                                        const int32_t* __restrict__ conn, const int nnodes, const uint8_t* __restrict__ mask,
                                                        const double* __restrict__ gate) {
   __shared__ double sXall[P2XL * PA_BLK];
   const int lane = threadIdx.x; const int64_t blk = EXA_P2_XCD ? xcd_block(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   if (gate != nullptr && gate[0] != 0.0) return;
   double* sX = sXall + lane;
   double XR[P2ND - P2XL]; double dbar = 0.0;
   constexpr int NPR = CMP ? PAC_PAIRS_GEO : PA_PAIRS;
   double2 rec[NPR];
   auto load = [&](int q) {
      const double2* r = reinterpret_cast<const double2*>(pa + (CMP ? pac_off<PAC_PAIRS_GEO>(blk, P2N, q, 0) : pa_off(blk, P2N, q, 0))) + lane;
#pragma unroll
      for (int pr = 0; pr < NPR; pr++) rec[pr] = ld_rec(&r[pr * PA_BLK]);
   };
   load(0);
   int gidx[P2N];
#pragma unroll
   for (int a = 0; a < P2N; a++) gidx[a] = conn[a + P2N * e];
#pragma unroll
   for (int a = 0; a < P2N; a++) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
         const int i = a + P2N * c;
         const double xv = x[gidx[a] + (int64_t)nnodes * c];
         if (i < P2XL) sX[i * PA_BLK] = xv; else XR[i < P2XL ? 0 : i - P2XL] = xv;
      }
      if (a % 9 == 8) __builtin_amdgcn_sched_barrier(0);
   }
   if (mask != nullptr) {   // essential dofs enter as zeros (second pass: keeps the gather free of branches)
      uint8_t mk[P2ND];
#pragma unroll
      for (int i = 0; i < P2ND; i++) mk[i] = mask[gidx[i % P2N] + (int64_t)nnodes * (i / P2N)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < P2ND; i++) {
         if (i < P2XL) { if (mk[i]) sX[i * PA_BLK] = 0.0; } else XR[i < P2XL ? 0 : i - P2XL] = mk[i] ? 0.0 : XR[i < P2XL ? 0 : i - P2XL];
      }
   }
   if (BBAR) {
#pragma unroll
      for (int i = 0; i < P2ND; i++) {
         dbar += eDS[eds_off(e, P2N, i % P2N, i / P2N)] * P2X(i);
         if (i % 9 == 8) __builtin_amdgcn_sched_barrier(0);
      }
   }
   double Y[P2ND];
#pragma unroll
   for (int i = 0; i < P2ND; i++) Y[i] = 0.0;
   double sbar = 0.0;
   const cptr t1 = as_const(T1);
   Rows1D r, rn;
   auto rows = [&](Rows1D& o, int q) { p2::load_rows(t1, q, o); };
   rows(r, 0);
   // one point: gather, point arithmetic, request of the next record into the registers just consumed, scatter.  PRE is a
   // compile-time flag and the last point is peeled: a conditional prefetch would turn the record into a loop-carried phi of old and
   // new values (92 register copies per point, spills).  Two records in flight were measured too: no gain, the launch runs at the
   // HBM rate of its traffic already.
   auto step = [&](const int q, auto pre) {
      rows(rn, q + 1 < P2N ? q + 1 : q);      // scalar loads for the next point, a whole point ahead of their use
      double gx[3][3], T[3][3];
      p2::gather(r, [&](int a, int c) { return P2X(a + P2N * c); }, gx);
      mf_point_p2<BBAR, TRANS, CMP>(rec, gx, W[q], dbar, sbar, T);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (decltype(pre)::value) load(q + 1);
      __builtin_amdgcn_sched_barrier(0);
      p2::scatter(r, T, Y);
      r = rn;
   };
#pragma unroll 1
   for (int q = 0; q < P2N - 1; q++) step(q, std::true_type{});
   step(P2N - 1, std::false_type{});
   // re-read conn and eDS here: carried across the point loop they cost 27 + 162 registers, i.e. scratch traffic inside the loop
   // (and a scratch load waits for the record prefetch in front of it)
   const double* eDS2 = eDS; asm volatile("" : "+s"(eDS2));
#pragma unroll
   for (int a = 0; a < P2N; a++) {
      const int gi = conn[a + P2N * e];
#pragma unroll
      for (int c = 0; c < 3; c++) {
         double v = Y[a + P2N * c];
         if (BBAR) v += eDS2[eds_off(e, P2N, a, c)] * sbar;
         atomicAdd(&y[gi + (int64_t)nnodes * c], v);
      }
      if (a % 9 == 8) __builtin_amdgcn_sched_barrier(0);
   }
}
#undef P2X

// ---- geometry pre-pass of the p = 2 constitutive launch ---------------------------------------------------------------------------------------------
// The fused p = 2 launch lets every quadrature point gather its element's 27 nodes itself: 27 points x 162 gathered values per element, each behind a
// connectivity load, in front of every wave's arithmetic (round 5: ~1.2 ms of a 2.6 ms launch at 64^3; SQ counters round 6: VALU busy 61 %, waiting 44 %,
// 1 245 B/point at the L2 boundary against 920 algorithmic).  Here one lane owns one ELEMENT: it gathers the 27 nodes once per field and component,
// contracts them towards all 27 points by sum factorisation (i, then j and k per point row) and writes, per point, the Jacobian J(i,j) = dx_i/dxi_j and
// the reference-space velocity gradient dv_c/dxi_d, both (3,3,Q,E) element-blocked - 512-byte rows per value.  The constitutive launch then reads
// 18 doubles per point (k_model_setup, VG) instead of 189 scattered ones.  Same multiply-add nesting as p2::gather: the Jacobians carry the bits of the fused form.
__global__ __launch_bounds__(PA_BLK) void k_geom_p2(const int E, const double* __restrict__ T1, const double* __restrict__ xl, const double* __restrict__ vl,
                                                    const int32_t* __restrict__ conn, const int nnodes, double* __restrict__ Jout, double* __restrict__ Lxout) {
   const int lane = threadIdx.x; const int64_t blk = EXA_P2_XCD ? xcd_block(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   const cptr t1 = as_const(T1);
   double B[3][3], D[3][3];   // [1D point][1D node] (wave-uniform: scalar registers)
#pragma unroll
   for (int q = 0; q < 3; q++)
#pragma unroll
      for (int i = 0; i < 3; i++) { B[q][i] = t1[6 * q + i]; D[q][i] = t1[6 * q + 3 + i]; }
   int g[P2N];
#pragma unroll
   for (int a = 0; a < P2N; a++) g[a] = conn[a + P2N * e];
#pragma unroll 1
   for (int f = 0; f < 2; f++) {
      const double* __restrict__ src = f == 0 ? xl : vl; double* __restrict__ out = f == 0 ? Jout : Lxout;
      if (src == nullptr || out == nullptr) continue;   // (uniform)
#pragma unroll 1
      for (int c = 0; c < 3; c++) {
         double X[27];
#pragma unroll
         for (int l = 0; l < 27; l++) X[l] = src[g[p2::NAT[l]] + (int64_t)nnodes * c];   // lexicographic l = i + 3 j + 9 k
         double UB[3][9], UD[3][9];   // [qi][j + 3 k]
#pragma unroll
         for (int jk = 0; jk < 9; jk++) {
            const double x0 = X[3 * jk], x1 = X[3 * jk + 1], x2 = X[3 * jk + 2];
#pragma unroll
            for (int qi = 0; qi < 3; qi++) {
               UB[qi][jk] = fma(B[qi][2], x2, fma(B[qi][1], x1, B[qi][0] * x0));
               UD[qi][jk] = fma(D[qi][2], x2, fma(D[qi][1], x1, D[qi][0] * x0));
            }
         }
#pragma unroll
         for (int qi = 0; qi < 3; qi++)
#pragma unroll
            for (int qj = 0; qj < 3; qj++) {
               double bb[3], bd[3], db[3];
#pragma unroll
               for (int k = 0; k < 3; k++) {
                  double a = 0.0, b = 0.0, d = 0.0;
#pragma unroll
                  for (int j = 0; j < 3; j++) { a = fma(B[qj][j], UB[qi][j + 3 * k], a); b = fma(D[qj][j], UB[qi][j + 3 * k], b); d = fma(B[qj][j], UD[qi][j + 3 * k], d); }
                  bb[k] = a; bd[k] = b; db[k] = d;
               }
#pragma unroll
               for (int qk = 0; qk < 3; qk++) {
                  const int q = qi + 3 * qj + 9 * qk;
                  double* o = out + (((blk * P2N + q) * 9 + c) << 6) + lane;      // entry (c, d) of the point at [(c + 3 d) * 64]
                  o[0] = fma(B[qk][2], db[2], fma(B[qk][1], db[1], B[qk][0] * db[0]));
                  o[3 * 64] = fma(B[qk][2], bd[2], fma(B[qk][1], bd[1], B[qk][0] * bd[0]));
                  o[6 * 64] = fma(D[qk][2], bb[2], fma(D[qk][1], bb[1], D[qk][0] * bb[0]));
               }
            }
      }
   }
}

// ---- residual on L-vectors for p = 2, plain or B-bar (ExaNLFIntegrator / ICExaNLFIntegrator AssemblePA + AddMultPA fused with E->L) ---
// y_a,c += sum_q W detJ B(-bar)_a,c . sigma.  One lane per element, points in sequence, y_e (81 doubles) in registers, the node
// contraction as three one-dimensional passes (p2::scatter); 15 doubles per point from HBM, element-blocked or reference layout.
template <bool BBAR, bool QB>
__global__ __launch_bounds__(PA_BLK) void k_residual_p2(const int E, const double* __restrict__ T1, const double* __restrict__ W, const double* __restrict__ J,
                                                        const double* __restrict__ S, const double* __restrict__ eDS, double* __restrict__ y,
                                                        const int32_t* __restrict__ conn, const int nnodes) {
   const int lane = threadIdx.x; const int64_t e = (EXA_P2_XCD ? xcd_block(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x) * PA_BLK + lane;
   if (e >= E) return;
   double Y[P2ND];
#pragma unroll
   for (int i = 0; i < P2ND; i++) Y[i] = 0.0;
   double sbar = 0.0;
   const cptr t1 = as_const(T1);
#pragma unroll 1
   for (int q = 0; q < P2N; q++) {
      Rows1D r; p2::load_rows(t1, q, r);
      const QView vJ = qview<QB>(9, P2N, e, q), vS = qview<QB>(6, P2N, e, q);
      double adj[9], detJ; adj_det(J + vJ.base, adj, detJ, vJ.stride);
      const double* sp = S + vS.base; const double w = W[q];
      const double sg[6] = { w * sp[0], w * sp[vS.stride], w * sp[2 * vS.stride], w * sp[3 * vS.stride], w * sp[4 * vS.stride], w * sp[5 * vS.stride] };
      const double Sm[3][3] = { { sg[0], sg[5], sg[4] }, { sg[5], sg[1], sg[3] }, { sg[4], sg[3], sg[2] } };
      double T[3][3];
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
         for (int c = 0; c < 3; c++) T[j][c] = adj[3 * j] * Sm[0][c] + adj[3 * j + 1] * Sm[1][c] + adj[3 * j + 2] * Sm[2][c];
      if (BBAR) {   // (gbar - b)/3 tr(sigma): the -b part folds into T, the gbar part is element-constant
         const double tr3 = (sg[0] + sg[1] + sg[2]) * (1.0 / 3.0);
         sbar += detJ * tr3;
#pragma unroll
         for (int j = 0; j < 3; j++)
#pragma unroll
            for (int c = 0; c < 3; c++) T[j][c] -= adj[3 * j + c] * tr3;
      }
      p2::scatter(r, T, Y);
   }
#pragma unroll
   for (int a = 0; a < P2N; a++) {
      const int gi = conn[a + P2N * e];
#pragma unroll
      for (int c = 0; c < 3; c++) {
         double v = Y[a + P2N * c];
         if (BBAR) v += eDS[eds_off(e, P2N, a, c)] * sbar;
         atomicAdd(&y[gi + (int64_t)nnodes * c], v);
      }
      if (a % 9 == 8) __builtin_amdgcn_sched_barrier(0);
   }
}

__global__ __launch_bounds__(PA_BLK) void k_ea_diag_gen(const int E, const int nd, const double* __restrict__ emat, double* __restrict__ y) {
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   for (int j = 0; j < nd; j++) y[j + (int64_t)nd * e] += emat[eag_off(blk, nd, j, j) + lane];
}

__global__ __launch_bounds__(PA_BLK) void k_ea_export_gen(const int E, const int nd, const double* __restrict__ emat, double* __restrict__ out) {
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   for (int j = 0; j < nd; j++) for (int i = 0; i < nd; i++) out[i + (int64_t)nd * (j + (int64_t)nd * e)] = emat[eag_off(blk, nd, j, i) + lane];
}

// generic partial-assembly gradient action: stage 1 (one thread per point) T = adj ( Ct : sym( (G x_e) adj ) )
__global__ void k_pa_apply_stage1(const int Q, const int n, const int64_t P, const double* __restrict__ G, const double* __restrict__ pa,
                                  const double* __restrict__ X, double* __restrict__ Tb) {
   extern __shared__ double sG_lds[];
   const double* sG = stage_shape(sG_lds, G, n, Q);
   const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (ip >= P) return;
   const int q = (int)(ip % Q); const int64_t e = ip / Q;
   const int64_t blk = e / PA_BLK; const int lane = (int)(e % PA_BLK);
   const double* rec = pa + pa_off(blk, Q, q, 0) + 2 * lane;
   auto slot = [&](int s) { return rec[(int64_t)(s >> 1) * PA_BLK * 2 + (s & 1)]; };
   double adj[9]; for (int i = 0; i < 9; i++) adj[i] = slot(36 + i);
   const double* x = X + (int64_t)3 * n * e; const double* Gq = sG + 3 * n * q;
   double gx[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
   for (int a = 0; a < n; a++) {
      const double g[3] = { Gq[a], Gq[a + n], Gq[a + 2 * n] };
      for (int c = 0; c < 3; c++) { const double xv = x[a + n * c]; for (int j = 0; j < 3; j++) gx[c][j] += g[j] * xv; }
   }
   double h[3][3];
   for (int c = 0; c < 3; c++) for (int t = 0; t < 3; t++) h[c][t] = gx[c][0] * adj[t] + gx[c][1] * adj[3 + t] + gx[c][2] * adj[6 + t];
   const double eps[6] = { h[0][0], h[1][1], h[2][2], h[1][2] + h[2][1], h[0][2] + h[2][0], h[0][1] + h[1][0] };
   double sg[6];
   for (int i = 0; i < 6; i++) { double s = 0; for (int j = 0; j < 6; j++) s += slot(i + 6 * j) * eps[j]; sg[i] = s; }
   const double Sm[3][3] = { { sg[0], sg[5], sg[4] }, { sg[5], sg[1], sg[3] }, { sg[4], sg[3], sg[2] } };
   for (int j = 0; j < 3; j++) for (int c = 0; c < 3; c++) Tb[9 * ip + j + 3 * c] = adj[3 * j] * Sm[0][c] + adj[3 * j + 1] * Sm[1][c] + adj[3 * j + 2] * Sm[2][c];
}

// diag(a,c,e) += sum_q b(a)^T Ct_c b(a) for any order (reference src/mechanics_integrators.cpp:702-743)
__global__ void k_pa_diag_gen(const int Q, const int n, const int E, const double* __restrict__ G, const double* __restrict__ pa, double* __restrict__ Y) {
   extern __shared__ double sG_lds[];
   const double* sG = stage_shape(sG_lds, G, n, Q);
   const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= (int64_t)n * E) return;
   const int a = (int)(t % n); const int64_t e = t / n;
   const int64_t blk = e / PA_BLK; const int lane = (int)(e % PA_BLK);
   constexpr int R[3][3] = { { 0, 5, 4 }, { 5, 1, 3 }, { 4, 3, 2 } };
   double y[3] = { 0, 0, 0 };
   for (int q = 0; q < Q; q++) {
      const double* rec = pa + pa_off(blk, Q, q, 0) + 2 * lane;
      auto slot = [&](int s) { return rec[(int64_t)(s >> 1) * PA_BLK * 2 + (s & 1)]; };
      const double g0 = sG[a + n * (3 * q)], g1 = sG[a + n * (3 * q + 1)], g2 = sG[a + n * (3 * q + 2)];
      double b[3];
      for (int c = 0; c < 3; c++) b[c] = g0 * slot(36 + c) + g1 * slot(39 + c) + g2 * slot(42 + c);
      for (int c = 0; c < 3; c++) { double s = 0; for (int r = 0; r < 3; r++) for (int u = 0; u < 3; u++) s += b[r] * slot(R[c][r] + 6 * R[c][u]) * b[u]; y[c] += s; }
   }
   for (int c = 0; c < 3; c++) Y[a + n * (c + 3 * e)] += y[c];
}

inline unsigned nblk(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

int exa_launch_residual_apply_from(exa_ctx* ctx, const double* D, double* Y, hipStream_t s);   // pa_kernels.hip
int exa_launch_eds(exa_ctx* ctx, const double* J, hipStream_t s);

int exa_launch_eds(exa_ctx* ctx, const double* J, hipStream_t s) {
   if (ctx->qblk) hipLaunchKernelGGL(k_eds<true>, dim3(nblk((int64_t)ctx->n * ctx->E, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->E, ctx->W_dev, ctx->G_dev, J, ctx->eDS);
   else hipLaunchKernelGGL(k_eds<false>, dim3(nblk((int64_t)ctx->n * ctx->E, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->E, ctx->W_dev, ctx->G_dev, J, ctx->eDS);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_residual_bbar(exa_ctx* ctx, const double* J, const double* S, double* Y, hipStream_t s) {
   hipLaunchKernelGGL(k_residual_bbar, dim3(nblk((int64_t)ctx->n * ctx->E, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->E, ctx->W_dev, ctx->G_dev, J, S, ctx->eDS, Y);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_assemble_ea_gen(exa_ctx* ctx, hipStream_t s) {
   const bool bbar = ctx->cfg.integ == EXA_INTEG_BBAR;
   const dim3 grid(nblk(ctx->E, PA_BLK), 3 * ctx->n); const size_t lds = exa_g_lds_bytes(ctx->n, ctx->Q);
   if (ctx->n == 8) {
      if (bbar) hipLaunchKernelGGL((k_assemble_ea_gen<8, true>), grid, dim3(PA_BLK), lds, s, ctx->E, ctx->pa, ctx->G_dev, ctx->W_dev, ctx->eDS, ctx->emat);
      else hipLaunchKernelGGL((k_assemble_ea_gen<8, false>), grid, dim3(PA_BLK), lds, s, ctx->E, ctx->pa, ctx->G_dev, ctx->W_dev, ctx->eDS, ctx->emat);
   } else if (ctx->n == 27) {
      if (bbar) hipLaunchKernelGGL((k_assemble_ea_gen<27, true>), grid, dim3(PA_BLK), lds, s, ctx->E, ctx->pa, ctx->G_dev, ctx->W_dev, ctx->eDS, ctx->emat);
      else hipLaunchKernelGGL((k_assemble_ea_gen<27, false>), grid, dim3(PA_BLK), lds, s, ctx->E, ctx->pa, ctx->G_dev, ctx->W_dev, ctx->eDS, ctx->emat);
   } else {   // any other order: run-time kernels
      const dim3 g3(nblk(ctx->E, PA_BLK), 3 * ctx->n, ctx->n);
      if (bbar) hipLaunchKernelGGL((k_assemble_ea_rt<true>), g3, dim3(PA_BLK), 0, s, ctx->n, ctx->Q, ctx->E, ctx->pa, ctx->G_dev, ctx->W_dev, ctx->eDS, ctx->emat);
      else hipLaunchKernelGGL((k_assemble_ea_rt<false>), g3, dim3(PA_BLK), 0, s, ctx->n, ctx->Q, ctx->E, ctx->pa, ctx->G_dev, ctx->W_dev, ctx->eDS, ctx->emat);
   }
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_ea_apply_gen(exa_ctx* ctx, const double* x, double* y, bool lvec, const uint8_t* mask, const double* gate, hipStream_t s) {
   const unsigned nb = nblk(ctx->E, PA_BLK);
   if (ctx->n == 8) {
      if (lvec) hipLaunchKernelGGL((k_ea_apply_gen<8, true>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->emat, x, y, ctx->conn, ctx->nnodes, mask, gate);
      else hipLaunchKernelGGL((k_ea_apply_gen<8, false>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->emat, x, y, ctx->conn, ctx->nnodes, mask, gate);
   } else if (ctx->n == 27) {
      if (lvec) hipLaunchKernelGGL((k_ea_apply_gen<27, true>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->emat, x, y, ctx->conn, ctx->nnodes, mask, gate);
      else hipLaunchKernelGGL((k_ea_apply_gen<27, false>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->emat, x, y, ctx->conn, ctx->nnodes, mask, gate);
   } else {
      const dim3 g2(nb, 3 * ctx->n);
      if (lvec) hipLaunchKernelGGL((k_ea_apply_rt<true>), g2, dim3(PA_BLK), 0, s, ctx->n, ctx->E, ctx->emat, x, y, ctx->conn, ctx->nnodes, mask, gate);
      else hipLaunchKernelGGL((k_ea_apply_rt<false>), g2, dim3(PA_BLK), 0, s, ctx->n, ctx->E, ctx->emat, x, y, ctx->conn, ctx->nnodes, mask, gate);
   }
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
// one-dimensional basis tables of the p = 2 kernels + a check of the node numbering they have compiled in
int exa_ensure_p2_tables(exa_ctx* ctx) {
   if (ctx->T1_dev) return EXA_OK;
   std::vector<double> t1; std::vector<int> nat;
   exa_build_1d_tables(2, t1, nat);
   static const int expect[27] = { 0, 8, 1, 11, 20, 9, 3, 10, 2, 16, 21, 17, 24, 26, 22, 19, 23, 18, 4, 12, 5, 15, 25, 13, 7, 14, 6 };
   for (int i = 0; i < 27; i++) if (nat[i] != expect[i]) { ctx->err = "p = 2 kernels: node numbering mismatch"; return EXA_ERR_STATE; }
   EXA_HIP_CHECK(ctx, hipMalloc(&ctx->T1_dev, sizeof(double) * t1.size()));
   EXA_HIP_CHECK(ctx, hipMemcpy(ctx->T1_dev, t1.data(), sizeof(double) * t1.size(), hipMemcpyHostToDevice));
   return EXA_OK;
}
// geometry pre-pass of the p = 2 constitutive launch on the element-blocked layout: J (and, with vl, the reference-space velocity gradient) for every point
int exa_launch_geom_p2(exa_ctx* ctx, const double* xl, const double* vl, double* J, double* Lx, hipStream_t s) {
   if (ctx->n != 27 || !ctx->qblk) { ctx->err = "geometry pre-pass: p = 2, element-blocked layout"; return EXA_ERR_UNSUPPORTED; }
   if (int rc = exa_ensure_p2_tables(ctx)) return rc;
   hipLaunchKernelGGL(k_geom_p2, dim3(nblk(ctx->E, PA_BLK)), dim3(PA_BLK), 0, s, ctx->E, ctx->T1_dev, xl, vl, ctx->conn, ctx->nnodes, J, Lx);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
// element-average gradients of the B-bar integrator from a Jacobian field (driver: the record route has no exa_grad_setup pass that would refresh them)
int exa_grad_refresh_bbar(exa_ctx* ctx, const double* J, hipStream_t s) {
   if (ctx->cfg.integ != EXA_INTEG_BBAR) return EXA_OK;
   if (!ctx->eDS) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->eDS, sizeof(double) * 3 * ctx->n * PA_BLK * (size_t)((ctx->E + PA_BLK - 1) / PA_BLK)));
   return exa_launch_eds(ctx, J, s);
}
// residual on L-vectors at p = 2 (B-bar: element-average gradients refreshed from J first)
int exa_launch_residual_p2(exa_ctx* ctx, const double* J, const double* S, double* y, hipStream_t s) {
   if (int rc = exa_ensure_p2_tables(ctx)) return rc;
   const bool bbar = ctx->cfg.integ == EXA_INTEG_BBAR; const unsigned nb = nblk(ctx->E, PA_BLK);
   if (bbar) { if (int rc = exa_launch_eds(ctx, J, s)) return rc; }
#define RES_LAUNCH(B, QBV) hipLaunchKernelGGL((k_residual_p2<B, QBV>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->T1_dev, ctx->W_dev, J, S, ctx->eDS, y, ctx->conn, ctx->nnodes)
   if (bbar) { if (ctx->qblk) RES_LAUNCH(true, true); else RES_LAUNCH(true, false); }
   else { if (ctx->qblk) RES_LAUNCH(false, true); else RES_LAUNCH(false, false); }
#undef RES_LAUNCH
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
// matrix-free p = 2 action on L-vectors; `trans` selects the operator of the assembled matrices (EA) instead of the PA one
int exa_launch_mf_apply_p2(exa_ctx* ctx, const double* x, double* y, const uint8_t* mask, const double* gate, bool trans, hipStream_t s) {
   if (ctx->n != 27 || ctx->Q != 27) { ctx->err = "matrix-free action is built for p = 2 (27 nodes, 27 points)"; return EXA_ERR_UNSUPPORTED; }
   const unsigned nb = nblk(ctx->E, PA_BLK); const bool bbar = ctx->cfg.integ == EXA_INTEG_BBAR;
   if (int rc = exa_ensure_p2_tables(ctx)) return rc;
#define MF_LAUNCH(B, T, CM) hipLaunchKernelGGL((k_mf_apply_p2<B, T, CM>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, CM ? ctx->pa_c : ctx->pa, ctx->T1_dev, ctx->W_dev, ctx->eDS, x, y, ctx->conn, ctx->nnodes, mask, gate)
   const bool cm = ctx->pa_c != nullptr && ctx->pac_pairs == PAC_PAIRS_GEO;
   if (cm) {
      if (bbar) { if (trans) MF_LAUNCH(true, true, true); else MF_LAUNCH(true, false, true); }
      else { if (trans) MF_LAUNCH(false, true, true); else MF_LAUNCH(false, false, true); }
   } else {
      if (bbar) { if (trans) MF_LAUNCH(true, true, false); else MF_LAUNCH(true, false, false); }
      else { if (trans) MF_LAUNCH(false, true, false); else MF_LAUNCH(false, false, false); }
   }
#undef MF_LAUNCH
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_ea_diag_gen(exa_ctx* ctx, double* y, hipStream_t s) {
   hipLaunchKernelGGL(k_ea_diag_gen, dim3(nblk(ctx->E, PA_BLK)), dim3(PA_BLK), 0, s, ctx->E, 3 * ctx->n, ctx->emat, y);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_ea_export_gen(exa_ctx* ctx, double* out, hipStream_t s) {
   hipLaunchKernelGGL(k_ea_export_gen, dim3(nblk(ctx->E, PA_BLK)), dim3(PA_BLK), 0, s, ctx->E, 3 * ctx->n, ctx->emat, out);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
// generic PA action on E-vectors: stage 1 into the (3,3,Q,E) scratch `dmat`, stage 2 = the AddMultPA contraction
int exa_launch_pa_apply_gen(exa_ctx* ctx, const double* x, double* y, hipStream_t s) {
   hipLaunchKernelGGL(k_pa_apply_stage1, dim3(nblk(ctx->P, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->P, ctx->G_dev, ctx->pa, x, ctx->tbuf);
   EXA_HIP_CHECK(ctx, hipGetLastError());
   return exa_launch_residual_apply_from(ctx, ctx->tbuf, y, s);
}
int exa_launch_pa_diag_gen(exa_ctx* ctx, double* y, hipStream_t s) {
   hipLaunchKernelGGL(k_pa_diag_gen, dim3(nblk((int64_t)ctx->n * ctx->E, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->E, ctx->G_dev, ctx->pa, y);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
