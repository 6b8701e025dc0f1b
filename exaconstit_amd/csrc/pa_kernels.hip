// Integrator kernels (gfx950) behind the ExaNLFIntegrator seam of the reference:
//   AssemblePA / AddMultPA            reference src/mechanics_integrators.cpp:160-314, 518-557      (residual B^T sigma)
//   TransformMatGradTo4D+AssembleGradPA  src/mechanics_model.cpp:949-1061, src/mechanics_integrators.cpp:331-513
//   AddMultGradPA                     src/mechanics_integrators.cpp:562-622                          (the PCG inner kernel)
//   AssembleGradDiagonalPA            src/mechanics_integrators.cpp:625-748
//   AssembleEA + element mat-vec      src/mechanics_integrators.cpp:756-1017, spec src/mechanics_operator_ext.cpp:303-314
//   geometric factors / grad_calc     src/mechanics_operator.cpp:350-391, src/mechanics_kernels.cpp:7-78
//
// Design (HBM-bound): the gradient action never materialises the reference's 81-double C4 and 81-double D4 per point.
// grad_setup writes one 46-double record per point (36 scaled tangent + 9 adj(J) + weight) in an element-blocked
// AoSoA layout (exa_internal.hpp), and the apply kernel — one lane per element, one wave per 64-element block —
// streams it with 16-byte loads that are contiguous across the wave, keeps x_e / y_e (24 doubles each) in registers,
// and contracts  y_e += G^T adj ( Ct : sym( (G x_e) adj ) )  per point.  The p=1 shape-derivative table is a
// compile-time constant.  L-vector variants fuse the gather (connectivity table) and the scatter-add (FP64 atomics).
#include "exa_internal.hpp"
#include <cstdlib>
#include <cstring>

namespace {

// ---- p = 1 reference table as compile-time constants ---------------------------------------------------------------
constexpr double GL0 = 0.21132486540518713, GL1 = 0.78867513459481287;   // (1 -/+ 1/sqrt 3)/2
constexpr double gl_pt(int i) { return i == 0 ? GL0 : GL1; }
constexpr int VX[8] = { 0, 1, 1, 0, 0, 1, 1, 0 }, VY[8] = { 0, 0, 1, 1, 0, 0, 1, 1 }, VZ[8] = { 0, 0, 0, 0, 1, 1, 1, 1 };
constexpr double n1(int v, double x) { return v ? x : 1.0 - x; }
constexpr double d1(int v) { return v ? 1.0 : -1.0; }
// dN_a/dxi_j at quadrature point q (x fastest)
constexpr double G1(int a, int j, int q) {
   const double x = gl_pt(q & 1), y = gl_pt((q >> 1) & 1), z = gl_pt((q >> 2) & 1);
   return j == 0 ? d1(VX[a]) * n1(VY[a], y) * n1(VZ[a], z) : (j == 1 ? n1(VX[a], x) * d1(VY[a]) * n1(VZ[a], z) : n1(VX[a], x) * n1(VY[a], y) * d1(VZ[a]));
}

__device__ __forceinline__ int64_t pa_off(int64_t blk, int Q, int q, int pair) { return (((blk * Q + q) * PA_PAIRS + pair) * PA_BLK) * 2; }

__device__ __forceinline__ void adj_det(const double* Jq, double adj[9], double& detJ, const int st = 1) {
   const double J11 = Jq[0], J21 = Jq[st], J31 = Jq[2 * st], J12 = Jq[3 * st], J22 = Jq[4 * st], J32 = Jq[5 * st], J13 = Jq[6 * st], J23 = Jq[7 * st], J33 = Jq[8 * st];
   adj[0] = J22 * J33 - J23 * J32; adj[1] = J32 * J13 - J12 * J33; adj[2] = J12 * J23 - J22 * J13;
   adj[3] = J31 * J23 - J21 * J33; adj[4] = J11 * J33 - J13 * J31; adj[5] = J21 * J13 - J11 * J23;
   adj[6] = J21 * J32 - J31 * J22; adj[7] = J31 * J12 - J11 * J32; adj[8] = J11 * J22 - J12 * J21;
   detJ = J11 * adj[0] + J21 * adj[1] + J31 * adj[2];
}

// ---- generic-order kernels -------------------------------------------------------------------------------------------
// MFEM GeometricFactors::J, laid out (Q,3,3,E) with J(q,i,j,e) = dx_i/dxi_j, into the (3,3,Q,E) array of the integrators
// (reference src/mechanics_operator.cpp:377-391: jac_view(l,k,j,i) = geom_j_view(j,l,k,i))
template <bool QB>
__global__ void k_jacobians_from_geom(const int Q, const int64_t P, const double* __restrict__ gj, double* __restrict__ J) {
   const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (ip >= P) return;
   const int q = (int)(ip % Q); const int64_t e = ip / Q;
   const QView v = qview<QB>(9, Q, e, q);
   for (int c = 0; c < 9; c++) J[v.base + (int64_t)c * v.stride] = gj[q + (int64_t)Q * (c + 9 * e)];   // c = l + 3 k on both sides
}

template <bool QB>
__global__ void k_jacobians(const int Q, const int n, const int64_t P, const double* __restrict__ G, const double* __restrict__ xe, double* __restrict__ J) {
   extern __shared__ double sG_lds[];
   const double* sG = stage_shape(sG_lds, G, n, Q);
   const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (ip >= P) return;
   const int q = (int)(ip % Q); const int64_t e = ip / Q;
   const double* x = xe + (int64_t)3 * n * e; const double* Gq = sG + 3 * n * q;
   double Jl[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
   for (int a = 0; a < n; a++) {
      const double g0 = Gq[a], g1 = Gq[a + n], g2 = Gq[a + 2 * n];
      const double x0 = x[a], x1 = x[a + n], x2 = x[a + 2 * n];
      Jl[0] += x0 * g0; Jl[1] += x1 * g0; Jl[2] += x2 * g0;
      Jl[3] += x0 * g1; Jl[4] += x1 * g1; Jl[5] += x2 * g1;
      Jl[6] += x0 * g2; Jl[7] += x1 * g2; Jl[8] += x2 * g2;
   }
   const QView v = qview<QB>(9, Q, e, q);
   for (int i = 0; i < 9; i++) J[v.base + (int64_t)i * v.stride] = Jl[i];
}

template <bool QB>
__global__ void k_grad_calc(const int Q, const int n, const int64_t P, const double* __restrict__ J, const double* __restrict__ G,
                            const double* __restrict__ fe, double* __restrict__ out) {
   extern __shared__ double sG_lds[];
   const double* sG = stage_shape(sG_lds, G, n, Q);
   const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (ip >= P) return;
   const int q = (int)(ip % Q); const int64_t e = ip / Q;
   const QView v = qview<QB>(9, Q, e, q);
   double adj[9], detJ; adj_det(J + v.base, adj, detJ, v.stride);
   const double di = 1.0 / detJ;
   const double* f = fe + (int64_t)3 * n * e; const double* Gq = sG + 3 * n * q;
   double L[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
   for (int r = 0; r < n; r++) {
      const double g0 = Gq[r], g1 = Gq[r + n], g2 = Gq[r + 2 * n];
      double b[3];
      for (int t = 0; t < 3; t++) b[t] = (g0 * adj[t] + g1 * adj[3 + t] + g2 * adj[6 + t]) * di;
      const double v0 = f[r], v1 = f[r + n], v2 = f[r + 2 * n];
      for (int t = 0; t < 3; t++) { L[3 * t] += v0 * b[t]; L[3 * t + 1] += v1 * b[t]; L[3 * t + 2] += v2 * b[t]; }
   }
   for (int i = 0; i < 9; i++) out[v.base + (int64_t)i * v.stride] = L[i];
}

// D(j,k,q,e) = W_q sum_l sigma(k,l) adj(J)(j,l)
__global__ void k_residual_setup(const int Q, const int64_t P, const double* __restrict__ W, const double* __restrict__ J,
                                 const double* __restrict__ S, double* __restrict__ D) {
   const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (ip >= P) return;
   const int q = (int)(ip % Q);
   double adj[9], detJ; adj_det(J + 9 * ip, adj, detJ);
   const double* s = S + 6 * ip;
   const double sg[3][3] = { { s[0], s[5], s[4] }, { s[5], s[1], s[3] }, { s[4], s[3], s[2] } };
   const double w = W[q];
   for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++)
      D[9 * ip + j + 3 * k] = w * (sg[k][0] * adj[3 * j] + sg[k][1] * adj[3 * j + 1] + sg[k][2] * adj[3 * j + 2]);
}

// Y(i,k,e) += sum_q sum_j G(i,j,q) D(j,k,q,e);  one thread per (node, element)
__global__ void k_residual_apply(const int Q, const int n, const int E, const double* __restrict__ G, const double* __restrict__ D, double* __restrict__ Y) {
   extern __shared__ double sG_lds[];
   const double* sG = stage_shape(sG_lds, G, n, Q);
   const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= (int64_t)n * E) return;
   const int i = (int)(t % n); const int64_t e = t / n;
   double y0 = 0, y1 = 0, y2 = 0;
   for (int q = 0; q < Q; q++) {
      const double* d = D + 9 * (q + (int64_t)Q * e);
      const double g0 = sG[i + n * (3 * q)], g1 = sG[i + n * (3 * q + 1)], g2 = sG[i + n * (3 * q + 2)];
      y0 += g0 * d[0] + g1 * d[1] + g2 * d[2];
      y1 += g0 * d[3] + g1 * d[4] + g2 * d[5];
      y2 += g0 * d[6] + g1 * d[7] + g2 * d[8];
   }
   double* y = Y + (int64_t)3 * n * e;
   y[i] += y0; y[i + n] += y1; y[i + 2 * n] += y2;
}

// ---- compact tangent form -------------------------------------------------------------------------------------------------
// Every ExaCMech evptn model returns d sigma / d eps = V65 D V65^T + K m m^T (a 5 x 5 deviatoric block in the (vecd) basis the
// library works in, plus the bulk term; m = (1,1,1,0,0,0); ecm_device.hpp, end of point_update): 26 numbers instead of 36.  The
// geometry-recomputing action can stream those: 13 instead of 18 16-byte pairs per point.  V65^T of a Voigt 6-vector:

// pa record for every point of an element; one lane per element (coalesced 16-byte stores).  CMP: also the compact record.
// TRD: the compact record holds D^T (element-assembly contexts: their action applies C^T, and the kernels then run the same code)
// FULL = false: only the compact records are written - all that the L-vector action with recomputed geometry reads (46 of the 117 doubles this pass moves per
// point otherwise); the 46-double records are then built on demand from the same arrays (exa_launch_grad_setup_pa, pa_lazy).
// (Round 6, measured at 128^3 on the reference layout: one lane per POINT - contiguous 16-byte loads of its own 288-byte row, stores of 8 x 128-byte segments -
//  is slower than this form, 5.2 against 2.5 ms for the 46-double records: 64 separate lines per load instruction either way, and the stores lose their 1 KB rows.)
template <bool QB, int CMP, bool TRD, bool FULL = true>
__global__ __launch_bounds__(PA_BLK) void k_grad_setup_pa(const int Q, const int E, const double dt, const double* __restrict__ W,
                                                          const double* __restrict__ J, const double* __restrict__ C, double* __restrict__ pa, double* __restrict__ pac) {
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   for (int q = 0; q < Q; q++) {
      const QView vj = qview<QB>(9, Q, e, q), vc = qview<QB>(36, Q, e, q);
      double adj[9], detJ; adj_det(J + vj.base, adj, detJ, vj.stride);
      const double sc = dt * W[q] / detJ;
      const double* c = C + vc.base;
      if (FULL) {
      double2* rec = reinterpret_cast<double2*>(pa + pa_off(blk, Q, q, 0)) + lane;
#pragma unroll
      for (int pr = 0; pr < 18; pr++) rec[pr * PA_BLK] = make_double2(c[(2 * pr) * vc.stride] * sc, c[(2 * pr + 1) * vc.stride] * sc);
#pragma unroll
      for (int pr = 0; pr < 4; pr++) rec[(18 + pr) * PA_BLK] = make_double2(adj[2 * pr], adj[2 * pr + 1]);
      rec[22 * PA_BLK] = make_double2(adj[8], W[q] * detJ);
      }
      if (CMP) {   // 13 pairs: D, K;  18 pairs: D, K, adj(J), W detJ
         double D[36], Dn[25]; tangent_to_d55(c, vc.stride, Dn, D[25]);
#pragma unroll
         for (int l = 0; l < 5; l++)
#pragma unroll
            for (int k = 0; k < 5; k++) D[k + 5 * l] = (TRD ? Dn[l + 5 * k] : Dn[k + 5 * l]) * sc;
         D[25] *= sc;
         if (CMP == PAC_PAIRS_GEO) { for (int i = 0; i < 9; i++) D[26 + i] = adj[i]; D[35] = W[q] * detJ; }
         double2* rc = reinterpret_cast<double2*>(pac + pac_off<CMP ? CMP : 1>(blk, Q, q, 0)) + lane;
#pragma unroll
         for (int pr = 0; pr < CMP; pr++) rc[pr * PA_BLK] = make_double2(D[2 * pr], D[2 * pr + 1]);
      }
   }
}

// largest relative deviation of a tangent field from the compact form: max_points |C - (V65 D V65^T + K m m^T)|_max / |C|_max
template <bool QB>
__global__ void k_tangent_defect(const int Q, const int64_t P, const double* __restrict__ C, unsigned long long* __restrict__ out) {
   const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   double defect = 0.0;
   if (ip < P) {
      const QView vc = qview<QB>(36, Q, ip / Q, (int)(ip % Q));
      const double* c = C + vc.base;
      double D[25], K; tangent_to_d55(c, vc.stride, D, K);
      double cmax = 0.0, dmax = 0.0;
#pragma unroll
      for (int j = 0; j < 6; j++) {
         double eps[6] = { 0, 0, 0, 0, 0, 0 }, col[6]; eps[j] = 1.0;
         d55_apply(D, K, eps, col);
#pragma unroll
         for (int i = 0; i < 6; i++) { const double v = c[(i + 6 * j) * vc.stride]; cmax = fmax(cmax, fabs(v)); dmax = fmax(dmax, fabs(v - col[i])); }
      }
      defect = cmax > 0.0 ? dmax / cmax : 0.0;
   }
   // non-negative doubles order like their bit patterns
   for (int o = 32; o > 0; o >>= 1) defect = fmax(defect, __shfl_xor(defect, o));
   if ((threadIdx.x & 63) == 0 && defect > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(defect));
}

// ---- p = 1 specialised kernels: one lane per element, 64-element block per wave ------------------------------------
// GEO: adj(J) is recomputed per point from the nodal coordinates of the element (24 doubles gathered once per element, L2-resident)
// instead of being streamed from the record: 36 instead of 46 doubles per point from HBM for ~90 more FMAs per point.
// CMP (with GEO): the tangent arrives in its compact form (13 pairs per point, k_grad_setup_pa<.., true>).
// TRANS: C^T instead of C, i.e. the operator of the assembled element matrices (y_j += sum_i A_ij x_i with A = B^T C B, k_ea_apply_p1):
// the element-assembly action without the 24 x 24 matrices.
// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Remapped, XCD x works on one contiguous eighth of the element
// blocks, so the node rows neighbouring blocks share (gathers of x / coordinates, scatter atomics) stay within one L2.
// CG (E-vector action without coordinates): the 18-pair record - compact tangent (D, K) + adj(J) + W detJ (EXA_TANGENT_DEV5_BULK_GEO).
template <bool LVEC, bool GEO, bool CMP = false, bool TRANS = false, bool NT = true, bool CG = false>
__global__ __launch_bounds__(PA_BLK) void k_grad_apply_p1(const int E, const double* __restrict__ pa, const double* __restrict__ x, double* __restrict__ y,
                                                          const int32_t* __restrict__ conn, const int nnodes, const uint8_t* __restrict__ mask,
                                                          const double* __restrict__ gate, const double* __restrict__ coords, double* __restrict__ ev = nullptr,
                                                          const int blk0 = 0) {
   // blk0: first 64-element block of this launch (the multi-rank action runs the blocks that touch shared nodes first, then the interior
   // ones while the halo exchange of the first part is on the wire: host/driver.hip, NonlinearMechOperator::GradMult)
   const int lane = threadIdx.x; const int64_t blk = blk0 + xcd_block(blockIdx.x, gridDim.x); const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   if (gate != nullptr && gate[0] != 0.0) return;   // device-side "solver already converged" flag
   double X[3][8], Y[3][8];
   double XC[GEO ? 3 : 1][8];
   int g[8];
   if (LVEC) {
#pragma unroll
      for (int a = 0; a < 8; a++) g[a] = conn[a + 8 * e];
      if (GEO) {   // (requested with the x values: one round trip for all 72 gathers)
#pragma unroll
         for (int c = 0; c < 3; c++)
#pragma unroll
            for (int a = 0; a < 8; a++) XC[c][a] = coords[g[a] + (int64_t)nnodes * c];
      }
      // every value and every mask byte is requested before the first one is looked at (written as `mask[idx] ? 0 : x[idx]` the compiler makes each
      // x load conditional on its mask byte: 24 dependent round trips in front of the record stream of a wave - measured in round 5)
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int a = 0; a < 8; a++) X[c][a] = x[g[a] + (int64_t)nnodes * c];
      if (mask != nullptr) {
         uint8_t mk[3][8];
#pragma unroll
         for (int c = 0; c < 3; c++)
#pragma unroll
            for (int a = 0; a < 8; a++) mk[c][a] = mask[g[a] + (int64_t)nnodes * c];
         __builtin_amdgcn_sched_barrier(0);
#pragma unroll
         for (int c = 0; c < 3; c++)
#pragma unroll
            for (int a = 0; a < 8; a++) X[c][a] = mk[c][a] ? 0.0 : X[c][a];
      }
   } else {
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int a = 0; a < 8; a++) X[c][a] = x[a + 8 * (c + 3 * e)];
   }
#pragma unroll
   for (int c = 0; c < 3; c++)
#pragma unroll
      for (int a = 0; a < 8; a++) Y[c][a] = 0.0;
#pragma unroll
   for (int q = 0; q < 8; q++) {
      static_assert(!CMP || GEO, "the compact record carries no geometry");
      static_assert(!CG || (!GEO && !CMP && !TRANS), "the record with geometry is a form of its own");
      const double2* rec = reinterpret_cast<const double2*>(pa + (CG ? pac_off<PAC_PAIRS_GEO>(blk, 8, q, 0) : (CMP ? pac_off<PAC_PAIRS>(blk, 8, q, 0) : pa_off(blk, 8, q, 0)))) + lane;
      double v[PA_SLOTS];
#pragma unroll
      for (int pr = 0; pr < (CG ? PAC_PAIRS_GEO : (CMP ? PAC_PAIRS : (GEO ? 18 : PA_PAIRS))); pr++) { const double2 t = ld_rec<NT>(&rec[pr * PA_BLK]); v[2 * pr] = t.x; v[2 * pr + 1] = t.y; }
      if (GEO) {   // J(i,j) = sum_a x_a,i dN_a/dxi_j, then adj(J) exactly as grad_setup stored it
         double Jl[9];
#pragma unroll
         for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++) { double t = 0; for (int a = 0; a < 8; a++) t += G1(a, j, q) * XC[i][a]; Jl[i + 3 * j] = t; }
         double dj; adj_det(Jl, v + 36, dj);
      }
      const double* Ct = v; const double* adj = v + (CG ? 26 : 36);
      // gx[c][j] = sum_a G(a,j,q) X[c][a]
      double gx[3][3];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int j = 0; j < 3; j++) { double s = 0; for (int a = 0; a < 8; a++) s += G1(a, j, q) * X[c][a]; gx[c][j] = s; }
      // h[c][t] = sum_j gx[c][j] adj(j,t)
      double h[3][3];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int t = 0; t < 3; t++) h[c][t] = gx[c][0] * adj[t] + gx[c][1] * adj[3 + t] + gx[c][2] * adj[6 + t];
      const double eps[6] = { h[0][0], h[1][1], h[2][2], h[1][2] + h[2][1], h[0][2] + h[2][0], h[0][1] + h[1][0] };
      double sg[6];
      static_assert(!(CMP && TRANS), "compact records are stored in the orientation the action needs");
      if (CMP || CG) d55_apply(v, v[25], eps, sg);
      else {
#pragma unroll
         for (int i = 0; i < 6; i++) { double s = 0; for (int j = 0; j < 6; j++) s += (TRANS ? Ct[j + 6 * i] : Ct[i + 6 * j]) * eps[j]; sg[i] = s; }
      }
      const double S[3][3] = { { sg[0], sg[5], sg[4] }, { sg[5], sg[1], sg[3] }, { sg[4], sg[3], sg[2] } };
      // T[j][c] = sum_t adj(j,t) S(t,c)
      double T[3][3];
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
         for (int c = 0; c < 3; c++) T[j][c] = adj[3 * j] * S[0][c] + adj[3 * j + 1] * S[1][c] + adj[3 * j + 2] * S[2][c];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int a = 0; a < 8; a++) Y[c][a] += G1(a, 0, q) * T[0][c] + G1(a, 1, q) * T[1][c] + G1(a, 2, q) * T[2][c];
   }
   if (LVEC && ev != nullptr) {   // deterministic mode: element outputs to the scratch ([block][c][a][lane]), summed per node by k_e2l_gather
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int a = 0; a < 8; a++) ev[(blk * 24 + c * 8 + a) * PA_BLK + lane] = Y[c][a];
   } else if (LVEC) {
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int a = 0; a < 8; a++) atomicAdd(&y[g[a] + (int64_t)nnodes * c], Y[c][a]);
   } else {
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int a = 0; a < 8; a++) y[a + 8 * (c + 3 * e)] += Y[c][a];
   }
}

__global__ __launch_bounds__(PA_BLK) void k_grad_diag_p1(const int E, const double* __restrict__ pa, double* __restrict__ y) {
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   constexpr int R[3][3] = { { 0, 5, 4 }, { 5, 1, 3 }, { 4, 3, 2 } };
   double Y[3][8];
#pragma unroll
   for (int c = 0; c < 3; c++)
#pragma unroll
      for (int a = 0; a < 8; a++) Y[c][a] = 0.0;
#pragma unroll
   for (int q = 0; q < 8; q++) {
      const double2* rec = reinterpret_cast<const double2*>(pa + pa_off(blk, 8, q, 0)) + lane;
      double v[PA_SLOTS];
#pragma unroll
      for (int pr = 0; pr < PA_PAIRS; pr++) { const double2 t = ld_rec(&rec[pr * PA_BLK]); v[2 * pr] = t.x; v[2 * pr + 1] = t.y; }
      const double* Ct = v; const double* adj = v + 36;
#pragma unroll
      for (int a = 0; a < 8; a++) {
         double b[3];
#pragma unroll
         for (int t = 0; t < 3; t++) b[t] = G1(a, 0, q) * adj[t] + G1(a, 1, q) * adj[3 + t] + G1(a, 2, q) * adj[6 + t];
#pragma unroll
         for (int c = 0; c < 3; c++) {
            double s = 0;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
               for (int t = 0; t < 3; t++) s += b[r] * Ct[R[c][r] + 6 * R[c][t]] * b[t];
            Y[c][a] += s;
         }
      }
   }
#pragma unroll
   for (int c = 0; c < 3; c++)
#pragma unroll
      for (int a = 0; a < 8; a++) y[a + 8 * (c + 3 * e)] += Y[c][a];
}

// fused AssemblePA + AddMultPA + scatter-add for p = 1
// GEO: no Jacobian field - J(i,j) = sum_a x_a,i dN_a/dxi_j at the 8 points from the element's nodal coordinates (gathered once through the
// connectivity the scatter needs anyway): 72 B per point less to read, and the constitutive launch need not write them (exa_residual_lvec)
template <bool LVEC, bool QB, bool GEO = false>
__global__ __launch_bounds__(PA_BLK) void k_residual_p1(const int E, const double* __restrict__ W, const double* __restrict__ J, const double* __restrict__ S,
                                                        double* __restrict__ y, const int32_t* __restrict__ conn, const int nnodes, double* __restrict__ ev = nullptr,
                                                        const double* __restrict__ coords = nullptr) {
   const int64_t e = (int64_t)blockIdx.x * PA_BLK + threadIdx.x;
   if (e >= E) return;
   static_assert(!GEO || LVEC, "geometry from the nodes needs the connectivity");
   double XC[GEO ? 3 : 1][8];
   if (GEO) {
#pragma unroll
      for (int a = 0; a < 8; a++) {
         const int g = conn[a + 8 * e];
#pragma unroll
         for (int c = 0; c < 3; c++) XC[c][a] = coords[g + (int64_t)nnodes * c];
      }
   }
   double Y[3][8];
#pragma unroll
   for (int c = 0; c < 3; c++)
#pragma unroll
      for (int a = 0; a < 8; a++) Y[c][a] = 0.0;
#pragma unroll
   for (int q = 0; q < 8; q++) {
      const QView vj = qview<QB>(9, 8, e, q), vs = qview<QB>(6, 8, e, q);
      double adj[9], detJ;
      if (GEO) {
         double Jl[9];
#pragma unroll
         for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++) { double t = 0; for (int a = 0; a < 8; a++) t += G1(a, j, q) * XC[i][a]; Jl[i + 3 * j] = t; }
         adj_det(Jl, adj, detJ);
      } else adj_det(J + vj.base, adj, detJ, vj.stride);
      const double* sp = S + vs.base;
      const double s[6] = { sp[0], sp[vs.stride], sp[2 * vs.stride], sp[3 * vs.stride], sp[4 * vs.stride], sp[5 * vs.stride] };
      const double w = W[q];
      const double sg[3][3] = { { s[0], s[5], s[4] }, { s[5], s[1], s[3] }, { s[4], s[3], s[2] } };
      double D[3][3];   // D[j][k]
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
         for (int k = 0; k < 3; k++) D[j][k] = w * (sg[k][0] * adj[3 * j] + sg[k][1] * adj[3 * j + 1] + sg[k][2] * adj[3 * j + 2]);
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
         for (int a = 0; a < 8; a++) Y[k][a] += G1(a, 0, q) * D[0][k] + G1(a, 1, q) * D[1][k] + G1(a, 2, q) * D[2][k];
   }
   if (LVEC && ev != nullptr) {
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int a = 0; a < 8; a++) ev[((int64_t)blockIdx.x * 24 + c * 8 + a) * PA_BLK + threadIdx.x] = Y[c][a];
   } else if (LVEC) {
#pragma unroll
      for (int a = 0; a < 8; a++) {
         const int g = conn[a + 8 * e];
#pragma unroll
         for (int c = 0; c < 3; c++) atomicAdd(&y[g + (int64_t)nnodes * c], Y[c][a]);
      }
   } else {
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int a = 0; a < 8; a++) y[a + 8 * (c + 3 * e)] += Y[c][a];
   }
}

// ---- element assembly (p = 1): emat layout [block][col j (24)][row pair (12)][64 lanes][2] ---------------------------
__device__ __forceinline__ int64_t ea_off(int64_t blk, int j, int ipair) { return (((blk * 24 + j) * 12 + ipair) * PA_BLK) * 2; }

__global__ __launch_bounds__(PA_BLK) void k_assemble_ea_p1(const int E, const double* __restrict__ pa, double* __restrict__ emat) {
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   const int cj = blockIdx.y;          // column dof = node + 8 * comp
   if (e >= E) return;
   const int aj = cj & 7, kj = cj >> 3;
   double M[24];
#pragma unroll
   for (int i = 0; i < 24; i++) M[i] = 0.0;
#pragma unroll
   for (int q = 0; q < 8; q++) {
      const double2* rec = reinterpret_cast<const double2*>(pa + pa_off(blk, 8, q, 0)) + lane;
      double v[PA_SLOTS];
#pragma unroll
      for (int pr = 0; pr < PA_PAIRS; pr++) { const double2 t = ld_rec(&rec[pr * PA_BLK]); v[2 * pr] = t.x; v[2 * pr + 1] = t.y; }
      const double* Ct = v; const double* adj = v + 36;
      // b vectors (detJ * dN/dx) for every node; Ct already carries dt W / detJ
      double b[8][3];
#pragma unroll
      for (int a = 0; a < 8; a++)
#pragma unroll
         for (int t = 0; t < 3; t++) b[a][t] = G1(a, 0, q) * adj[t] + G1(a, 1, q) * adj[3 + t] + G1(a, 2, q) * adj[6 + t];
      // column strain vector of dof cj (runtime node/comp -> select)
      double bj[3] = { 0, 0, 0 };
#pragma unroll
      for (int a = 0; a < 8; a++) if (a == aj) { bj[0] = b[a][0]; bj[1] = b[a][1]; bj[2] = b[a][2]; }
      double epsj[6];
      epsj[0] = kj == 0 ? bj[0] : 0.0; epsj[1] = kj == 1 ? bj[1] : 0.0; epsj[2] = kj == 2 ? bj[2] : 0.0;
      epsj[3] = kj == 1 ? bj[2] : (kj == 2 ? bj[1] : 0.0);
      epsj[4] = kj == 0 ? bj[2] : (kj == 2 ? bj[0] : 0.0);
      epsj[5] = kj == 0 ? bj[1] : (kj == 1 ? bj[0] : 0.0);
      double cb[6];
#pragma unroll
      for (int u = 0; u < 6; u++) { double s = 0; for (int w = 0; w < 6; w++) s += Ct[u + 6 * w] * epsj[w]; cb[u] = s; }
#pragma unroll
      for (int a = 0; a < 8; a++) {
         M[a] += b[a][0] * cb[0] + b[a][2] * cb[4] + b[a][1] * cb[5];
         M[a + 8] += b[a][1] * cb[1] + b[a][2] * cb[3] + b[a][0] * cb[5];
         M[a + 16] += b[a][2] * cb[2] + b[a][1] * cb[3] + b[a][0] * cb[4];
      }
   }
   double2* out = reinterpret_cast<double2*>(emat + ea_off(blk, cj, 0)) + lane;
#pragma unroll
   for (int ipair = 0; ipair < 12; ipair++) out[ipair * PA_BLK] = make_double2(M[2 * ipair], M[2 * ipair + 1]);
}

// y(j,e) += sum_i A(i,j,e) x(i,e)
template <bool LVEC>
__global__ __launch_bounds__(PA_BLK) void k_ea_apply_p1(const int E, const double* __restrict__ emat, const double* __restrict__ x, double* __restrict__ y,
                                                        const int32_t* __restrict__ conn, const int nnodes, const uint8_t* __restrict__ mask,
                                                        const double* __restrict__ gate) {
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   if (gate != nullptr && gate[0] != 0.0) return;
   double X[24]; int g[8];
   if (LVEC) {
#pragma unroll
      for (int a = 0; a < 8; a++) g[a] = conn[a + 8 * e];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
         for (int a = 0; a < 8; a++) X[a + 8 * c] = x[g[a] + (int64_t)nnodes * c];
      if (mask != nullptr) {   // second pass: no x load waits for its mask byte (see k_grad_apply_p1)
         uint8_t mk[24];
#pragma unroll
         for (int c = 0; c < 3; c++)
#pragma unroll
            for (int a = 0; a < 8; a++) mk[a + 8 * c] = mask[g[a] + (int64_t)nnodes * c];
         __builtin_amdgcn_sched_barrier(0);
#pragma unroll
         for (int i = 0; i < 24; i++) X[i] = mk[i] ? 0.0 : X[i];
      }
   } else {
#pragma unroll
      for (int i = 0; i < 24; i++) X[i] = x[i + 24 * e];
   }
#pragma unroll
   for (int j = 0; j < 24; j++) {
      const double2* col = reinterpret_cast<const double2*>(emat + ea_off(blk, j, 0)) + lane;
      double s = 0;
#pragma unroll
      for (int ipair = 0; ipair < 12; ipair++) { const double2 t = ld_rec(&col[ipair * PA_BLK]); s += t.x * X[2 * ipair] + t.y * X[2 * ipair + 1]; }
      if (LVEC) atomicAdd(&y[g[j & 7] + (int64_t)nnodes * (j >> 3)], s);
      else y[j + 24 * e] += s;
   }
}

__global__ __launch_bounds__(PA_BLK) void k_ea_diag_p1(const int E, const double* __restrict__ emat, double* __restrict__ y) {
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   for (int j = 0; j < 24; j++) {
      const double* col = emat + ea_off(blk, j, j >> 1) + 2 * lane;
      y[j + 24 * e] += col[j & 1];
   }
}

// copy out in the reference layout (3n,3n,E) col-major
__global__ __launch_bounds__(PA_BLK) void k_ea_export_p1(const int E, const double* __restrict__ emat, double* __restrict__ out) {
   const int lane = threadIdx.x; const int64_t blk = blockIdx.x; const int64_t e = blk * PA_BLK + lane;
   if (e >= E) return;
   for (int j = 0; j < 24; j++) for (int i = 0; i < 24; i++) out[i + 24 * (j + 24 * e)] = emat[ea_off(blk, j, i >> 1) + 2 * lane + (i & 1)];
}

// ---- restriction & reductions -----------------------------------------------------------------------------------------
__global__ void k_restrict(const int n, const int E, const int nnodes, const int32_t* __restrict__ conn, const double* __restrict__ L, double* __restrict__ Ev) {
   const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= (int64_t)n * E) return;
   const int a = (int)(t % n); const int64_t e = t / n;
   const int g = conn[t];
   for (int c = 0; c < 3; c++) Ev[a + n * (c + 3 * e)] = L[g + (int64_t)nnodes * c];
}

__global__ void k_restrict_T(const int n, const int E, const int nnodes, const int32_t* __restrict__ conn, const double* __restrict__ Ev, double* __restrict__ L) {
   const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= (int64_t)n * E) return;
   const int a = (int)(t % n); const int64_t e = t / n;
   const int g = conn[t];
   for (int c = 0; c < 3; c++) atomicAdd(&L[g + (int64_t)nnodes * c], Ev[a + n * (c + 3 * e)]);
}

// Deterministic E->L: node i adds the contributions of its elements in the fixed order of the node -> (element, local node) table
// (ascending element index) instead of racing FP64 atomics: bit-reproducible L-vectors, hence bit-reproducible CG iterates.
// BLOCKED: contributions in the [block of 64 elements][c][a][lane] scratch the fused p = 1 kernels write; otherwise an E-vector (n,3,E).
template <bool BLOCKED>
__global__ void k_e2l_gather(const int n, const int nnodes, const int32_t* __restrict__ off, const int32_t* __restrict__ idx, const double* __restrict__ ev,
                             double* __restrict__ y, const double* __restrict__ gate) {
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nnodes) return;
   if (gate != nullptr && gate[0] != 0.0) return;
   double s0 = 0.0, s1 = 0.0, s2 = 0.0;
   for (int k = off[i]; k < off[i + 1]; k++) {
      const int code = idx[k]; const int64_t e = code / n; const int a = code % n;
      if (BLOCKED) {
         const int64_t b = ((e >> 6) * 24 + a) * PA_BLK + (e & 63);
         s0 += ev[b]; s1 += ev[b + 8 * PA_BLK]; s2 += ev[b + 16 * PA_BLK];
      } else {
         const int64_t b = a + (int64_t)n * 3 * e;
         s0 += ev[b]; s1 += ev[b + n]; s2 += ev[b + 2 * n];
      }
   }
   y[i] += s0; y[i + (int64_t)nnodes] += s1; y[i + 2 * (int64_t)nnodes] += s2;
}

// partial sums of W detJ * qf(c) and of W detJ: one block -> (vdim + 1) partials
template <bool QB>
__global__ void k_vol_avg_partial(const int Q, const int64_t P, const int vdim, const double* __restrict__ W, const double* __restrict__ J,
                                  const double* __restrict__ qf, double* __restrict__ partial) {
   extern __shared__ double sm[];   // blockDim.x
   const int nb = gridDim.x;
   for (int c = 0; c <= vdim; c++) {
      double acc = 0;
      for (int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ip < P; ip += (int64_t)nb * blockDim.x) {
         const int q = (int)(ip % Q); const int64_t e = ip / Q;
         const QView vj = qview<QB>(9, Q, e, q), vq = qview<QB>(vdim, Q, e, q);
         double adj[9], detJ; adj_det(J + vj.base, adj, detJ, vj.stride);
         const double w = W[q] * detJ;
         acc += (c < vdim) ? w * qf[vq.base + (int64_t)c * vq.stride] : w;
      }
      sm[threadIdx.x] = acc; __syncthreads();
      for (int s = blockDim.x / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
      if (threadIdx.x == 0) partial[c * nb + blockIdx.x] = sm[0];
      __syncthreads();
   }
}

}  // namespace

// ---- launchers ----------------------------------------------------------------------------------------------------------
static inline unsigned nblk(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

int exa_launch_jacobians(exa_ctx* ctx, const double* xe, double* J, hipStream_t s) {
   if (ctx->qblk) hipLaunchKernelGGL(k_jacobians<true>, dim3(nblk(ctx->P, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->P, ctx->G_dev, xe, J);
   else hipLaunchKernelGGL(k_jacobians<false>, dim3(nblk(ctx->P, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->P, ctx->G_dev, xe, J);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_jacobians_from_geom(exa_ctx* ctx, const double* gj, double* J, hipStream_t s) {
   if (ctx->qblk) hipLaunchKernelGGL(k_jacobians_from_geom<true>, dim3(nblk(ctx->P, 256)), dim3(256), 0, s, ctx->Q, ctx->P, gj, J);
   else hipLaunchKernelGGL(k_jacobians_from_geom<false>, dim3(nblk(ctx->P, 256)), dim3(256), 0, s, ctx->Q, ctx->P, gj, J);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_grad_calc(exa_ctx* ctx, const double* J, const double* fe, double* out, hipStream_t s) {
   if (ctx->qblk) hipLaunchKernelGGL(k_grad_calc<true>, dim3(nblk(ctx->P, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->P, J, ctx->G_dev, fe, out);
   else hipLaunchKernelGGL(k_grad_calc<false>, dim3(nblk(ctx->P, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->P, J, ctx->G_dev, fe, out);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_residual_setup(exa_ctx* ctx, const double* J, const double* S, hipStream_t s) {
   hipLaunchKernelGGL(k_residual_setup, dim3(nblk(ctx->P, 256)), dim3(256), 0, s, ctx->Q, ctx->P, ctx->W_dev, J, S, ctx->dmat);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_residual_apply(exa_ctx* ctx, double* Y, hipStream_t s) {
   hipLaunchKernelGGL(k_residual_apply, dim3(nblk((int64_t)ctx->n * ctx->E, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->E, ctx->G_dev, ctx->dmat, Y);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_residual_apply_from(exa_ctx* ctx, const double* D, double* Y, hipStream_t s) {
   hipLaunchKernelGGL(k_residual_apply, dim3(nblk((int64_t)ctx->n * ctx->E, 256)), dim3(256), exa_g_lds_bytes(ctx->n, ctx->Q), s, ctx->Q, ctx->n, ctx->E, ctx->G_dev, D, Y);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_det_prepare(exa_ctx* ctx);   // capi.hip: builds the node -> element table on first use
static int det_gather(exa_ctx* ctx, double* y, const double* gate, hipStream_t s) {
   hipLaunchKernelGGL(k_e2l_gather<true>, dim3(nblk(ctx->nnodes, 256)), dim3(256), 0, s, ctx->n, ctx->nnodes, ctx->n2e_off, ctx->n2e_idx, ctx->ev_det, y, gate);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_residual_p1(exa_ctx* ctx, const double* J, const double* S, double* y, bool lvec, hipStream_t s) {
   const unsigned nb = nblk(ctx->E, PA_BLK);
   double* ev = nullptr;
   if (lvec && ctx->det) { if (int rc = exa_det_prepare(ctx)) return rc; ev = ctx->ev_det; }
   if (lvec && J == nullptr) {   // geometry from the nodal coordinates registered with exa_grad_set_coords
      if (ctx->qblk) hipLaunchKernelGGL((k_residual_p1<true, true, true>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->W_dev, J, S, y, ctx->conn, ctx->nnodes, ev, ctx->coords_lvec);
      else hipLaunchKernelGGL((k_residual_p1<true, false, true>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->W_dev, J, S, y, ctx->conn, ctx->nnodes, ev, ctx->coords_lvec);
   } else if (lvec && ctx->qblk) hipLaunchKernelGGL((k_residual_p1<true, true>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->W_dev, J, S, y, ctx->conn, ctx->nnodes, ev);
   else if (lvec) hipLaunchKernelGGL((k_residual_p1<true, false>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->W_dev, J, S, y, ctx->conn, ctx->nnodes, ev);
   else if (ctx->qblk) hipLaunchKernelGGL((k_residual_p1<false, true>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->W_dev, J, S, y, ctx->conn, ctx->nnodes);
   else hipLaunchKernelGGL((k_residual_p1<false, false>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->W_dev, J, S, y, ctx->conn, ctx->nnodes);
   EXA_HIP_CHECK(ctx, hipGetLastError());
   if (ev) return det_gather(ctx, y, nullptr, s);
   return EXA_OK;
}
// the 46-double records of a context whose last exa_grad_setup wrote the compact ones only: built now, from the arrays that call was given
static int pa_full_on_demand(exa_ctx* ctx, hipStream_t s) {
   if (!ctx->pa_lazy) return EXA_OK;
   const dim3 grid(nblk(ctx->E, PA_BLK));
   if (ctx->qblk) hipLaunchKernelGGL((k_grad_setup_pa<true, 0, false>), grid, dim3(PA_BLK), 0, s, ctx->Q, ctx->E, ctx->lazy_dt, ctx->W_dev, ctx->lazy_J, ctx->lazy_C, ctx->pa, ctx->pa_c);
   else hipLaunchKernelGGL((k_grad_setup_pa<false, 0, false>), grid, dim3(PA_BLK), 0, s, ctx->Q, ctx->E, ctx->lazy_dt, ctx->W_dev, ctx->lazy_J, ctx->lazy_C, ctx->pa, ctx->pa_c);
   EXA_HIP_CHECK(ctx, hipGetLastError());
   ctx->pa_lazy = false;
   return EXA_OK;
}

int exa_launch_grad_setup_pa(exa_ctx* ctx, double dt, const double* J, const double* C, hipStream_t s) {
   ctx->pa_lazy = false;
   const dim3 grid(nblk(ctx->E, PA_BLK));
#define GS_LAUNCH(QBV, NP, TR) hipLaunchKernelGGL((k_grad_setup_pa<QBV, NP, TR>), grid, dim3(PA_BLK), 0, s, ctx->Q, ctx->E, dt, ctx->W_dev, J, C, ctx->pa, ctx->pa_c)
#define GS_LAUNCH2(NP, TR) do { if (ctx->qblk) GS_LAUNCH(true, NP, TR); else GS_LAUNCH(false, NP, TR); } while (0)
   const bool trd = ctx->cfg.assembly == EXA_ASSEMBLY_EA;
   static const bool lazy_off = [] { const char* e = std::getenv("EXA_GRAD_SETUP_LAZY"); return e && std::strcmp(e, "off") == 0; }();   // A/B switch: both record sets in one pass, as in rounds 3-5
   if (ctx->pa_c && ctx->pac_pairs == PAC_PAIRS && !trd && !lazy_off) {
      // p = 1 partial assembly: the L-vector action with recomputed geometry streams the compact records alone (exa_grad_set_coords); the diagonal, the
      // E-vector action and an action without coordinates need the 46-double ones, which are built on demand - J and C must stay as they are until the next
      // exa_grad_setup (they do in a Newton iteration: the reference calls AssembleGradDiagonalPA right after AssembleGradPA, src/mechanics_operator.cpp:436-443)
      if (ctx->qblk) hipLaunchKernelGGL((k_grad_setup_pa<true, PAC_PAIRS, false, false>), grid, dim3(PA_BLK), 0, s, ctx->Q, ctx->E, dt, ctx->W_dev, J, C, ctx->pa, ctx->pa_c);
      else hipLaunchKernelGGL((k_grad_setup_pa<false, PAC_PAIRS, false, false>), grid, dim3(PA_BLK), 0, s, ctx->Q, ctx->E, dt, ctx->W_dev, J, C, ctx->pa, ctx->pa_c);
      ctx->pa_lazy = true; ctx->lazy_J = J; ctx->lazy_C = C; ctx->lazy_dt = dt;
   }
   else if (ctx->pa_c && ctx->pac_pairs == PAC_PAIRS_GEO && ctx->n == 8 && !trd && !lazy_off) {      // p = 1, E-vector action on the compact records with geometry: same scheme
      if (ctx->qblk) hipLaunchKernelGGL((k_grad_setup_pa<true, PAC_PAIRS_GEO, false, false>), grid, dim3(PA_BLK), 0, s, ctx->Q, ctx->E, dt, ctx->W_dev, J, C, ctx->pa, ctx->pa_c);
      else hipLaunchKernelGGL((k_grad_setup_pa<false, PAC_PAIRS_GEO, false, false>), grid, dim3(PA_BLK), 0, s, ctx->Q, ctx->E, dt, ctx->W_dev, J, C, ctx->pa, ctx->pa_c);
      ctx->pa_lazy = true; ctx->lazy_J = J; ctx->lazy_C = C; ctx->lazy_dt = dt;
   }
   else if (ctx->pa_c && ctx->pac_pairs == PAC_PAIRS) { if (trd) GS_LAUNCH2(PAC_PAIRS, true); else GS_LAUNCH2(PAC_PAIRS, false); }
   else if (ctx->pa_c) { if (trd) GS_LAUNCH2(PAC_PAIRS_GEO, true); else GS_LAUNCH2(PAC_PAIRS_GEO, false); }
   else GS_LAUNCH2(0, false);
#undef GS_LAUNCH2
#undef GS_LAUNCH
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
// blk0 / nblk_range: sub-range of the 64-element blocks (nblk_range < 0: all of them)
int exa_launch_grad_apply_p1(exa_ctx* ctx, const double* x, double* y, bool lvec, const uint8_t* mask, const double* gate, hipStream_t s, bool trans, int blk0, int nblk_range) {
   const unsigned nball = nblk(ctx->E, PA_BLK);
   const bool ranged = nblk_range >= 0;
   if (ranged && (!lvec || ctx->det || blk0 < 0 || (unsigned)(blk0 + nblk_range) > nball)) { ctx->err = "exa_launch_grad_apply_p1: block ranges are for the atomic L-vector action"; return EXA_ERR_ARG; }
   const unsigned nb = ranged ? (unsigned)nblk_range : nball;
   if (!ranged) blk0 = 0;
   if (nb == 0) return EXA_OK;
   if (!(lvec && ctx->coords_lvec && ctx->pa_c && ctx->pac_pairs == PAC_PAIRS) && !(!lvec && ctx->pa_c && ctx->pac_pairs == PAC_PAIRS_GEO && ctx->n == 8)) {
      if (int rc = pa_full_on_demand(ctx, s)) return rc;   // every other form streams the 46-double records
   }
   double* ev = nullptr;
   if (lvec && ctx->det) { if (int rc = exa_det_prepare(ctx)) return rc; ev = ctx->ev_det; }
   // record stream of this launch: non-temporal loads only when it is larger than what the caches keep from one launch to the next (exa_internal.hpp, exa_stream_nt)
   const bool nt = exa_stream_nt((size_t)ctx->P * 16 * ((lvec && ctx->coords_lvec && ctx->pa_c && ctx->pac_pairs == PAC_PAIRS) ? PAC_PAIRS : ((lvec && ctx->coords_lvec) ? 18 : PA_PAIRS)));
#define GA_LAUNCH(G, CM, T, REC, CRD) do { if (nt) hipLaunchKernelGGL((k_grad_apply_p1<true, G, CM, T, true>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, REC, x, y, ctx->conn, ctx->nnodes, mask, gate, CRD, ev, blk0); \
      else hipLaunchKernelGGL((k_grad_apply_p1<true, G, CM, T, false>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, REC, x, y, ctx->conn, ctx->nnodes, mask, gate, CRD, ev, blk0); } while (0)
   const double* none = nullptr;
   if (lvec && ctx->coords_lvec && ctx->pa_c && ctx->pac_pairs == PAC_PAIRS) GA_LAUNCH(true, true, false, ctx->pa_c, ctx->coords_lvec);   // D or D^T in the record
   else if (lvec && ctx->coords_lvec) { if (trans) GA_LAUNCH(true, false, true, ctx->pa, ctx->coords_lvec); else GA_LAUNCH(true, false, false, ctx->pa, ctx->coords_lvec); }
   else if (lvec) { if (trans) GA_LAUNCH(false, false, true, ctx->pa, none); else GA_LAUNCH(false, false, false, ctx->pa, none); }
   else if (ctx->pa_c && ctx->pac_pairs == PAC_PAIRS_GEO && ctx->n == 8)      // E-vector action on the compact records with geometry
      hipLaunchKernelGGL((k_grad_apply_p1<false, false, false, false, true, true>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->pa_c, x, y, ctx->conn, ctx->nnodes, mask, gate, none);
   else hipLaunchKernelGGL((k_grad_apply_p1<false, false>), dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->pa, x, y, ctx->conn, ctx->nnodes, mask, gate, none);
#undef GA_LAUNCH
   EXA_HIP_CHECK(ctx, hipGetLastError());
   if (ev) return det_gather(ctx, y, gate, s);
   return EXA_OK;
}
// max over the points of the relative deviation of C from the compact tangent form; result as the bit pattern of a double in *out_dev
int exa_launch_tangent_defect(exa_ctx* ctx, const double* C, unsigned long long* out_dev, hipStream_t s) {
   EXA_HIP_CHECK(ctx, hipMemsetAsync(out_dev, 0, sizeof(unsigned long long), s));
   if (ctx->qblk) hipLaunchKernelGGL(k_tangent_defect<true>, dim3(nblk(ctx->P, 256)), dim3(256), 0, s, ctx->Q, ctx->P, C, out_dev);
   else hipLaunchKernelGGL(k_tangent_defect<false>, dim3(nblk(ctx->P, 256)), dim3(256), 0, s, ctx->Q, ctx->P, C, out_dev);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_grad_diag_p1(exa_ctx* ctx, double* y, hipStream_t s) {
   if (int rc = pa_full_on_demand(ctx, s)) return rc;
   hipLaunchKernelGGL(k_grad_diag_p1, dim3(nblk(ctx->E, PA_BLK)), dim3(PA_BLK), 0, s, ctx->E, ctx->pa, y);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_assemble_ea_p1(exa_ctx* ctx, hipStream_t s) {
   hipLaunchKernelGGL(k_assemble_ea_p1, dim3(nblk(ctx->E, PA_BLK), 24), dim3(PA_BLK), 0, s, ctx->E, ctx->pa, ctx->emat);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_ea_apply_p1(exa_ctx* ctx, const double* x, double* y, bool lvec, const uint8_t* mask, const double* gate, hipStream_t s) {
   const unsigned nb = nblk(ctx->E, PA_BLK);
   if (lvec) hipLaunchKernelGGL(k_ea_apply_p1<true>, dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->emat, x, y, ctx->conn, ctx->nnodes, mask, gate);
   else hipLaunchKernelGGL(k_ea_apply_p1<false>, dim3(nb), dim3(PA_BLK), 0, s, ctx->E, ctx->emat, x, y, ctx->conn, ctx->nnodes, mask, gate);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_ea_diag_p1(exa_ctx* ctx, double* y, hipStream_t s) {
   hipLaunchKernelGGL(k_ea_diag_p1, dim3(nblk(ctx->E, PA_BLK)), dim3(PA_BLK), 0, s, ctx->E, ctx->emat, y);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_ea_export_p1(exa_ctx* ctx, double* out, hipStream_t s) {
   hipLaunchKernelGGL(k_ea_export_p1, dim3(nblk(ctx->E, PA_BLK)), dim3(PA_BLK), 0, s, ctx->E, ctx->emat, out);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_restrict(exa_ctx* ctx, const double* L, double* Ev, hipStream_t s) {
   hipLaunchKernelGGL(k_restrict, dim3(nblk((int64_t)ctx->n * ctx->E, 256)), dim3(256), 0, s, ctx->n, ctx->E, ctx->nnodes, ctx->conn, L, Ev);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_restrict_T(exa_ctx* ctx, const double* Ev, double* L, hipStream_t s) {
   if (ctx->det) {
      if (int rc = exa_det_prepare(ctx)) return rc;
      hipLaunchKernelGGL(k_e2l_gather<false>, dim3(nblk(ctx->nnodes, 256)), dim3(256), 0, s, ctx->n, ctx->nnodes, ctx->n2e_off, ctx->n2e_idx, Ev, L, (const double*)nullptr);
      EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
   }
   hipLaunchKernelGGL(k_restrict_T, dim3(nblk((int64_t)ctx->n * ctx->E, 256)), dim3(256), 0, s, ctx->n, ctx->E, ctx->nnodes, ctx->conn, Ev, L);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
int exa_launch_vol_avg(exa_ctx* ctx, const double* J, const double* qf, int vdim, double* partial, int nb, hipStream_t s) {
   if (ctx->qblk) hipLaunchKernelGGL(k_vol_avg_partial<true>, dim3(nb), dim3(256), sizeof(double) * 256, s, ctx->Q, ctx->P, vdim, ctx->W_dev, J, qf, partial);
   else hipLaunchKernelGGL(k_vol_avg_partial<false>, dim3(nb), dim3(256), sizeof(double) * 256, s, ctx->Q, ctx->P, vdim, ctx->W_dev, J, qf, partial);
   EXA_HIP_CHECK(ctx, hipGetLastError()); return EXA_OK;
}
