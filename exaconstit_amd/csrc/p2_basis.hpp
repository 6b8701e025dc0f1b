// Triquadratic hexahedron (p = 2) on the device: node numbering, one-dimensional basis rows and the two node contractions
// (gather of a nodal field's reference gradient at a point, scatter of a point's contribution to the 27 nodes) as three
// one-dimensional passes.  Shared by the matrix-free gradient action, the L-vector residual and the fused constitutive launch.
#pragma once
#include <cstdint>

namespace p2 {

constexpr int N = 27, ND = 81;
// lexicographic (i,j,k) -> native node of the triquadratic hexahedron (host_tables.cpp native_order(2); checked when the tables are uploaded)
__device__ constexpr int NAT[27] = { 0, 8, 1, 11, 20, 9, 3, 10, 2, 16, 21, 17, 24, 26, 22, 19, 23, 18, 4, 12, 5, 15, 25, 13, 7, 14, 6 };

// The one-dimensional tables [1D point][B0 B1 B2 D0 D1 D2] are read-only for the life of the context: through the constant address
// space a wave-uniform read is a scalar load (as plain global loads the compiler issues them per lane).
typedef const __attribute__((address_space(4))) double* cptr;
__device__ __forceinline__ cptr as_const(const double* p) { return (cptr)(uintptr_t)p; }

// dN_a/dxi(q) = D[qi][i] B[qj][j] B[qk][k] etc.: 18 coefficients per point instead of an 81-double table row
struct Rows { double bx[3], dx[3], by[3], dy[3], bz[3], dz[3]; };

__device__ __forceinline__ void load_rows(cptr t1, const int q, Rows& o) {
   const int qi = q % 3, qj = (q / 3) % 3, qk = q / 9;
#pragma unroll
   for (int d = 0; d < 3; d++) {
      o.bx[d] = t1[6 * qi + d]; o.dx[d] = t1[6 * qi + 3 + d]; o.by[d] = t1[6 * qj + d]; o.dy[d] = t1[6 * qj + 3 + d];
      o.bz[d] = t1[6 * qk + d]; o.dz[d] = t1[6 * qk + 3 + d];
   }
}

// gx[c][d] = sum_a x(a, c) dN_a/dxi_d at one point; x(a, c) = value of component c at native node a
template <class F>
__device__ __forceinline__ void gather(const Rows& r, F&& x, double (&gx)[3][3]) {
#pragma unroll
   for (int c = 0; c < 3; c++) {
      double bb[3] = { 0, 0, 0 }, bd[3] = { 0, 0, 0 }, db[3] = { 0, 0, 0 };     // per k: (Bx By), (Bx Dy), (Dx By) contracted over i, j
#pragma unroll
      for (int k = 0; k < 3; k++) {
#pragma unroll
         for (int j = 0; j < 3; j++) {
            const double x0 = x(NAT[0 + 3 * j + 9 * k], c), x1 = x(NAT[1 + 3 * j + 9 * k], c), x2 = x(NAT[2 + 3 * j + 9 * k], c);
            const double ub = fma(r.bx[2], x2, fma(r.bx[1], x1, r.bx[0] * x0));
            const double ud = fma(r.dx[2], x2, fma(r.dx[1], x1, r.dx[0] * x0));
            bb[k] = fma(r.by[j], ub, bb[k]); bd[k] = fma(r.dy[j], ub, bd[k]); db[k] = fma(r.by[j], ud, db[k]);
         }
      }
      gx[c][0] = fma(r.bz[2], db[2], fma(r.bz[1], db[1], r.bz[0] * db[0]));
      gx[c][1] = fma(r.bz[2], bd[2], fma(r.bz[1], bd[1], r.bz[0] * bd[0]));
      gx[c][2] = fma(r.dz[2], bb[2], fma(r.dz[1], bb[1], r.dz[0] * bb[0]));
      __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from fetching all 81 values up front (162 VGPRs)
   }
}

// Y[a + 27 c] += sum_d dN_a/dxi_d T[d][c]
__device__ __forceinline__ void scatter(const Rows& r, const double (&T)[3][3], double (&Y)[ND]) {
#pragma unroll
   for (int c = 0; c < 3; c++) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
         const double a0 = r.bz[k] * T[0][c], a1 = r.bz[k] * T[1][c], a2 = r.dz[k] * T[2][c];
#pragma unroll
         for (int j = 0; j < 3; j++) {
            const double pd = r.by[j] * a0;                          // multiplies Dx[i]
            const double pb = fma(r.by[j], a2, r.dy[j] * a1);        // multiplies Bx[i]
#pragma unroll
            for (int i = 0; i < 3; i++) {
               const int idx = NAT[i + 3 * j + 9 * k] + N * c;
               Y[idx] = fma(r.bx[i], pb, fma(r.dx[i], pd, Y[idx]));
            }
         }
      }
   }
}

}  // namespace p2
