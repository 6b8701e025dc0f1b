// Experiment: do the small dense contractions of the tangent block (two 5x5x5 products per point, Q5 L Q5^T with a
// different Q5 per point, exaconstit_amd/csrc/ecm_device.hpp tail of point_update) pay on v_mfma_f64_16x16x4_f64?
//   valu : one thread per point, D = Q (L Q^T) in registers, 250 FMAs per point                 (what the product kernel does)
//   mfma : the only way to batch products with DIFFERENT matrices on the matrix core is block-diagonal embedding: three 5x5 blocks per
//          16x16 tile, K = 16 -> 4 MFMA instructions per product per 3 points (375 useful of 4096 multiply-adds, 9 %)
//   peak : back-to-back independent MFMAs / FMAs, the ceiling of each pipe
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_tangent scripts/experiments/mfma_tangent.hip && /tmp/mfma_tangent
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double dvec4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_valu(int reps, double* out) {
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   double Q[5][5], L[5][5];
   for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) { Q[i][j] = (i == j ? 1.0 : 0.0) + 1e-3 * ((t + 3 * i + 7 * j) % 11); L[i][j] = 1.0 + 1e-3 * ((t + 5 * i + j) % 13); }
   for (int r = 0; r < reps; r++) {
      double T[5][5], D[5][5];
#pragma unroll
      for (int i = 0; i < 5; i++)
#pragma unroll
         for (int j = 0; j < 5; j++) { double s = 0; for (int k = 0; k < 5; k++) s += L[i][k] * Q[j][k]; T[i][j] = s; }
#pragma unroll
      for (int i = 0; i < 5; i++)
#pragma unroll
         for (int j = 0; j < 5; j++) { double s = 0; for (int k = 0; k < 5; k++) s += Q[i][k] * T[k][j]; D[i][j] = s; }
#pragma unroll
      for (int i = 0; i < 5; i++)
#pragma unroll
         for (int j = 0; j < 5; j++) L[i][j] = D[i][j] * 0.25;      // feed back so that nothing is hoisted
   }
   double s = 0; for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) s += L[i][j];
   out[t] = s;
}

// block-diagonal embedding: lane l of the wave holds A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15] of each K = 4 slice
__global__ void __launch_bounds__(256) k_mfma(int reps, double* out) {
   const int t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
   const int rc = lane & 15, kk = lane >> 4;
   double a[4], b[4];                                  // 4 K-slices of the two operands (block-diagonal: zero outside the 5x5 blocks)
   for (int s = 0; s < 4; s++) { const int k = 4 * s + kk; const bool in = (rc / 5 == k / 5) && rc < 15 && k < 15; a[s] = in ? 1.0 + 1e-3 * ((t + rc + k) % 7) : 0.0; b[s] = in ? 1.0 + 1e-3 * ((t + 2 * rc + k) % 5) : 0.0; }
   dvec4 acc = { 0, 0, 0, 0 };
   for (int r = 0; r < reps; r++) {
      dvec4 c = { 0, 0, 0, 0 };
#pragma unroll
      for (int s = 0; s < 4; s++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], c, 0, 0, 0);       // T = L Q^T  (3 points)
      // the second product needs T as an OPERAND: C/D layout (4 rows per lane) != A/B layout (1 value per lane): a cross-lane transpose
      // through LDS or DPP in a real kernel; here the cheapest possible stand-in (reuse one register) so that only the MFMA cost is measured
      dvec4 d = { 0, 0, 0, 0 };
#pragma unroll
      for (int s = 0; s < 4; s++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], c[s], d, 0, 0, 0);       // D = Q T
      acc += d; a[0] += 1e-9 * d[0];
   }
   out[t] = acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ void __launch_bounds__(256) k_mfma_peak(int reps, double* out) {
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   double a = 1.0 + 1e-6 * t, b = 1.0 - 1e-6 * t;
   dvec4 c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
   for (int r = 0; r < reps; r++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
   }
   out[t] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void __launch_bounds__(256) k_fma_peak(int reps, double* out) {
   const int t = blockIdx.x * blockDim.x + threadIdx.x;
   double x[8]; for (int i = 0; i < 8; i++) x[i] = 1.0 + 1e-6 * (t + i);
   const double m = 1.0 - 1e-9, c = 1e-9;
   for (int r = 0; r < reps; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = x[i] * m + c;
   }
   double s = 0; for (int i = 0; i < 8; i++) s += x[i];
   out[t] = s;
}

template <class F> static float timeit(F f) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); f(); (void)hipDeviceSynchronize(); (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
   const int blocks = 256 * 16, threads = 256, reps = 200;        // 4 waves per SIMD on every CU
   const long long nthreads = (long long)blocks * threads, nwaves = nthreads / 64;
   double* out; CK(hipMalloc(&out, sizeof(double) * nthreads));
   const float t_valu = timeit([&] { hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(threads), 0, 0, reps, out); });
   const float t_mfma = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, reps, out); });
   const float t_mp = timeit([&] { hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(threads), 0, 0, reps * 4, out); });
   const float t_fp = timeit([&] { hipLaunchKernelGGL(k_fma_peak, dim3(blocks), dim3(threads), 0, 0, reps * 16, out); });
   const double pts_valu = (double)nthreads * reps, pts_mfma = (double)nwaves * 3 * reps;
   printf("valu  (thread per point)      : %8.3f ms  %10.3e tangent rotations/s  (%.1f TFLOP/s useful, 500 flop per point)\n", t_valu, pts_valu / (t_valu * 1e-3), pts_valu * 500 / (t_valu * 1e-3) / 1e12);
   printf("mfma  (3 points per 16x16)    : %8.3f ms  %10.3e tangent rotations/s  (%.1f TFLOP/s useful; %.1f TFLOP/s issued)\n", t_mfma, pts_mfma / (t_mfma * 1e-3), pts_mfma * 500 / (t_mfma * 1e-3) / 1e12,
          (double)nwaves * reps * 8 * 2048 / (t_mfma * 1e-3) / 1e12);
   printf("peak  v_mfma_f64_16x16x4      : %8.3f ms  %.1f TFLOP/s\n", t_mp, (double)nwaves * reps * 4 * 4 * 2048 / (t_mp * 1e-3) / 1e12);
   printf("peak  v_fma_f64               : %8.3f ms  %.1f TFLOP/s\n", t_fp, (double)nthreads * reps * 16 * 8 * 2 / (t_fp * 1e-3) / 1e12);
   printf("ratio valu / mfma (rotations per second): %.1f\n", (pts_valu / t_valu) / (pts_mfma / t_mfma));
   return 0;
}
