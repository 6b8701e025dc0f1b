// Experiment (north star "lanes cooperating on a point"): the slip-system part of one residual + Jacobian evaluation of the
// Voce point problem - resolved shear stresses of 12 systems, power law (exponent 49 by repeated squaring) with its derivative, plastic
// deformation rate / spin (5 + 3), Jacobian blocks A = P G P^T (15) and B = Q G P^T (15) - with 1, 2 or 4 lanes per quadrature point.
//   LPP = 1 : one lane owns all 12 systems and all 38 outputs (what k_model_setup does)
//   LPP = 2 : each lane evaluates 6 systems and their contributions to all 38 sums, one butterfly round (ds_bpermute / DPP) completes them
//   LPP = 4 : 3 systems per lane, two butterfly rounds
// (after the all-reduce every lane of a point holds all 38 sums, which is what the 5x5 factorisation that follows needs)  Reported: points per second
// at full occupancy and the registers each variant needs.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/coop_lanes scripts/experiments/coop_lanes.hip && /tmp/coop_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__constant__ double cP[5][12], cQ[3][12];

__device__ __forceinline__ void kin(double tau, double g_i, double& gd, double& dg) {
   const double t = tau * g_i, at = fabs(t);
   double b = at, p = at;                       // |t|^49: 49 = 110001b
   b *= b; b *= b; b *= b; b *= b; p *= b;      // ^16 -> p = at^17
   b *= b; p *= b;                              // ^32 -> p = at^49
   const double temp = 1.0e0 * p;
   gd = temp * t; dg = temp * 50.0 * g_i;
}

// slip tables as compile-time constants (the production kernel folds its integer-coefficient tables into immediates the same way)
constexpr double TP(int c, int a) { return 0.1 * (((3 * c + 5 * a) % 9) - 4); }
constexpr double TQ(int c, int a) { return 0.1 * (((7 * c + 2 * a) % 9) - 4); }

template <int LPP>
__global__ void __launch_bounds__(256) k_eval(int reps, double* out) {
   constexpr int NS = 12 / LPP;                 // systems per lane
   const int t = blockIdx.x * blockDim.x + threadIdx.x, sub = threadIdx.x % LPP, pt = t / LPP;
   double k[5]; for (int c = 0; c < 5; c++) k[c] = 1e-2 * (1.0 + 0.1 * ((pt + 3 * c) % 7));
   const double g_i = 1.0 / 0.05;
   // LPP > 1: the lanes of a point run the same code on different systems, so their table rows are DATA (registers filled once), not immediates
   double P[NS][5], Q[NS][3];
   if (LPP > 1) for (int j = 0; j < NS; j++) { for (int c = 0; c < 5; c++) P[j][c] = cP[c][sub * NS + j]; for (int c = 0; c < 3; c++) Q[j][c] = cQ[c][sub * NS + j]; }
   double acc[38]; for (int i = 0; i < 38; i++) acc[i] = 0.0;
   for (int r = 0; r < reps; r++) {
      double o[38];
#pragma unroll
      for (int i = 0; i < 38; i++) o[i] = 0.0;
#pragma unroll
      for (int j = 0; j < NS; j++) {
         double pc[5], qc[3];
#pragma unroll
         for (int c = 0; c < 5; c++) pc[c] = (LPP == 1) ? TP(c, j) : P[j][c];
#pragma unroll
         for (int c = 0; c < 3; c++) qc[c] = (LPP == 1) ? TQ(c, j) : Q[j][c];
         double tau = 0;
#pragma unroll
         for (int c = 0; c < 5; c++) tau += pc[c] * k[c];
         double gd, dg; kin(tau, g_i, gd, dg);
         double gp[5];
#pragma unroll
         for (int c = 0; c < 5; c++) { o[c] += pc[c] * gd; gp[c] = dg * pc[c]; }
#pragma unroll
         for (int c = 0; c < 3; c++) o[5 + c] += qc[c] * gd;
         int e = 8;
#pragma unroll
         for (int i = 0; i < 5; i++)
#pragma unroll
            for (int jj = i; jj < 5; jj++) o[e++] += pc[i] * gp[jj];        // A, packed symmetric (15)
#pragma unroll
         for (int i = 0; i < 3; i++)
#pragma unroll
            for (int jj = 0; jj < 5; jj++) o[e++] += qc[i] * gp[jj];        // B (15)
      }
      if (LPP > 1) {                            // butterfly all-reduce of the 38 partial sums over the lanes of the point
#pragma unroll
         for (int m = 1; m < LPP; m <<= 1)
#pragma unroll
            for (int i = 0; i < 38; i++) o[i] += __shfl_xor(o[i], m, 64);
      }
#pragma unroll
      for (int i = 0; i < 38; i++) acc[i] += o[i];
      k[0] += 1e-12 * acc[0];                  // feed back
   }
   double s = 0; for (int i = 0; i < 38; i++) s += acc[i];
   out[t] = s;
}

template <int LPP> static void run(int blocks, int reps, double* out) {
   hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
   hipLaunchKernelGGL(k_eval<LPP>, dim3(blocks), dim3(256), 0, 0, reps, out); (void)hipDeviceSynchronize();
   (void)hipEventRecord(a); hipLaunchKernelGGL(k_eval<LPP>, dim3(blocks), dim3(256), 0, 0, reps, out); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
   float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
   hipFuncAttributes at; (void)hipFuncGetAttributes(&at, (const void*)k_eval<LPP>);
   const double pts = (double)blocks * 256 / LPP * reps;
   printf("lanes per point %d : %8.3f ms  %10.3e point-evaluations/s   %3d VGPRs  scratch %zu B\n", LPP, ms, pts / (ms * 1e-3), at.numRegs, (size_t)at.localSizeBytes);
}

int main() {
   double hP[5][12], hQ[3][12];
   for (int c = 0; c < 5; c++) for (int a = 0; a < 12; a++) hP[c][a] = 0.1 * (((3 * c + 5 * a) % 9) - 4);
   for (int c = 0; c < 3; c++) for (int a = 0; a < 12; a++) hQ[c][a] = 0.1 * (((7 * c + 2 * a) % 9) - 4);
   (void)hipMemcpyToSymbol(HIP_SYMBOL(cP), hP, sizeof(hP)); (void)hipMemcpyToSymbol(HIP_SYMBOL(cQ), hQ, sizeof(hQ));
   const int blocks = 256 * 16, reps = 200;
   double* out; if (hipMalloc(&out, sizeof(double) * blocks * 256) != hipSuccess) return 1;
   run<1>(blocks, reps, out); run<2>(blocks, reps, out); run<4>(blocks, reps, out);
   return 0;
}
