// Experiment (not product code): issue rate of v_fma_f64 on gfx950 as a function of the number of INDEPENDENT dependency chains per wave
// and of the waves per SIMD.  The constitutive kernel runs two waves per SIMD; if a dependent v_fma_f64 cannot issue every 8 cycles the
// serial sections of the point solve (LDL^T, triangular solves, Horner chains) leave the SIMD idle however few instructions they have.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/fma_latency scripts/experiments/fma_latency.hip ; run: /tmp/fma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NCH, typename T>
__global__ void k_chain(T* out, int iters, T a, T b) {
   T x[NCH];
   for (int c = 0; c < NCH; c++) x[c] = (T)(threadIdx.x + c);
   for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 16; u++)
#pragma unroll
         for (int c = 0; c < NCH; c++) x[c] = __builtin_fma(x[c], a, b);
   }
   T s = 0; for (int c = 0; c < NCH; c++) s += x[c];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH, typename T>
double run(int waves_per_simd, int iters) {
   hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
   const int cus = p.multiProcessorCount;
   const int threads = 64 * 4 * waves_per_simd;   // one block per CU
   T* out; (void)hipMalloc(&out, sizeof(T) * cus * threads);
   hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
   hipLaunchKernelGGL((k_chain<NCH, T>), dim3(cus), dim3(threads), 0, 0, out, 10, (T)0.999999, (T)1e-7);
   (void)hipEventRecord(e0);
   hipLaunchKernelGGL((k_chain<NCH, T>), dim3(cus), dim3(threads), 0, 0, out, iters, (T)0.999999, (T)1e-7);
   (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
   float ms; (void)hipEventElapsedTime(&ms, e0, e1);
   (void)hipFree(out);
   const double insts_per_wave = (double)iters * 16 * NCH;
   // cycles per instruction per SIMD at an assumed clock: report ns per (instruction x waves on the SIMD)
   return ms * 1e6 / (insts_per_wave * waves_per_simd);   // ns per VALU instruction issued on one SIMD
}
int main() {
   const int it = 20000;
   printf("ns per v_fma issued per SIMD (4 cycles at 2.4 GHz = 1.67 ns; at 1.9 GHz = 2.1 ns)\n");
   printf("%-8s %-6s %8s %8s %8s %8s %8s\n", "type", "waves", "1 chain", "2", "3", "4", "8");
   for (int w = 1; w <= 2; w++)
      printf("%-8s %-6d %8.2f %8.2f %8.2f %8.2f %8.2f\n", "f64", w, run<1, double>(w, it), run<2, double>(w, it), run<3, double>(w, it), run<4, double>(w, it), run<8, double>(w, it));
   for (int w = 1; w <= 2; w++)
      printf("%-8s %-6d %8.2f %8.2f %8.2f %8.2f %8.2f\n", "f32", w, run<1, float>(w, it), run<2, float>(w, it), run<3, float>(w, it), run<4, float>(w, it), run<8, float>(w, it));
   return 0;
}
