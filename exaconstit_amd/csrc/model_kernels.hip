// Fused constitutive kernels (gfx950): one launch replaces the seven passes of
// ExaCMechModel::ModelSetup (reference src/mechanics_ecmech.cpp:192-258):
//   StressSetup/StateVarsSetup copies, matGrad/vel_grad zero fills, grad_calc (src/mechanics_kernels.cpp:36-77),
//   kernel_setup (src/mechanics_ecmech.cpp:42-99), getResponseECM, kernel_postprocessing (:116-171).
// One thread per quadrature point; the reference shape-derivative table is staged in LDS; per-point state is read
// and written once (928 B per point algorithmic traffic).  Variants (templates): E-vector + Jacobian inputs (the reference's
// ModelSetup signature) or fused L-vector gathers that also write the Jacobians; reference (AOS) or element-blocked quadrature-function
// layout; run-time tail split of long local solves into a dense second launch.
#include "model_kernel.hpp"

int exa_launch_model_setup_aos(exa_ctx*, double, double*, const double*, const double*, const double*, const double*, double*, double*, double*, hipStream_t);   // model_kernels_aos.hip
int exa_launch_model_setup_p2(exa_ctx*, double, double*, const double*, const double*, const double*, const double*, double*, double*, double*, bool, hipStream_t);   // model_kernels_p2.hip

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

// one launch of the sequence (launch_levels): the full launch (mode 0) or a dense launch over a list of handed-over points
template <int KIN, bool LVEC, int NFIX, bool QB>
static void launch_level(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                         double* stress1, double* state1, double* cmat, int64_t blocks, int kcap, int* list, int mode, int* list_out, const double* rs_in, double* rs_out, hipStream_t s) {
   const double* G = NFIX == 27 ? ctx->T1_dev : ctx->G_dev;
   hipLaunchKernelGGL((k_model_setup<KIN, LVEC, NFIX, QB>), dim3((unsigned)blocks), dim3(EXA_MODEL_BS), model_lds_bytes(ctx, ecmdev::kin_is_km(KIN), NFIX == 27, QB, mode, LVEC && NFIX == 8), s, ctx->mp, ctx->Q, ctx->n, ctx->P, dt, J, G, vel, xl, ctx->conn,
                      ctx->nnodes, stress0, state0, stress1, state1, cmat, ctx->fail_count_dev, kcap, list, mode, (const double*)nullptr, 0, list_out, rs_in, rs_out);
}

template <int KIN, bool LVEC, int NFIX, bool QB>
static void launch_model_q(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                         double* stress1, double* state1, double* cmat, hipStream_t s) {
   const int bs = EXA_MODEL_BS;
   // QB: one wave per (64-element block, q)
   const int64_t nb = QB ? (((int64_t)((ctx->E + 63) / 64) * ctx->Q) + (bs / 64) - 1) / (bs / 64) : (ctx->P + bs - 1) / bs;
   launch_levels(ctx, nb, [&](int64_t blocks, int kcap, int* list, int mode, int* list_out, const double* rs_in, double* rs_out) {
      launch_level<KIN, LVEC, NFIX, QB>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, blocks, kcap, list, mode, list_out, rs_in, rs_out, s);
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

// lists and solver-state buffers of the tail split, allocated on first use; the list counters are cleared for the coming launch sequence
int exa_prepare_tail_lists(exa_ctx* ctx, hipStream_t s) {
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

// element-blocked layout (the reference layout is served by model_kernels_aos.hip)
template <int KIN, bool LVEC>
static void launch_model(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                         double* stress1, double* state1, double* cmat, hipStream_t s) {
   if (LVEC && ctx->n == 8) launch_model_q<KIN, LVEC, 8, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
   else if (LVEC && ctx->n == 27) launch_model_q<KIN, true, 27, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);   // T1 uploaded by the caller
   else launch_model_q<KIN, LVEC, 0, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
}

// xl == nullptr: J is an input and vel an E-vector; otherwise xl / vel are L-vectors (byNODES) and J is written
int exa_launch_model_setup(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                           double* stress1, double* state1, double* cmat, hipStream_t s) {
   if (!ctx->qblk) return exa_launch_model_setup_aos(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);   // reference layout: staged or per-lane launches
   EXA_HIP_CHECK(ctx, hipMemsetAsync(ctx->fail_count_dev, 0, sizeof(int), s));
   if (ctx->newton_cap > 0) {
      if (int rc = exa_prepare_tail_lists(ctx, s)) return rc;
   }
   const bool lv = xl != nullptr;
   // p = 2, element-blocked, L-vector form: geometry pre-pass + point launch on its velocity gradients (model_kernels_p2.hip); EXA_P2_PREPASS=off keeps the
   // launch in which every point gathers its element's 27 nodes (A/B switch)
   if (lv && ctx->n == 27 && J != nullptr) {
      const char* e = std::getenv("EXA_P2_PREPASS");
      if (!(e && std::strcmp(e, "off") == 0)) return exa_launch_model_setup_p2(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, false, s);
   }
   if (lv && ctx->n == 27) { if (int rc = exa_ensure_p2_tables(ctx)) return rc; }
   switch (ctx->mp.kin) {
      case KIN_VOCE:
         if (lv && (ctx->n == 8 || ctx->n == 27) && voce_xn49(ctx)) {   // element-blocked fused launches with the exponent compiled in
            if (ctx->n == 8) launch_model_q<KIN_VOCE | KIN_XN49, true, 8, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
            else launch_model_q<KIN_VOCE | KIN_XN49, true, 27, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         } else if (lv) launch_model<KIN_VOCE, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         else launch_model<KIN_VOCE, false>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         break;
      case KIN_VOCE_NL:
         if (lv && (ctx->n == 8 || ctx->n == 27) && voce_xn49(ctx)) {
            if (ctx->n == 8) launch_model_q<KIN_VOCE_NL | KIN_XN49, true, 8, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
            else launch_model_q<KIN_VOCE_NL | KIN_XN49, true, 27, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         } else if (lv) launch_model<KIN_VOCE_NL, true>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         else launch_model<KIN_VOCE_NL, false>(ctx, dt, J, vel, xl, stress0, state0, stress1, state1, cmat, s);
         break;
#ifdef EXA_VARIANT_VOCE_ONLY
      default: ctx->err = "variant build without Kocks-Mecking kernels"; return EXA_ERR_UNSUPPORTED;
#else
      default:
         if (lv && ctx->n == 8 && km_pq1(ctx)) {   // p = 1 element-blocked route with the tangent field (Jacobi / element-assembly set-ups)
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
