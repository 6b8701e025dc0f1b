// The p = 2 constitutive launch of element-blocked contexts behind its geometry pre-pass (gen_kernels.hip, k_geom_p2): Jacobians and reference-space
// velocity gradients are formed once per ELEMENT and read back per point (k_model_setup, VG), instead of every one of an element's 27 points gathering the
// 27 nodes itself.  Reference: NonlinearMechOperator::Setup + ExaCMechModel::ModelSetup (src/mechanics_operator.cpp:310-391, src/mechanics_ecmech.cpp:192-258) at
// p_refinement = 2 (BASELINE config 5).  With records = true the launch writes the 18-pair records of the matrix-free p = 2 action (AssembleGradPA /
// AssembleEA fused in, reference src/mechanics_integrators.cpp:331-513, 756-1017) instead of the tangent field.  Own translation unit (compile time).
#include "model_kernel.hpp"

int exa_prepare_tail_lists(exa_ctx*, hipStream_t);                                                          // model_kernels.hip
int exa_launch_geom_p2(exa_ctx*, const double*, const double*, double*, double*, hipStream_t);              // gen_kernels.hip

namespace {

template <int KIN, bool REC>
void launch_p2(exa_ctx* ctx, double dt, double* J, const double* Lx, const double* stress0, const double* state0, double* stress1, double* state1, double* cmat, hipStream_t s) {
   const int bs = EXA_MODEL_BS;
   const int64_t nb = (((int64_t)((ctx->E + 63) / 64) * ctx->Q) + (bs / 64) - 1) / (bs / 64);      // one wave per (64-element block, q)
   const int trd = ctx->cfg.assembly == EXA_ASSEMBLY_EA;
   launch_levels(ctx, nb, [&](int64_t blocks, int kcap, int* list, int mode, int* list_out, const double* rs_in, double* rs_out) {
      hipLaunchKernelGGL((k_model_setup<KIN, false, 27, true, REC>), dim3((unsigned)blocks), dim3(bs), model_lds_bytes(ctx, ecmdev::kin_is_km(KIN), true, true, mode, false), s,
                         ctx->mp, ctx->Q, ctx->n, ctx->P, dt, J, (const double*)nullptr, Lx, (const double*)nullptr, ctx->conn, ctx->nnodes, stress0, state0, stress1, state1,
                         REC ? ctx->pa_c : cmat, ctx->fail_count_dev, kcap, list, mode, ctx->W_dev, trd, list_out, rs_in, rs_out);
   });
}

template <int KIN>
void launch_p2_kind(exa_ctx* ctx, bool rec, double dt, double* J, const double* Lx, const double* stress0, const double* state0, double* stress1, double* state1, double* cmat, hipStream_t s) {
   if (rec) launch_p2<KIN, true>(ctx, dt, J, Lx, stress0, state0, stress1, state1, cmat, s);
   else launch_p2<KIN, false>(ctx, dt, J, Lx, stress0, state0, stress1, state1, cmat, s);
}

}  // namespace

// xl / vel: L-vectors (byNODES); J: the Jacobian field, written by the pre-pass (the integrator kernels of p = 2 read it); cmat: tangent field (records = false)
int exa_launch_model_setup_p2(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                              double* stress1, double* state1, double* cmat, bool records, hipStream_t s) {
   if (ctx->n != 27 || !ctx->qblk || !J) { ctx->err = "p = 2 launch behind the geometry pre-pass: element-blocked layout and a Jacobian field"; return EXA_ERR_UNSUPPORTED; }
   if (!ctx->vgrad_ref) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->vgrad_ref, sizeof(double) * exa_qf_doubles(ctx, 9)));
   if (int rc = exa_launch_geom_p2(ctx, xl, vel, J, ctx->vgrad_ref, s)) return rc;
   EXA_HIP_CHECK(ctx, hipMemsetAsync(ctx->fail_count_dev, 0, sizeof(int), s));
   if (ctx->newton_cap > 0) { if (int rc = exa_prepare_tail_lists(ctx, s)) return rc; }
   const double* Lx = ctx->vgrad_ref;
   switch (ctx->mp.kin) {
      case KIN_VOCE:
         if (voce_xn49(ctx)) launch_p2_kind<KIN_VOCE | KIN_XN49>(ctx, records, dt, J, Lx, stress0, state0, stress1, state1, cmat, s);
         else launch_p2_kind<KIN_VOCE>(ctx, records, dt, J, Lx, stress0, state0, stress1, state1, cmat, s);
         break;
      case KIN_VOCE_NL:
         if (voce_xn49(ctx)) launch_p2_kind<KIN_VOCE_NL | KIN_XN49>(ctx, records, dt, J, Lx, stress0, state0, stress1, state1, cmat, s);
         else launch_p2_kind<KIN_VOCE_NL>(ctx, records, dt, J, Lx, stress0, state0, stress1, state1, cmat, s);
         break;
#ifdef EXA_VARIANT_VOCE_ONLY
      default: ctx->err = "variant build without Kocks-Mecking kernels"; return EXA_ERR_UNSUPPORTED;
#else
      default:
         if (ECM_KM_DEFER && ctx->mp.with_g_athermal) launch_p2_kind<KIN_KMBALD_GA>(ctx, records, dt, J, Lx, stress0, state0, stress1, state1, cmat, s);
         else launch_p2_kind<KIN_KMBALD>(ctx, records, dt, J, Lx, stress0, state0, stress1, state1, cmat, s);
         break;
#endif
   }
   EXA_HIP_CHECK(ctx, hipGetLastError());
   return EXA_OK;
}
