// Staged launches of the fused constitutive kernel for contexts in the reference's quadrature-function layout (EXA_QLAYOUT_AOS): what the MFEM adapters
// reach through exa_model_setup / exa_model_setup_lvec (reference: ExaCMechModel::ModelSetup, src/mechanics_ecmech.cpp:192-258, on QuadratureFunctions laid out
// (vdim, Q, E), src/mechanics_model.cpp:209-214).  The kernel is the one of model_kernel.hpp with STG = true: a wave owns 64 consecutive points, moves their
// contiguous rows with coalesced 16-byte accesses and transposes them through its LDS stash region.  Own translation unit: these instantiations compile
// beside the ones of model_kernels.hip.
#include "model_kernel.hpp"

int exa_prepare_tail_lists(exa_ctx*, hipStream_t);       // model_kernels.hip

namespace {

struct Args { double dt; double* J; const double* vel; const double* xl; const double* stress0; const double* state0; double* stress1; double* state1; double* cmat; };

// The launch sequence (full launch, dense launches of a tail split) of one instantiation.  STG: the staged kernel; its dense launches run the same kernel
// in its per-lane mode, so that a listed point gets the bits the full launch would have given it (tests/test_gpu_parity.py, test_tail_split_is_bitwise_neutral).
template <int KIN, bool LVEC, int NFIX, bool STG>
void launch_aos(exa_ctx* ctx, const Args& a, hipStream_t s) {
   const int bs = EXA_MODEL_BS;
   const int64_t nb = (ctx->P + bs - 1) / bs;
   launch_levels(ctx, nb, [&](int64_t blocks, int kcap, int* list, int mode, int* list_out, const double* rs_in, double* rs_out) {
      hipLaunchKernelGGL((k_model_setup<KIN, LVEC, NFIX, false, false, STG>), dim3((unsigned)blocks), dim3(bs),
                         model_lds_bytes(ctx, ecmdev::kin_is_km(KIN), false, false, mode, LVEC && NFIX == 8, STG && NFIX == 8), s,
                         ctx->mp, ctx->Q, ctx->n, ctx->P, a.dt, a.J, ctx->G_dev, a.vel, a.xl, ctx->conn, ctx->nnodes, a.stress0, a.state0, a.stress1, a.state1, a.cmat, ctx->fail_count_dev,
                         kcap, list, mode, (const double*)nullptr, 0, list_out, rs_in, rs_out);
   });
}

// Staged: trilinear elements get the unrolled node loops and the instantiations with the kinetics' exponents compiled in (the ones the element-blocked
// launches use); other orders the run-time node loop with the base instantiation of the model.
// Per lane (exa_set_aos_staging(ctx, 0), or arrays that do not start on a 16-byte boundary): the launches of rounds 1-5 - base instantiation, unrolled node
// loops for the trilinear L-vector form only.
template <int KIN, int KIN_CT>
void launch_kind(exa_ctx* ctx, bool ct, bool lv, const Args& a, bool staged, hipStream_t s) {
   if (!staged) {
      if (lv && ctx->n == 8) launch_aos<KIN, true, 8, false>(ctx, a, s);
      else if (lv) launch_aos<KIN, true, 0, false>(ctx, a, s);
      else launch_aos<KIN, false, 0, false>(ctx, a, s);
   } else if (ctx->n == 8) {
      if (ct) { if (lv) launch_aos<KIN_CT, true, 8, true>(ctx, a, s); else launch_aos<KIN_CT, false, 8, true>(ctx, a, s); }
      else { if (lv) launch_aos<KIN, true, 8, true>(ctx, a, s); else launch_aos<KIN, false, 8, true>(ctx, a, s); }
   } else { if (lv) launch_aos<KIN, true, 0, true>(ctx, a, s); else launch_aos<KIN, false, 0, true>(ctx, a, s); }
}

// the fused p = 1 launch that writes the compact gradient records (exa_model_setup_lvec_records) on the reference layout: rows staged, records per lane
template <int KIN>
void launch_aos_rec(exa_ctx* ctx, const Args& a, hipStream_t s) {
   const int bs = EXA_MODEL_BS;
   const int64_t nb = (ctx->P + bs - 1) / bs;
   const int trd = ctx->cfg.assembly == EXA_ASSEMBLY_EA;
   launch_levels(ctx, nb, [&](int64_t blocks, int kcap, int* list, int mode, int* list_out, const double* rs_in, double* rs_out) {
      hipLaunchKernelGGL((k_model_setup<KIN, true, 8, false, true, true>), dim3((unsigned)blocks), dim3(bs), model_lds_bytes(ctx, ecmdev::kin_is_km(KIN), false, false, mode, true, true), s,
                         ctx->mp, ctx->Q, ctx->n, ctx->P, a.dt, a.J, ctx->G_dev, a.vel, a.xl, ctx->conn, ctx->nnodes, a.stress0, a.state0, a.stress1, a.state1, ctx->pa_c, ctx->fail_count_dev,
                         kcap, list, mode, ctx->W_dev, trd, list_out, rs_in, rs_out);
   });
}

}  // namespace

int exa_launch_model_setup_aos_rec(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                                   double* stress1, double* state1, hipStream_t s) {
   if (ctx->qblk || ctx->n != 8 || !xl) return EXA_ERR_UNSUPPORTED;
   bool staged = ctx->aos_stage && ECM_STASH_STRIDE == 64;
   for (const void* p : { (const void*)J, (const void*)stress0, (const void*)state0, (const void*)stress1, (const void*)state1 })
      if (reinterpret_cast<uintptr_t>(p) & 15u) staged = false;
   if (!staged) { ctx->err = "exa_model_setup_lvec_records on the reference layout is the staged launch: exa_set_aos_staging must be on and the arrays 16-byte aligned"; return EXA_ERR_UNSUPPORTED; }
   EXA_HIP_CHECK(ctx, hipMemsetAsync(ctx->fail_count_dev, 0, sizeof(int), s));
   if (ctx->newton_cap > 0) { if (int rc = exa_prepare_tail_lists(ctx, s)) return rc; }
   const Args a{ dt, J, vel, xl, stress0, state0, stress1, state1, nullptr };
   switch (ctx->mp.kin) {
      case KIN_VOCE: if (voce_xn49(ctx)) launch_aos_rec<KIN_VOCE | KIN_XN49>(ctx, a, s); else launch_aos_rec<KIN_VOCE>(ctx, a, s); break;
      case KIN_VOCE_NL: if (voce_xn49(ctx)) launch_aos_rec<KIN_VOCE_NL | KIN_XN49>(ctx, a, s); else launch_aos_rec<KIN_VOCE_NL>(ctx, a, s); break;
#ifdef EXA_VARIANT_VOCE_ONLY
      default: ctx->err = "variant build without Kocks-Mecking kernels"; return EXA_ERR_UNSUPPORTED;
#else
      default:
         if (ECM_KM_DEFER && ctx->mp.with_g_athermal) { if (km_pq1(ctx)) launch_aos_rec<KIN_KMBALD_GA | KIN_PQ1>(ctx, a, s); else launch_aos_rec<KIN_KMBALD_GA>(ctx, a, s); }
         else { if (km_pq1(ctx)) launch_aos_rec<KIN_KMBALD | KIN_PQ1>(ctx, a, s); else launch_aos_rec<KIN_KMBALD>(ctx, a, s); }
         break;
#endif
   }
   EXA_HIP_CHECK(ctx, hipGetLastError());
   return EXA_OK;
}

// xl == nullptr: J is an input and vel an E-vector; otherwise xl / vel are L-vectors (byNODES) and J is written
int exa_launch_model_setup_aos(exa_ctx* ctx, double dt, double* J, const double* vel, const double* xl, const double* stress0, const double* state0,
                               double* stress1, double* state1, double* cmat, hipStream_t s) {
   // 16-byte pieces need every array on a 16-byte boundary (hipMalloc gives 256; a caller's sub-array might not: per-lane accesses then)
   bool staged = ctx->aos_stage && ECM_STASH_STRIDE == 64;
   for (const void* p : { (const void*)J, (const void*)stress0, (const void*)state0, (const void*)stress1, (const void*)state1, (const void*)cmat, (const void*)(xl ? nullptr : vel) })
      if (reinterpret_cast<uintptr_t>(p) & 15u) staged = false;
   EXA_HIP_CHECK(ctx, hipMemsetAsync(ctx->fail_count_dev, 0, sizeof(int), s));
   if (ctx->newton_cap > 0) { if (int rc = exa_prepare_tail_lists(ctx, s)) return rc; }
   const bool lv = xl != nullptr;
   const Args a{ dt, J, vel, xl, stress0, state0, stress1, state1, cmat };
   switch (ctx->mp.kin) {
      case KIN_VOCE: launch_kind<KIN_VOCE, KIN_VOCE | KIN_XN49>(ctx, voce_xn49(ctx), lv, a, staged, s); break;
      case KIN_VOCE_NL: launch_kind<KIN_VOCE_NL, KIN_VOCE_NL | KIN_XN49>(ctx, voce_xn49(ctx), lv, a, staged, s); break;
#ifdef EXA_VARIANT_VOCE_ONLY
      default: ctx->err = "variant build without Kocks-Mecking kernels"; return EXA_ERR_UNSUPPORTED;
#else
      default:
         if (ECM_KM_DEFER && ctx->mp.with_g_athermal) launch_kind<KIN_KMBALD_GA, KIN_KMBALD_GA | KIN_PQ1>(ctx, km_pq1(ctx), lv, a, staged, s);
         else launch_kind<KIN_KMBALD, KIN_KMBALD | KIN_PQ1>(ctx, km_pq1(ctx), lv, a, staged, s);
         break;
#endif
   }
   EXA_HIP_CHECK(ctx, hipGetLastError());
   return EXA_OK;
}
