// C ABI of libexaconstit_hip.so — see include/exaconstit_hip.h for the reference interfaces each entry replaces.
#include <climits>
#include "exa_internal.hpp"
#include <cstring>
#include <cstdio>

int exa_launch_model_setup(exa_ctx*, double, double*, const double*, const double*, const double*, const double*, double*, double*, double*, hipStream_t);
int exa_launch_model_setup_rec(exa_ctx*, double, double*, const double*, const double*, const double*, const double*, double*, double*, hipStream_t);
int exa_launch_model_setup_p2(exa_ctx*, double, double*, const double*, const double*, const double*, const double*, double*, double*, double*, bool, hipStream_t);
int exa_launch_model_setup_aos_rec(exa_ctx*, double, double*, const double*, const double*, const double*, const double*, double*, double*, hipStream_t);
int exa_launch_init_state(exa_ctx*, double*, const double*, const double*, hipStream_t);
int exa_launch_state_normalize(exa_ctx*, double*, hipStream_t);
int exa_launch_nfev_hist(exa_ctx*, const double*, int*, hipStream_t);
int exa_launch_selftest_km_math(const double*, double*, int, hipStream_t);
int exa_launch_calc_dp(exa_ctx*, const double*, double*, hipStream_t);
int exa_launch_jacobians(exa_ctx*, const double*, double*, hipStream_t);
int exa_launch_jacobians_from_geom(exa_ctx*, const double*, double*, hipStream_t);
int exa_launch_grad_calc(exa_ctx*, const double*, const double*, double*, hipStream_t);
int exa_launch_residual_setup(exa_ctx*, const double*, const double*, hipStream_t);
int exa_launch_residual_apply(exa_ctx*, double*, hipStream_t);
int exa_launch_residual_p1(exa_ctx*, const double*, const double*, double*, bool, hipStream_t);
int exa_launch_grad_setup_pa(exa_ctx*, double, const double*, const double*, hipStream_t);
int exa_launch_grad_apply_p1(exa_ctx*, const double*, double*, bool, const uint8_t*, const double*, hipStream_t, bool trans = false, int blk0 = 0, int nblk_range = -1);
int exa_launch_grad_diag_p1(exa_ctx*, double*, hipStream_t);
int exa_launch_assemble_ea_p1(exa_ctx*, hipStream_t);
int exa_launch_ea_apply_p1(exa_ctx*, const double*, double*, bool, const uint8_t*, const double*, hipStream_t);
int exa_launch_ea_diag_p1(exa_ctx*, double*, hipStream_t);
int exa_launch_ea_export_p1(exa_ctx*, double*, hipStream_t);
int exa_launch_restrict(exa_ctx*, const double*, double*, hipStream_t);
int exa_launch_restrict_T(exa_ctx*, const double*, double*, hipStream_t);
int exa_launch_vol_avg(exa_ctx*, const double*, const double*, int, double*, int, hipStream_t);
int exa_launch_eds(exa_ctx*, const double*, hipStream_t);
int exa_launch_residual_bbar(exa_ctx*, const double*, const double*, double*, hipStream_t);
int exa_launch_assemble_ea_gen(exa_ctx*, hipStream_t);
int exa_launch_ea_apply_gen(exa_ctx*, const double*, double*, bool, const uint8_t*, const double*, hipStream_t);
int exa_launch_mf_apply_p2(exa_ctx*, const double*, double*, const uint8_t*, const double*, bool, hipStream_t);
int exa_launch_residual_p2(exa_ctx*, const double*, const double*, double*, hipStream_t);
int exa_launch_tangent_defect(exa_ctx*, const double*, unsigned long long*, hipStream_t);
int exa_launch_ea_diag_gen(exa_ctx*, double*, hipStream_t);
int exa_launch_ea_export_gen(exa_ctx*, double*, hipStream_t);
int exa_launch_pa_apply_gen(exa_ctx*, const double*, double*, hipStream_t);
int exa_launch_pa_diag_gen(exa_ctx*, double*, hipStream_t);

namespace {
constexpr int VOL_AVG_BLOCKS = 512;
inline hipStream_t S(exa_stream s) { return reinterpret_cast<hipStream_t>(s); }
int fail(exa_ctx* ctx, int code, const char* msg) { if (ctx) ctx->err = msg; return code; }
}

extern "C" {

exa_ctx* exa_create(const exa_config* cfg, int* err) {
   auto set = [&](int e) { if (err) *err = e; };
   // p = 1 and p = 2 have tuned kernels for every entry point; orders 3 ... 6 (what the reference's own unit tests run at:
   // test/mechanics_test.cpp:54,187,313,471,630) go through the run-time-order kernels.  Point ids of the tail-split list are 32-bit
   if (!cfg || cfg->nelems <= 0 || cfg->order < 1 || cfg->order > 6) { set(EXA_ERR_ARG); return nullptr; }
   { const int64_t np1 = cfg->order + 1; if ((int64_t)cfg->nelems * np1 * np1 * np1 >= (int64_t)INT32_MAX) { set(EXA_ERR_ARG); return nullptr; } }
   exa_ctx* ctx = new exa_ctx();
   ctx->cfg = *cfg; ctx->cfg.props = nullptr;
   if (!exa_fill_mat_params(*cfg, ctx->mp, ctx->hist_init, ctx->err)) { std::fprintf(stderr, "exa_create: %s\n", ctx->err.c_str()); delete ctx; set(EXA_ERR_ARG); return nullptr; }
   if (cfg->integ != EXA_INTEG_FULL && cfg->integ != EXA_INTEG_BBAR) { delete ctx; set(EXA_ERR_ARG); return nullptr; }
   // the reference has no partial-assembly gradient for B-bar (README.md:20; ICExaNLFIntegrator does not override AddMultGradPA)
   if (cfg->integ == EXA_INTEG_BBAR && cfg->assembly != EXA_ASSEMBLY_EA) { std::fprintf(stderr, "exa_create: integ_model BBAR requires element (or full) assembly\n"); delete ctx; set(EXA_ERR_UNSUPPORTED); return nullptr; }
   ctx->p = cfg->order; const int np = ctx->p + 1; ctx->n = np * np * np; ctx->Q = ctx->n; ctx->E = cfg->nelems; ctx->P = (int64_t)ctx->E * ctx->Q;
   ctx->nstatev = ecmdev::NSTATEV;
   if (cfg->device >= 0) { if (hipSetDevice(cfg->device) != hipSuccess) { delete ctx; set(EXA_ERR_HIP); return nullptr; } }
   if (hipGetDevice(&ctx->device) != hipSuccess) { std::fprintf(stderr, "exa_create: no HIP device available\n"); delete ctx; set(EXA_ERR_HIP); return nullptr; }
   if (ecmdev::kin_is_km(ctx->mp.kin) && (ctx->mp.p != 1.0 || ctx->mp.q != 1.0)) {
      // the one reference-held vector that exercises this regime (test/data/mtsdd_full_auto_stress.txt) is not reproduced by the CPU restatement
      // the kernels are checked against: say so once per process instead of running silently
      static bool told = false;
      if (!told) { told = true; std::fprintf(stderr, "exaconstit_hip: Kocks-Mecking kinetics with p = %g, q = %g: the regime p, q != 1 is not pinned to a reference vector (DESIGN.md section 5)\n", ctx->mp.p, ctx->mp.q); }
   }
   exa_build_ref_elem(ctx->p, ctx->G_host, ctx->W_host);
   bool ok = true;
   ok = ok && hipMalloc(&ctx->G_dev, sizeof(double) * ctx->G_host.size()) == hipSuccess;
   ok = ok && hipMalloc(&ctx->W_dev, sizeof(double) * ctx->W_host.size()) == hipSuccess;
   ok = ok && hipMalloc(&ctx->fail_count_dev, sizeof(int)) == hipSuccess;
   ctx->scratch_bytes = sizeof(double) * VOL_AVG_BLOCKS * 64;
   ok = ok && hipMalloc(&ctx->scratch_dev, ctx->scratch_bytes) == hipSuccess;
   if (ok) {
      ok = ok && hipMemcpy(ctx->G_dev, ctx->G_host.data(), sizeof(double) * ctx->G_host.size(), hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMemcpy(ctx->W_dev, ctx->W_host.data(), sizeof(double) * ctx->W_host.size(), hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMemset(ctx->fail_count_dev, 0, sizeof(int)) == hipSuccess;
   }
   if (!ok) { std::fprintf(stderr, "exa_create: device allocation failed\n"); exa_destroy(ctx); set(EXA_ERR_HIP); return nullptr; }
   set(EXA_OK);
   return ctx;
}

void exa_destroy(exa_ctx* ctx) {
   if (!ctx) return;
   (void)hipFree(ctx->n2e_off); (void)hipFree(ctx->n2e_idx); (void)hipFree(ctx->ev_det);
   (void)hipFree(ctx->G_dev); (void)hipFree(ctx->W_dev); (void)hipFree(ctx->fail_count_dev); (void)hipFree(ctx->tail_dev); (void)hipFree(ctx->tail2_dev); (void)hipFree(ctx->resume_dev[0]); (void)hipFree(ctx->resume_dev[1]); (void)hipFree(ctx->scratch_dev);
   (void)hipFree(ctx->dmat); (void)hipFree(ctx->pa); (void)hipFree(ctx->emat); (void)hipFree(ctx->eDS); (void)hipFree(ctx->T1_dev); (void)hipFree(ctx->pa_c); (void)hipFree(ctx->tbuf); (void)hipFree(ctx->vgrad_ref);
   delete ctx;
}

const char* exa_last_error(const exa_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int exa_num_state_vars(const exa_ctx* ctx) { return ctx ? ctx->nstatev : EXA_ERR_ARG; }
int exa_nodes_per_elem(const exa_ctx* ctx) { return ctx ? ctx->n : EXA_ERR_ARG; }
int exa_qpts_per_elem(const exa_ctx* ctx) { return ctx ? ctx->Q : EXA_ERR_ARG; }

int exa_shape_table(const exa_ctx* ctx, double* G_host, double* W_host) {
   if (!ctx) return EXA_ERR_ARG;
   if (G_host) std::memcpy(G_host, ctx->G_host.data(), sizeof(double) * ctx->G_host.size());
   if (W_host) std::memcpy(W_host, ctx->W_host.data(), sizeof(double) * ctx->W_host.size());
   return EXA_OK;
}

int exa_set_quadrature_layout(exa_ctx* ctx, int layout) {
   if (!ctx || (layout != EXA_QLAYOUT_AOS && layout != EXA_QLAYOUT_EB64)) return fail(ctx, EXA_ERR_ARG, "exa_set_quadrature_layout: bad argument");
   if (layout == EXA_QLAYOUT_EB64 && !((ctx->p == 1 && ctx->cfg.integ == EXA_INTEG_FULL) || ctx->p == 2))
      return fail(ctx, EXA_ERR_UNSUPPORTED, "exa_set_quadrature_layout: the element-blocked layout is built for p = 1 full integration and for p = 2");
   ctx->qblk = (layout == EXA_QLAYOUT_EB64); ctx->have_resid = false; ctx->have_grad = false;
   return EXA_OK;
}
int exa_set_aos_staging(exa_ctx* ctx, int on) { if (!ctx) return EXA_ERR_ARG; ctx->aos_stage = on != 0; return EXA_OK; }
int exa_get_aos_staging(const exa_ctx* ctx) { return (ctx && ctx->aos_stage) ? 1 : 0; }
int exa_get_quadrature_layout(const exa_ctx* ctx) { return (ctx && ctx->qblk) ? EXA_QLAYOUT_EB64 : EXA_QLAYOUT_AOS; }
int64_t exa_qf_size(const exa_ctx* ctx, int vdim) { return (ctx && vdim > 0) ? (int64_t)exa_qf_doubles(ctx, vdim) : -1; }

int exa_init_state(exa_ctx* ctx, double* state0, const double* quats, exa_stream s) {
   if (!ctx || !state0 || !quats) return fail(ctx, EXA_ERR_ARG, "exa_init_state: null pointer");
   double* hist_dev = ctx->scratch_dev;   // 26 doubles
   EXA_HIP_CHECK(ctx, hipMemcpyAsync(hist_dev, ctx->hist_init, sizeof(double) * ecmdev::NUM_HIST, hipMemcpyHostToDevice, S(s)));
   return exa_launch_init_state(ctx, state0, quats, hist_dev, S(s));
}

int exa_state_normalize(exa_ctx* ctx, double* state, exa_stream s) {
   if (!ctx || !state) return fail(ctx, EXA_ERR_ARG, "exa_state_normalize: null pointer");
   return exa_launch_state_normalize(ctx, state, S(s));
}

// exa_set_newton_cap_auto: the cap of the NEXT launches from the evaluation counts this launch left in state1 (slot 3).  The distribution drifts slowly: the first
// four launches and every fourth one after them are looked at (a histogram launch + one 256-byte read-back, which waits for the stream) - the schedule and the cost
// model of the stand-alone driver (host/driver.hip, NonlinearMechOperator::Setup, choose_newton_cap), which keeps its own call so that its launch timers stay clean.
extern "C" int exa_choose_newton_cap(const int* hist64, double tail_cost);   // host/driver_capi.hip (include/exaconstit_driver.h)
static int cap_controller(exa_ctx* ctx, int rc, const double* state1, exa_stream s) {
   if (rc != EXA_OK || !ctx->cap_auto) return rc;
   ctx->cap_calls++;
   if (!(ctx->cap_calls <= 4 || ctx->cap_calls % 4 == 0)) return rc;
   int h[64];
   if (int r = exa_model_nfev_hist(ctx, state1, h, s)) return r;
   return exa_set_newton_caps(ctx, exa_choose_newton_cap(h, ctx->cap_tail_cost), 0, ctx->tail_resume);
}

int exa_set_newton_cap_auto(exa_ctx* ctx, int mode, double tail_cost) {
   if (!ctx || mode < 0 || mode > 2) return fail(ctx, EXA_ERR_ARG, "exa_set_newton_cap_auto: mode 0 (off), 1 (Kocks-Mecking models only) or 2 (every model)");
   if (mode && ctx->P >= (int64_t)INT32_MAX) return fail(ctx, EXA_ERR_ARG, "exa_set_newton_cap_auto: the deferred-point list holds 32-bit point ids (P < 2^31)");
   const bool km = ecmdev::kin_is_km(ctx->mp.kin);
   ctx->cap_auto = (mode == 2 || (mode == 1 && km)) ? 1 : 0;
   ctx->cap_tail_cost = tail_cost > 0.0 ? tail_cost : (km ? 1.5 : 4.0);   // dense launch's cost per evaluation relative to the full launch's (measured at 128^3, host/driver.hip)
   ctx->cap_calls = 0;
   if (!mode) { ctx->newton_cap = 0; ctx->newton_cap2 = 0; }
   return EXA_OK;
}
int exa_get_newton_cap(exa_ctx* ctx) { return ctx ? ctx->newton_cap : EXA_ERR_ARG; }

int exa_model_setup(exa_ctx* ctx, double dt, const double* J, const double* vel, const double* stress0, const double* state0,
                    double* stress1, double* state1, double* ddsdde, exa_stream s) {
   if (!ctx || !J || !vel || !stress0 || !state0 || !stress1 || !state1 || !ddsdde) return fail(ctx, EXA_ERR_ARG, "exa_model_setup: null pointer");
   if (!(dt > 0.0)) return fail(ctx, EXA_ERR_ARG, "exa_model_setup: dt must be positive");
   return cap_controller(ctx, exa_launch_model_setup(ctx, dt, const_cast<double*>(J), vel, nullptr, stress0, state0, stress1, state1, ddsdde, S(s)), state1, s);
}

int exa_model_setup_lvec(exa_ctx* ctx, double dt, const double* x_lvec, const double* v_lvec, const double* stress0, const double* state0,
                         double* stress1, double* state1, double* ddsdde, double* J_out, exa_stream s) {
   if (!ctx || !x_lvec || !v_lvec || !stress0 || !state0 || !stress1 || !state1 || !ddsdde || !J_out) return fail(ctx, EXA_ERR_ARG, "exa_model_setup_lvec: null pointer");
   if (!ctx->conn) return fail(ctx, EXA_ERR_STATE, "exa_model_setup_lvec: call exa_set_connectivity first");
   if (!(dt > 0.0)) return fail(ctx, EXA_ERR_ARG, "exa_model_setup_lvec: dt must be positive");
   return cap_controller(ctx, exa_launch_model_setup(ctx, dt, J_out, v_lvec, x_lvec, stress0, state0, stress1, state1, ddsdde, S(s)), state1, s);
}

int exa_model_setup_lvec_records(exa_ctx* ctx, double dt, const double* x_lvec, const double* v_lvec, const double* stress0, const double* state0,
                                 double* stress1, double* state1, double* J_out, exa_stream s) {
   if (!ctx || !x_lvec || !v_lvec || !stress0 || !state0 || !stress1 || !state1) return fail(ctx, EXA_ERR_ARG, "exa_model_setup_lvec_records: null pointer");   // J_out may be null
   if (!ctx->conn) return fail(ctx, EXA_ERR_STATE, "exa_model_setup_lvec_records: call exa_set_connectivity first");
   if (!(dt > 0.0)) return fail(ctx, EXA_ERR_ARG, "exa_model_setup_lvec_records: dt must be positive");
   const bool p1 = ctx->p == 1 && ctx->cfg.integ == EXA_INTEG_FULL, p2 = ctx->p == 2;      // p = 2: plain or B-bar, behind the geometry pre-pass (a Jacobian field is part of it)
   if (!(p1 || p2) || (!ctx->qblk && !p1) || ctx->tangent_form != EXA_TANGENT_DEV5_BULK || !(ctx->cfg.assembly == EXA_ASSEMBLY_PA || ctx->ea_matfree))
      return fail(ctx, EXA_ERR_UNSUPPORTED, "exa_model_setup_lvec_records: needs p = 1 full integration (either layout) or p = 2 (element-blocked layout), the compact tangent form and PA or matrix-free EA");
   if (p2 && !J_out) return fail(ctx, EXA_ERR_ARG, "exa_model_setup_lvec_records: at p = 2 the Jacobian field is not optional (the pre-pass writes it, the launch and the residual read it)");
   const int npair = p1 ? PAC_PAIRS : PAC_PAIRS_GEO;
   if (ctx->pa_c && ctx->pac_pairs != npair) return fail(ctx, EXA_ERR_STATE, "exa_model_setup_lvec_records: compact records of another shape exist");
   if (!ctx->pa_c) {
      ctx->pac_pairs = npair;
      EXA_HIP_CHECK(ctx, hipMalloc(&ctx->pa_c, (size_t)((ctx->E + PA_BLK - 1) / PA_BLK) * ctx->Q * 2 * npair * PA_BLK * sizeof(double)));
      EXA_HIP_CHECK(ctx, hipMemsetAsync(ctx->pa_c, 0, (size_t)((ctx->E + PA_BLK - 1) / PA_BLK) * ctx->Q * 2 * npair * PA_BLK * sizeof(double), S(s)));
   }
   const int rc = p2 ? exa_launch_model_setup_p2(ctx, dt, J_out, v_lvec, x_lvec, stress0, state0, stress1, state1, nullptr, true, S(s))
                     : !ctx->qblk ? exa_launch_model_setup_aos_rec(ctx, dt, J_out, v_lvec, x_lvec, stress0, state0, stress1, state1, S(s))
                     : exa_launch_model_setup_rec(ctx, dt, J_out, v_lvec, x_lvec, stress0, state0, stress1, state1, S(s));
   if (rc == EXA_OK) { ctx->have_grad = true; ctx->emat_valid = false; ctx->grad_records_only = true; }
   return cap_controller(ctx, rc, state1, s);
}

// driver-internal: device address of the failed-point counter of the last constitutive launch (consumed on the device by the residual norm)
const int* exa_model_fail_counter_dev(exa_ctx* ctx) { return ctx ? ctx->fail_count_dev : nullptr; }

int exa_model_status(exa_ctx* ctx, exa_stream s) {
   if (!ctx) return EXA_ERR_ARG;
   int h = 0;
   EXA_HIP_CHECK(ctx, hipMemcpyAsync(&h, ctx->fail_count_dev, sizeof(int), hipMemcpyDeviceToHost, S(s)));
   EXA_HIP_CHECK(ctx, hipStreamSynchronize(S(s)));
   return h;
}

// the return convention SURVEY 8(b) specifies for the model seam in one call: < 0 error, 0 ok, > 0 quadrature points whose local solve failed
int exa_model_setup_checked(exa_ctx* ctx, double dt, const double* J, const double* vel, const double* stress0, const double* state0,
                            double* stress1, double* state1, double* ddsdde, exa_stream s) {
   const int rc = exa_model_setup(ctx, dt, J, vel, stress0, state0, stress1, state1, ddsdde, s);
   return rc != EXA_OK ? rc : exa_model_status(ctx, s);
}

int exa_set_newton_cap(exa_ctx* ctx, int max_evals) {
   if (!ctx || (max_evals != 0 && max_evals < 2)) return fail(ctx, EXA_ERR_ARG, "exa_set_newton_cap: 0 (off) or >= 2");
   if (max_evals && ctx->P >= (int64_t)INT32_MAX) return fail(ctx, EXA_ERR_ARG, "exa_set_newton_cap: the deferred-point list holds 32-bit point ids (P < 2^31)");
   ctx->newton_cap = max_evals; ctx->newton_cap2 = 0; return EXA_OK;
}
int exa_set_newton_caps(exa_ctx* ctx, int max_evals, int max_evals_2, int resume) {
   if (ctx && max_evals_2 != 0 && (max_evals == 0 || max_evals_2 <= max_evals || !resume)) return fail(ctx, EXA_ERR_ARG, "exa_set_newton_caps: the second cap needs a first one below it and resume = 1");
   if (int rc = exa_set_newton_cap(ctx, max_evals)) return rc;
   ctx->newton_cap2 = max_evals_2; ctx->tail_resume = resume ? 1 : 0; return EXA_OK;
}
int exa_model_nfev_hist(exa_ctx* ctx, const double* state, int* hist64_host, exa_stream s) {
   if (!ctx || !state || !hist64_host) return fail(ctx, EXA_ERR_ARG, "exa_model_nfev_hist: null pointer");
   int* hd = reinterpret_cast<int*>(ctx->scratch_dev);
   int rc = exa_launch_nfev_hist(ctx, state, hd, S(s));
   if (rc) return rc;
   EXA_HIP_CHECK(ctx, hipMemcpyAsync(hist64_host, hd, sizeof(int) * 64, hipMemcpyDeviceToHost, S(s)));
   EXA_HIP_CHECK(ctx, hipStreamSynchronize(S(s)));
   return EXA_OK;
}
int exa_selftest_km_math(const double* x_dev, double* out_dev, int n, exa_stream s) {
   if (!x_dev || !out_dev || n <= 0) return EXA_ERR_ARG;
   return exa_launch_selftest_km_math(x_dev, out_dev, n, S(s));
}
int exa_model_tail_count(exa_ctx* ctx, exa_stream s) {
   if (!ctx) return EXA_ERR_ARG;
   if (!ctx->tail_dev || ctx->newton_cap <= 0) return 0;
   int h = 0;
   EXA_HIP_CHECK(ctx, hipMemcpyAsync(&h, ctx->tail_dev, sizeof(int), hipMemcpyDeviceToHost, S(s)));
   EXA_HIP_CHECK(ctx, hipStreamSynchronize(S(s)));
   return h;
}

int exa_calc_dp(exa_ctx* ctx, const double* state, double* dp, exa_stream s) {
   if (!ctx || !state || !dp) return fail(ctx, EXA_ERR_ARG, "exa_calc_dp: null pointer");
   return exa_launch_calc_dp(ctx, state, dp, S(s));
}

int exa_jacobians(exa_ctx* ctx, const double* xe, double* J, exa_stream s) {
   if (!ctx || !xe || !J) return fail(ctx, EXA_ERR_ARG, "exa_jacobians: null pointer");
   return exa_launch_jacobians(ctx, xe, J, S(s));
}

int exa_jacobians_from_geom(exa_ctx* ctx, const double* gj, double* J, exa_stream s) {
   if (!ctx || !gj || !J) return fail(ctx, EXA_ERR_ARG, "exa_jacobians_from_geom: null pointer");
   return exa_launch_jacobians_from_geom(ctx, gj, J, S(s));
}

int exa_grad_calc(exa_ctx* ctx, const double* J, const double* fe, double* out, exa_stream s) {
   if (!ctx || !J || !fe || !out) return fail(ctx, EXA_ERR_ARG, "exa_grad_calc: null pointer");
   return exa_launch_grad_calc(ctx, J, fe, out, S(s));
}

int exa_residual_setup(exa_ctx* ctx, const double* J, const double* stress1, exa_stream s) {
   if (!ctx || !J || !stress1) return fail(ctx, EXA_ERR_ARG, "exa_residual_setup: null pointer");
   if (ctx->qblk) return fail(ctx, EXA_ERR_UNSUPPORTED, "exa_residual_setup: E-vector residual is AOS-only; with the element-blocked layout use exa_residual_lvec");
   ctx->have_resid = true;
   if (ctx->cfg.integ == EXA_INTEG_BBAR) {      // ICExaNLFIntegrator::AssemblePA: element-average gradient; J and sigma are read by AddMultPA
      if (!ctx->eDS) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->eDS, sizeof(double) * 3 * ctx->n * PA_BLK * (size_t)((ctx->E + PA_BLK - 1) / PA_BLK)));
      ctx->resid_J = J; ctx->resid_S = stress1;
      return exa_launch_eds(ctx, J, S(s));
   }
   if (!ctx->dmat) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->dmat, sizeof(double) * 9 * ctx->P));
   return exa_launch_residual_setup(ctx, J, stress1, S(s));
}

int exa_residual_apply(exa_ctx* ctx, double* y, exa_stream s) {
   if (!ctx || !y) return fail(ctx, EXA_ERR_ARG, "exa_residual_apply: null pointer");
   if (!ctx->have_resid) return fail(ctx, EXA_ERR_STATE, "exa_residual_apply called before exa_residual_setup");
   if (ctx->cfg.integ == EXA_INTEG_BBAR) return exa_launch_residual_bbar(ctx, ctx->resid_J, ctx->resid_S, y, S(s));
   return exa_launch_residual_apply(ctx, y, S(s));
}

// element-assembly contexts whose L-vector action is computed from the point records (exa_set_ea_matrix_free)
static bool ea_from_records(const exa_ctx* ctx) {
   return ctx->ea_matfree && (ctx->n == 27 || (ctx->p == 1 && ctx->cfg.integ == EXA_INTEG_FULL));
}

// element matrices from the point records (and eDS) of the last exa_grad_setup
static int assemble_ea(exa_ctx* ctx, hipStream_t s) {
   if (ctx->emat_valid) return EXA_OK;
   const size_t nblocks = (size_t)((ctx->E + PA_BLK - 1) / PA_BLK), nd = 3 * (size_t)ctx->n;
   if (!ctx->emat) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->emat, nblocks * nd * nd * PA_BLK * sizeof(double)));
   const int rc = ctx->ea_generic ? exa_launch_assemble_ea_gen(ctx, s) : exa_launch_assemble_ea_p1(ctx, s);
   ctx->emat_valid = (rc == EXA_OK);
   return rc;
}

int exa_set_tangent_form(exa_ctx* ctx, int form) {
   if (!ctx || (form != EXA_TANGENT_FULL && form != EXA_TANGENT_DEV5_BULK && form != EXA_TANGENT_DEV5_BULK_GEO)) return fail(ctx, EXA_ERR_ARG, "exa_set_tangent_form: bad argument");
   if (form == EXA_TANGENT_DEV5_BULK_GEO && !(ctx->p == 1 && ctx->cfg.integ == EXA_INTEG_FULL && ctx->cfg.assembly == EXA_ASSEMBLY_PA))
      return fail(ctx, EXA_ERR_UNSUPPORTED, "exa_set_tangent_form: the compact record with geometry serves the E-vector action of p = 1 partial assembly");
   if (form != ctx->tangent_form && ctx->pa_c) { (void)hipFree(ctx->pa_c); ctx->pa_c = nullptr; ctx->pac_pairs = 0; ctx->have_grad = false; }      // (records of another shape)
   ctx->tangent_form = form;
   return EXA_OK;
}

int exa_grad_tangent_defect(exa_ctx* ctx, const double* C, double* defect_host, exa_stream s) {
   if (!ctx || !C || !defect_host) return fail(ctx, EXA_ERR_ARG, "exa_grad_tangent_defect: null pointer");
   unsigned long long* slot = reinterpret_cast<unsigned long long*>(ctx->scratch_dev);
   int rc = exa_launch_tangent_defect(ctx, C, slot, S(s));
   if (rc) return rc;
   unsigned long long bits = 0;
   EXA_HIP_CHECK(ctx, hipMemcpyAsync(&bits, slot, sizeof(bits), hipMemcpyDeviceToHost, S(s)));
   EXA_HIP_CHECK(ctx, hipStreamSynchronize(S(s)));
   std::memcpy(defect_host, &bits, sizeof(double));
   return EXA_OK;
}

int exa_set_deterministic(exa_ctx* ctx, int on) {
   if (!ctx) return EXA_ERR_ARG;
   ctx->det = on != 0; return EXA_OK;
}

int exa_set_ea_matrix_free(exa_ctx* ctx, int on) {
   if (!ctx) return EXA_ERR_ARG;
   ctx->ea_matfree = on != 0; return EXA_OK;
}

int exa_grad_setup(exa_ctx* ctx, double dt, const double* J, const double* C, exa_stream s) {
   if (!ctx || !J || !C) return fail(ctx, EXA_ERR_ARG, "exa_grad_setup: null pointer");
   if (!ctx->pa) { EXA_HIP_CHECK(ctx, hipMalloc(&ctx->pa, pa_bytes(ctx->E, ctx->Q))); EXA_HIP_CHECK(ctx, hipMemsetAsync(ctx->pa, 0, pa_bytes(ctx->E, ctx->Q), S(s))); }
   if (ctx->tangent_form == EXA_TANGENT_DEV5_BULK_GEO && !ctx->pa_c) {
      ctx->pac_pairs = PAC_PAIRS_GEO;
      EXA_HIP_CHECK(ctx, hipMalloc(&ctx->pa_c, (size_t)((ctx->E + PA_BLK - 1) / PA_BLK) * ctx->Q * 2 * ctx->pac_pairs * PA_BLK * sizeof(double)));
   }
   if (ctx->tangent_form == EXA_TANGENT_DEV5_BULK && !ctx->pa_c) {   // compact records for the actions that can stream them
      const bool p1 = ctx->p == 1 && ctx->cfg.integ == EXA_INTEG_FULL && (ctx->cfg.assembly == EXA_ASSEMBLY_PA || ctx->ea_matfree);   // k_grad_apply_p1<.., GEO, CMP>
      const bool p2 = ctx->n == 27 && (ctx->cfg.assembly == EXA_ASSEMBLY_PA || ctx->ea_matfree);               // k_mf_apply_p2<.., CMP>
      if (p1 || p2) {
         ctx->pac_pairs = p1 ? PAC_PAIRS : PAC_PAIRS_GEO;
         EXA_HIP_CHECK(ctx, hipMalloc(&ctx->pa_c, (size_t)((ctx->E + PA_BLK - 1) / PA_BLK) * ctx->Q * 2 * ctx->pac_pairs * PA_BLK * sizeof(double)));
      }
   }
   int rc = exa_launch_grad_setup_pa(ctx, dt, J, C, S(s));
   if (rc) return rc;
   ctx->grad_records_only = false;
   if (ctx->cfg.assembly == EXA_ASSEMBLY_EA) {
      const bool bbar = ctx->cfg.integ == EXA_INTEG_BBAR;
      ctx->ea_generic = bbar || ctx->p != 1;
      if (bbar) {
         if (!ctx->eDS) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->eDS, sizeof(double) * 3 * ctx->n * PA_BLK * (size_t)((ctx->E + PA_BLK - 1) / PA_BLK)));
         rc = exa_launch_eds(ctx, J, S(s));
         if (rc) return rc;
      }
      ctx->emat_valid = false;
      if (!ea_from_records(ctx)) rc = assemble_ea(ctx, S(s));   // matrix-free: assembled on demand only
   } else if (ctx->p != 1) {
      if (!ctx->tbuf) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->tbuf, sizeof(double) * 9 * ctx->P));
   }
   ctx->have_grad = (rc == EXA_OK);
   return rc;
}

int exa_grad_apply(exa_ctx* ctx, const double* x, double* y, exa_stream s) {
   if (!ctx || !x || !y) return fail(ctx, EXA_ERR_ARG, "exa_grad_apply: null pointer");
   if (!ctx->have_grad) return fail(ctx, EXA_ERR_STATE, "exa_grad_apply called before exa_grad_setup");
   if (ctx->grad_records_only) return fail(ctx, EXA_ERR_STATE, "exa_grad_apply: the gradient data are the compact records of exa_model_setup_lvec_records; call exa_grad_setup for the full records");
   if (ctx->cfg.assembly == EXA_ASSEMBLY_EA) {
      if (int rc = assemble_ea(ctx, S(s))) return rc;
      return ctx->ea_generic ? exa_launch_ea_apply_gen(ctx, x, y, false, nullptr, nullptr, S(s)) : exa_launch_ea_apply_p1(ctx, x, y, false, nullptr, nullptr, S(s));
   }
   if (ctx->p != 1) return exa_launch_pa_apply_gen(ctx, x, y, S(s));
   return exa_launch_grad_apply_p1(ctx, x, y, false, nullptr, nullptr, S(s));
}

int exa_grad_diagonal(exa_ctx* ctx, double* d, exa_stream s) {
   if (!ctx || !d) return fail(ctx, EXA_ERR_ARG, "exa_grad_diagonal: null pointer");
   if (!ctx->have_grad) return fail(ctx, EXA_ERR_STATE, "exa_grad_diagonal called before exa_grad_setup");
   if (ctx->grad_records_only) return fail(ctx, EXA_ERR_STATE, "exa_grad_diagonal: the gradient data are the compact records of exa_model_setup_lvec_records; call exa_grad_setup for the full records");
   if (ctx->cfg.assembly == EXA_ASSEMBLY_EA) {
      if (int rc = assemble_ea(ctx, S(s))) return rc;
      return ctx->ea_generic ? exa_launch_ea_diag_gen(ctx, d, S(s)) : exa_launch_ea_diag_p1(ctx, d, S(s));
   }
   if (ctx->p != 1) return exa_launch_pa_diag_gen(ctx, d, S(s));
   return exa_launch_grad_diag_p1(ctx, d, S(s));
}

int exa_grad_get_ea(exa_ctx* ctx, double* emat, exa_stream s) {
   if (!ctx || !emat) return fail(ctx, EXA_ERR_ARG, "exa_grad_get_ea: null pointer");
   if (!ctx->have_grad || ctx->cfg.assembly != EXA_ASSEMBLY_EA) return fail(ctx, EXA_ERR_STATE, "exa_grad_get_ea: no element matrices assembled");
   if (ctx->grad_records_only) return fail(ctx, EXA_ERR_STATE, "exa_grad_get_ea: the gradient data are the compact records of exa_model_setup_lvec_records; call exa_grad_setup first");
   if (int rc = assemble_ea(ctx, S(s))) return rc;
   return ctx->ea_generic ? exa_launch_ea_export_gen(ctx, emat, S(s)) : exa_launch_ea_export_p1(ctx, emat, S(s));
}

int exa_set_connectivity(exa_ctx* ctx, const int32_t* conn, int nnodes) {
   if (!ctx || !conn || nnodes <= 0) return fail(ctx, EXA_ERR_ARG, "exa_set_connectivity: bad argument");
   ctx->conn = conn; ctx->nnodes = nnodes;
   return EXA_OK;
}

int exa_restrict(exa_ctx* ctx, const double* L, double* Ev, exa_stream s) {
   if (!ctx || !L || !Ev) return fail(ctx, EXA_ERR_ARG, "exa_restrict: null pointer");
   if (!ctx->conn) return fail(ctx, EXA_ERR_STATE, "exa_restrict: connectivity not set");
   return exa_launch_restrict(ctx, L, Ev, S(s));
}

int exa_restrict_transpose_add(exa_ctx* ctx, const double* Ev, double* L, exa_stream s) {
   if (!ctx || !L || !Ev) return fail(ctx, EXA_ERR_ARG, "exa_restrict_transpose_add: null pointer");
   if (!ctx->conn) return fail(ctx, EXA_ERR_STATE, "exa_restrict_transpose_add: connectivity not set");
   return exa_launch_restrict_T(ctx, Ev, L, S(s));
}

}  // extern "C"
// node -> (element, local node) table of the current connectivity, entries of a node in ascending element order (built once per
// connectivity on the host: one device->host copy of the element table)
int exa_det_prepare(exa_ctx* ctx) {
   if (ctx->n2e_off && ctx->det_conn == ctx->conn && ctx->det_nnodes == ctx->nnodes) return EXA_OK;
   if (!ctx->conn) return fail(ctx, EXA_ERR_STATE, "deterministic E->L: connectivity not set");
   const size_t ne = (size_t)ctx->n * ctx->E;
   if (ne >= (size_t)INT32_MAX) return fail(ctx, EXA_ERR_UNSUPPORTED, "deterministic E->L: element table too large for 32-bit entries");
   std::vector<int32_t> conn(ne);
   EXA_HIP_CHECK(ctx, hipMemcpy(conn.data(), ctx->conn, sizeof(int32_t) * ne, hipMemcpyDeviceToHost));
   std::vector<int32_t> off((size_t)ctx->nnodes + 1, 0), idx(ne);
   for (size_t k = 0; k < ne; k++) { if (conn[k] < 0 || conn[k] >= ctx->nnodes) return fail(ctx, EXA_ERR_ARG, "deterministic E->L: node index out of range"); off[(size_t)conn[k] + 1]++; }
   for (int i = 0; i < ctx->nnodes; i++) off[(size_t)i + 1] += off[i];
   { std::vector<int32_t> cur(off.begin(), off.end() - 1); for (size_t k = 0; k < ne; k++) idx[(size_t)cur[conn[k]]++] = (int32_t)k; }   // k = a + n e ascending -> element order
   (void)hipFree(ctx->n2e_off); (void)hipFree(ctx->n2e_idx); ctx->n2e_off = nullptr; ctx->n2e_idx = nullptr;
   EXA_HIP_CHECK(ctx, hipMalloc(&ctx->n2e_off, sizeof(int32_t) * off.size())); EXA_HIP_CHECK(ctx, hipMalloc(&ctx->n2e_idx, sizeof(int32_t) * ne));
   EXA_HIP_CHECK(ctx, hipMemcpy(ctx->n2e_off, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice));
   EXA_HIP_CHECK(ctx, hipMemcpy(ctx->n2e_idx, idx.data(), sizeof(int32_t) * ne, hipMemcpyHostToDevice));
   if (!ctx->ev_det) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->ev_det, sizeof(double) * 24 * PA_BLK * (size_t)((ctx->E + PA_BLK - 1) / PA_BLK)));
   ctx->det_conn = ctx->conn; ctx->det_nnodes = ctx->nnodes;
   return EXA_OK;
}
extern "C" {

// driver-internal variant: `gate` (nullable) is a device flag; a non-zero value turns the launch into a no-op so that a
// PCG loop whose scalars live on the device can be enqueued without host synchronisation.
// driver-internal: the p = 1 record-based action on the 64-element blocks [blk0, blk0 + nblk) only (partial assembly and element assembly
// from records; atomic scatter).  Returns EXA_ERR_UNSUPPORTED for contexts that have no such kernel: the caller then runs the whole action.
int exa_grad_apply_lvec_blocks(exa_ctx* ctx, const double* x, double* y, const uint8_t* mask, const double* gate, int blk0, int nblk, exa_stream s) {
   if (!ctx || !x || !y) return fail(ctx, EXA_ERR_ARG, "exa_grad_apply_lvec_blocks: null pointer");
   if (!ctx->conn || !ctx->have_grad) return fail(ctx, EXA_ERR_STATE, "exa_grad_apply_lvec_blocks: connectivity / gradient data not set");
   if (ctx->grad_records_only && !ctx->coords_lvec) return fail(ctx, EXA_ERR_STATE, "exa_grad_apply_lvec_blocks: name the nodal coordinates with exa_grad_set_coords");
   if (ctx->det || ctx->p != 1 || ctx->cfg.integ != EXA_INTEG_FULL) return EXA_ERR_UNSUPPORTED;
   if (ctx->cfg.assembly == EXA_ASSEMBLY_EA) {
      if (!ea_from_records(ctx)) return EXA_ERR_UNSUPPORTED;
      return exa_launch_grad_apply_p1(ctx, x, y, true, mask, gate, S(s), true, blk0, nblk);
   }
   return exa_launch_grad_apply_p1(ctx, x, y, true, mask, gate, S(s), false, blk0, nblk);
}

int exa_grad_apply_lvec_gated(exa_ctx* ctx, const double* x, double* y, const uint8_t* mask, const double* gate, exa_stream s) {
   if (!ctx || !x || !y) return fail(ctx, EXA_ERR_ARG, "exa_grad_apply_lvec: null pointer");
   if (!ctx->conn) return fail(ctx, EXA_ERR_STATE, "exa_grad_apply_lvec: connectivity not set");
   if (!ctx->have_grad) return fail(ctx, EXA_ERR_STATE, "exa_grad_apply_lvec called before exa_grad_setup");
   if (ctx->grad_records_only && !ctx->coords_lvec) return fail(ctx, EXA_ERR_STATE, "exa_grad_apply_lvec: name the nodal coordinates with exa_grad_set_coords (the compact records hold no geometry)");
   if (ctx->det && (ctx->p != 1 || ctx->cfg.integ != EXA_INTEG_FULL))
      return fail(ctx, EXA_ERR_UNSUPPORTED, "deterministic mode: the fused L-vector action is ordered for p = 1 full integration only; use exa_restrict + exa_grad_apply + exa_restrict_transpose_add");
   if (ctx->cfg.assembly == EXA_ASSEMBLY_EA) {
      if (ctx->ea_matfree && ctx->n == 27) return exa_launch_mf_apply_p2(ctx, x, y, mask, gate, true, S(s));
      if (ea_from_records(ctx)) return exa_launch_grad_apply_p1(ctx, x, y, true, mask, gate, S(s), true);
      if (ctx->det) return fail(ctx, EXA_ERR_UNSUPPORTED, "deterministic mode: the L-vector action of assembled element matrices scatters with atomics; use exa_set_ea_matrix_free or exa_grad_apply + exa_restrict_transpose_add");
      if (int rc = assemble_ea(ctx, S(s))) return rc;
      return ctx->ea_generic ? exa_launch_ea_apply_gen(ctx, x, y, true, mask, gate, S(s)) : exa_launch_ea_apply_p1(ctx, x, y, true, mask, gate, S(s));
   }
   if (ctx->n == 27) return exa_launch_mf_apply_p2(ctx, x, y, mask, gate, false, S(s));
   if (ctx->p != 1) return fail(ctx, EXA_ERR_UNSUPPORTED, "fused L-vector partial-assembly action is built for p = 1 and p = 2; use exa_restrict + exa_grad_apply");
   return exa_launch_grad_apply_p1(ctx, x, y, true, mask, gate, S(s));
}

int exa_grad_set_coords(exa_ctx* ctx, const double* coords_lvec) {
   if (!ctx) return EXA_ERR_ARG;
   ctx->coords_lvec = coords_lvec; return EXA_OK;
}

int exa_grad_apply_lvec(exa_ctx* ctx, const double* x, double* y, const uint8_t* mask, exa_stream s) {
   return exa_grad_apply_lvec_gated(ctx, x, y, mask, nullptr, s);
}

int exa_residual_lvec(exa_ctx* ctx, const double* J, const double* stress1, double* y, exa_stream s) {
   if (!ctx || !stress1 || !y) return fail(ctx, EXA_ERR_ARG, "exa_residual_lvec: null pointer");
   if (!ctx->conn) return fail(ctx, EXA_ERR_STATE, "exa_residual_lvec: connectivity not set");
   if (!J && !(ctx->p == 1 && ctx->cfg.integ == EXA_INTEG_FULL && ctx->coords_lvec))
      return fail(ctx, EXA_ERR_ARG, "exa_residual_lvec: a null Jacobian field needs p = 1 full integration and nodal coordinates (exa_grad_set_coords)");
   if (ctx->det && ctx->p == 2) return fail(ctx, EXA_ERR_UNSUPPORTED, "deterministic mode: the fused p = 2 residual scatters with atomics; use exa_residual_setup/apply + exa_restrict_transpose_add");
   if (ctx->p == 2) {
      if (ctx->cfg.integ == EXA_INTEG_BBAR && !ctx->eDS) EXA_HIP_CHECK(ctx, hipMalloc(&ctx->eDS, sizeof(double) * 3 * ctx->n * PA_BLK * (size_t)((ctx->E + PA_BLK - 1) / PA_BLK)));
      return exa_launch_residual_p2(ctx, J, stress1, y, S(s));
   }
   if (ctx->p != 1 || ctx->cfg.integ != EXA_INTEG_FULL) return fail(ctx, EXA_ERR_UNSUPPORTED, "exa_residual_lvec: fused path is built for p = 1 full integration and p = 2; use exa_residual_setup/apply + exa_restrict_transpose_add");
   return exa_launch_residual_p1(ctx, J, stress1, y, true, S(s));
}

int exa_vol_avg(exa_ctx* ctx, const double* J, const double* qf, int vdim, int normalise, double* out_host, exa_stream s) {
   if (!ctx || !J || !qf || !out_host || vdim < 1 || vdim > 63) return fail(ctx, EXA_ERR_ARG, "exa_vol_avg: bad argument");
   const int nb = VOL_AVG_BLOCKS;
   int rc = exa_launch_vol_avg(ctx, J, qf, vdim, ctx->scratch_dev, nb, S(s));
   if (rc) return rc;
   std::vector<double> part((size_t)(vdim + 1) * nb);
   EXA_HIP_CHECK(ctx, hipMemcpyAsync(part.data(), ctx->scratch_dev, sizeof(double) * part.size(), hipMemcpyDeviceToHost, S(s)));
   EXA_HIP_CHECK(ctx, hipStreamSynchronize(S(s)));
   for (int c = 0; c <= vdim; c++) { double a = 0; for (int b = 0; b < nb; b++) a += part[(size_t)c * nb + b]; out_host[c] = a; }
   if (normalise) for (int c = 0; c < vdim; c++) out_host[c] /= out_host[vdim];
   return EXA_OK;
}

}  // extern "C"
