// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points so tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can drive the CPU restatement through ctypes.  Never linked into the product library.
#include "driver_port.hpp"
#include <cstdint>

extern "C" {

struct orc_case {
   int nx, ny, nz, p; double sx, sy, sz;
   int xtal, kin, nprops; const double* props; double temp_k;
   int ngrains; const int* elem_grain; const double* quats;
   int nsteps; const double* dts;
   int nbc; const int* bc_step; const int* bc_nids; const int* bc_ids; const int* bc_comps; const double* bc_vals;   // flattened
   const double* bc_vgrad;   // 9 per BC set (row-major essential_vel_grad), may be null
   int assembly, nl_solver, precond, integ;
   double newton_rel, newton_abs; int newton_iter;
   double krylov_rel, krylov_abs; int krylov_iter;
   int additional_avgs, second_order_terms, use_input_temperature, verbose;
   int dt_auto; double dt_start, dt_min, dt_scale, t_final;   // Time.Auto; nsteps is then the row capacity of the result arrays
};

struct orc_result {
   double* avg_stress; double* avg_def_grad; double* avg_pl_work; double* avg_dp_tensor;   // caller-allocated, nsteps rows
   int* newton_iters; int* krylov_iters; int* model_calls;
   int64_t qpt_updates; double t_model, t_krylov, t_total; int failed;
   int steps_done; double* dts_used;   // auto time stepping: rows actually produced, dt per row (may be null)
};

static void fill_config(const orc_case* c, drv::Config& cfg) {
   cfg.nx = c->nx; cfg.ny = c->ny; cfg.nz = c->nz; cfg.p = c->p; cfg.sx = c->sx; cfg.sy = c->sy; cfg.sz = c->sz;
   cfg.xtal = c->xtal; cfg.kin = c->kin; cfg.props.assign(c->props, c->props + c->nprops); cfg.temp_k = c->temp_k;
   const int E = c->nx * c->ny * c->nz;
   cfg.elem_grain.assign(c->elem_grain, c->elem_grain + E);
   cfg.quats.assign(c->quats, c->quats + 4 * c->ngrains);
   if (c->dt_auto) { cfg.dt_auto = true; cfg.dt_start = c->dt_start; cfg.dt_min = c->dt_min; cfg.dt_scale = c->dt_scale; cfg.t_final = c->t_final; cfg.max_steps = c->nsteps; }
   else cfg.dts.assign(c->dts, c->dts + c->nsteps);
   int off = 0;
   for (int b = 0; b < c->nbc; b++) {
      drv::BCSet bc; bc.step = c->bc_step[b];
      for (int i = 0; i < c->bc_nids[b]; i++) {
         bc.ids.push_back(c->bc_ids[off + i]); bc.comps.push_back(c->bc_comps[off + i]);
         for (int k = 0; k < 3; k++) bc.vals.push_back(c->bc_vals[3 * (off + i) + k]);
      }
      off += c->bc_nids[b];
      if (c->bc_vgrad) for (int k = 0; k < 9; k++) bc.vgrad[k] = c->bc_vgrad[9 * b + k];
      cfg.bcs.push_back(bc);
   }
   cfg.assembly = c->assembly; cfg.nl_solver = c->nl_solver; cfg.precond = c->precond; cfg.integ = c->integ;
   cfg.newton_rel = c->newton_rel; cfg.newton_abs = c->newton_abs; cfg.newton_iter = c->newton_iter;
   cfg.krylov_rel = c->krylov_rel; cfg.krylov_abs = c->krylov_abs; cfg.krylov_iter = c->krylov_iter;
   cfg.additional_avgs = c->additional_avgs != 0; cfg.second_order_terms = c->second_order_terms != 0;
   cfg.use_input_temperature = c->use_input_temperature != 0; cfg.verbose = c->verbose;
}

int orc_run_case(const orc_case* c, orc_result* r) {
   drv::Config cfg; fill_config(c, cfg);
   drv::Result res;
   drv::run_case(cfg, res);
   const int ns = (int)(res.avg_stress.size() / 6);   // completed steps (== c->nsteps unless auto time stepping stopped early)
   r->steps_done = ns;
   if (r->dts_used) for (size_t i = 0; i < res.dts_used.size() && (int)i < c->nsteps; i++) r->dts_used[i] = res.dts_used[i];
   for (int i = 0; i < 6 * ns; i++) r->avg_stress[i] = res.avg_stress[i];
   if (cfg.additional_avgs) {
      for (int i = 0; i < 9 * ns; i++) r->avg_def_grad[i] = res.avg_def_grad[i];
      for (int i = 0; i < ns; i++) r->avg_pl_work[i] = res.avg_pl_work[i];
      for (int i = 0; i < 6 * ns; i++) r->avg_dp_tensor[i] = res.avg_dp_tensor[i];
   }
   for (int i = 0; i < ns && i < (int)res.newton_iters.size(); i++) { r->newton_iters[i] = res.newton_iters[i]; r->krylov_iters[i] = res.krylov_iters[i]; r->model_calls[i] = res.model_calls[i]; }
   r->qpt_updates = res.qpt_updates; r->t_model = res.t_model; r->t_krylov = res.t_krylov; r->t_total = res.t_total; r->failed = res.failed;
   return res.failed;
}

// number of OpenMP threads of the constitutive loop (1 = the reference's serial CPU path); returns the value in effect
int orc_set_threads(int n) { fem::model_threads() = n < 1 ? 1 : n; return fem::model_threads(); }

// ---- reference element / mesh helpers ------------------------------------------------------
int orc_ref_elem(int p, double* G /*(n,3,Q)*/, double* W /*(Q)*/) {
   fem::RefElem re; fem::ref_elem_init(re, p);
   if (G) std::memcpy(G, re.G.data(), sizeof(double) * re.G.size());
   if (W) std::memcpy(W, re.W.data(), sizeof(double) * re.W.size());
   return re.n;
}

void orc_mesh(int p, int nx, int ny, int nz, double sx, double sy, double sz, int* conn /*(n,E)*/, double* X /*(NN,3)*/) {
   fem::RefElem re; fem::ref_elem_init(re, p);
   fem::Mesh m; fem::mesh_init(m, re, nx, ny, nz, sx, sy, sz);
   if (conn) std::memcpy(conn, m.conn.data(), sizeof(int) * m.conn.size());
   if (X) std::memcpy(X, m.X.data(), sizeof(double) * m.X.size());
}

void orc_jacobians(int p, int E, const double* xe, double* J) { fem::RefElem re; fem::ref_elem_init(re, p); fem::jacobians(re, E, xe, J); }
void orc_grad_calc(int Q, int E, int n, const double* J, const double* G, const double* field, double* out) { fem::grad_calc(Q, E, n, J, G, field, out); }

// ---- model ------------------------------------------------------------------------------------
int orc_model_setup(int xtal, int kin, const double* props, int nprops, int Q, int E, int n, int nstatev, double dt, double temp_k,
                    const double* J, const double* G, const double* vel_e, const double* stress0, const double* state0,
                    double* stress1, double* state1, double* ddsdde, double* vgrad_out, int transpose_tangent,
                    int second_order_terms, int use_input_temperature) {
   ecm::Model mdl; if (!ecm::model_init(mdl, xtal, kin, props, nprops)) return -1;
   fem::ModelOpts mo; mo.transpose_tangent = transpose_tangent != 0; mo.po.second_order_terms = second_order_terms != 0;
   mo.po.use_input_temperature = use_input_temperature != 0;
   return fem::model_setup(mdl, Q, E, n, nstatev, dt, temp_k, J, G, vel_e, stress0, state0, stress1, state1, ddsdde, vgrad_out, mo);
}

void orc_hist_init(int xtal, int kin, const double* props, int nprops, double* hist26) {
   ecm::Model mdl; ecm::model_init(mdl, xtal, kin, props, nprops); ecm::hist_init(mdl, hist26);
}

void orc_slip_geom(int xtal, double* P /*(5,12) row-major*/, double* Qv /*(3,12)*/) {
   ecm::Model mdl; std::memset(&mdl, 0, sizeof(mdl)); mdl.xtal = xtal; ecm::slip_geom_init(mdl);
   std::memcpy(P, mdl.P, sizeof(mdl.P)); std::memcpy(Qv, mdl.Q, sizeof(mdl.Q));
}

int orc_point_response(int xtal, int kin, const double* props, int nprops, double dt, const double* d_svec_p, const double* w_vec,
                       const double* vol_ratio, double* e_int, double* stress_svec_p, double* hist, double* tkelv, double* sdd, double* mtan,
                       int second_order_terms, int use_input_temperature) {
   ecm::Model mdl; if (!ecm::model_init(mdl, xtal, kin, props, nprops)) return -1;
   ecm::PointOpts po; po.second_order_terms = second_order_terms != 0; po.use_input_temperature = use_input_temperature != 0;
   return ecm::get_response_sngl(mdl, dt, d_svec_p, w_vec, vol_ratio, e_int, stress_svec_p, hist, *tkelv, sdd, mtan, po);
}

// ---- integrators --------------------------------------------------------------------------------
void orc_assemble_pa(int Q, int E, const double* W, const double* J, const double* stress1, double* dmat) { fem::assemble_pa(Q, E, W, J, stress1, dmat); }
void orc_add_mult_pa(int Q, int E, int n, const double* G, const double* dmat, double* Y) { fem::add_mult_pa(Q, E, n, G, dmat, Y); }
void orc_transform_4d(int64_t P, const double* C, double* C4) { fem::transform_matgrad_4d((size_t)P, C, C4); }
void orc_assemble_grad_pa(int Q, int E, double dt, const double* W, const double* J, const double* C4, double* D4) { fem::assemble_grad_pa(Q, E, dt, W, J, C4, D4); }
void orc_add_mult_grad_pa(int Q, int E, int n, const double* G, const double* D4, const double* X, double* Y) { fem::add_mult_grad_pa(Q, E, n, G, D4, X, Y); }
void orc_assemble_grad_diag_pa(int Q, int E, int n, double dt, const double* W, const double* G, const double* J, const double* K, double* Y) { fem::assemble_grad_diag_pa(Q, E, n, dt, W, G, J, K, Y); }
void orc_assemble_ea(int Q, int E, int n, double dt, const double* W, const double* G, const double* J, const double* K, double* emat) { fem::assemble_ea(Q, E, n, dt, W, G, J, K, emat); }
void orc_ea_mult(int E, int n, const double* emat, const double* X, double* Y) { fem::ea_mult(E, n, emat, X, Y); }
void orc_ea_diag(int E, int n, const double* emat, double* Y) { fem::ea_diag(E, n, emat, Y); }
void orc_element_vector(int Q, int E, int n, const double* W, const double* G, const double* J, const double* stress1, double* Y) { fem::element_vector(Q, E, n, W, G, J, stress1, Y); }
void orc_element_eds(int Q, int E, int n, const double* W, const double* G, const double* J, double* eDS) { fem::element_eds(Q, E, n, W, G, J, eDS); }
void orc_add_mult_pa_bbar(int Q, int E, int n, const double* W, const double* G, const double* J, const double* eDS, const double* S, double* Y) { fem::add_mult_pa_bbar(Q, E, n, W, G, J, eDS, S, Y); }
void orc_assemble_ea_bbar(int Q, int E, int n, double dt, const double* W, const double* G, const double* J, const double* eDS, const double* K, double* emat) { fem::assemble_ea_bbar(Q, E, n, dt, W, G, J, eDS, K, emat); }
void orc_element_vector_bbar(int Q, int E, int n, const double* W, const double* G, const double* J, const double* eDS, const double* s, double* Y) { fem::element_vector_bbar(Q, E, n, W, G, J, eDS, s, Y); }
void orc_vol_avg(int Q, int E, int vdim, const double* W, const double* J, const double* qf, double* out, int normalise) { fem::vol_avg(Q, E, vdim, W, J, qf, out, normalise != 0); }
void orc_calc_dp_mat(int xtal, int64_t P, int nstatev, const double* state1, double* dp) {
   ecm::Model mdl; std::memset(&mdl, 0, sizeof(mdl)); mdl.xtal = xtal; ecm::slip_geom_init(mdl);
   fem::calc_dp_mat(mdl, (size_t)P, nstatev, state1, dp);
}


// Time.Auto replay guided by a golden sigma_33 column (driver_port.hpp: run_case_replay).  ks_out: chosen Newton count per row.
int orc_run_case_replay(const orc_case* c, const double* target33, int nrows, orc_result* r, int* ks_out, int increments) {
   drv::Config cfg; fill_config(c, cfg);
   drv::Result res; std::vector<int> ks;
   drv::run_case_replay(cfg, std::vector<double>(target33, target33 + nrows), res, ks, increments != 0);
   const int ns = (int)(res.avg_stress.size() / 6);
   r->steps_done = ns;
   for (int i = 0; i < 6 * ns; i++) r->avg_stress[i] = res.avg_stress[i];
   for (int i = 0; i < ns; i++) { if (r->dts_used) r->dts_used[i] = res.dts_used[i]; ks_out[i] = ks[i]; }
   r->failed = res.failed;
   return res.failed;
}

}  // extern "C"
