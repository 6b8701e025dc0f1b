/* exaconstit_hip.h — C ABI of libexaconstit_hip.so (MI355X / gfx950).
 *
 * The reference (LLNL/ExaConstit v0.7.0) has no C ABI: its replaceable seam for the hot path is two C++ class
 * interfaces.  Each entry point below cites the reference interface it stands in for; INTEGRATION.md shows the
 * MFEM-side adapters (HipExaModel : ExaModel, HipExaNLFIntegrator : ExaNLFIntegrator) that bind them.
 *
 * Conventions
 *   - every pointer marked "dev" is a device pointer owned by the caller; "host" pointers are host memory;
 *   - every array is FP64 and uses the reference's layouts (column-major = first index fastest):
 *       E-vector        (node, comp, elem)            reference src/mechanics_kernels.cpp:28-29
 *       Jacobians       (3, 3, Q, E), J(i,j)=dx_i/dxi_j   reference src/mechanics_operator.cpp:379-390
 *       shape table     (node, dir, qpt)              reference src/mechanics_operator.cpp:249-260
 *       quadrature data (vdim, Q, E)                  reference src/mechanics_model.cpp:209-214
 *       tangent         (6, 6, Q, E) column-major, Voigt (11,22,33,23,13,12), engineering shear
 *   - every function returns int: 0 ok, <0 invalid argument / HIP error (see exa_last_error), and the work is
 *     enqueued on the given stream without host synchronisation unless stated otherwise;
 *   - no global state; one exa_ctx per (mesh partition, material); re-entrant per context.
 */
#ifndef EXACONSTIT_HIP_H
#define EXACONSTIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct exa_ctx exa_ctx;
typedef void* exa_stream;   /* hipStream_t */

/* model ids: reference src/mechanics_ecmech.hpp:407-414 (Voce), :460-463 (KM-DD); chosen in
 * reference src/mechanics_operator.cpp:49-210 from Model.ExaCMech.{xtal_type,slip_type} */
enum { EXA_FCC_VOCE = 0, EXA_FCC_VOCE_NL = 1, EXA_BCC_VOCE = 2, EXA_BCC_VOCE_NL = 3, EXA_FCC_KMDD = 4, EXA_BCC_KMDD = 5 };
/* reference src/option_types.hpp Assembly / IntegrationType */
enum { EXA_ASSEMBLY_PA = 0, EXA_ASSEMBLY_EA = 1 };
enum { EXA_INTEG_FULL = 0, EXA_INTEG_BBAR = 1 };

enum { EXA_OK = 0, EXA_ERR_ARG = -1, EXA_ERR_HIP = -2, EXA_ERR_STATE = -3, EXA_ERR_UNSUPPORTED = -4 };
/* Which entry points are legal in which context (everything else returns EXA_ERR_UNSUPPORTED with a message in exa_last_error):
 *
 *   route                         layout   entry points                                                               orders / integrators
 *   ----------------------------  -------  -------------------------------------------------------------------------  ----------------------------------
 *   A  E-vector (MFEM adapters)   AOS      exa_jacobians[_from_geom], exa_model_setup, exa_residual_setup/apply,      p = 1..6, full and B-bar (B-bar:
 *                                          exa_grad_setup, exa_grad_apply, exa_grad_diagonal, exa_grad_get_ea,        element assembly only, like the
 *                                          exa_restrict / exa_restrict_transpose_add, exa_calc_dp, exa_vol_avg        reference)
 *   B  fused L-vector             AOS or   A's set-up calls + exa_set_connectivity, exa_model_setup_lvec,             p = 1 full integration; p = 2 full
 *                                 EB64     exa_residual_lvec, exa_grad_apply_lvec (PA, EA from matrices or - with     and B-bar (EB64: L-vector entries
 *                                          exa_set_ea_matrix_free - from the point records), exa_grad_set_coords,     only, exa_residual_setup/apply stay
 *                                          exa_set_tangent_form                                                       AOS); p >= 3: exa_grad_apply_lvec
 *                                                                                                                     for element assembly only
 *   C  record route (driver)      EB64     exa_model_setup_lvec_records + exa_grad_set_coords + exa_grad_apply_lvec   p = 1 full integration, PA or
 *                                          + exa_residual_lvec (Jacobian field optional)                              matrix-free EA, compact tangent form
 *
 *   exa_set_deterministic: ordered sums for route B / C at p = 1 full integration; elsewhere the fused entries refuse and route A + exa_restrict_transpose_add is
 *   the reproducible path.  An adapter that does not care about any of this uses route A and never sees EXA_ERR_UNSUPPORTED. */

typedef struct {
   int model;            /* EXA_* model id */
   int nprops;           /* 17 (Voce), 18 (Voce NL), 24 (KM-DD) — reference src/option_parser.cpp:396-478 */
   const double* props;  /* host; parameter order of reference src/mechanics_ecmech.hpp:395-405,444-458 */
   double temp_k;        /* Properties.temperature (the library overrides it with its EOS temperature, as ExaCMech does) */
   int order;            /* H1 order p: n = Q = (p+1)^3 */
   int nelems;           /* elements of this partition */
   int assembly;         /* EXA_ASSEMBLY_* */
   int integ;            /* EXA_INTEG_* */
   int device;           /* HIP device ordinal; -1 = current */
} exa_config;

/* lifetime -------------------------------------------------------------------------------------------------- */
exa_ctx* exa_create(const exa_config* cfg, int* err);          /* ECMechXtalModel ctor, src/mechanics_ecmech.hpp:126-262 */
void     exa_destroy(exa_ctx* ctx);
const char* exa_last_error(const exa_ctx* ctx);                /* MFEM_ABORT text equivalent */
const char* exa_build_id(void);                                /* 12 hex digits: hash of the library's sources at build time */
const char* exa_kernel_build_id(void);                         /* ... of the device code and its compile flags alone: stamped into the counter files under profiles/ */
int exa_num_state_vars(const exa_ctx* ctx);                    /* numHist + ne + 1 = 28, src/mechanics_ecmech.hpp:136-141 */
int exa_nodes_per_elem(const exa_ctx* ctx);
int exa_qpts_per_elem(const exa_ctx* ctx);

/* reference-element tables built on the host once: src/mechanics_operator.cpp:237-261, src/mechanics_integrators.cpp:184-197 */
int exa_shape_table(const exa_ctx* ctx, double* G_host /*(n,3,Q)*/, double* W_host /*(Q)*/);

/* Layout of every quadrature function passed to this context (jacobian, stress, state, ddsdde, dp, any exa_vol_avg field):
 *   EXA_QLAYOUT_AOS  (default) the reference's QuadratureFunction layout (vdim, Q, E), first index fastest;
 *   EXA_QLAYOUT_EB64 [block of 64 elements][q][component][lane = element], exa_qf_size(ctx, vdim) doubles per field.  It is the
 *                    internal layout of the stand-alone driver (every per-value access of a wave is one contiguous 512-byte row:
 *                    the constitutive launch is 1.4x faster at 128^3); p = 1 full integration and p = 2 (plain or B-bar) with
 *                    the L-vector entry points only (exa_residual_setup / exa_residual_apply stay AOS).  An MFEM adapter keeps AOS. */
enum { EXA_QLAYOUT_AOS = 0, EXA_QLAYOUT_EB64 = 1 };
int exa_set_quadrature_layout(exa_ctx* ctx, int layout);
/* EXA_QLAYOUT_AOS, constitutive launches (exa_model_setup, exa_model_setup_lvec): the 64 points of a wave are 64 x 28 / 6 / 9 / 36 CONTIGUOUS doubles of
 * the state / stress / Jacobian / tangent arrays, so the wave moves them with coalesced 16-byte accesses and transposes them through the per-lane LDS
 * stash the kernel owns anyway (inputs once, outputs in three rounds: two tangent halves, then state + stress) instead of 8-byte accesses at a
 * 224 / 288-byte lane stride - the reference's layout at the element-blocked layout's speed, same bits out.  On by default; 0 restores the
 * per-lane strided accesses (A/B runs).  exa_get_aos_staging: the setting (the staged kernels exist for every order and model; the dense
 * launches of a tail split keep strided accesses - their lanes are scattered points). */
int exa_set_aos_staging(exa_ctx* ctx, int on);
int exa_get_aos_staging(const exa_ctx* ctx);
int exa_get_quadrature_layout(const exa_ctx* ctx);               /* EXA_QLAYOUT_* currently selected */
int64_t exa_qf_size(const exa_ctx* ctx, int vdim);

/* ExaModel seam ------------------------------------------------------------------------------------------------ */
/* getHistInfo + init_state_vars (src/mechanics_ecmech.hpp:248-300) with setStateVarData's quaternion splice
 * (src/mechanics_driver.cpp:1058-1154): fills state0 (28,Q,E) from one quaternion per element. */
int exa_init_state(exa_ctx* ctx, double* state0_dev, const double* quats_per_elem_dev /*(4,E)*/, exa_stream s);

/* State layout ------------------------------------------------------------------------------------------------- */
/* The 28 state variables of a quadrature point, all six models (index table of src/mechanics_ecmech.hpp:165-185; numHist + ne + 1,
 * :136-141).  Slot = offset inside the (28, Q, E) quadrature function:
 *
 *    0        shrateEff   effective shear rate of the step, sum_a |gdot_a|            (ecmech::evptn::iHistA_shrateEff, "shrateEff")
 *    1        shrEff      accumulated effective shear, slot 1 += slot 0 * dt          (iHistA_shrEff, "shrEff")
 *    2        flowStr     ExaCMech's flow strength; ExaConstit stores the accumulated plastic work there ("pl_work",
 *                         src/mechanics_ecmech.cpp:154-160) - so does this library
 *    3        nFEval      residual evaluations the local Newton solve (SNLS trust-region dog-leg) of this point took in this step
 *                         (iHistA_nFEval); 2 for an elastic point
 *    4 .. 8   deviatoric elastic strain in the lattice frame, 5-vector              (iHistLbE, "elas_strain")
 *    9 .. 12  lattice orientation, unit quaternion, scalar first                    (iHistLbQ, "quats")
 *    13       hardness: slip-system strength g (Voce kinds) or dislocation density rho (Kocks-Mecking kinds)   (iHistLbH, "hardness")
 *    14 .. 25 the 12 slip rates gdot_a of the converged step                          (iHistLbGdot, "gdot")
 *    26       relative volume                                                         ("rel_vol")
 *    27       internal energy per reference volume                                    ("int_eng")
 *
 * Two slots carry more than a value:
 *    slot 0  is an INPUT of the next step: the hardness update takes the begin-of-step effective shear rate from it instead of
 *            re-reading the 12 slip rates (88 B per point less traffic).  Invariant a caller-supplied state0 must satisfy:
 *            slot 0 == sum_a |slot 14+a|.  It holds for every state this library wrote and for exa_init_state; a state that
 *            comes from elsewhere (restart file, another code) is brought into this form by exa_state_normalize.
 *    slot 3  follows the iteration PATH of the local solve, not only its result.  The library reproduces the reference solver's
 *            path: against a CPU restatement of that solver the count is equal at >= 99.9 % of the points and never differs by more than one
 *            (ulp-level ties of the trust-region acceptance tests; asserted by tests/test_gpu_parity.py, test_gpu_point_fixtures.py,
 *            test_gpu_fullsize.py; measured 99.999 %, profiles/r04_nfev_agreement.txt).  The tail split (exa_set_newton_cap) does not
 *            change it. */
/* state[0] = sum_a |state[14+a]| at every point of a state array in the context's layout (see above) */
int exa_state_normalize(exa_ctx* ctx, double* state_dev, exa_stream s);

/* ExaCMechModel::ModelSetup (src/mechanics_ecmech.cpp:192-258), i.e. StressSetup/StateVarsSetup, grad_calc,
 * kernel_setup, getResponseECM, kernel_postprocessing fused into one launch.  On return (stream order) stress1, state1
 * and the column-major tangent d sigma/d eps hold end-of-step values.  Points whose local solve did not converge are
 * counted; read the count with exa_model_status. */
int exa_model_setup(exa_ctx* ctx, double dt, const double* jacobian_dev /*(3,3,Q,E)*/, const double* vel_evec_dev /*(n,3,E)*/,
                    const double* stress0_dev, const double* state0_dev,
                    double* stress1_dev, double* state1_dev, double* ddsdde_dev, exa_stream s);
/* exa_model_setup + exa_model_status in one call, with the return value the model seam is specified with: < 0 error, 0 every local
 * solve converged, > 0 the number of quadrature points whose ExaCMech solve failed (the reference aborts the run on one: ECMECH_FAIL behind
 * getResponseECM, src/mechanics_ecmech.cpp:176-186).  Synchronises the stream (one 4-byte read-back); the MFEM adapter's ModelSetup uses it. */
int exa_model_setup_checked(exa_ctx* ctx, double dt, const double* jacobian_dev, const double* vel_evec_dev,
                            const double* stress0_dev, const double* state0_dev,
                            double* stress1_dev, double* state1_dev, double* ddsdde_dev, exa_stream s);
/* Same update driven from L-vectors (needs exa_set_connectivity): gathers nodal coordinates / velocities itself and also
 * WRITES the Jacobians (3,3,Q,E) the integrator calls need, i.e. NonlinearMechOperator::Setup's two L->E restrictions and
 * SetupJacobianTerms (src/mechanics_operator.cpp:310-391) fused into the constitutive launch. */
int exa_model_setup_lvec(exa_ctx* ctx, double dt, const double* coords_lvec_dev /*(nnodes,3) byNODES*/, const double* vel_lvec_dev,
                         const double* stress0_dev, const double* state0_dev,
                         double* stress1_dev, double* state1_dev, double* ddsdde_dev, double* jacobian_out_dev, exa_stream s);
/* The same launch with AssembleGradPA (src/mechanics_integrators.cpp:331-414) fused in as well: instead of the 36 tangent entries it
 * writes, for every point, the compact record the L-vector gradient action streams (EXA_TANGENT_DEV5_BULK: the tangent's 5 x 5
 * deviatoric block and bulk term times dt W_q / detJ; D^T in element-assembly contexts), so that neither a tangent field nor a
 * separate exa_grad_setup pass exists on this path.  After it exa_grad_apply_lvec is valid once exa_grad_set_coords has named the
 * coordinates (the same coords_lvec).  Preconditions: p = 1 full integration, EXA_QLAYOUT_EB64, exa_set_connectivity,
 * exa_set_tangent_form(EXA_TANGENT_DEV5_BULK), partial assembly or matrix-free element assembly.  The entry points that read the
 * full 46-double records (exa_grad_apply on E-vectors, exa_grad_diagonal, exa_grad_get_ea) still need exa_grad_setup.
 * jacobian_out_dev may be NULL: the Jacobians are then not written at all (72 B per point less traffic) - on this route both integrator
 * actions can take the geometry from the nodal coordinates (exa_grad_set_coords + exa_residual_lvec with a NULL Jacobian field); a
 * caller that wants volume averages afterwards fills a Jacobian field once with exa_restrict + exa_jacobians. */
int exa_model_setup_lvec_records(exa_ctx* ctx, double dt, const double* coords_lvec_dev, const double* vel_lvec_dev,
                                 const double* stress0_dev, const double* state0_dev,
                                 double* stress1_dev, double* state1_dev, double* jacobian_out_dev, exa_stream s);
/* Tail split of the constitutive launch (0 = off, the default).  The local Newton solve needs 3-6 evaluations at most points and
 * 10-19 at a few per cent of them, and a wave waits for its slowest lane (measured wave max / mean = 1.2 ... 2.7).  With max_evals = K
 * the launch stops a point after K residual evaluations and a second, dense launch of the same kernel redoes exactly those points from
 * scratch without a cap, so every output - the evaluation count in state slot 3 included - is the one of the uncapped solve.
 * exa_model_tail_count (synchronises) returns how many points the last launch handed over. */
int exa_set_newton_cap(exa_ctx* ctx, int max_evals);
/* The same with the dense launch resuming instead of starting over (resume = 1, the default of a context): a point that is cut off leaves
 * its solver state - accepted iterate, trust radius, evaluation count; 80 bytes, stored by list slot - and the dense launch restores (r, J)
 * with one evaluation at that iterate and carries on, bit for bit the uncapped iteration.  max_evals_2 > max_evals (0 = off) adds a second
 * level: the dense launch stops at max_evals_2 evaluations and a third launch finishes what is left (the evaluation counts of a
 * Kocks-Mecking RVE have a second mode near 10 and a thin tail up to 20: two dense launches waste fewer idle lanes than one).
 * With resume on, the capped launch of a Kocks-Mecking model also lists a point at the moment one of its trial steps is rejected (such a
 * point sits at its accepted iterate with a smaller trust radius - a resumable state - and would be listed a few evaluations later anyway),
 * so that no wave of the full launch pays the re-evaluation of the accepted point; same bits again.
 * exa_set_newton_cap(ctx, K) is exa_set_newton_caps(ctx, K, 0, <current resume setting>). */
int exa_set_newton_caps(exa_ctx* ctx, int max_evals, int max_evals_2, int resume);
int exa_model_tail_count(exa_ctx* ctx, exa_stream s);
/* Let the library choose max_evals itself (what the stand-alone driver does for its own launches, host/driver.hip): after the first four launches
 * of exa_model_setup / _lvec / _lvec_records and after every fourth one from then on, the evaluation counts the launch left in state1 are binned
 * (exa_model_nfev_hist: one small launch and a 256-byte read-back that waits for the stream) and the cap of the next launches is the K that
 * minimises  E[max of 64 draws of min(n, K)] + 0.2 + (share of points above K) * tail_cost * (E[max of 64 draws | n > K] + 2)  - or no cap when that
 * does not beat the uncapped launch by 3 %.  mode: 0 = off (and the cap is cleared), 1 = for the Kocks-Mecking models only - the driver's default:
 * at 128^3 the BCC launch takes 7.2 ms with it and 17.3 ms without; the Voce launches never find a paying cap in steady state -, 2 = every model.
 * tail_cost <= 0: the measured defaults (1.5 Kocks-Mecking, 4 Voce).  Results do not depend on the cap (see above), only the launch time does.
 * The MFEM adapters (include/exaconstit_mfem_adapters.hpp) switch mode 1 on. */
int exa_set_newton_cap_auto(exa_ctx* ctx, int mode, double tail_cost);
int exa_get_newton_cap(exa_ctx* ctx);   /* the cap in force (0 = none) */
/* histogram (64 bins, last bin = 63 and more) of the evaluation counts stored in slot 3 of a state array; synchronises.  The driver
 * picks max_evals from it (host/driver.hip, choose_newton_cap). */
int exa_model_nfev_hist(exa_ctx* ctx, const double* state_dev, int* hist64_host, exa_stream s);
/* synchronises the stream; returns the number of non-converged points of the last exa_model_setup (>= 0) */
int exa_model_status(exa_ctx* ctx, exa_stream s);

/* calcDpMat (src/mechanics_ecmech.hpp:303-357): dp (3,3,Q,E) from the slip rates and orientation stored in state */
int exa_calc_dp(exa_ctx* ctx, const double* state_dev, double* dp_dev, exa_stream s);

/* geometry helpers (MFEM GeometricFactors + re-layout, src/mechanics_operator.cpp:350-391; grad_calc on any field,
 * src/mechanics_kernels.cpp:7-78, used for the deformation gradient src/mechanics_operator.cpp:393-427) */
int exa_jacobians(exa_ctx* ctx, const double* coords_evec_dev /*(n,3,E)*/, double* jacobian_dev, exa_stream s);
/* Re-layout of mfem::GeometricFactors::J (Q,3,3,E) into the (3,3,Q,E) Jacobian array (src/mechanics_operator.cpp:377-391,
 * src/mechanics_integrators.cpp:225-238): what an MFEM-side adapter calls instead of exa_jacobians. */
int exa_jacobians_from_geom(exa_ctx* ctx, const double* geom_J_dev /*(Q,3,3,E)*/, double* jacobian_dev, exa_stream s);
int exa_grad_calc(exa_ctx* ctx, const double* jacobian_dev, const double* field_evec_dev, double* grad_dev /*(3,3,Q,E), overwritten*/, exa_stream s);

/* ExaNLFIntegrator seam ---------------------------------------------------------------------------------------- */
/* AssemblePA (src/mechanics_integrators.cpp:160-314): D = W sigma adj(J)^T kept inside the context */
int exa_residual_setup(exa_ctx* ctx, const double* jacobian_dev, const double* stress1_dev, exa_stream s);
/* AddMultPA (src/mechanics_integrators.cpp:518-557): y_evec += G^T D */
int exa_residual_apply(exa_ctx* ctx, double* y_evec_dev, exa_stream s);
/* PA: TransformMatGradTo4D + AssembleGradPA (src/mechanics_model.cpp:949-1061, src/mechanics_integrators.cpp:331-513)
 * EA: AssembleEA (src/mechanics_integrators.cpp:756-1017).  The tangent is the column-major matGrad of exa_model_setup. */
int exa_grad_setup(exa_ctx* ctx, double dt, const double* jacobian_dev, const double* ddsdde_dev, exa_stream s);
/* AddMultGradPA (src/mechanics_integrators.cpp:562-622) | element mat-vec (spec src/mechanics_operator_ext.cpp:303-314):
 * y_evec += K_e x_evec */
int exa_grad_apply(exa_ctx* ctx, const double* x_evec_dev, double* y_evec_dev, exa_stream s);
/* AssembleGradDiagonalPA (src/mechanics_integrators.cpp:625-748) | EA diagonal (spec src/mechanics_operator_ext.cpp:246-252):
 * diag_evec += diag(K_e) */
int exa_grad_diagonal(exa_ctx* ctx, double* diag_evec_dev, exa_stream s);
/* EA only: copy the element matrices out in the reference layout (3n,3n,E) column-major, dof = node + n*comp */
int exa_grad_get_ea(exa_ctx* ctx, double* emat_dev, exa_stream s);

/* L-vector conveniences for callers without MFEM (element restriction, spec src/mechanics_operator_ext.cpp:149-157) -- */
/* connectivity (n,E) of local node -> L-vector node; L-vectors are byNODES: [x0..xN, y0.., z0..] with nnodes entries per comp */
int exa_set_connectivity(exa_ctx* ctx, const int32_t* conn_dev /*(n,E)*/, int nnodes);
int exa_restrict(exa_ctx* ctx, const double* lvec_dev, double* evec_dev, exa_stream s);                 /* L -> E */
int exa_restrict_transpose_add(exa_ctx* ctx, const double* evec_dev, double* lvec_dev, exa_stream s);   /* L += E^T */
/* fused gather / apply / scatter-add of the gradient action on L-vectors (the PCG inner kernel): y_L += K x_L.
 * mask_dev (nullable, 3*nnodes bytes): essential dofs — x is read as 0 there (spec src/mechanics_operator_ext.cpp:143-146). */
int exa_grad_apply_lvec(exa_ctx* ctx, const double* x_lvec_dev, double* y_lvec_dev, const uint8_t* mask_dev, exa_stream s);
/* Optional (p = 1 partial assembly, L-vector action): the nodal coordinates (nnodes,3 byNODES) the Jacobians given to exa_grad_setup
 * were computed from.  When set, exa_grad_apply_lvec recomputes adj(J) from them instead of streaming it from its per-point record
 * (36 instead of 46 doubles per point from HBM); the array must stay unchanged until the next exa_grad_setup.  NULL switches it off. */
int exa_grad_set_coords(exa_ctx* ctx, const double* coords_lvec_dev);
/* Form of the tangent the p = 1 partial-assembly action streams when the geometry is recomputed (exa_grad_set_coords):
 *   EXA_TANGENT_FULL       (default) the 36 entries of ddsdde;
 *   EXA_TANGENT_DEV5_BULK  ddsdde = V65 D V65^T + K m m^T, m = (1,1,1,0,0,0): a 5 x 5 block in the deviatoric vector basis of ExaCMech
 *                          plus a bulk term - what every ExaCMech evptn model returns (26 numbers, 13 instead of 18 16-byte loads per point).
 *                          exa_grad_setup projects ddsdde onto that form; the projection is exact only if the tangent has the form, which
 *                          exa_grad_tangent_defect measures (max over the points of |C - projection|_max / |C|_max; ~1e-16 for ExaCMech
 *                          tangents).  The caller is responsible for checking it; the stand-alone driver does on every GetGradient.
 *                          p = 1 partial assembly: exa_grad_setup then writes the compact records only (71 instead of 117 doubles moved per point); the
 *                          46-double records that exa_grad_diagonal, exa_grad_apply on E-vectors and an L-vector action without exa_grad_set_coords
 *                          read are built when one of them is first called, from the jacobian_dev / ddsdde_dev arrays of that exa_grad_setup - which
 *                          must therefore stay unchanged until then (they do within a Newton iteration).
 *   EXA_TANGENT_DEV5_BULK_GEO  the same compact tangent with adj(J) and W detJ kept in the record (36 numbers, 18 pairs): for the E-vector action
 *                          exa_grad_apply at p = 1 partial assembly, which has no nodal coordinates to recompute the geometry from - 360 instead of 440 bytes per
 *                          point and action (what HipExaNLFIntegrator selects after the same defect check); the 46-double records are built on demand as above. */
enum { EXA_TANGENT_FULL = 0, EXA_TANGENT_DEV5_BULK = 1, EXA_TANGENT_DEV5_BULK_GEO = 2 };
int exa_set_tangent_form(exa_ctx* ctx, int form);
int exa_grad_tangent_defect(exa_ctx* ctx, const double* ddsdde_dev, double* defect_host, exa_stream s);
/* Element assembly without the element matrices: with `on` != 0 exa_grad_setup stops after the per-point records (and the
 * element-average gradients of the B-bar integrator) and exa_grad_apply_lvec computes the action of the element matrices from them -
 * the same operator (y_j += sum_i A_ij x_i, A = B^T C B, i.e. B^T C^T B x), from 0.3-0.4 KB per point instead of 4.6 KB (p = 1,
 * 24 x 24) or 52 KB (p = 2, 81 x 81) per element.  Built for p = 1 full integration and for p = 2 (plain and B-bar); other
 * contexts keep the assembled path.  The matrices themselves are assembled on the first call that needs them (exa_grad_apply on
 * E-vectors, exa_grad_diagonal, exa_grad_get_ea).  Default: off. */
int exa_set_ea_matrix_free(exa_ctx* ctx, int on);
/* self-test hook (tests/test_gpu_point_fixtures.py): the Kocks-Mecking kinetics' own exp and near-1 log evaluated on the device,
 * out[0..n) = exp(x), out[n..2n) = log(x) (the latter meaningful for x in [0.75, 1.25]); no context needed */
int exa_selftest_km_math(const double* x_dev, double* out_dev, int n, exa_stream s);
/* Bit-reproducible E->L sums (reference: mfem::ElementRestriction::MultTranspose; the fused kernels here scatter with FP64 atomics, whose
 * order - and therefore the last bits of the L-vector, and from there every CG iterate - changes from run to run).  With `on` != 0
 * exa_residual_lvec, exa_grad_apply_lvec (partial assembly and element assembly from the point records) and exa_restrict_transpose_add
 * write per-element outputs and add them node by node in ascending element order (node -> element table built once per connectivity).
 * Costs one extra write + read of the element outputs (24 doubles per element).  The fused L-vector entries are ordered for p = 1 full
 * integration; in other contexts they return EXA_ERR_UNSUPPORTED while the mode is on and the E-vector entries + exa_restrict_transpose_add
 * are the reproducible route (what the stand-alone driver then takes).  Default off. */
int exa_set_deterministic(exa_ctx* ctx, int on);
/* fused AssemblePA + AddMultPA + E->L: y_L += B^T sigma (p = 1 full integration; p = 2 full integration and B-bar, where the
 * element-average gradients are refreshed from the Jacobians first: ICExaNLFIntegrator::AssemblePA + AddMultPA).
 * p = 1 full integration: jacobian_dev may be NULL when exa_grad_set_coords has named the nodal coordinates of the configuration -
 * adj(J) is then recomputed from them (the scatter gathers the connectivity anyway) instead of read from a Jacobian field. */
int exa_residual_lvec(exa_ctx* ctx, const double* jacobian_dev, const double* stress1_dev, double* y_lvec_dev, exa_stream s);
/* volume average  sum_q W detJ val / sum_q W detJ  (src/mechanics_kernels.hpp:19-134); out_host[vdim] (+ volume in out_host[vdim]).
 * Synchronises the stream. */
int exa_vol_avg(exa_ctx* ctx, const double* jacobian_dev, const double* qf_dev, int vdim, int normalise, double* out_host, exa_stream s);

#ifdef __cplusplus
}
#endif
#endif
