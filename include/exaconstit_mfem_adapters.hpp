// MFEM-side adapters that put libexaconstit_hip.so behind ExaConstit's two plug-in seams:
//   HipExaModel          : ExaModel           reference src/mechanics_model.hpp:17-241 (ctor :70-75, ModelSetup :109-111, calcDpMat :233)
//                                             replaces ECMechXtalModel<...>, src/mechanics_ecmech.hpp:111-363
//   HipExaNLFIntegrator  : ExaNLFIntegrator   reference src/mechanics_integrators.hpp:14-76
// selected in NonlinearMechOperator's model / integrator switch (src/mechanics_operator.cpp:49-210).
// A second pair serves the same two seams with L-vectors instead of E-vectors (the fused kernels of the stand-alone driver):
//   HipExaModelLVec          : HipExaModel        ModelSetup(.., vel = velocity L-vector): node gathers, Jacobians and constitutive update in one launch
//   HipExaNLFIntegratorLVec  : ExaNLFIntegrator   AddMultPA / AddMultGradPA on L-vectors: gather, element action and scatter-add in one launch
// for an operator that skips its element restrictions ("The L-vector pair").
//
// Header-only.  In an ExaConstit build it is included after mfem.hpp and ExaConstit's own headers.  MFEM is not part of this
// repository's image; tests/test_adapters.py compiles this very file against tests/mock_mfem/ (a small restatement of the part of the
// mfem::Vector / QuadratureFunction / GeometricFactors / NonlinearFormIntegrator and ExaModel / ExaNLFIntegrator surface used below)
// and drives ModelSetup -> AssemblePA/AddMultPA -> AssembleGradPA/AddMultGradPA/diagonal -> AssembleEA through it on the GPU.
//
// Device pointers: with mfem::Device::Configure("hip") Vector::Read()/Write()/ReadWrite() return device pointers, which is what the C ABI
// expects.  Streams: nullptr (the default stream) keeps MFEM's implicit ordering.
#pragma once

#if defined(EXA_ADAPTER_MOCK_MFEM)
#include "mock_mfem.hpp"               // tests/mock_mfem/mock_mfem.hpp
#elif __has_include("mfem.hpp")
#include "mfem.hpp"
#include "mechanics_model.hpp"         // ExaConstit: ExaModel, Assembly
#include "mechanics_integrators.hpp"   // ExaConstit: ExaNLFIntegrator
#else
#error "exaconstit_mfem_adapters.hpp needs MFEM (mfem.hpp) and ExaConstit's mechanics_model.hpp / mechanics_integrators.hpp on the include path"
#endif

#include <hip/hip_runtime.h>   // hipMemsetAsync (the L-vector pair zeroes an E-vector scratch on the stream)
#include <stdexcept>
#include <string>
#include "exaconstit_hip.h"

#ifndef EXA_ADAPTER_VERIFY
#define EXA_ADAPTER_VERIFY(cond, msg) do { if (!(cond)) throw std::runtime_error(std::string("exaconstit_hip adapter: ") + (msg)); } while (0)
#endif

// ExaConstit's (xtal_type, slip_type) -> library model id (src/mechanics_ecmech.hpp:407-414,460-463)
inline int exa_model_id(bool bcc, int slip /* 0 powervoce, 1 powervocenl, 2 mtsdd */) {
   return slip == 0 ? (bcc ? EXA_BCC_VOCE : EXA_FCC_VOCE) : (slip == 1 ? (bcc ? EXA_BCC_VOCE_NL : EXA_FCC_VOCE_NL) : (bcc ? EXA_BCC_KMDD : EXA_FCC_KMDD));
}

class HipExaModel : public ExaModel {
   exa_ctx* ctx_ = nullptr;
   bool check_local_solves_;
 public:
   HipExaModel(mfem::QuadratureFunction* q_stress0, mfem::QuadratureFunction* q_stress1, mfem::QuadratureFunction* q_matGrad,
               mfem::QuadratureFunction* q_matVars0, mfem::QuadratureFunction* q_matVars1, mfem::ParGridFunction* beg_coords,
               mfem::ParGridFunction* end_coords, mfem::Vector* props, int nProps, int nStateVars, double temp_k, int model_id, int order,
               int nelems, Assembly assembly_, bool bbar = false, bool check_local_solves = true)
      : ExaModel(q_stress0, q_stress1, q_matGrad, q_matVars0, q_matVars1, beg_coords, end_coords, props, nProps, nStateVars, assembly_),
        check_local_solves_(check_local_solves) {
      exa_config cfg;
      cfg.model = model_id; cfg.nprops = nProps; cfg.props = props->HostRead(); cfg.temp_k = temp_k; cfg.order = order; cfg.nelems = nelems;
      cfg.assembly = (assembly_ == Assembly::PA) ? EXA_ASSEMBLY_PA : EXA_ASSEMBLY_EA;      // FULL assembles the same element operator
      cfg.integ = bbar ? EXA_INTEG_BBAR : EXA_INTEG_FULL; cfg.device = -1;
      int err = 0;
      ctx_ = exa_create(&cfg, &err);
      EXA_ADAPTER_VERIFY(ctx_ != nullptr, "exa_create failed with code " + std::to_string(err));
      EXA_ADAPTER_VERIFY(exa_num_state_vars(ctx_) == nStateVars, "state variable count mismatch");
      // tail split of the constitutive launch, chosen by the library from its own launches' evaluation counts (Kocks-Mecking models only: 128^3 BCC 17.3 -> 7.2 ms;
      // results do not depend on it, include/exaconstit_hip.h).  exa_set_newton_cap_auto(ctx(), 0, 0) switches it off.
      EXA_ADAPTER_VERIFY(exa_set_newton_cap_auto(ctx_, 1, 0.0) == EXA_OK, exa_last_error(ctx_));
   }
   ~HipExaModel() override { exa_destroy(ctx_); }
   HipExaModel(const HipExaModel&) = delete; HipExaModel& operator=(const HipExaModel&) = delete;

   // ECMechXtalModel::init_state_vars (src/mechanics_ecmech.hpp:264-300) with one quaternion per element
   void InitStateVars(const mfem::Vector& quats_per_elem) {
      EXA_ADAPTER_VERIFY(exa_init_state(ctx_, matVars0->ReadWrite(), quats_per_elem.Read(), nullptr) == EXA_OK, exa_last_error(ctx_));
   }

   // src/mechanics_model.hpp:109-111, called from NonlinearMechOperator::Setup (src/mechanics_operator.cpp:339-347).
   // The library owns the reference-element gradient table (exa_shape_table), so loc_grad is not needed.
   void ModelSetup(const int nqpts, const int nelems, const int /*space_dim*/, const int nnodes, const mfem::Vector& jacobian,
                   const mfem::Vector& /*loc_grad*/, const mfem::Vector& vel) override {
      EXA_ADAPTER_VERIFY(nqpts == exa_qpts_per_elem(ctx_) && nnodes == exa_nodes_per_elem(ctx_), "element order does not match the context");
      (void)nelems;
      // ExaCMech fails the run when a local solve does not converge (ECMECH_FAIL): exa_model_setup_checked returns that count (> 0) at the price
      // of one 4-byte read-back; check_local_solves = false keeps the launch asynchronous (exa_model_status can be asked later)
      const int rc = check_local_solves_
         ? exa_model_setup_checked(ctx_, dt, jacobian.Read(), vel.Read(), stress0->Read(), matVars0->Read(), stress1->Write(), matVars1->Write(), matGrad->Write(), nullptr)
         : exa_model_setup(ctx_, dt, jacobian.Read(), vel.Read(), stress0->Read(), matVars0->Read(), stress1->Write(), matVars1->Write(), matGrad->Write(), nullptr);
      EXA_ADAPTER_VERIFY(rc >= EXA_OK, exa_last_error(ctx_));
      EXA_ADAPTER_VERIFY(rc == 0, "the constitutive update did not converge at " + std::to_string(rc) + " quadrature point(s)");
   }
   void UpdateModelVars() override {}
   void calcDpMat(mfem::QuadratureFunction& DpMat) const override {            // src/mechanics_model.hpp:233, src/mechanics_ecmech.hpp:302-363
      EXA_ADAPTER_VERIFY(exa_calc_dp(ctx_, matVars1->Read(), DpMat.Write(), nullptr) == EXA_OK, exa_last_error(ctx_));
   }
   exa_ctx* ctx() const { return ctx_; }
   bool checks_local_solves() const { return check_local_solves_; }
};

// TransformMatGradTo4D() (called for PA at src/mechanics_operator.cpp:297-300) is not needed with this integrator: the 4-D tensor is never
// materialised, and the tangent in matGrad is already column-major (no transpose pass, cf. src/mechanics_ecmech.cpp:155-170).
class HipExaNLFIntegrator : public ExaNLFIntegrator {
   HipExaModel* hmodel_;
   bool compact_;              // p = 1 partial assembly: AddMultGradPA streams the tangent as its 5 x 5 deviatoric block + bulk term with adj(J) (EXA_TANGENT_DEV5_BULK_GEO:
                               // 36 instead of 46 doubles per point), after a check of matGrad per AssembleGradPA; other contexts refuse the form and keep the full records
   mfem::Vector jac_;          // (3,3,Q,E)
   // geometric factors of the current (end-of-step) mesh nodes, re-laid-out as src/mechanics_integrators.cpp:225-238 does
   void RefreshJacobians(const mfem::FiniteElementSpace& fes) {
      const mfem::FiniteElement& el = *fes.GetFE(0);
      const mfem::IntegrationRule* ir = &(mfem::IntRules.Get(el.GetGeomType(), 2 * el.GetOrder() + 1));
      const mfem::GeometricFactors* g = fes.GetMesh()->GetGeometricFactors(*ir, mfem::GeometricFactors::JACOBIANS);
      if (jac_.Size() != g->J.Size()) { jac_.SetSize(g->J.Size()); jac_.UseDevice(true); }
      EXA_ADAPTER_VERIFY(exa_jacobians_from_geom(hmodel_->ctx(), g->J.Read(), jac_.Write(), nullptr) == EXA_OK, exa_last_error(hmodel_->ctx()));
   }
 public:
   explicit HipExaNLFIntegrator(HipExaModel* m, bool compact_tangent = true)
      : ExaNLFIntegrator(m), hmodel_(m), compact_(compact_tangent && exa_set_tangent_form(m->ctx(), EXA_TANGENT_DEV5_BULK_GEO) == EXA_OK) {}
   using ExaNLFIntegrator::AssemblePA;
   void AssemblePA(const mfem::FiniteElementSpace& fes) override {                                   // src/mechanics_integrators.cpp:160-314
      RefreshJacobians(fes);
      EXA_ADAPTER_VERIFY(exa_residual_setup(hmodel_->ctx(), jac_.Read(), model->GetStress1()->Read(), nullptr) == EXA_OK, exa_last_error(hmodel_->ctx()));
   }
   void AddMultPA(const mfem::Vector& /*x*/, mfem::Vector& y) const override {                       // :518-557
      EXA_ADAPTER_VERIFY(exa_residual_apply(hmodel_->ctx(), y.ReadWrite(), nullptr) == EXA_OK, exa_last_error(hmodel_->ctx()));
   }
   void AssembleGradPA(const mfem::Vector& /*x*/, const mfem::FiniteElementSpace& fes) override { AssembleGradPA(fes); }
   void AssembleGradPA(const mfem::FiniteElementSpace& fes) override {                               // :331-513
      RefreshJacobians(fes);
      if (compact_) {   // exact for every ExaCMech tangent; a matGrad of another form switches it off for good
         double defect = 0.0;
         EXA_ADAPTER_VERIFY(exa_grad_tangent_defect(hmodel_->ctx(), model->GetMatGrad()->Read(), &defect, nullptr) == EXA_OK, exa_last_error(hmodel_->ctx()));
         if (!(defect < 1e-11)) { compact_ = false; EXA_ADAPTER_VERIFY(exa_set_tangent_form(hmodel_->ctx(), EXA_TANGENT_FULL) == EXA_OK, exa_last_error(hmodel_->ctx())); }
      }
      EXA_ADAPTER_VERIFY(exa_grad_setup(hmodel_->ctx(), model->GetModelDt(), jac_.Read(), model->GetMatGrad()->Read(), nullptr) == EXA_OK, exa_last_error(hmodel_->ctx()));
   }
   void AddMultGradPA(const mfem::Vector& x, mfem::Vector& y) const override {                       // :562-622
      EXA_ADAPTER_VERIFY(exa_grad_apply(hmodel_->ctx(), x.Read(), y.ReadWrite(), nullptr) == EXA_OK, exa_last_error(hmodel_->ctx()));
   }
   void AssembleGradDiagonalPA(mfem::Vector& diag) const override {                                  // :625-748
      EXA_ADAPTER_VERIFY(exa_grad_diagonal(hmodel_->ctx(), diag.ReadWrite(), nullptr) == EXA_OK, exa_last_error(hmodel_->ctx()));
   }
   void AssembleGradEA(const mfem::Vector& /*x*/, const mfem::FiniteElementSpace& fes, mfem::Vector& emat) override { AssembleEA(fes, emat); }
   void AssembleEA(const mfem::FiniteElementSpace& fes, mfem::Vector& emat) override {               // :756-1017
      RefreshJacobians(fes);
      EXA_ADAPTER_VERIFY(exa_grad_setup(hmodel_->ctx(), model->GetModelDt(), jac_.Read(), model->GetMatGrad()->Read(), nullptr) == EXA_OK, exa_last_error(hmodel_->ctx()));
      EXA_ADAPTER_VERIFY(exa_grad_get_ea(hmodel_->ctx(), emat.Write(), nullptr) == EXA_OK, exa_last_error(hmodel_->ctx()));   // reference layout (3n,3n,E)
   }
};

// ---- the L-vector pair ---------------------------------------------------------------------------------------------------------------------------------
// Same seams, other vectors.  The reference's operator restricts coordinates and velocity to E-vectors and refreshes MFEM's geometric factors before
// ModelSetup (src/mechanics_operator.cpp:310-348, 350-391), and its gradient extension wraps AddMultGradPA in elem_restrict->Mult / MultTranspose
// (src/mechanics_operator_ext.cpp:143-157) - except where the space has no element restriction, in which case the integrators are handed the global
// vectors (:159-165).  This pair takes that second form: ModelSetup receives the velocity L-vector (what the operator already passes to a UMAT model,
// src/mechanics_operator.cpp:339-341, after P->Mult), gathers nodes through the element -> node table, computes the Jacobians itself and keeps them for the
// integrator; AddMultPA / AddMultGradPA act on L-vectors (byNODES, the ordering of the reference's nodal space, src/mechanics_driver.cpp:336).
// The quadrature functions stay MFEM's, in MFEM's (vdim, Q, E) layout.
class HipExaModelLVec : public HipExaModel {
   mfem::Vector jac_;            // (3,3,Q,E) of the end-of-step configuration, written by the constitutive launch
   mfem::Array<int> conn_;       // (n, E) element -> node, ElementDofOrdering::NATIVE (src/mechanics_operator.cpp:228)
   int nnodes_ = 0; bool fused_ = false;
 public:
   HipExaModelLVec(mfem::QuadratureFunction* q_stress0, mfem::QuadratureFunction* q_stress1, mfem::QuadratureFunction* q_matGrad,
                   mfem::QuadratureFunction* q_matVars0, mfem::QuadratureFunction* q_matVars1, mfem::ParGridFunction* beg_coords,
                   mfem::ParGridFunction* end_coords, mfem::Vector* props, int nProps, int nStateVars, double temp_k, int model_id,
                   const mfem::FiniteElementSpace& fes, Assembly assembly_, bool bbar = false, bool check_local_solves = true, bool fused_records = false)
      : HipExaModel(q_stress0, q_stress1, q_matGrad, q_matVars0, q_matVars1, beg_coords, end_coords, props, nProps, nStateVars, temp_k, model_id,
                    fes.GetFE(0)->GetOrder(), fes.GetNE(), assembly_, bbar, check_local_solves) {
      const int n = exa_nodes_per_elem(ctx()), E = fes.GetNE();
      nnodes_ = fes.GetNDofs();
      conn_.SetSize(n * E);
      int* c = conn_.HostWrite();
      mfem::Array<int> dofs;
      for (int e = 0; e < E; e++) {
         fes.GetElementDofs(e, dofs);            // scalar dofs = node numbers, native element order
         EXA_ADAPTER_VERIFY(dofs.Size() == n, "element with an unexpected number of nodes");
         for (int a = 0; a < n; a++) c[a + n * e] = dofs[a];
      }
      static_assert(sizeof(int) == sizeof(int32_t), "the connectivity table is 32-bit");
      EXA_ADAPTER_VERIFY(exa_set_connectivity(ctx(), conn_.Read(), nnodes_) == EXA_OK, exa_last_error(ctx()));
      jac_.SetSize(9 * exa_qpts_per_elem(ctx()) * E); jac_.UseDevice(true);
      // fused_records (p = 1 full integration, the reference's identity "Jacobi"): ModelSetup writes the compact records the L-vector action streams instead of
      // matGrad - AssembleGradPA rides in the constitutive launch and matGrad is NOT written; the diagonal and AssembleEA then have nothing to read
      fused_ = fused_records;
      if (fused_) EXA_ADAPTER_VERIFY(exa_set_tangent_form(ctx(), EXA_TANGENT_DEV5_BULK) == EXA_OK, exa_last_error(ctx()));
   }
   bool FusedRecords() const { return fused_; }
   // vel: the velocity L-vector (3 * nnodes, byNODES).  jacobian and loc_grad are not read: the launch gathers end_coords and writes Jacobians().
   void ModelSetup(const int nqpts, const int /*nelems*/, const int /*space_dim*/, const int nnodes, const mfem::Vector& /*jacobian*/,
                   const mfem::Vector& /*loc_grad*/, const mfem::Vector& vel) override {
      EXA_ADAPTER_VERIFY(nqpts == exa_qpts_per_elem(ctx()) && nnodes == exa_nodes_per_elem(ctx()), "element order does not match the context");
      EXA_ADAPTER_VERIFY(vel.Size() == 3 * nnodes_ && end_coords->Size() == 3 * nnodes_, "HipExaModelLVec::ModelSetup takes the velocity L-vector");
      int rc = fused_ ? exa_model_setup_lvec_records(ctx(), dt, end_coords->Read(), vel.Read(), stress0->Read(), matVars0->Read(), stress1->Write(), matVars1->Write(),
                                                     jac_.Write(), nullptr)
                      : exa_model_setup_lvec(ctx(), dt, end_coords->Read(), vel.Read(), stress0->Read(), matVars0->Read(), stress1->Write(), matVars1->Write(),
                                             matGrad->Write(), jac_.Write(), nullptr);
      EXA_ADAPTER_VERIFY(rc >= EXA_OK, exa_last_error(ctx()));
      if (checks_local_solves()) {
         rc = exa_model_status(ctx(), nullptr);
         EXA_ADAPTER_VERIFY(rc == 0, "the constitutive update did not converge at " + std::to_string(rc) + " quadrature point(s)");
      }
   }
   const mfem::Vector& Jacobians() const { return jac_; }
   const mfem::ParGridFunction* EndCoords() const { return end_coords; }
   int NumNodes() const { return nnodes_; }
};

class HipExaNLFIntegratorLVec : public ExaNLFIntegrator {
   HipExaModelLVec* hmodel_;
   bool compact_;                    // stream the tangent as its 5 x 5 deviatoric block + bulk term (exaconstit_hip.h, EXA_TANGENT_DEV5_BULK), checked per AssembleGradPA
   mutable mfem::Vector ev_;         // E-vector scratch of the diagonal
   exa_ctx* ctx() const { return hmodel_->ctx(); }
   void GradSetup() {
      if (hmodel_->FusedRecords()) {   // the records were written by ModelSetup; the action only needs the coordinates they belong to
         EXA_ADAPTER_VERIFY(exa_grad_set_coords(ctx(), hmodel_->EndCoords()->Read()) == EXA_OK, exa_last_error(ctx()));
         return;
      }
      if (compact_) {                // valid for every ExaCMech tangent; a matGrad that is not of the form (another model behind the seam) switches it off for good
         double defect = 0.0;
         EXA_ADAPTER_VERIFY(exa_grad_tangent_defect(ctx(), model->GetMatGrad()->Read(), &defect, nullptr) == EXA_OK, exa_last_error(ctx()));
         if (!(defect < 1e-11)) { compact_ = false; EXA_ADAPTER_VERIFY(exa_set_tangent_form(ctx(), EXA_TANGENT_FULL) == EXA_OK, exa_last_error(ctx())); }
      }
      EXA_ADAPTER_VERIFY(exa_grad_setup(ctx(), model->GetModelDt(), hmodel_->Jacobians().Read(), model->GetMatGrad()->Read(), nullptr) == EXA_OK, exa_last_error(ctx()));
      // p = 1: the action recomputes adj(J) from the coordinates the Jacobians came from (unchanged until the next ModelSetup) instead of streaming it
      EXA_ADAPTER_VERIFY(exa_grad_set_coords(ctx(), hmodel_->EndCoords()->Read()) == EXA_OK, exa_last_error(ctx()));
   }
 public:
   explicit HipExaNLFIntegratorLVec(HipExaModelLVec* m, bool compact_tangent = true) : ExaNLFIntegrator(m), hmodel_(m), compact_(compact_tangent || m->FusedRecords()) {
      if (compact_) EXA_ADAPTER_VERIFY(exa_set_tangent_form(ctx(), EXA_TANGENT_DEV5_BULK) == EXA_OK, exa_last_error(ctx()));
   }
   using ExaNLFIntegrator::AssemblePA;
   void AssemblePA(const mfem::FiniteElementSpace& /*fes*/) override {}                              // the residual action reads sigma and J itself
   void AddMultPA(const mfem::Vector& /*x*/, mfem::Vector& y) const override {                       // y_L += B^T sigma  (AssemblePA + AddMultPA + E->L, :160-314, :518-557)
      EXA_ADAPTER_VERIFY(y.Size() == 3 * hmodel_->NumNodes(), "HipExaNLFIntegratorLVec acts on L-vectors");
      EXA_ADAPTER_VERIFY(exa_residual_lvec(ctx(), hmodel_->Jacobians().Read(), model->GetStress1()->Read(), y.ReadWrite(), nullptr) == EXA_OK, exa_last_error(ctx()));
   }
   void AssembleGradPA(const mfem::Vector& /*x*/, const mfem::FiniteElementSpace& fes) override { AssembleGradPA(fes); }
   void AssembleGradPA(const mfem::FiniteElementSpace& /*fes*/) override { GradSetup(); }            // :331-513
   void AddMultGradPA(const mfem::Vector& x, mfem::Vector& y) const override {                       // y_L += K x_L  (L->E + :562-622 + E->L)
      EXA_ADAPTER_VERIFY(x.Size() == 3 * hmodel_->NumNodes() && y.Size() == x.Size(), "HipExaNLFIntegratorLVec acts on L-vectors");
      EXA_ADAPTER_VERIFY(exa_grad_apply_lvec(ctx(), x.Read(), y.ReadWrite(), nullptr, nullptr) == EXA_OK, exa_last_error(ctx()));
   }
   void AssembleGradDiagonalPA(mfem::Vector& diag) const override {                                  // diag_L += diag(K)  (:625-748 + E->L)
      const int nev = 3 * exa_nodes_per_elem(ctx()) * (hmodel_->Jacobians().Size() / (9 * exa_qpts_per_elem(ctx())));
      if (ev_.Size() != nev) { ev_.SetSize(nev); ev_.UseDevice(true); }
      EXA_ADAPTER_VERIFY(hipMemsetAsync(ev_.Write(), 0, sizeof(double) * nev, nullptr) == hipSuccess, "hipMemsetAsync");
      EXA_ADAPTER_VERIFY(exa_grad_diagonal(ctx(), ev_.ReadWrite(), nullptr) == EXA_OK, exa_last_error(ctx()));
      EXA_ADAPTER_VERIFY(exa_restrict_transpose_add(ctx(), ev_.Read(), diag.ReadWrite(), nullptr) == EXA_OK, exa_last_error(ctx()));
   }
   void AssembleGradEA(const mfem::Vector& /*x*/, const mfem::FiniteElementSpace& fes, mfem::Vector& emat) override { AssembleEA(fes, emat); }
   void AssembleEA(const mfem::FiniteElementSpace& /*fes*/, mfem::Vector& emat) override {           // :756-1017; the element matrices in the reference layout (3n,3n,E)
      GradSetup();
      EXA_ADAPTER_VERIFY(exa_grad_get_ea(ctx(), emat.Write(), nullptr) == EXA_OK, exa_last_error(ctx()));
   }
};
