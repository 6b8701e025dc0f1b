// TEST INFRASTRUCTURE.  Minimal stand-in for the part of MFEM's and ExaConstit's class surface that
// include/exaconstit_mfem_adapters.hpp touches, so that the adapters are compiled and run in an image without MFEM:
//   mfem::Vector (device-resident, Read/Write/ReadWrite/HostRead/HostWrite/SetSize/UseDevice/Size), QuadratureFunction, ParGridFunction,
//   FiniteElement (GetGeomType/GetOrder), IntRules.Get, GeometricFactors (J laid out (Q,3,3,E)), Mesh::GetGeometricFactors,
//   FiniteElementSpace (GetFE/GetMesh/GetNE/GetNDofs/GetElementDofs), Array<int> (SetSize/HostWrite/Read/Size/operator[]), NonlinearFormIntegrator (the PA / EA virtuals of the MFEM fork ExaConstit builds on),
//   Assembly, ExaModel (reference src/mechanics_model.hpp:17-241: members, ctor :70-75, ModelSetup :109-111, accessors),
//   ExaNLFIntegrator (reference src/mechanics_integrators.hpp:14-76).
// Signatures follow the reference headers; bodies are the least that works.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdexcept>
#include <vector>

namespace mfem {

class Vector {
   double* d_ = nullptr; int n_ = 0; mutable std::vector<double> h_;
 public:
   Vector() = default;
   explicit Vector(int n) { SetSize(n); }
   Vector(const Vector&) = delete; Vector& operator=(const Vector&) = delete;
   ~Vector() { if (d_) (void)hipFree(d_); }
   void SetSize(int n) { if (d_) (void)hipFree(d_); d_ = nullptr; n_ = n; if (n && hipMalloc(&d_, sizeof(double) * n) != hipSuccess) throw std::runtime_error("mock Vector: hipMalloc"); if (n) (void)hipMemset(d_, 0, sizeof(double) * n); }
   void UseDevice(bool) {}
   int Size() const { return n_; }
   const double* Read() const { return d_; }
   double* Write() { return d_; }
   double* ReadWrite() { return d_; }
   const double* HostRead() const { h_.resize(n_); if (n_) (void)hipMemcpy(h_.data(), d_, sizeof(double) * n_, hipMemcpyDeviceToHost); return h_.data(); }
   void FromHost(const double* p) { if (n_) (void)hipMemcpy(d_, p, sizeof(double) * n_, hipMemcpyHostToDevice); }
};

// mfem::Array<T>: host array with a device mirror that Read() brings up to date (the L-vector adapter's element -> node table)
template <class T> class Array {
   std::vector<T> h_; mutable T* d_ = nullptr; mutable int dn_ = 0;
 public:
   Array() = default; Array(const Array&) = delete; Array& operator=(const Array&) = delete;
   ~Array() { if (d_) (void)hipFree(d_); }
   void SetSize(int n) { h_.resize(n); }
   int Size() const { return (int)h_.size(); }
   T* HostWrite() { return h_.data(); }
   T& operator[](int i) { return h_[i]; }
   const T& operator[](int i) const { return h_[i]; }
   const T* Read() const {
      if (dn_ != (int)h_.size()) { if (d_) (void)hipFree(d_); d_ = nullptr; dn_ = (int)h_.size(); if (dn_ && hipMalloc(&d_, sizeof(T) * dn_) != hipSuccess) throw std::runtime_error("mock Array: hipMalloc"); }
      if (dn_) (void)hipMemcpy(d_, h_.data(), sizeof(T) * dn_, hipMemcpyHostToDevice);
      return d_;
   }
};

class QuadratureFunction : public Vector { int vdim_ = 1; public: QuadratureFunction(int npts, int vdim) : Vector(npts * vdim), vdim_(vdim) {} int GetVDim() const { return vdim_; } };
class ParGridFunction : public Vector { public: using Vector::Vector; };

struct Geometry { enum Type { CUBE = 5 }; };
class IntegrationRule { public: int order = 0; };
class IntegrationRules { IntegrationRule r_; public: const IntegrationRule& Get(int /*geom*/, int order) { r_.order = order; return r_; } };
static IntegrationRules IntRules;

class FiniteElement { int p_; public: explicit FiniteElement(int p) : p_(p) {} Geometry::Type GetGeomType() const { return Geometry::CUBE; } int GetOrder() const { return p_; } };

class GeometricFactors { public: enum FactorFlags { COORDINATES = 1, JACOBIANS = 2, DETERMINANTS = 4 }; Vector J; };
class Mesh {
   GeometricFactors gf_;
 public:
   GeometricFactors& factors() { return gf_; }      // the test fills J (Q,3,3,E)
   const GeometricFactors* GetGeometricFactors(const IntegrationRule&, int /*flags*/) const { return &gf_; }
};
class FiniteElementSpace {
   Mesh* mesh_; FiniteElement fe_;
   std::vector<int> conn_; int ne_ = 0, ndofs_ = 0, npe_ = 0;      // element -> scalar dof (node) table, native element order
 public:
   FiniteElementSpace(Mesh* m, int p) : mesh_(m), fe_(p) {}
   void SetElementDofs(const int* conn, int npe, int ne, int ndofs) { conn_.assign(conn, conn + (size_t)npe * ne); npe_ = npe; ne_ = ne; ndofs_ = ndofs; }   // (mock only)
   const FiniteElement* GetFE(int) const { return &fe_; }
   Mesh* GetMesh() const { return mesh_; }
   int GetNE() const { return ne_; }
   int GetNDofs() const { return ndofs_; }
   void GetElementDofs(int e, Array<int>& dofs) const { dofs.SetSize(npe_); for (int a = 0; a < npe_; a++) dofs[a] = conn_[a + (size_t)npe_ * e]; }
};

class NonlinearFormIntegrator {
 public:
   virtual ~NonlinearFormIntegrator() {}
   virtual void AssemblePA(const FiniteElementSpace&) {}
   virtual void AssemblePA(const FiniteElementSpace&, const FiniteElementSpace&) {}
   virtual void AddMultPA(const Vector&, Vector&) const {}
   virtual void AssembleGradPA(const Vector&, const FiniteElementSpace&) {}
   virtual void AssembleGradPA(const FiniteElementSpace&) {}
   virtual void AddMultGradPA(const Vector&, Vector&) const {}
   virtual void AssembleGradDiagonalPA(Vector&) const {}
   virtual void AssembleGradEA(const Vector&, const FiniteElementSpace&, Vector&) {}
   virtual void AssembleEA(const FiniteElementSpace&, Vector&) {}
};

}  // namespace mfem

// ---- ExaConstit side (reference src/option_types.hpp, src/mechanics_model.hpp, src/mechanics_integrators.hpp) ----
enum class Assembly { FULL, PA, EA, NOTYPE };

class ExaModel {
 public:
   int numProps; int numStateVars; bool init_step = false;
 protected:
   double dt = 0, t = 0;
   mfem::ParGridFunction* beg_coords; mfem::ParGridFunction* end_coords;
   mfem::QuadratureFunction* stress0; mfem::QuadratureFunction* stress1; mfem::QuadratureFunction* matGrad;
   mfem::QuadratureFunction* matVars0; mfem::QuadratureFunction* matVars1;
   mfem::Vector* matProps; Assembly assembly;
 public:
   ExaModel(mfem::QuadratureFunction* q_stress0, mfem::QuadratureFunction* q_stress1, mfem::QuadratureFunction* q_matGrad, mfem::QuadratureFunction* q_matVars0,
            mfem::QuadratureFunction* q_matVars1, mfem::ParGridFunction* _beg_coords, mfem::ParGridFunction* _end_coords, mfem::Vector* props, int nProps,
            int nStateVars, Assembly _assembly)
      : numProps(nProps), numStateVars(nStateVars), beg_coords(_beg_coords), end_coords(_end_coords), stress0(q_stress0), stress1(q_stress1), matGrad(q_matGrad),
        matVars0(q_matVars0), matVars1(q_matVars1), matProps(props), assembly(_assembly) {}
   virtual ~ExaModel() {}
   virtual void ModelSetup(const int nqpts, const int nelems, const int space_dim, const int nnodes, const mfem::Vector& jacobian, const mfem::Vector& loc_grad,
                           const mfem::Vector& vel) = 0;
   virtual void UpdateModelVars() = 0;
   virtual void calcDpMat(mfem::QuadratureFunction& DpMat) const = 0;
   void SetModelDt(const double dtime) { dt = dtime; }
   double GetModelDt() { return dt; }
   mfem::QuadratureFunction* GetStress0() { return stress0; }
   mfem::QuadratureFunction* GetStress1() { return stress1; }
   mfem::QuadratureFunction* GetMatGrad() { return matGrad; }
   mfem::QuadratureFunction* GetMatVars0() { return matVars0; }
};

class ExaNLFIntegrator : public mfem::NonlinearFormIntegrator {
 protected:
   ExaModel* model;
 public:
   ExaNLFIntegrator(ExaModel* m) : model(m) {}
   virtual ~ExaNLFIntegrator() {}
   using mfem::NonlinearFormIntegrator::AssemblePA;
   void AssembleGradPA(const mfem::Vector&, const mfem::FiniteElementSpace&) override {}
   void AssembleGradPA(const mfem::FiniteElementSpace&) override {}
   void AddMultGradPA(const mfem::Vector&, mfem::Vector&) const override {}
   void AssemblePA(const mfem::FiniteElementSpace&) override {}
   void AddMultPA(const mfem::Vector&, mfem::Vector&) const override {}
   void AssembleGradDiagonalPA(mfem::Vector&) const override {}
   void AssembleGradEA(const mfem::Vector&, const mfem::FiniteElementSpace&, mfem::Vector&) override {}
   void AssembleEA(const mfem::FiniteElementSpace&, mfem::Vector&) override {}
};
