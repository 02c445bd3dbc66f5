// rdc_host.h — C++ host layer above the C-ABI (include/rdc_assembly.h) that mirrors the slice of the
// libMesh interface the reference's assemble callbacks live in, so that the drop-in replacements
// keep the reference's names, argument meaning and call order:
//
//     model.attach_assemble_function(assemble_pihna);     // src/pihna.C:35
//     model.solve();                                      // src/pihna.C:80  -> assemble() -> callback
//     void assemble_pihna(EquationSystems& es, const std::string& system_name);   // src/pihna.C:318
//
// libMesh itself is not available in this build environment (README.md:35 pins d3bda6c), so this is
// NOT libMesh: it is the minimal set of types with the same member names that the callbacks and the
// time loop touch (EquationSystems::parameters / get_system / get_mesh, System::n_vars,
// TransientLinearImplicitSystem::{old_local_solution, matrix, rhs, attach_assemble_function}).
// A libMesh build uses integration/libmesh_adapter.C instead, which binds the same C-ABI calls to
// the real libMesh objects.
//
// Header-only, C++17, depends only on the C-ABI.  Errors: the reference aborts through
// libmesh_error(); here every failing C-ABI status becomes a std::runtime_error carrying
// rdc_last_error().
#ifndef RDC_HOST_H
#define RDC_HOST_H

#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <ostream>
#include <set>
#include <sstream>
#include <string>
#include <variant>
#include <vector>

#include "../../include/rdc_assembly.h"

namespace rdc {
namespace host {

using Real = double;
using Number = double;
using dof_id_type = uint32_t;

inline void check(rdc_ctx* c, int rc, const char* what) {
  if (rc != RDC_OK) throw std::runtime_error(std::string(what) + ": " + rdc_last_error(c));
}

// libMesh::Point as far as the parameters need it ("BC/<id>/displacement", src/solid.C:246-256)
struct Point {
  double c[3] = {0.0, 0.0, 0.0};
  Point() = default;
  Point(double x, double y, double z) : c{x, y, z} {}
  double& operator()(int d) { return c[d]; }
  double operator()(int d) const { return c[d]; }
};

// src/utils.h:268-288: the integers of a blank-separated string ("BCs", "materials", "loading_time_points")
inline std::set<int> export_integers(const std::string& s) {
  std::set<int> numbers;
  std::stringstream ss(s);
  std::string tmp;
  while (ss >> tmp) {
    int n;
    if (std::stringstream(tmp) >> n) numbers.insert(n);
  }
  return numbers;
}

// ---- libMesh::Parameters (string-keyed, typed) ------------------------------------------------
class Parameters {
 public:
  template <class T> T& set(const std::string& key) {
    auto& v = map_[key];
    if (!std::holds_alternative<T>(v)) v = T();
    return std::get<T>(v);
  }
  template <class T> const T& get(const std::string& key) const {
    auto it = map_.find(key);
    if (it == map_.end()) throw std::runtime_error("Parameters::get: no parameter '" + key + "'");
    if (!std::holds_alternative<T>(it->second)) throw std::runtime_error("Parameters::get: type mismatch for '" + key + "'");
    return std::get<T>(it->second);
  }
  template <class T> bool have_parameter(const std::string& key) const {
    auto it = map_.find(key);
    return it != map_.end() && std::holds_alternative<T>(it->second);
  }
 private:
  std::map<std::string, std::variant<Real, int, bool, std::string, Point>> map_;
};

// ---- mesh: FIRST-order TET4 / HEX8, libMesh node order -----------------------------------------
class Mesh {
 public:
  Mesh(int elem_type, std::vector<uint32_t> conn, std::vector<double> xyz)
      : elem_type_(elem_type), conn_(std::move(conn)), xyz_(std::move(xyz)) {
    if (elem_type != RDC_TET4 && elem_type != RDC_HEX8) throw std::runtime_error("Mesh: TET4 or HEX8 only");
    if (conn_.size() % elem_type || xyz_.size() % 3) throw std::runtime_error("Mesh: ragged arrays");
  }
  unsigned int mesh_dimension() const { return 3; }
  int64_t n_elem() const { return (int64_t)conn_.size() / elem_type_; }
  int64_t n_nodes() const { return (int64_t)xyz_.size() / 3; }
  int elem_type() const { return elem_type_; }
  const std::vector<uint32_t>& connectivity() const { return conn_; }
  std::vector<double>& coordinates() { return xyz_; }              // moving mesh (SolidSystem::update)
  const std::vector<double>& coordinates() const { return xyz_; }
  // elem->subdomain_id() (first Gmsh tag of the volume elements); all 0 unless set
  void set_subdomain_ids(std::vector<int32_t> ids) {
    if ((int64_t)ids.size() != n_elem()) throw std::runtime_error("Mesh: one subdomain id per element");
    subdomain_ = std::move(ids);
  }
  int32_t subdomain_id(int64_t e) const { return subdomain_.empty() ? 0 : subdomain_[(size_t)e]; }
  // get_boundary_info(): (element, libMesh side number, boundary id) of the tagged boundary faces
  struct BoundarySide { int64_t elem; int32_t side; int32_t id; };
  void add_side(int64_t elem, int32_t side, int32_t id) { sides_.push_back({elem, side, id}); }
  const std::vector<BoundarySide>& boundary_sides() const { return sides_; }
 private:
  int elem_type_;
  std::vector<uint32_t> conn_;
  std::vector<double> xyz_;
  std::vector<int32_t> subdomain_;
  std::vector<BoundarySide> sides_;
};

// ---- NumericVector / SparseMatrix (host copies; the GPU owns the assembled values) --------------
class NumericVector {
 public:
  void init(int64_t n) { v_.assign((size_t)n, 0.0); }
  int64_t size() const { return (int64_t)v_.size(); }
  Number operator()(int64_t i) const { return v_[(size_t)i]; }
  void set(int64_t i, Number x) { v_[(size_t)i] = x; }
  void zero() { std::fill(v_.begin(), v_.end(), 0.0); }
  Real l2_norm() const { long double s = 0; for (double x : v_) s += (long double)x * x; return (Real)std::sqrt((double)s); }
  NumericVector& operator=(const NumericVector&) = default;
  std::vector<double>& raw() { return v_; }
  const std::vector<double>& raw() const { return v_; }
 private:
  std::vector<double> v_;
};

class SparseMatrix {  // scalar CSR == PETSc AIJ, row = node*nvar + var
 public:
  std::vector<int64_t> row_ptr;
  std::vector<int32_t> col_idx;
  std::vector<double> val;
  int64_t m() const { return (int64_t)row_ptr.size() - 1; }
  void zero() { std::fill(val.begin(), val.end(), 0.0); }
  Number operator()(int64_t i, int64_t j) const {
    for (int64_t k = row_ptr[(size_t)i]; k < row_ptr[(size_t)i + 1]; k++) if (col_idx[(size_t)k] == j) return val[(size_t)k];
    return 0.0;
  }
  void vector_mult(std::vector<double>& y, const std::vector<double>& x) const {
    y.assign((size_t)m(), 0.0);
    for (int64_t i = 0; i < m(); i++) {
      double s = 0.0;
      for (int64_t k = row_ptr[(size_t)i]; k < row_ptr[(size_t)i + 1]; k++) s += val[(size_t)k] * x[(size_t)col_idx[(size_t)k]];
      y[(size_t)i] = s;
    }
  }
};

class EquationSystems;

// ---- System / TransientLinearImplicitSystem -----------------------------------------------------
class System {
 public:
  System(EquationSystems& es, std::string name) : es_(es), name_(std::move(name)) {}
  virtual ~System() = default;
  const std::string& name() const { return name_; }
  unsigned int add_variable(const std::string& var) { vars_.push_back(var); return (unsigned)vars_.size() - 1; }
  unsigned int n_vars() const { return (unsigned)vars_.size(); }
  unsigned int variable_number(const std::string& var) const {
    for (size_t v = 0; v < vars_.size(); v++) if (vars_[v] == var) return (unsigned)v;
    throw std::runtime_error("System::variable_number: no variable '" + var + "' in " + name_);
  }
  EquationSystems& get_equation_systems() { return es_; }
  // dof = node * n_vars + var (libMesh variable-group numbering)
  NumericVector solution, current_local_solution;
  Number current_solution(dof_id_type dof) const { return current_local_solution((int64_t)dof); }
  bool elemental = false;  // CONSTANT MONOMIAL variables (one value per element): "Tracts", fibres
  virtual void init(int64_t n_nodes) {
    solution.init(n_nodes * n_vars());
    current_local_solution.init(n_nodes * n_vars());
  }
  virtual void update() { current_local_solution = solution; }  // serial: ghost update is the identity
  virtual void init_data() {}                                   // called by EquationSystems::init() after the vectors exist
 protected:
  EquationSystems& es_;
  std::string name_;
  std::vector<std::string> vars_;
};

using ExplicitSystem = System;

class TransientExplicitSystem : public System {   // "SolidSystem::auxiliary" (src/solid.C:34-37)
 public:
  using System::System;
  NumericVector old_local_solution, older_local_solution;
  void init(int64_t n) override {
    System::init(n);
    old_local_solution.init(n * n_vars());
    older_local_solution.init(n * n_vars());
  }
};

// a system with a matrix and a right-hand side: EquationSystems::init() gives it an assembly context and the pattern
class ImplicitSystem : public System {
 public:
  using System::System;
  NumericVector rhs_storage;
  SparseMatrix matrix_storage;
  NumericVector* rhs = &rhs_storage;
  SparseMatrix* matrix = &matrix_storage;
  void init(int64_t n_nodes) override {
    System::init(n_nodes);
    rhs_storage.init(n_nodes * n_vars());
  }
};

// BiCGStab preconditioned with ILU(0) of the assembled matrix: the stand-in for "linear_solver->solve(matrix,
// solution, rhs)" (the reference hands the system to PETSc KSP -- GMRES + ILU(0) by default --, which stays on the
// host and is out of this project's scope).  Returns the iteration count.
inline int bicgstab_ilu0(const SparseMatrix& A, const std::vector<double>& b, std::vector<double>& x, Real tol, int max_its);

class TransientLinearImplicitSystem : public ImplicitSystem {
 public:
  using AssembleFn = void (*)(EquationSystems&, const std::string&);
  using ImplicitSystem::ImplicitSystem;
  NumericVector old_local_solution, older_local_solution;
  Real time = 0.0;
  std::vector<int64_t> handback_log;   // node ranges [n0, n1) in the order the chunked hand-back delivered them ("rdc/handback_chunks")
  void attach_assemble_function(AssembleFn f) { assemble_fn_ = f; }
  Number old_solution(dof_id_type dof) const { return old_local_solution((int64_t)dof); }
  SparseMatrix& get_system_matrix() { return matrix_storage; }
  void init(int64_t n_nodes) override {
    ImplicitSystem::init(n_nodes);
    old_local_solution.init(n_nodes * n_vars());
    older_local_solution.init(n_nodes * n_vars());
  }
  // ImplicitSystem::assemble(): zero matrix and rhs, then System::user_assembly() -> the callback
  void assemble() {
    if (!assemble_fn_) throw std::runtime_error("no assemble function attached to system " + name_);
    matrix_storage.zero();
    rhs_storage.zero();
    assemble_fn_(es_, name_);
  }
  // LinearImplicitSystem::solve(): assemble, linear solve (stand-in: the reference hands the system
  // to PETSc KSP, which stays on the host and is out of this project's scope), update()
  int solve(Real tol = 1e-12, int max_its = 5000);
 private:
  AssembleFn assemble_fn_ = nullptr;
};

// ---- EquationSystems ----------------------------------------------------------------------------
class EquationSystems {
 public:
  explicit EquationSystems(Mesh& mesh, int device = 0) : mesh_(mesh), device_(device) {}
  ~EquationSystems() { for (auto& kv : ctx_) rdc_ctx_destroy(kv.second); }
  EquationSystems(const EquationSystems&) = delete;
  Parameters parameters;
  Mesh& get_mesh() { return mesh_; }
  template <class T> T& add_system(const std::string& name) {
    auto p = std::make_unique<T>(*this, name);
    T& ref = *p;
    systems_[name] = std::move(p);
    return ref;
  }
  template <class T> T& get_system(const std::string& name) {
    auto it = systems_.find(name);
    if (it == systems_.end()) throw std::runtime_error("EquationSystems::get_system: no system '" + name + "'");
    T* p = dynamic_cast<T*>(it->second.get());
    if (!p) throw std::runtime_error("EquationSystems::get_system: wrong type for '" + name + "'");
    return *p;
  }
  bool has_system(const std::string& name) const { return systems_.count(name) != 0; }
  // es.init(): DoF numbering, sparsity pattern, vector allocation (src/pihna.C:48)
  void init() {
    for (auto& kv : systems_) {
      kv.second->init(kv.second->elemental ? mesh_.n_elem() : mesh_.n_nodes());
      if (auto* t = dynamic_cast<ImplicitSystem*>(kv.second.get())) {
        rdc_ctx* c = context(t->name(), (int)t->n_vars());
        int64_t n_rows = 0, nnz = 0;
        check(c, rdc_csr_dims(c, &n_rows, &nnz), "rdc_csr_dims");
        t->matrix_storage.row_ptr.resize((size_t)n_rows + 1);
        t->matrix_storage.col_idx.resize((size_t)nnz);
        t->matrix_storage.val.assign((size_t)nnz, 0.0);
        check(c, rdc_csr_pattern_download(c, t->matrix_storage.row_ptr.data(), t->matrix_storage.col_idx.data()), "pattern");
      }
    }
    for (auto& kv : systems_) kv.second->init_data();
  }
  void reinit() {}   // es.reinit() (src/solid_system.C:391): nothing to redistribute on a fixed serial mesh
  // one assembly context (GPU-resident mesh + pattern) per implicit system
  rdc_ctx* context(const std::string& system, int nvar) {
    auto it = ctx_.find(system);
    if (it != ctx_.end()) return it->second;
    rdc_ctx* c = nullptr;
    int rc = rdc_ctx_create(device_, &c);
    if (rc != RDC_OK) throw std::runtime_error(std::string("rdc_ctx_create: ") + rdc_last_error(nullptr));
    ctx_[system] = c;
    check(c, rdc_mesh_upload(c, mesh_.elem_type(), mesh_.n_elem(), mesh_.n_nodes(), mesh_.n_nodes(),
                             mesh_.connectivity().data(), mesh_.coordinates().data(), nvar), "rdc_mesh_upload");
    return c;
  }
 private:
  Mesh& mesh_;
  int device_;
  std::map<std::string, std::unique_ptr<System>> systems_;
  std::map<std::string, rdc_ctx*> ctx_;
};

inline int bicgstab_ilu0(const SparseMatrix& A, const std::vector<double>& b, std::vector<double>& x, Real tol, int max_its) {
  const size_t n = b.size();
  // ILU(0) on the pattern of A (columns ascending within a row, diagonal present)
  std::vector<double> lu = A.val;
  std::vector<int64_t> diag(n, -1);
  for (size_t i = 0; i < n; i++)
    for (int64_t k = A.row_ptr[i]; k < A.row_ptr[i + 1]; k++) if ((size_t)A.col_idx[(size_t)k] == i) diag[i] = k;
  for (size_t i = 0; i < n; i++) if (diag[i] < 0) throw std::runtime_error("bicgstab_ilu0: missing diagonal entry");
  {
    std::vector<int64_t> pos(n, -1);
    for (size_t i = 0; i < n; i++) {
      for (int64_t k = A.row_ptr[i]; k < A.row_ptr[i + 1]; k++) pos[(size_t)A.col_idx[(size_t)k]] = k;
      for (int64_t k = A.row_ptr[i]; k < diag[i]; k++) {
        const size_t j = (size_t)A.col_idx[(size_t)k];
        const double piv = lu[(size_t)diag[j]];
        if (piv == 0.0) throw std::runtime_error("bicgstab_ilu0: zero pivot");
        const double l = lu[(size_t)k] / piv;
        lu[(size_t)k] = l;
        for (int64_t m = diag[j] + 1; m < A.row_ptr[j + 1]; m++) {
          const int64_t q = pos[(size_t)A.col_idx[(size_t)m]];
          if (q >= 0) lu[(size_t)q] -= l * lu[(size_t)m];
        }
      }
      for (int64_t k = A.row_ptr[i]; k < A.row_ptr[i + 1]; k++) pos[(size_t)A.col_idx[(size_t)k]] = -1;
    }
  }
  auto precond = [&](const std::vector<double>& r, std::vector<double>& z) {
    z = r;
    for (size_t i = 0; i < n; i++) {
      double s = z[i];
      for (int64_t k = A.row_ptr[i]; k < diag[i]; k++) s -= lu[(size_t)k] * z[(size_t)A.col_idx[(size_t)k]];
      z[i] = s;
    }
    for (size_t i = n; i-- > 0;) {
      double s = z[i];
      for (int64_t k = diag[i] + 1; k < A.row_ptr[i + 1]; k++) s -= lu[(size_t)k] * z[(size_t)A.col_idx[(size_t)k]];
      z[i] = s / lu[(size_t)diag[i]];
    }
  };
  std::vector<double> r(n), r0(n), p(n, 0.0), v(n, 0.0), s(n), t(n), y(n), z(n), tmp;
  A.vector_mult(tmp, x);
  double bn = 0.0;
  for (size_t i = 0; i < n; i++) { r[i] = b[i] - tmp[i]; r0[i] = r[i]; bn += b[i] * b[i]; }
  bn = std::sqrt(bn) > 0 ? std::sqrt(bn) : 1.0;
  double rho = 1, alpha = 1, omega = 1;
  int it = 0;
  for (; it < max_its; it++) {
    double rn = 0, rho1 = 0;
    for (size_t i = 0; i < n; i++) { rn += r[i] * r[i]; rho1 += r0[i] * r[i]; }
    if (std::sqrt(rn) <= tol * bn) break;
    if (rho1 == 0.0) {  // breakdown: restart with the current residual as the shadow vector
      r0 = r;
      rho1 = rn;
      std::fill(p.begin(), p.end(), 0.0);
      std::fill(v.begin(), v.end(), 0.0);
      rho = alpha = omega = 1;
    }
    const double beta = (rho1 / rho) * (alpha / omega);
    rho = rho1;
    for (size_t i = 0; i < n; i++) p[i] = r[i] + beta * (p[i] - omega * v[i]);
    precond(p, y);
    A.vector_mult(v, y);
    double r0v = 0;
    for (size_t i = 0; i < n; i++) r0v += r0[i] * v[i];
    alpha = rho / r0v;
    for (size_t i = 0; i < n; i++) s[i] = r[i] - alpha * v[i];
    precond(s, z);
    A.vector_mult(t, z);
    double ts = 0, tt = 0;
    for (size_t i = 0; i < n; i++) { ts += t[i] * s[i]; tt += t[i] * t[i]; }
    omega = tt > 0 ? ts / tt : 0.0;
    for (size_t i = 0; i < n; i++) { x[i] += alpha * y[i] + omega * z[i]; r[i] = s[i] - omega * t[i]; }
    if (omega == 0.0) break;
  }
  return it;
}

inline int TransientLinearImplicitSystem::solve(Real tol, int max_its) {
  assemble();
  const int it = bicgstab_ilu0(matrix_storage, rhs_storage.raw(), solution.raw(), tol, max_its);
  update();
  return it;
}

// ---- the drop-in callbacks: same signature as the reference's static assemble_* -----------------
namespace detail {
// Hand-back of the assembled owned rows (what add_matrix / add_vector did element by element, src/pihna.C:754-755).
// "rdc/handback_chunks" > 1 (es.parameters, int) selects the pipelined form the libMesh adapter uses (push_results_chunked in
// integration/libmesh_adapter.C): the node range is cut into chunks, two chunk downloads are kept in flight on the context's
// copy stream, and every chunk is consumed (there: one MatSetValues per node block; here: `consume`) while the next travels.
template <class Consume>
inline void pull_results_chunked(rdc_ctx* c, TransientLinearImplicitSystem& sys, int n_chunks, Consume&& consume) {
  std::vector<double>& val = sys.matrix_storage.val;
  std::vector<double>& rhs = sys.rhs_storage.raw();
  const int64_t nv = (int64_t)sys.n_vars();
  const int64_t n_nodes = (int64_t)rhs.size() / nv;
  check(c, rdc_host_pin(c, val.data(), val.size() * sizeof(double)), "rdc_host_pin");
  check(c, rdc_host_pin(c, rhs.data(), rhs.size() * sizeof(double)), "rdc_host_pin");
  auto bound = [&](int k) { return n_nodes * k / n_chunks; };
  int ticket[2] = {-1, -1};
  check(c, rdc_csr_download_rows_async(c, bound(0), bound(1), val.data(), rhs.data(), &ticket[0]), "rdc_csr_download_rows_async");
  for (int k = 0; k < n_chunks; k++) {
    if (k + 1 < n_chunks)
      check(c, rdc_csr_download_rows_async(c, bound(k + 1), bound(k + 2), val.data(), rhs.data(), &ticket[(k + 1) & 1]), "rdc_csr_download_rows_async");
    check(c, rdc_ticket_wait(c, ticket[k & 1]), "rdc_ticket_wait");
    consume(bound(k), bound(k + 1));     // rows of nodes [bound(k), bound(k + 1)) are in host memory
  }
  check(c, rdc_host_unpin(c, val.data()), "rdc_host_unpin");
  check(c, rdc_host_unpin(c, rhs.data()), "rdc_host_unpin");
}
inline void pull_results(rdc_ctx* c, TransientLinearImplicitSystem& sys) {
  const Parameters& P = sys.get_equation_systems().parameters;
  const int chunks = P.have_parameter<int>("rdc/handback_chunks") ? P.get<int>("rdc/handback_chunks") : 1;
  if (chunks > 1) {
    // the mirror's matrix IS the destination array: the consumer only records which rows it was handed, in which order
    std::vector<int64_t>& log = sys.handback_log;
    log.clear();
    pull_results_chunked(c, sys, chunks, [&](int64_t n0, int64_t n1) { log.push_back(n0); log.push_back(n1); });
    return;
  }
  check(c, rdc_csr_download(c, sys.matrix_storage.val.data(), sys.rhs_storage.raw().data()), "rdc_csr_download");
}
inline Real getR(const Parameters& p, const char* k) { return p.get<Real>(k); }
}  // namespace detail

// src/pihna.C:318-758
inline void assemble_pihna(EquationSystems& es, const std::string& system_name) {
  auto& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  if (system.n_vars() != 5) throw std::runtime_error("assemble_pihna: system must have 5 variables (n,c,h,v,a)");
  const Parameters& P = es.parameters;
  rdc_pihna_params p;
  p.time_step = detail::getR(P, "time_step");
  p.cells_min_capacity = detail::getR(P, "cells_min_capacity");
  p.cells_max_capacity = detail::getR(P, "cells_max_capacity");
  p.cytokines_max_capacity = detail::getR(P, "cytokines_max_capacity");
  p.cells_max_capacity_exponent = detail::getR(P, "cells_max_capacity/exponent");
  p.necrosis_c = detail::getR(P, "necrosis/c"); p.necrosis_h = detail::getR(P, "necrosis/h"); p.necrosis_v = detail::getR(P, "necrosis/v");
  p.diffuse_c = detail::getR(P, "diffuse/c"); p.taxis_c = detail::getR(P, "taxis/c");
  p.diffuse_h = detail::getR(P, "diffuse/h"); p.taxis_h = detail::getR(P, "taxis/h");
  p.produce_c = detail::getR(P, "produce/c");
  p.switch_c2h = detail::getR(P, "switch/c/to/h"); p.switch_h2c = detail::getR(P, "switch/h/to/c"); p.switch_h2n = detail::getR(P, "switch/h/to/n");
  p.diffuse_v = detail::getR(P, "diffuse/v"); p.taxis_v = detail::getR(P, "taxis/v"); p.produce_v = detail::getR(P, "produce/v");
  p.secrete_a_c = detail::getR(P, "secrete/a/from/c"); p.secrete_a_h = detail::getR(P, "secrete/a/from/h");
  p.uptake_a_v = detail::getR(P, "uptake/a/from/v"); p.decay_a = detail::getR(P, "decay/a");
  rdc_ctx* c = es.context(system_name, 5);
  check(c, rdc_field_upload(c, RDC_FIELD_OLD_SOLUTION, system.old_local_solution.raw().data(), system.old_local_solution.size()), "old solution");
  check(c, rdc_assemble_pihna(c, &p), "rdc_assemble_pihna");
  detail::pull_results(c, system);
}

// src/ripf.C:337-673: reads the "RIPF-TimeDeriv" (vars 1,2) and "RT" (var 2) systems too
inline void assemble_ripf(EquationSystems& es, const std::string& system_name) {
  auto& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  if (system.n_vars() != 3) throw std::runtime_error("assemble_ripf: system must have 3 variables (HU,cc,fb)");
  System& TD = es.get_system<System>("RIPF-TimeDeriv");
  System& RT = es.get_system<System>("RT");
  const Parameters& P = es.parameters;
  rdc_ripf_params p{};
  p.time_step = detail::getR(P, "time_step");
  p.VolFr_stroma = detail::getR(P, "volume_fraction/stroma"); p.VolFr_parenchyma = detail::getR(P, "volume_fraction/parenchyma");
  p.VolFr_exponent = detail::getR(P, "volume_fraction/exponent"); p.VolFr_min_vacant = detail::getR(P, "volume_fraction/min_vacant");
  p.VolFr_max_vacant = detail::getR(P, "volume_fraction/max_vacant");
  p.phi_cc_B = detail::getR(P, "HU/phi/cc/build"); p.phi_cc_D = detail::getR(P, "HU/phi/cc/decay"); p.phi_cc = detail::getR(P, "HU/phi/cc/rate");
  p.phi_fb_B = detail::getR(P, "HU/phi/fb/build"); p.phi_fb_D = detail::getR(P, "HU/phi/fb/decay"); p.phi_fb = detail::getR(P, "HU/phi/fb/rate");
  p.phi_tol = detail::getR(P, "HU/phi/tolerance");
  p.kappa = detail::getR(P, "cc/kappa"); p.kappa_RT_c = detail::getR(P, "cc/kappa/RT/c");
  p.delta = detail::getR(P, "cc/delta"); p.delta_RT_a = detail::getR(P, "cc/delta/RT/a"); p.delta_RT_b = detail::getR(P, "cc/delta/RT/b");
  p.lambda = detail::getR(P, "fb/lambda"); p.lambda_RT_r = detail::getR(P, "fb/lambda/RT/r"); p.lambda_HU_r = detail::getR(P, "fb/lambda/HU/r");
  p.omicro = detail::getR(P, "fb/omicro"); p.omicro_RT_r = detail::getR(P, "fb/omicro/RT/r"); p.omicro_fb_b = detail::getR(P, "fb/omicro/fb/b");
  p.omega = detail::getR(P, "fb/omega"); p.diffusion = detail::getR(P, "fb/diffusion");
  p.haptotaxis = detail::getR(P, "fb/haptotaxis"); p.radiotaxis = detail::getR(P, "fb/radiotaxis");
  p.RT_dose_total_max = P.get<int>("RT_dose/total/max");
  const int64_t nn = es.get_mesh().n_nodes();
  std::vector<double> aux((size_t)nn * 3);
  for (int64_t n = 0; n < nn; n++) {
    aux[(size_t)n * 3 + 0] = TD.current_solution((dof_id_type)(n * 3 + 1));  // cc__dtime, src/ripf.C:470
    aux[(size_t)n * 3 + 1] = TD.current_solution((dof_id_type)(n * 3 + 2));  // fb__dtime, :471
    aux[(size_t)n * 3 + 2] = RT.current_solution((dof_id_type)(n * 3 + 2));  // RT_dose/total, :477
  }
  rdc_ctx* c = es.context(system_name, 3);
  check(c, rdc_field_upload(c, RDC_FIELD_OLD_SOLUTION, system.old_local_solution.raw().data(), system.old_local_solution.size()), "old solution");
  check(c, rdc_field_upload(c, RDC_FIELD_AUX_NODAL, aux.data(), (int64_t)aux.size()), "aux fields");
  check(c, rdc_assemble_ripf(c, &p), "rdc_assemble_ripf");
  detail::pull_results(c, system);
}

// src/coupled_hcc.C:414-649: assembled on the CURRENT node positions of the (moving) mesh
inline void assemble_hcc(EquationSystems& es, const std::string& system_name) {
  auto& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  if (system.n_vars() != 3) throw std::runtime_error("assemble_hcc: system must have 3 variables (l,c,n)");
  const Parameters& P = es.parameters;
  rdc_hcc_params p{};
  p.time_step = detail::getR(P, "time_step");
  p.cells_min_capacity = detail::getR(P, "cells/min_capacity"); p.cells_max_capacity = detail::getR(P, "cells/max_capacity");
  p.cells_max_capacity_exponent = detail::getR(P, "cells/max_capacity/exponent");
  p.produce_l = detail::getR(P, "produce/l");
  p.diffuse_c = detail::getR(P, "diffuse/c"); p.mechano_c = detail::getR(P, "mechano/c"); p.produce_c = detail::getR(P, "produce/c");
  p.necrosis_l = detail::getR(P, "necrosis/l"); p.necrosis_c = detail::getR(P, "necrosis/c"); p.necrosis_pressure = detail::getR(P, "necrosis/pressure");
  rdc_ctx* c = es.context(system_name, 3);
  check(c, rdc_mesh_update_coords(c, es.get_mesh().coordinates().data()), "rdc_mesh_update_coords");
  check(c, rdc_field_upload(c, RDC_FIELD_OLD_SOLUTION, system.old_local_solution.raw().data(), system.old_local_solution.size()), "old solution");
  check(c, rdc_assemble_hcc(c, &p), "rdc_assemble_hcc");
  detail::pull_results(c, system);
}

// src/adpm.C:324-652: reads the elemental "Tracts" system (3 variables per element, :448-453) and system.time
inline void assemble_adpm(EquationSystems& es, const std::string& system_name) {
  auto& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  if (system.n_vars() != 3) throw std::runtime_error("assemble_adpm: system must have 3 variables (PrP,A_b,Tau)");
  System& tracts = es.get_system<System>("Tracts");
  const Parameters& P = es.parameters;
  rdc_adpm_params p{};
  p.time_step = detail::getR(P, "time_step");
  p.time = system.time;
  p.decay_PrP_time_exponent = detail::getR(P, "decay/PrP/time_exponent");
  auto triple = [&](double* dst, const std::string& key, const char* kind) {
    dst[0] = detail::getR(P, key.c_str());
    dst[1] = detail::getR(P, (key + "/" + kind + "/0").c_str());
    dst[2] = detail::getR(P, (key + "/" + kind + "/1").c_str());
  };
  auto trapezoid = [&](double* dst, const std::string& key) {
    dst[0] = detail::getR(P, key.c_str());
    for (int i = 0; i < 4; i++) dst[1 + i] = detail::getR(P, (key + "/trapezoid/" + std::to_string(i)).c_str());
  };
  triple(p.decay_PrP, "decay/PrP", "pulse");
  trapezoid(p.transform_A_b, "transform/A_b"); trapezoid(p.transform_Tau, "transform/Tau");
  triple(p.diffuse_A_b, "diffuse/A_b", "pulse"); triple(p.taxis1_A_b, "taxis_1/A_b", "pulse"); triple(p.taxis2_A_b, "taxis_2/A_b", "pulse");
  triple(p.produce_A_b, "produce/A_b", "sigmoid"); triple(p.decay_A_b, "decay/A_b", "pulse");
  triple(p.diffuse_Tau, "diffuse/Tau", "pulse"); triple(p.taxis1_Tau, "taxis_1/Tau", "pulse"); triple(p.taxis2_Tau, "taxis_2/Tau", "pulse");
  triple(p.produce_Tau, "produce/Tau", "sigmoid"); triple(p.decay_Tau, "decay/Tau", "pulse");
  p.taxis_A_b_angle = detail::getR(P, "taxis/A_b/angle");  // radians: input() stores degrees_to_radians(...), src/adpm.C:193
  p.taxis_Tau_angle = detail::getR(P, "taxis/Tau/angle");
  rdc_ctx* c = es.context(system_name, 3);
  check(c, rdc_field_upload(c, RDC_FIELD_OLD_SOLUTION, system.old_local_solution.raw().data(), system.old_local_solution.size()), "old solution");
  check(c, rdc_field_upload(c, RDC_FIELD_ELEM_TRACTS, tracts.solution.raw().data(), tracts.solution.size()), "tracts");
  check(c, rdc_assemble_adpm(c, &p), "rdc_assemble_adpm");
  detail::pull_results(c, system);
}

// src/proteas.C:338-705: reads the nodal "AUX" system (HU, RTD)
inline void assemble_proteas_model(EquationSystems& es, const std::string& system_name) {
  auto& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  if (system.n_vars() != 5) throw std::runtime_error("assemble_proteas_model: system must have 5 variables");
  System& AUX = es.get_system<System>("AUX");
  const Parameters& P = es.parameters;
  rdc_proteas_params p{};
  p.time_step = detail::getR(P, "time_step");
  p.cells_total_capacity = detail::getR(P, "cells/total_capacity"); p.RT_max_dosage = detail::getR(P, "radiotherapy/max_dosage");
  p.host_proliferation = detail::getR(P, "host/proliferation"); p.host_vsc_threshold = detail::getR(P, "host/vsc_threshold");
  p.host_RT_death_rate = detail::getR(P, "host/RT_death_rate"); p.host_RT_exp_a = detail::getR(P, "host/RT_exp_a");
  p.host_RT_exp_b = detail::getR(P, "host/RT_exp_b"); p.host_necrosis_rate = detail::getR(P, "host/necrosis_rate");
  p.tumour_diffusion = detail::getR(P, "tumour/diffusion"); p.tumour_diffusion_host = detail::getR(P, "tumour/diffusion_host");
  p.tumour_proliferation = detail::getR(P, "tumour/proliferation"); p.tumour_vsc_threshold = detail::getR(P, "tumour/vsc_threshold");
  p.tumour_RT_death_rate = detail::getR(P, "tumour/RT_death_rate"); p.tumour_RT_exp_a = detail::getR(P, "tumour/RT_exp_a");
  p.tumour_RT_exp_b = detail::getR(P, "tumour/RT_exp_b"); p.tumour_necrosis_rate = detail::getR(P, "tumour/necrosis_rate");
  p.necrosis_clearance = detail::getR(P, "necrosis/clearance"); p.necrosis_slope = detail::getR(P, "necrosis/slope");
  p.necrosis_vsc_threshold = detail::getR(P, "necrosis/vsc_threshold");
  p.vascular_proliferation = detail::getR(P, "vascular/proliferation"); p.vascular_necrosis_rate = detail::getR(P, "vascular/necrosis_rate");
  p.oedema_diffusion = detail::getR(P, "oedema/diffusion"); p.oedema_proliferation = detail::getR(P, "oedema/proliferation");
  p.oedema_vsc_threshold = detail::getR(P, "oedema/vsc_threshold"); p.oedema_RT_coeff = detail::getR(P, "oedema/RT_coeff");
  p.oedema_RT_exp = detail::getR(P, "oedema/RT_exp"); p.oedema_reabsorption_rate = detail::getR(P, "oedema/reabsorption_rate");
  const int64_t nn = es.get_mesh().n_nodes();
  std::vector<double> aux((size_t)nn * 3, 0.0);
  for (int64_t n = 0; n < nn; n++) {
    aux[(size_t)n * 3 + 0] = AUX.current_solution((dof_id_type)(n * 2 + 0));  // "HU": the only AUX variable the assembly reads (:472,481)
    aux[(size_t)n * 3 + 1] = AUX.current_solution((dof_id_type)(n * 2 + 1));  // "RTD"
  }
  rdc_ctx* c = es.context(system_name, 5);
  check(c, rdc_field_upload(c, RDC_FIELD_OLD_SOLUTION, system.old_local_solution.raw().data(), system.old_local_solution.size()), "old solution");
  check(c, rdc_field_upload(c, RDC_FIELD_AUX_NODAL, aux.data(), (int64_t)aux.size()), "aux fields");
  check(c, rdc_assemble_proteas(c, &p), "rdc_assemble_proteas");
  detail::pull_results(c, system);
}

// check_solution of PIHNA / HCC / ADPM-style models: negativity clamp of the freshly solved state
// (src/pihna.C:760-803, src/coupled_hcc.C:695-731, src/proteas.C:712-750), on the device
inline void check_solution(EquationSystems& es, const std::string& system_name) {
  auto& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  rdc_ctx* c = es.context(system_name, (int)system.n_vars());
  check(c, rdc_field_upload(c, RDC_FIELD_OLD_SOLUTION, system.solution.raw().data(), system.solution.size()), "solution");
  check(c, rdc_clamp_nonnegative(c, RDC_FIELD_OLD_SOLUTION), "rdc_clamp_nonnegative");
  check(c, rdc_field_download(c, RDC_FIELD_OLD_SOLUTION, system.solution.raw().data(), system.solution.size()), "solution");
  system.update();
}

// save_solution of PIHNA (src/pihna.C:842-976): one CSV line per call -- time, number of dofs and the four
// thresholded element-volume sums, which the device computes from the current solution; header at time 0
inline void save_solution_pihna(std::ostream& csv, EquationSystems& es) {
  auto& system = es.get_system<TransientLinearImplicitSystem>("PIHNA");
  const Parameters& P = es.parameters;
  rdc_pihna_ranges r;
  r.active_tumor_min = detail::getR(P, "range/active_tumor/min"); r.active_tumor_max = detail::getR(P, "range/active_tumor/max");
  r.necrotic_min = detail::getR(P, "range/necrotic/min"); r.necrotic_max = detail::getR(P, "range/necrotic/max");
  r.vascularity_min = detail::getR(P, "range/vascularity/min"); r.vascularity_max = detail::getR(P, "range/vascularity/max");
  r.total_cell_min = detail::getR(P, "range/total_cell/min"); r.total_cell_max = detail::getR(P, "range/total_cell/max");
  r.cells_max_capacity = detail::getR(P, "cells_max_capacity");
  rdc_ctx* c = es.context("PIHNA", 5);
  double v[4];
  check(c, rdc_field_upload(c, RDC_FIELD_OLD_SOLUTION, system.solution.raw().data(), system.solution.size()), "solution");
  check(c, rdc_pihna_volume_integrals(c, &r, -1, v), "rdc_pihna_volume_integrals");
  if (0.0 == system.time)
    csv << "\"TIME\",\"DEGREES_OF_FREEDOM\",\"ACTIVE_TUMOR_VOLUME\",\"NECROTIC_VOLUME\",\"VASCULARITY_VOLUME\",\"TOTAL_CELL_VOLUME\"" << std::endl;
  csv << system.time << ',' << (system.n_vars() * es.get_mesh().n_nodes()) << ',' << v[0] << ',' << v[1] << ',' << v[2] << ',' << v[3] << std::endl;
}

// save_solution of RIPF (src/ripf.C:777-866): time, tumour volume, fibrosis volume (upstream writes no header)
inline void save_solution_ripf(std::ostream& csv, EquationSystems& es) {
  auto& system = es.get_system<TransientLinearImplicitSystem>("RIPF");
  const Parameters& P = es.parameters;
  rdc_ripf_ranges r;
  r.cc_HU_min = detail::getR(P, "range_cc/HU/min"); r.cc_HU_max = detail::getR(P, "range_cc/HU/max"); r.cc_min = detail::getR(P, "range_cc/min");
  r.fb_HU_min = detail::getR(P, "range_fb/HU/min"); r.fb_HU_max = detail::getR(P, "range_fb/HU/max"); r.fb_min = detail::getR(P, "range_fb/min");
  rdc_ctx* c = es.context("RIPF", 3);
  double v[2];
  check(c, rdc_field_upload(c, RDC_FIELD_OLD_SOLUTION, system.solution.raw().data(), system.solution.size()), "solution");
  check(c, rdc_ripf_volume_integrals(c, &r, -1, v), "rdc_ripf_volume_integrals");
  csv << system.time << ',' << v[0] << ',' << v[1] << std::endl;
}


// ---- SolidSystem: the FEMSystem subclass of src/solid_system.h:30-84 -----------------------------------------
// The reference implements the per-element virtuals element_time_derivative / side_time_derivative
// (src/solid_system.C:146-371) and lets libMesh's FEMSystem::assembly() loop over the elements and sides and add
// their results into the global Jacobian and residual, once per Newton iteration.  The GPU drop-in replaces that
// WHOLE loop: assembly(get_residual, get_jacobian) marshals what the virtuals read -- the current node positions
// (the unknowns; the mesh IS the solution, :103-123), the undeformed positions of "SolidSystem::auxiliary"
// (:221-229), the reference fibre of "SolidSystem::fibre" (:204-216), the material of each subdomain (:183-190), the
// boundary sides named by "BCs" with their prescribed displacement (:294-306), "pseudo_time",
// "BCs/displacement_penalty", "solver/assembly_use_symmetry" -- and makes ONE call, rdc_solid_assemble.
// solve() is a plain Newton iteration with the options init_data() reads (:84-100), standing in for libMesh's
// NewtonSolver + PETSc; everything else keeps the reference's names: init_data, save_initial_mesh, update,
// run_solver, post_process, update_data, var[], undefo_var[], deltat.
class SolidSystem : public ImplicitSystem {
 public:
  using ImplicitSystem::ImplicitSystem;
  unsigned int var[3] = {0, 1, 2}, undefo_var[3] = {0, 1, 2};
  Real deltat = 0.0;
  // DiffSolver options (src/solid_system.C:84-100)
  bool quiet = true, require_residual_reduction = false;
  int max_nonlinear_iterations = 10, max_linear_iterations = 50000;
  Real relative_step_tolerance = 1e-3, relative_residual_tolerance = 1e-8, absolute_residual_tolerance = 1e-8,
       initial_linear_tolerance = 1e-3;
  // what the last solve() did (the reference prints this through NewtonSolver's verbose mode)
  struct NewtonLog { int nonlinear_iterations = 0, linear_iterations = 0, assemblies = 0; Real first_residual = 0, last_residual = 0; bool converged = false; };
  NewtonLog last;

  std::string system_type() const { return "SolidSystem"; }

  void init_data() override;                                   // src/solid_system.C:49-101
  void save_initial_mesh();                                    // :26-47
  void update() override;                                      // :103-123
  void assembly(bool get_residual, bool get_jacobian);         // FEMSystem::assembly -> the :146-371 virtuals
  void solve();                                                // FEMSystem::solve() -> SteadySolver -> NewtonSolver
  void run_solver();                                           // :373-392
  void post_process();                                         // :394-538
  void update_data();                                          // :540-557
  void rebind() { bound_ = false; }                            // materials / BC sides are re-read from es.parameters

 private:
  bool bound_ = false;
  void bind(rdc_ctx* c);
};

inline void SolidSystem::init_data() {
  Parameters& P = es_.parameters;
  var[0] = variable_number("x"); var[1] = variable_number("y"); var[2] = variable_number("z");
  System& aux = es_.get_system<System>("SolidSystem::auxiliary");
  undefo_var[0] = aux.variable_number("undeformed_x");
  undefo_var[1] = aux.variable_number("undeformed_y");
  undefo_var[2] = aux.variable_number("undeformed_z");
  deltat = P.get<Real>("loading_step");
  if (!P.have_parameter<Real>("BCs/displacement_penalty")) P.set<Real>("BCs/displacement_penalty") = 1.0e+5;
  // mesh_position_get(): the unknowns start as the node positions
  solution.raw() = es_.get_mesh().coordinates();
  current_local_solution = solution;
  quiet = P.get<bool>("solver/quiet");
  max_nonlinear_iterations = P.get<int>("solver/nonlinear/max_nonlinear_iterations");
  relative_step_tolerance = P.get<Real>("solver/nonlinear/relative_step_tolerance");
  relative_residual_tolerance = P.get<Real>("solver/nonlinear/relative_residual_tolerance");
  absolute_residual_tolerance = P.get<Real>("solver/nonlinear/absolute_residual_tolerance");
  if (!P.have_parameter<bool>("solver/assembly_use_symmetry")) P.set<bool>("solver/assembly_use_symmetry") = false;
  require_residual_reduction = P.get<bool>("solver/nonlinear/require_reduction");
  max_linear_iterations = P.get<int>("solver/linear/max_linear_iterations");
  initial_linear_tolerance = P.get<Real>("solver/linear/initial_linear_tolerance");
}

inline void SolidSystem::save_initial_mesh() {
  System& aux = es_.get_system<System>("SolidSystem::auxiliary");
  aux.current_local_solution = current_local_solution;  // same (node, component) numbering in both systems
  aux.solution = aux.current_local_solution;
}

inline void SolidSystem::update() {
  System::update();
  es_.get_mesh().coordinates() = current_local_solution.raw();   // mesh_position_set()
  System& disp = es_.get_system<System>("SolidSystem::displacement");
  System& aux = es_.get_system<System>("SolidSystem::auxiliary");
  disp.current_local_solution = current_local_solution;
  for (size_t i = 0; i < disp.current_local_solution.raw().size(); i++)
    disp.current_local_solution.raw()[i] -= aux.current_local_solution.raw()[i];
  disp.solution = disp.current_local_solution;
}

// one-time marshalling of what does not change between Newton iterations
inline void SolidSystem::bind(rdc_ctx* c) {
  const Parameters& P = es_.parameters;
  const Mesh& mesh = es_.get_mesh();
  // materials: one table entry per subdomain id present in the mesh (string-keyed lookups per element upstream)
  std::map<int32_t, int32_t> index;
  std::vector<rdc_solid_material> table;
  std::vector<int32_t> elem_material((size_t)mesh.n_elem());
  for (int64_t e = 0; e < mesh.n_elem(); e++) {
    const int32_t id = mesh.subdomain_id(e);
    auto it = index.find(id);
    if (it == index.end()) {
      const std::string k = "material/" + std::to_string(id) + "/Hyperelastic/";
      rdc_solid_material m;
      m.Young = P.get<Real>(k + "Young"); m.Poisson = P.get<Real>(k + "Poisson"); m.FibreStiffness = P.get<Real>(k + "FibreStiffness");
      for (int d = 0; d < 3; d++) m.rate[d] = P.get<Real>(k + "VolumetricStretchRatio/rate_" + std::to_string(d));
      it = index.emplace(id, (int32_t)table.size()).first;
      table.push_back(m);
    }
    elem_material[(size_t)e] = it->second;
  }
  check(c, rdc_solid_set_materials(c, elem_material.data(), (int32_t)table.size(), table.data()), "rdc_solid_set_materials");
  // sides: every boundary side whose id is in "BCs", with "BC/<id>/displacement" (NaN component = unconstrained)
  const std::set<int> bcs = export_integers(P.get<std::string>("BCs"));
  std::vector<int64_t> se;
  std::vector<int32_t> si;
  std::vector<double> sd;
  for (int bc : bcs) {   // ascending id, then mesh order: the order FEMSystem::assembly meets them does not matter for a sum
    const Point& u = P.get<Point>("BC/" + std::to_string(bc) + "/displacement");
    for (const Mesh::BoundarySide& s : mesh.boundary_sides()) {
      if (s.id != bc) continue;
      se.push_back(s.elem); si.push_back(s.side);
      for (int d = 0; d < 3; d++) sd.push_back(u(d));
    }
  }
  check(c, rdc_solid_set_sides(c, (int64_t)se.size(), se.data(), si.data(), sd.data()), "rdc_solid_set_sides");
  bound_ = true;
}

inline void SolidSystem::assembly(bool get_residual, bool get_jacobian) {
  (void)get_residual;   // the residual always comes with the call, as in element_time_derivative
  const Parameters& P = es_.parameters;
  rdc_ctx* c = es_.context(name_, 3);
  if (!bound_) bind(c);
  System& aux = es_.get_system<System>("SolidSystem::auxiliary");
  System& fibre = es_.get_system<System>("SolidSystem::fibre");
  // FEMContext::pre_fe_reinit moves the element's nodes to the current iterate: current coordinates = unknowns
  check(c, rdc_mesh_update_coords(c, current_local_solution.raw().data()), "rdc_mesh_update_coords");
  check(c, rdc_field_upload(c, RDC_FIELD_UNDEFORMED_XYZ, aux.current_local_solution.raw().data(), aux.current_local_solution.size()), "undeformed coordinates");
  const int64_t ne = es_.get_mesh().n_elem();
  const unsigned int nf = fibre.n_vars();   // 6: reference x,y,z then current x,y,z (src/solid.C:44-50)
  std::vector<double> eta((size_t)ne * 3);
  for (int64_t e = 0; e < ne; e++)
    for (int d = 0; d < 3; d++) eta[(size_t)e * 3 + d] = fibre.current_local_solution(e * nf + d);
  check(c, rdc_field_upload(c, RDC_FIELD_ELEM_FIBRE, eta.data(), (int64_t)eta.size()), "fibre field");
  rdc_solid_params p{};
  p.pseudo_time = P.get<Real>("pseudo_time");
  p.displacement_penalty = P.get<Real>("BCs/displacement_penalty");
  p.use_symmetry = P.get<bool>("solver/assembly_use_symmetry") ? 1 : 0;
  check(c, rdc_solid_assemble(c, &p, get_jacobian ? 1 : 0), "rdc_solid_assemble");
  check(c, rdc_csr_download(c, get_jacobian ? matrix_storage.val.data() : nullptr, rhs_storage.raw().data()), "rdc_csr_download");
  last.assemblies++;
}

inline void SolidSystem::solve() {
  last = NewtonLog();
  std::vector<double> delta(solution.raw().size()), minus_r(delta.size());
  for (int it = 0;; it++) {
    assembly(true, true);
    const Real rn = rhs_storage.l2_norm();
    if (it == 0) last.first_residual = rn;
    last.last_residual = rn;
    if (rn <= absolute_residual_tolerance || rn <= relative_residual_tolerance * last.first_residual) { last.converged = true; break; }
    if (it >= max_nonlinear_iterations) break;
    for (size_t i = 0; i < delta.size(); i++) { minus_r[i] = -rhs_storage.raw()[i]; delta[i] = 0.0; }
    last.linear_iterations += bicgstab_ilu0(matrix_storage, minus_r, delta, initial_linear_tolerance, max_linear_iterations);
    Real step = 1.0;
    const std::vector<double> u0 = solution.raw();
    for (;;) {   // full Newton step; halved while the residual does not go down if "require_reduction" is set
      for (size_t i = 0; i < delta.size(); i++) solution.raw()[i] = u0[i] + step * delta[i];
      update();
      if (!require_residual_reduction || step < 1e-3) break;
      assembly(true, false);
      if (rhs_storage.l2_norm() < rn) break;
      step *= 0.5;
    }
    last.nonlinear_iterations = it + 1;
    long double dn = 0, un = 0;
    for (size_t i = 0; i < delta.size(); i++) { dn += (long double)delta[i] * delta[i] * step * step; un += (long double)solution.raw()[i] * solution.raw()[i]; }
    if (std::sqrt((double)dn) <= relative_step_tolerance * std::sqrt((double)un)) {
      assembly(true, false);
      last.last_residual = rhs_storage.l2_norm();
      last.converged = true;
      break;
    }
  }
}

inline void SolidSystem::run_solver() {
  System& aux = es_.get_system<System>("SolidSystem::auxiliary");
  solve();
  aux.solution = aux.current_local_solution;
  es_.reinit();
}

inline void SolidSystem::post_process() {
  const Parameters& P = es_.parameters;
  rdc_ctx* c = es_.context(name_, 3);
  if (!bound_) bind(c);
  System& aux = es_.get_system<System>("SolidSystem::auxiliary");
  System& fibre = es_.get_system<System>("SolidSystem::fibre");
  System& press = es_.get_system<System>("SolidSystem::pressure");
  System& vm = es_.get_system<System>("SolidSystem::von_mises");
  const int64_t ne = es_.get_mesh().n_elem();
  const unsigned int nf = fibre.n_vars();
  check(c, rdc_mesh_update_coords(c, current_local_solution.raw().data()), "rdc_mesh_update_coords");
  check(c, rdc_field_upload(c, RDC_FIELD_UNDEFORMED_XYZ, aux.current_local_solution.raw().data(), aux.current_local_solution.size()), "undeformed coordinates");
  std::vector<double> eta((size_t)ne * 3), cur((size_t)ne * 3);
  for (int64_t e = 0; e < ne; e++)
    for (int d = 0; d < 3; d++) eta[(size_t)e * 3 + d] = fibre.current_local_solution(e * nf + d);
  check(c, rdc_field_upload(c, RDC_FIELD_ELEM_FIBRE, eta.data(), (int64_t)eta.size()), "fibre field");
  rdc_solid_params p{};
  p.pseudo_time = P.get<Real>("pseudo_time");
  p.displacement_penalty = P.get<Real>("BCs/displacement_penalty");
  check(c, rdc_solid_post_process(c, &p, press.solution.raw().data(), vm.solution.raw().data(), cur.data()), "rdc_solid_post_process");
  for (int64_t e = 0; e < ne; e++)
    for (int d = 0; d < 3; d++) fibre.solution.set(e * nf + 3 + d, cur[(size_t)e * 3 + d]);   // fibre_current_* (:526-527)
}

inline void SolidSystem::update_data() {
  auto& aux = es_.get_system<TransientExplicitSystem>("SolidSystem::auxiliary");
  aux.older_local_solution = aux.old_local_solution;
  aux.old_local_solution = aux.current_local_solution;
}

}  // namespace host
}  // namespace rdc
#endif
