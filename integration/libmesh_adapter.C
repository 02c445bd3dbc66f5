// libmesh_adapter.C — the reference-side binding: drop-in replacements for the static
// assemble_<model>(EquationSystems&, const std::string&) callbacks of rdcFEs that forward to the
// C-ABI of librdc_assembly.so.
//
// NOT COMPILED IN THIS REPOSITORY'S BUILD: it needs libMesh (d3bda6c) and PETSc (746207a), neither of
// which exists in the build image.  It is written against the libMesh API the reference itself uses
// (src/pihna.C:318-395, :752-755) and is exercised here only through its twin over the mock types,
// rdcfes_amd/host/rdc_host.h, which makes exactly the same C-ABI calls in the same order.
//
// Usage in the reference tree:
//   1. add this file to src/ (Makefile:7 globs src/*.C), link with -lrdc_assembly;
//   2. in src/pihna.C replace   model.attach_assemble_function(assemble_pihna);          (:35)
//      by                       model.attach_assemble_function(rdc_gpu::assemble_pihna);
//      (same for src/ripf.C:27, src/coupled_hcc.C:37, src/adpm.C, src/proteas.C).  Nothing else changes:
//      libMesh still zeroes matrix/rhs, calls the callback once per time step on every rank, closes
//      matrix/rhs and hands them to PETSc KSP.
//   3. solid mechanics: replace   es.add_system<SolidSystem>("SolidSystem")           (src/solid.C:27, src/coupled_hcc.C:40)
//      by                         es.add_system<rdc_gpu::SolidSystemGPU>("SolidSystem")
//      -- a subclass of the reference's SolidSystem that overrides FEMSystem::assembly(), so that one
//      rdc_solid_assemble call replaces the per-element element_time_derivative / side_time_derivative
//      virtuals (src/solid_system.C:146-371) of every Newton iteration.
//   4. MORE THAN ONE MPI RANK: the GPU context of a rank holds its local elements PLUS the neighbouring elements that
//      touch one of its nodes, and reads the old solution (and the TD / RT / AUX / auxiliary systems) at all their
//      nodes.  libMesh's default algebraic ghosting (send_list) covers the dofs of local elements only, so before
//      es.init() call   rdc_gpu::add_ghost_layer(system)   for every system the callbacks read (the model system and,
//      where used, "RIPF-TimeDeriv", "RT", "AUX", "SolidSystem::auxiliary"): it attaches a PointNeighborCoupling with one
//      level to the system's DofMap, which puts those dofs into the ghosted vectors.  Without it sys.old_solution(dof)
//      on such a node is an out-of-range read in opt mode (an assert in dbg).  Serial runs need nothing.
//      The GPU of a rank is  node-local MPI rank % rdc_device_count()  (one rank per GPU).
#include "libmesh/boundary_info.h"
#include "libmesh/dof_map.h"
#include "libmesh/elem.h"
#include "libmesh/equation_systems.h"
#include "libmesh/mesh_base.h"
#include "libmesh/numeric_vector.h"
#include "libmesh/petsc_matrix.h"
#include "libmesh/point_neighbor_coupling.h"
#include "libmesh/transient_system.h"

#include "./solid_system.h"   // the reference's own header (src/solid_system.h): SolidSystemGPU derives from it
#include "./utils.h"          // export_integers (src/utils.h:268)

#include <map>
#include <unordered_map>
#include <vector>

#include "rdc_assembly.h"

using namespace libMesh;

namespace rdc_gpu {

// One GPU context per (rank, system): built on the first call, reused every time step.
struct Binding {
  rdc_ctx* ctx = nullptr;
  std::vector<dof_id_type> local_to_global_node;   // local node id -> libMesh node id (owned first, then ghosts)
  std::vector<dof_id_type> elem_ids;               // binding element -> libMesh element id (local elements + ghost layer)
  std::vector<PetscInt> row_ptr, col_glob;         // owned-row CSR with GLOBAL dof column ids
  std::vector<PetscInt> row_glob;                  // global dof id of owned row r = local node * nvar + var
  std::vector<double> val, rhs, u_old;
  dof_id_type n_owned = 0;
  bool pinned = false;                             // val / rhs registered with the HIP runtime (rdc_host_pin): async chunk copies
  bool pattern_frozen = false;                     // MAT_NEW_NONZERO_LOCATIONS switched off after the first assembly
};

static std::map<std::string, Binding> g_bindings;

static void fail(rdc_ctx* c, const char* what) { libmesh_error_msg(std::string(what) + ": " + rdc_last_error(c)); }

// N > 1 ranks: ghost the dofs of every element that shares a point with a local element (see the header, item 4).
// Call before es.init() for every system the GPU callbacks read.
void add_ghost_layer(System& sys) {
  static std::vector<std::unique_ptr<PointNeighborCoupling>> keep;   // the DofMap stores a reference
  keep.emplace_back(new PointNeighborCoupling());
  keep.back()->set_n_levels(1);
  sys.get_dof_map().add_algebraic_ghosting_functor(*keep.back());
}

// one rank per GPU: the device of this rank is its rank among the ranks of the same host
static int device_of_this_rank(const Parallel::Communicator& comm) {
  int ndev = 0;
  if (rdc_device_count(&ndev) != RDC_OK || ndev <= 0) libmesh_error_msg(rdc_last_error(nullptr));
  int local_rank = 0;
#ifdef LIBMESH_HAVE_MPI
  MPI_Comm node;
  MPI_Comm_split_type(comm.get(), MPI_COMM_TYPE_SHARED, (int)comm.rank(), MPI_INFO_NULL, &node);
  MPI_Comm_rank(node, &local_rank);
  MPI_Comm_free(&node);
#endif
  return local_rank % ndev;
}

// Marshal the rank's partition once: owned nodes first, then ghost nodes; elements = every active
// element touching an owned node (libMesh's active_local elements plus one ghost layer, which a
// DistributedMesh / ghosted ReplicatedMesh already holds).
static Binding& bind(EquationSystems& es, const std::string& name, unsigned int nvar) {
  Binding& B = g_bindings[name];
  if (B.ctx) return B;
  const MeshBase& mesh = es.get_mesh();
  const System& sys = es.get_system(name);
  const processor_id_type me = mesh.processor_id();
  std::unordered_map<dof_id_type, uint32_t> g2l;
  for (const auto& node : mesh.local_node_ptr_range()) { g2l[node->id()] = (uint32_t)B.local_to_global_node.size(); B.local_to_global_node.push_back(node->id()); }
  B.n_owned = (dof_id_type)B.local_to_global_node.size();
  std::vector<uint32_t> conn;
  std::vector<const Elem*> elems;
  for (const auto& elem : mesh.active_element_ptr_range()) {
    bool touches = false;
    for (unsigned int i = 0; i < elem->n_nodes(); i++) touches |= (elem->node_ref(i).processor_id() == me);
    if (touches) elems.push_back(elem);
  }
  const int nen = (int)elems.front()->n_nodes();   // TET4 (4) or HEX8 (8), one type per mesh
  for (const Elem* elem : elems) B.elem_ids.push_back(elem->id());
  for (const Elem* elem : elems)
    for (int i = 0; i < nen; i++) {
      const dof_id_type g = elem->node_id(i);
      auto it = g2l.find(g);
      if (it == g2l.end()) { it = g2l.emplace(g, (uint32_t)B.local_to_global_node.size()).first; B.local_to_global_node.push_back(g); }
      conn.push_back(it->second);
    }
  std::vector<double> xyz(3 * B.local_to_global_node.size());
  for (size_t l = 0; l < B.local_to_global_node.size(); l++)
    for (int d = 0; d < 3; d++) xyz[3 * l + d] = mesh.node_ref(B.local_to_global_node[l])(d);
  if (rdc_ctx_create(device_of_this_rank(mesh.comm()), &B.ctx) != RDC_OK) libmesh_error_msg(rdc_last_error(nullptr));
  if (rdc_mesh_upload(B.ctx, nen, (int64_t)elems.size(), (int64_t)B.local_to_global_node.size(), B.n_owned, conn.data(),
                      xyz.data(), (int)nvar) != RDC_OK) fail(B.ctx, "rdc_mesh_upload");
  // pattern with LOCAL column ids -> global PETSc dof ids (dof_number(sys, var, 0))
  int64_t n_rows = 0, nnz = 0;
  rdc_csr_dims(B.ctx, &n_rows, &nnz);
  std::vector<int64_t> rp(n_rows + 1);
  std::vector<int32_t> cl(nnz);
  rdc_csr_pattern_download(B.ctx, rp.data(), cl.data());
  B.row_ptr.assign(rp.begin(), rp.end());
  B.col_glob.resize(nnz);
  for (int64_t k = 0; k < nnz; k++) {
    const Node& nd = mesh.node_ref(B.local_to_global_node[cl[k] / nvar]);
    B.col_glob[k] = (PetscInt)nd.dof_number(sys.number(), cl[k] % nvar, 0);
  }
  B.val.resize(nnz); B.rhs.resize(n_rows); B.u_old.resize(nvar * B.local_to_global_node.size());
  B.row_glob.resize(n_rows);
  for (int64_t r = 0; r < n_rows; r++)
    B.row_glob[r] = (PetscInt)mesh.node_ref(B.local_to_global_node[r / nvar]).dof_number(sys.number(), r % nvar, 0);
  return B;
}

// old_local_solution (ghosted) -> [local node][var], the layout of RDC_FIELD_OLD_SOLUTION
static void gather_old_solution(const EquationSystems& es, const TransientLinearImplicitSystem& sys, Binding& B, unsigned int nvar) {
  const MeshBase& mesh = es.get_mesh();
  for (size_t l = 0; l < B.local_to_global_node.size(); l++) {
    const Node& nd = mesh.node_ref(B.local_to_global_node[l]);
    for (unsigned int v = 0; v < nvar; v++) B.u_old[l * nvar + v] = sys.old_solution(nd.dof_number(sys.number(), v, 0));   // src/pihna.C:433
  }
}

// assembled owned rows -> system.matrix / system.rhs (what add_matrix / add_vector did, src/pihna.C:754-755)
static void push_results(const EquationSystems& es, TransientLinearImplicitSystem& sys, Binding& B, unsigned int nvar) {
  if (rdc_csr_download(B.ctx, B.val.data(), B.rhs.data()) != RDC_OK) fail(B.ctx, "rdc_csr_download");
  Mat A = cast_ref<PetscMatrix<Number>&>(*sys.matrix).mat();
  const MeshBase& mesh = es.get_mesh();
  for (dof_id_type l = 0; l < B.n_owned; l++) {
    const Node& nd = mesh.node_ref(B.local_to_global_node[l]);
    for (unsigned int a = 0; a < nvar; a++) {
      const PetscInt row = (PetscInt)nd.dof_number(sys.number(), a, 0);
      const PetscInt r = (PetscInt)(l * nvar + a), b = B.row_ptr[r], n = B.row_ptr[r + 1] - b;
      MatSetValues(A, 1, &row, n, &B.col_glob[b], &B.val[b], INSERT_VALUES);   // complete rows: no off-rank stash traffic
      sys.rhs->set(row, B.rhs[r]);
    }
  }
}

// The same hand-back, pipelined (es.parameters "rdc/handback_chunks" > 1; mirrored and tested in rdcfes_amd/host/rdc_host.h,
// detail::pull_results_chunked): the node range is cut into chunks, two chunk downloads are kept in flight on the context's
// copy stream (rdc_csr_download_rows_async: behind the work enqueued so far, independent of later work), and every chunk is
// inserted while the next travels.  The nvar rows of a node share one column set and lie back to back in the CSR array, so
// they are ONE dense nvar x ncols block for MatSetValues: one call per node block instead of one per row (rows of different
// nodes have different column sets: a chunk as a whole is not a dense block, so it cannot be a single call).
// After the first assembly the pattern is frozen (MAT_NEW_NONZERO_LOCATIONS off): PETSc then skips the search for new
// locations and any pattern drift is an error instead of a silent reallocation.
static void push_results_chunked(const EquationSystems& es, TransientLinearImplicitSystem& sys, Binding& B, unsigned int nvar, int n_chunks) {
  Mat A = cast_ref<PetscMatrix<Number>&>(*sys.matrix).mat();
  const MeshBase& mesh = es.get_mesh();
  if (!B.pinned) {
    if (rdc_host_pin(B.ctx, B.val.data(), B.val.size() * sizeof(double)) != RDC_OK) fail(B.ctx, "rdc_host_pin");
    if (rdc_host_pin(B.ctx, B.rhs.data(), B.rhs.size() * sizeof(double)) != RDC_OK) fail(B.ctx, "rdc_host_pin");
    B.pinned = true;
  }
  const int64_t n_nodes = (int64_t)B.n_owned;
  auto bound = [&](int k) { return n_nodes * k / n_chunks; };
  int ticket[2] = {-1, -1};
  if (rdc_csr_download_rows_async(B.ctx, bound(0), bound(1), B.val.data(), B.rhs.data(), &ticket[0]) != RDC_OK) fail(B.ctx, "rdc_csr_download_rows_async");
  std::vector<PetscInt> rows(nvar);
  for (int k = 0; k < n_chunks; k++) {
    if (k + 1 < n_chunks &&
        rdc_csr_download_rows_async(B.ctx, bound(k + 1), bound(k + 2), B.val.data(), B.rhs.data(), &ticket[(k + 1) & 1]) != RDC_OK)
      fail(B.ctx, "rdc_csr_download_rows_async");
    if (rdc_ticket_wait(B.ctx, ticket[k & 1]) != RDC_OK) fail(B.ctx, "rdc_ticket_wait");
    for (int64_t l = bound(k); l < bound(k + 1); l++) {
      const Node& nd = mesh.node_ref(B.local_to_global_node[(size_t)l]);
      for (unsigned int a = 0; a < nvar; a++) rows[a] = (PetscInt)nd.dof_number(sys.number(), a, 0);
      const PetscInt b = B.row_ptr[(size_t)(l * nvar)], n = B.row_ptr[(size_t)(l * nvar) + 1] - b;
      // rows l*nvar .. l*nvar + nvar - 1: the same n columns (B.col_glob[b ..]), values row after row from B.val[b]
      MatSetValues(A, (PetscInt)nvar, rows.data(), n, &B.col_glob[(size_t)b], &B.val[(size_t)b], INSERT_VALUES);
      for (unsigned int a = 0; a < nvar; a++) sys.rhs->set(rows[a], B.rhs[(size_t)(l * nvar + a)]);
    }
  }
  if (!B.pattern_frozen) {   // takes effect for the NEXT assembly (this one may still have created locations)
    MatSetOption(A, MAT_NEW_NONZERO_LOCATIONS, PETSC_FALSE);
    B.pattern_frozen = true;
  }
}

#if defined(PETSC_HAVE_HIP) && defined(RDC_ADAPTER_DEVICE_HANDOFF)
// Device-pointer hand-off: with one rank and a MATSEQAIJHIPSPARSE system matrix PETSc's CSR value
// array IS the context's (same row order, columns ascending within a row -- rdc_csr_pattern_download is what the adapter
// preallocated from), so the hand-back is one device-to-device copy and the 5 GB never cross PCIe.  With several ranks PETSc
// keeps the diagonal and the off-diagonal block of MPIAIJ separately; the chunked host path above stays the portable one.
#include <hip/hip_runtime_api.h>
static bool push_results_device(TransientLinearImplicitSystem& sys, Binding& B) {
  Mat A = cast_ref<PetscMatrix<Number>&>(*sys.matrix).mat();
  PetscBool is_seq_hip = PETSC_FALSE;
  PetscObjectTypeCompare((PetscObject)A, MATSEQAIJHIPSPARSE, &is_seq_hip);
  if (!is_seq_hip) return false;
  double *d_val = nullptr, *d_rhs = nullptr;
  if (rdc_csr_values_device_ptr(B.ctx, &d_val, &d_rhs) != RDC_OK) fail(B.ctx, "rdc_csr_values_device_ptr");
  PetscScalar* a = nullptr;
  MatSeqAIJHIPSPARSEGetArrayWrite(A, &a);                 // device pointer of the CSR values, write access
  if (rdc_synchronize(B.ctx) != RDC_OK) fail(B.ctx, "rdc_synchronize");
  if (hipMemcpy(a, d_val, B.val.size() * sizeof(double), hipMemcpyDeviceToDevice) != hipSuccess) libmesh_error_msg("device hand-off copy failed");
  MatSeqAIJHIPSPARSERestoreArrayWrite(A, &a);
  // the rhs is small (nvar doubles per node): through the host as before
  if (rdc_csr_download(B.ctx, nullptr, B.rhs.data()) != RDC_OK) fail(B.ctx, "rdc_csr_download");
  for (size_t r = 0; r < B.rhs.size(); r++) sys.rhs->set(B.row_glob[r], B.rhs[r]);
  return true;
}
#endif

// what every assemble_<model> callback does with the finished rows
static void hand_back(const EquationSystems& es, TransientLinearImplicitSystem& sys, Binding& B, unsigned int nvar) {
#if defined(PETSC_HAVE_HIP) && defined(RDC_ADAPTER_DEVICE_HANDOFF)
  if (B.pattern_frozen && push_results_device(sys, B)) return;   // from the second assembly on (the first one creates the pattern on the host path)
#endif
  const int chunks = es.parameters.have_parameter<int>("rdc/handback_chunks") ? es.parameters.get<int>("rdc/handback_chunks") : 8;
  if (chunks > 1) push_results_chunked(es, sys, B, nvar, chunks);
  else push_results(es, sys, B, nvar);
}

void assemble_pihna(EquationSystems& es, const std::string& system_name) {
  TransientLinearImplicitSystem& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  libmesh_assert_equal_to(system.n_vars(), 5);
  Binding& B = bind(es, system_name, 5);
  rdc_pihna_params p;   // the same keys assemble_pihna reads, src/pihna.C:358-381
  p.time_step = es.parameters.get<Real>("time_step");
  p.cells_min_capacity = es.parameters.get<Real>("cells_min_capacity");
  p.cells_max_capacity = es.parameters.get<Real>("cells_max_capacity");
  p.cytokines_max_capacity = es.parameters.get<Real>("cytokines_max_capacity");
  p.cells_max_capacity_exponent = es.parameters.get<Real>("cells_max_capacity/exponent");
  p.necrosis_c = es.parameters.get<Real>("necrosis/c"); p.necrosis_h = es.parameters.get<Real>("necrosis/h"); p.necrosis_v = es.parameters.get<Real>("necrosis/v");
  p.diffuse_c = es.parameters.get<Real>("diffuse/c"); p.taxis_c = es.parameters.get<Real>("taxis/c");
  p.diffuse_h = es.parameters.get<Real>("diffuse/h"); p.taxis_h = es.parameters.get<Real>("taxis/h");
  p.produce_c = es.parameters.get<Real>("produce/c");
  p.switch_c2h = es.parameters.get<Real>("switch/c/to/h"); p.switch_h2c = es.parameters.get<Real>("switch/h/to/c"); p.switch_h2n = es.parameters.get<Real>("switch/h/to/n");
  p.diffuse_v = es.parameters.get<Real>("diffuse/v"); p.taxis_v = es.parameters.get<Real>("taxis/v"); p.produce_v = es.parameters.get<Real>("produce/v");
  p.secrete_a_c = es.parameters.get<Real>("secrete/a/from/c"); p.secrete_a_h = es.parameters.get<Real>("secrete/a/from/h");
  p.uptake_a_v = es.parameters.get<Real>("uptake/a/from/v"); p.decay_a = es.parameters.get<Real>("decay/a");
  gather_old_solution(es, system, B, 5);
  if (rdc_field_upload(B.ctx, RDC_FIELD_OLD_SOLUTION, B.u_old.data(), (int64_t)B.u_old.size()) != RDC_OK) fail(B.ctx, "rdc_field_upload");
  if (rdc_assemble_pihna(B.ctx, &p) != RDC_OK) fail(B.ctx, "rdc_assemble_pihna");
  hand_back(es, system, B, 5);
}

// src/coupled_hcc.C:414-649.  The mesh is the CURRENT configuration (SolidSystem::update moved the nodes,
// src/coupled_hcc.C:98-114), so the coordinates are refreshed before every assembly.
void assemble_hcc(EquationSystems& es, const std::string& system_name) {
  TransientLinearImplicitSystem& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  libmesh_assert_equal_to(system.n_vars(), 3);
  Binding& B = bind(es, system_name, 3);
  rdc_hcc_params p;   // src/coupled_hcc.C:450-461
  p.time_step = es.parameters.get<Real>("time_step");
  p.cells_min_capacity = es.parameters.get<Real>("cells/min_capacity");
  p.cells_max_capacity = es.parameters.get<Real>("cells/max_capacity");
  p.cells_max_capacity_exponent = es.parameters.get<Real>("cells/max_capacity/exponent");
  p.produce_l = es.parameters.get<Real>("produce/l");
  p.diffuse_c = es.parameters.get<Real>("diffuse/c"); p.mechano_c = es.parameters.get<Real>("mechano/c"); p.produce_c = es.parameters.get<Real>("produce/c");
  p.necrosis_l = es.parameters.get<Real>("necrosis/l"); p.necrosis_c = es.parameters.get<Real>("necrosis/c");
  p.necrosis_pressure = es.parameters.get<Real>("necrosis/pressure");
  const MeshBase& mesh = es.get_mesh();
  std::vector<double> xyz(3 * B.local_to_global_node.size());
  for (size_t l = 0; l < B.local_to_global_node.size(); l++) {
    const Node& nd = mesh.node_ref(B.local_to_global_node[l]);
    for (unsigned int d = 0; d < 3; d++) xyz[3 * l + d] = nd(d);
  }
  if (rdc_mesh_update_coords(B.ctx, xyz.data()) != RDC_OK) fail(B.ctx, "rdc_mesh_update_coords");
  gather_old_solution(es, system, B, 3);
  if (rdc_field_upload(B.ctx, RDC_FIELD_OLD_SOLUTION, B.u_old.data(), (int64_t)B.u_old.size()) != RDC_OK) fail(B.ctx, "rdc_field_upload");
  if (rdc_assemble_hcc(B.ctx, &p) != RDC_OK) fail(B.ctx, "rdc_assemble_hcc");
  hand_back(es, system, B, 3);
}

// src/ripf.C:337-673: additionally reads TD vars 1, 2 (src/ripf.C:470-471) and RT var 2 (:477-478)
void assemble_ripf(EquationSystems& es, const std::string& system_name) {
  TransientLinearImplicitSystem& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  libmesh_assert_equal_to(system.n_vars(), 3);
  const System& TD = es.get_system<System>("RIPF-TimeDeriv");
  const System& RT = es.get_system<ExplicitSystem>("RT");
  Binding& B = bind(es, system_name, 3);
  rdc_ripf_params p;   // src/ripf.C:377-408
  auto R = [&](const char* k) { return es.parameters.get<Real>(k); };
  p.time_step = R("time_step");
  p.VolFr_stroma = R("volume_fraction/stroma"); p.VolFr_parenchyma = R("volume_fraction/parenchyma");
  p.VolFr_exponent = R("volume_fraction/exponent"); p.VolFr_min_vacant = R("volume_fraction/min_vacant");
  p.VolFr_max_vacant = R("volume_fraction/max_vacant");
  p.phi_cc_B = R("HU/phi/cc/build"); p.phi_cc_D = R("HU/phi/cc/decay"); p.phi_cc = R("HU/phi/cc/rate");
  p.phi_fb_B = R("HU/phi/fb/build"); p.phi_fb_D = R("HU/phi/fb/decay"); p.phi_fb = R("HU/phi/fb/rate");
  p.phi_tol = R("HU/phi/tolerance");
  p.kappa = R("cc/kappa"); p.kappa_RT_c = R("cc/kappa/RT/c");
  p.delta = R("cc/delta"); p.delta_RT_a = R("cc/delta/RT/a"); p.delta_RT_b = R("cc/delta/RT/b");
  p.lambda = R("fb/lambda"); p.lambda_RT_r = R("fb/lambda/RT/r"); p.lambda_HU_r = R("fb/lambda/HU/r");
  p.omicro = R("fb/omicro"); p.omicro_RT_r = R("fb/omicro/RT/r"); p.omicro_fb_b = R("fb/omicro/fb/b");
  p.omega = R("fb/omega"); p.diffusion = R("fb/diffusion"); p.haptotaxis = R("fb/haptotaxis"); p.radiotaxis = R("fb/radiotaxis");
  p.RT_dose_total_max = es.parameters.get<int>("RT_dose/total/max");
  const MeshBase& mesh = es.get_mesh();
  std::vector<double> aux(3 * B.local_to_global_node.size());
  for (size_t l = 0; l < B.local_to_global_node.size(); l++) {
    const Node& nd = mesh.node_ref(B.local_to_global_node[l]);
    aux[3 * l + 0] = TD.current_solution(nd.dof_number(TD.number(), 1, 0));
    aux[3 * l + 1] = TD.current_solution(nd.dof_number(TD.number(), 2, 0));
    aux[3 * l + 2] = RT.current_solution(nd.dof_number(RT.number(), 2, 0));
  }
  gather_old_solution(es, system, B, 3);
  if (rdc_field_upload(B.ctx, RDC_FIELD_OLD_SOLUTION, B.u_old.data(), (int64_t)B.u_old.size()) != RDC_OK) fail(B.ctx, "rdc_field_upload");
  if (rdc_field_upload(B.ctx, RDC_FIELD_AUX_NODAL, aux.data(), (int64_t)aux.size()) != RDC_OK) fail(B.ctx, "rdc_field_upload(aux)");
  if (rdc_assemble_ripf(B.ctx, &p) != RDC_OK) fail(B.ctx, "rdc_assemble_ripf");
  hand_back(es, system, B, 3);
}

// current_local_solution of any nodal system at the binding's nodes -> [local node][first..first+n) per node
static void gather_nodal(const EquationSystems& es, const System& sys, const Binding& B, unsigned int first_var, unsigned int n,
                         unsigned int stride, unsigned int offset, std::vector<double>& out) {
  const MeshBase& mesh = es.get_mesh();
  for (size_t l = 0; l < B.local_to_global_node.size(); l++) {
    const Node& nd = mesh.node_ref(B.local_to_global_node[l]);
    for (unsigned int v = 0; v < n; v++) out[l * stride + offset + v] = sys.current_solution(nd.dof_number(sys.number(), first_var + v, 0));
  }
}

// src/adpm.C:324-652: unknowns PrP, A_b, Tau; the elemental "Tracts" system (3 CONSTANT MONOMIAL variables, :448-453)
// becomes RDC_FIELD_ELEM_TRACTS; decay/PrP is scaled with system.time (:367-413)
void assemble_adpm(EquationSystems& es, const std::string& system_name) {
  TransientLinearImplicitSystem& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  libmesh_assert_equal_to(system.n_vars(), 3);
  const System& tracts = es.get_system<System>("Tracts");
  Binding& B = bind(es, system_name, 3);
  auto R = [&](const std::string& k) { return es.parameters.get<Real>(k); };
  rdc_adpm_params p;
  p.time_step = R("time_step");
  p.time = system.time;
  p.decay_PrP_time_exponent = R("decay/PrP/time_exponent");
  auto triple = [&](double* d, const std::string& key, const char* kind) {
    d[0] = R(key); d[1] = R(key + "/" + kind + "/0"); d[2] = R(key + "/" + kind + "/1");
  };
  auto trapezoid = [&](double* d, const std::string& key) {
    d[0] = R(key);
    for (int i = 0; i < 4; i++) d[1 + i] = R(key + "/trapezoid/" + std::to_string(i));
  };
  triple(p.decay_PrP, "decay/PrP", "pulse");
  trapezoid(p.transform_A_b, "transform/A_b"); trapezoid(p.transform_Tau, "transform/Tau");
  triple(p.diffuse_A_b, "diffuse/A_b", "pulse"); triple(p.taxis1_A_b, "taxis_1/A_b", "pulse"); triple(p.taxis2_A_b, "taxis_2/A_b", "pulse");
  triple(p.produce_A_b, "produce/A_b", "sigmoid"); triple(p.decay_A_b, "decay/A_b", "pulse");
  triple(p.diffuse_Tau, "diffuse/Tau", "pulse"); triple(p.taxis1_Tau, "taxis_1/Tau", "pulse"); triple(p.taxis2_Tau, "taxis_2/Tau", "pulse");
  triple(p.produce_Tau, "produce/Tau", "sigmoid"); triple(p.decay_Tau, "decay/Tau", "pulse");
  p.taxis_A_b_angle = R("taxis/A_b/angle");   // already radians in es.parameters (src/adpm.C:193)
  p.taxis_Tau_angle = R("taxis/Tau/angle");
  // per-element tract vectors in the binding's element order
  const MeshBase& mesh = es.get_mesh();
  std::vector<double> tr(3 * B.elem_ids.size());
  for (size_t e = 0; e < B.elem_ids.size(); e++) {
    const Elem& elem = mesh.elem_ref(B.elem_ids[e]);
    for (unsigned int d = 0; d < 3; d++) tr[3 * e + d] = tracts.current_solution(elem.dof_number(tracts.number(), d, 0));
  }
  gather_old_solution(es, system, B, 3);
  if (rdc_field_upload(B.ctx, RDC_FIELD_OLD_SOLUTION, B.u_old.data(), (int64_t)B.u_old.size()) != RDC_OK) fail(B.ctx, "rdc_field_upload");
  if (rdc_field_upload(B.ctx, RDC_FIELD_ELEM_TRACTS, tr.data(), (int64_t)tr.size()) != RDC_OK) fail(B.ctx, "rdc_field_upload(tracts)");
  if (rdc_assemble_adpm(B.ctx, &p) != RDC_OK) fail(B.ctx, "rdc_assemble_adpm");
  hand_back(es, system, B, 3);
}

// src/proteas.C:338-705: unknowns hos, tum, nec, vsc, oed; the nodal "AUX" system {HU, RTD} -> RDC_FIELD_AUX_NODAL
// (only variable 0 is read by the assembly, at local node 1 of every element: src/proteas.C:472,481)
void assemble_proteas_model(EquationSystems& es, const std::string& system_name) {
  TransientLinearImplicitSystem& system = es.get_system<TransientLinearImplicitSystem>(system_name);
  libmesh_assert_equal_to(system.n_vars(), 5);
  const System& AUX = es.get_system<System>("AUX");
  Binding& B = bind(es, system_name, 5);
  auto R = [&](const char* k) { return es.parameters.get<Real>(k); };
  rdc_proteas_params p;   // src/proteas.C:376-409
  p.time_step = R("time_step");
  p.cells_total_capacity = R("cells/total_capacity"); p.RT_max_dosage = R("radiotherapy/max_dosage");
  p.host_proliferation = R("host/proliferation"); p.host_vsc_threshold = R("host/vsc_threshold");
  p.host_RT_death_rate = R("host/RT_death_rate"); p.host_RT_exp_a = R("host/RT_exp_a"); p.host_RT_exp_b = R("host/RT_exp_b");
  p.host_necrosis_rate = R("host/necrosis_rate");
  p.tumour_diffusion = R("tumour/diffusion"); p.tumour_diffusion_host = R("tumour/diffusion_host");
  p.tumour_proliferation = R("tumour/proliferation"); p.tumour_vsc_threshold = R("tumour/vsc_threshold");
  p.tumour_RT_death_rate = R("tumour/RT_death_rate"); p.tumour_RT_exp_a = R("tumour/RT_exp_a"); p.tumour_RT_exp_b = R("tumour/RT_exp_b");
  p.tumour_necrosis_rate = R("tumour/necrosis_rate");
  p.necrosis_clearance = R("necrosis/clearance"); p.necrosis_slope = R("necrosis/slope"); p.necrosis_vsc_threshold = R("necrosis/vsc_threshold");
  p.vascular_proliferation = R("vascular/proliferation"); p.vascular_necrosis_rate = R("vascular/necrosis_rate");
  p.oedema_diffusion = R("oedema/diffusion"); p.oedema_proliferation = R("oedema/proliferation"); p.oedema_vsc_threshold = R("oedema/vsc_threshold");
  p.oedema_RT_coeff = R("oedema/RT_coeff"); p.oedema_RT_exp = R("oedema/RT_exp"); p.oedema_reabsorption_rate = R("oedema/reabsorption_rate");
  std::vector<double> aux(3 * B.local_to_global_node.size(), 0.0);
  gather_nodal(es, AUX, B, 0, 2, 3, 0, aux);
  gather_old_solution(es, system, B, 5);
  if (rdc_field_upload(B.ctx, RDC_FIELD_OLD_SOLUTION, B.u_old.data(), (int64_t)B.u_old.size()) != RDC_OK) fail(B.ctx, "rdc_field_upload");
  if (rdc_field_upload(B.ctx, RDC_FIELD_AUX_NODAL, aux.data(), (int64_t)aux.size()) != RDC_OK) fail(B.ctx, "rdc_field_upload(aux)");
  if (rdc_assemble_proteas(B.ctx, &p) != RDC_OK) fail(B.ctx, "rdc_assemble_proteas");
  hand_back(es, system, B, 5);
}

// ---- SolidSystem: FEMSystem::assembly() replaced wholesale -------------------------------------------------------
// libMesh's FEMSystem::assembly(get_residual, get_jacobian, ...) zeroes rhs / matrix, loops over the local elements
// (and their boundary sides) calling the reference's element_time_derivative / side_time_derivative virtuals
// (src/solid_system.C:146-371) and adds each element's residual and Jacobian into the global ones.  Here the loop is
// ONE GPU call on data marshalled from exactly what those virtuals read.  NewtonSolver, the PETSc linear solve,
// SolidSystem::update() (moves the mesh), run_solver / post_process / update_data stay the reference's code.
class SolidSystemGPU : public SolidSystem {
 public:
  SolidSystemGPU(EquationSystems& es, std::string name, unsigned int number) : SolidSystem(es, name, number) {}

  virtual void assembly(bool get_residual, bool get_jacobian, bool /*apply_heterogeneous_constraints*/ = false,
                        bool /*apply_no_constraints*/ = false) override {
    EquationSystems& es = this->get_equation_systems();
    const MeshBase& mesh = es.get_mesh();
    Binding& B = bind(es, this->name(), 3);
    const System& aux = es.get_system<System>("SolidSystem::auxiliary");
    const System& fibre = es.get_system<System>("SolidSystem::fibre");
    const size_t nn = B.local_to_global_node.size(), ne = B.elem_ids.size();
    if (!solid_bound_) {   // what does not change between Newton iterations
      // subdomain -> material table ("material/<id>/Hyperelastic/...", src/solid_system.C:183-190)
      std::map<subdomain_id_type, int32_t> index;
      std::vector<rdc_solid_material> table;
      std::vector<int32_t> em(ne);
      for (size_t e = 0; e < ne; e++) {
        const subdomain_id_type id = mesh.elem_ref(B.elem_ids[e]).subdomain_id();
        auto it = index.find(id);
        if (it == index.end()) {
          const std::string k = "material/" + std::to_string(id) + "/Hyperelastic/";
          rdc_solid_material m;
          m.Young = es.parameters.get<Real>(k + "Young"); m.Poisson = es.parameters.get<Real>(k + "Poisson");
          m.FibreStiffness = es.parameters.get<Real>(k + "FibreStiffness");
          for (int d = 0; d < 3; d++) m.rate[d] = es.parameters.get<Real>(k + "VolumetricStretchRatio/rate_" + std::to_string(d));
          it = index.emplace(id, (int32_t)table.size()).first;
          table.push_back(m);
        }
        em[e] = it->second;
      }
      if (rdc_solid_set_materials(B.ctx, em.data(), (int32_t)table.size(), table.data()) != RDC_OK) fail(B.ctx, "rdc_solid_set_materials");
      // boundary sides whose id is in "BCs" (src/solid_system.C:294-306) of EVERY element of the binding, ghost layer
      // included: the library adds a side's penalty terms only into rows of nodes this rank owns, so each rank ends up
      // with the complete rows of its nodes (a side next to the partition boundary is listed on both ranks)
      const std::set<int> bcs = export_integers(es.parameters.get<std::string>("BCs"));
      std::vector<int64_t> se;
      std::vector<int32_t> si;
      std::vector<double> sd;
      for (int bc : bcs) {
        const Point u = es.parameters.get<Point>("BC/" + std::to_string(bc) + "/displacement");
        for (size_t e = 0; e < ne; e++) {
          const Elem& elem = mesh.elem_ref(B.elem_ids[e]);
          for (auto s : elem.side_index_range())
            if (mesh.get_boundary_info().has_boundary_id(&elem, s, cast_int<boundary_id_type>(bc))) {
              se.push_back((int64_t)e); si.push_back((int32_t)s);
              for (int d = 0; d < 3; d++) sd.push_back(u(d));
            }
        }
      }
      if (rdc_solid_set_sides(B.ctx, (int64_t)se.size(), se.data(), si.data(), sd.data()) != RDC_OK) fail(B.ctx, "rdc_solid_set_sides");
      solid_bound_ = true;
    }
    // current node positions = the unknowns of the current Newton iterate (FEMContext::pre_fe_reinit moves the element's
    // nodes there); undeformed positions; reference fibre
    std::vector<double> x(3 * nn), X(3 * nn), eta(3 * ne);
    for (size_t l = 0; l < nn; l++) {
      const Node& nd = mesh.node_ref(B.local_to_global_node[l]);
      for (unsigned int d = 0; d < 3; d++) {
        x[3 * l + d] = this->current_solution(nd.dof_number(this->number(), this->var[d], 0));
        X[3 * l + d] = aux.current_solution(nd.dof_number(aux.number(), this->undefo_var[d], 0));   // :221-229
      }
    }
    for (size_t e = 0; e < ne; e++) {
      const Elem& elem = mesh.elem_ref(B.elem_ids[e]);
      for (unsigned int d = 0; d < 3; d++) eta[3 * e + d] = fibre.current_solution(elem.dof_number(fibre.number(), d, 0));   // :204-216
    }
    if (rdc_mesh_update_coords(B.ctx, x.data()) != RDC_OK) fail(B.ctx, "rdc_mesh_update_coords");
    if (rdc_field_upload(B.ctx, RDC_FIELD_UNDEFORMED_XYZ, X.data(), (int64_t)X.size()) != RDC_OK) fail(B.ctx, "rdc_field_upload(undeformed)");
    if (rdc_field_upload(B.ctx, RDC_FIELD_ELEM_FIBRE, eta.data(), (int64_t)eta.size()) != RDC_OK) fail(B.ctx, "rdc_field_upload(fibre)");
    rdc_solid_params p;
    p.pseudo_time = es.parameters.get<Real>("pseudo_time");
    p.displacement_penalty = es.parameters.get<Real>("BCs/displacement_penalty");
    p.use_symmetry = es.parameters.get<bool>("solver/assembly_use_symmetry") ? 1 : 0;
    p._pad = 0;
    if (rdc_solid_assemble(B.ctx, &p, get_jacobian ? 1 : 0) != RDC_OK) fail(B.ctx, "rdc_solid_assemble");
    if (rdc_csr_download(B.ctx, get_jacobian ? B.val.data() : nullptr, B.rhs.data()) != RDC_OK) fail(B.ctx, "rdc_csr_download");
    // owned rows -> system.matrix / system.rhs (FEMSystem::assembly zeroes them first; complete rows: no stash traffic)
    if (get_residual) this->rhs->zero();
    if (get_jacobian) this->matrix->zero();
    Mat A = get_jacobian ? cast_ref<PetscMatrix<Number>&>(*this->matrix).mat() : nullptr;
    for (dof_id_type l = 0; l < B.n_owned; l++) {
      const Node& nd = mesh.node_ref(B.local_to_global_node[l]);
      for (unsigned int a = 0; a < 3; a++) {
        const PetscInt row = (PetscInt)nd.dof_number(this->number(), this->var[a], 0);
        const PetscInt r = (PetscInt)(l * 3 + a), b = B.row_ptr[r], n = B.row_ptr[r + 1] - b;
        if (get_jacobian) MatSetValues(A, 1, &row, n, &B.col_glob[b], &B.val[b], INSERT_VALUES);
        if (get_residual) this->rhs->set(row, B.rhs[r]);
      }
    }
    if (get_residual) this->rhs->close();
    if (get_jacobian) this->matrix->close();
  }

 private:
  bool solid_bound_ = false;
};

}  // namespace rdc_gpu
