// Test driver of the SolidSystem side of the C++ host mirror (rdcfes_amd/host/rdc_host.h): the call order of the
// reference's coupled driver, src/coupled_hcc.C:28-130 (and of src/solid.C:27-109 when no HCC step is wanted):
//
//   es.add_system<TransientLinearImplicitSystem>("HCC") + attach_assemble_function(assemble_hcc)
//   es.add_system<SolidSystem>("SolidSystem"), "::auxiliary", "::displacement", "::fibre", "::pressure", "::von_mises"
//   es.init();  model_sb.save_initial_mesh();
//   for t: time += dt; if loading step: pseudo_time += deltat
//          shift old/older; model_rds.solve();  check_solution(es);          <- assemble_hcc on the CURRENT mesh
//          if loading step: model_sb.run_solver(); post_process(); update_data();   <- Newton: rdc_solid_assemble per iteration
//
// Reads a case written by tests/test_gpu_solid_mirror.py, writes the states the test checks against the oracle.
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../rdcfes_amd/host/rdc_host.h"

using namespace rdc::host;

template <class T> std::vector<T> read_raw(const std::string& f) {
  std::ifstream in(f, std::ios::binary | std::ios::ate);
  if (!in) throw std::runtime_error("cannot open " + f);
  const std::streamsize n = in.tellg();
  in.seekg(0);
  std::vector<T> v((size_t)n / sizeof(T));
  in.read(reinterpret_cast<char*>(v.data()), n);
  return v;
}
template <class T> void write_raw(const std::string& f, const std::vector<T>& v) {
  std::ofstream out(f, std::ios::binary);
  out.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: driver <dir> <elem_type>\n"); return 2; }
  const std::string dir = argv[1];
  const int elem_type = std::atoi(argv[2]);
  try {
    Mesh mesh(elem_type, read_raw<uint32_t>(dir + "/conn.bin"), read_raw<double>(dir + "/xyz.bin"));
    mesh.set_subdomain_ids(read_raw<int32_t>(dir + "/subdomain.bin"));
    {
      const std::vector<int64_t> s = read_raw<int64_t>(dir + "/sides.bin");   // [n][3] = elem, side, boundary id
      for (size_t k = 0; k + 2 < s.size(); k += 3) mesh.add_side(s[k], (int32_t)s[k + 1], (int32_t)s[k + 2]);
    }
    EquationSystems es(mesh);
    {  // input(): GetPot key/value -> es.parameters with the types of src/coupled_hcc.C:144-373 / src/solid.C:114-283
      std::ifstream in(dir + "/params.txt");
      std::string type, key, val;
      while (in >> type >> key) {
        std::getline(in, val);
        while (!val.empty() && val.front() == ' ') val.erase(val.begin());
        if (type == "real") es.parameters.set<Real>(key) = std::atof(val.c_str());
        else if (type == "int") es.parameters.set<int>(key) = std::atoi(val.c_str());
        else if (type == "bool") es.parameters.set<bool>(key) = (val == "true" || val == "1");
        else if (type == "string") es.parameters.set<std::string>(key) = val;
        else if (type == "point") {
          Point p;
          std::stringstream ss(val);
          for (int d = 0; d < 3; d++) { std::string w; ss >> w; p(d) = (w == "nan" || w == "NAN") ? std::nan("") : std::atof(w.c_str()); }
          es.parameters.set<Point>(key) = p;
        } else throw std::runtime_error("params.txt: unknown type " + type);
      }
    }
    const bool with_rd = es.parameters.get<bool>("test/with_rd");

    TransientLinearImplicitSystem* model_rds = nullptr;
    if (with_rd) {
      model_rds = &es.add_system<TransientLinearImplicitSystem>("HCC");
      for (const char* v : {"l", "c", "n"}) model_rds->add_variable(v);
      model_rds->attach_assemble_function(assemble_hcc);
    }
    SolidSystem& model_sb = es.add_system<SolidSystem>("SolidSystem");
    for (const char* v : {"x", "y", "z"}) model_sb.add_variable(v);
    TransientExplicitSystem& aux_sys = es.add_system<TransientExplicitSystem>("SolidSystem::auxiliary");
    for (const char* v : {"undeformed_x", "undeformed_y", "undeformed_z"}) aux_sys.add_variable(v);
    ExplicitSystem& disp_sys = es.add_system<ExplicitSystem>("SolidSystem::displacement");
    for (const char* v : {"u_x", "u_y", "u_z"}) disp_sys.add_variable(v);
    ExplicitSystem& fibre_sys = es.add_system<ExplicitSystem>("SolidSystem::fibre");
    fibre_sys.elemental = true;
    for (const char* v : {"fibre_reference_x", "fibre_reference_y", "fibre_reference_z", "fibre_current_x", "fibre_current_y", "fibre_current_z"})
      fibre_sys.add_variable(v);
    ExplicitSystem& press_sys = es.add_system<ExplicitSystem>("SolidSystem::pressure");
    press_sys.elemental = true;
    press_sys.add_variable("p");
    ExplicitSystem& von_mises_sys = es.add_system<ExplicitSystem>("SolidSystem::von_mises");
    von_mises_sys.elemental = true;
    von_mises_sys.add_variable("VM");

    es.init();
    {  // initial_fibres(): the reference fibre of every element
      const std::vector<double> f = read_raw<double>(dir + "/fibre.bin");   // [n_elem][3]
      for (int64_t e = 0; e < mesh.n_elem(); e++)
        for (int d = 0; d < 3; d++) fibre_sys.solution.set(e * 6 + d, f[(size_t)e * 3 + d]);
      fibre_sys.update();
    }
    if (with_rd) {  // initial_hcc()
      model_rds->solution.raw() = read_raw<double>(dir + "/u.bin");
      model_rds->update();
    }
    model_sb.save_initial_mesh();

    const std::set<int> ltp = export_integers(es.parameters.get<std::string>("loading_time_points"));
    es.parameters.set<Real>("time") = 0.0;
    es.parameters.set<Real>("pseudo_time") = 0.0;
    const int n_time_step = es.parameters.get<int>("number_of_time_steps");
    std::ofstream log(dir + "/log.txt");
    log.precision(17);
    for (int t = 1; t <= n_time_step; t++) {
      es.parameters.set<Real>("time") += es.parameters.get<Real>("time_step");
      const bool loading = ltp.end() != ltp.find(t);
      if (loading) es.parameters.set<Real>("pseudo_time") += model_sb.deltat;
      if (with_rd) {
        model_rds->time = es.parameters.get<Real>("time");
        model_rds->older_local_solution = model_rds->old_local_solution;
        model_rds->old_local_solution = model_rds->current_local_solution;
        write_raw(dir + "/rd_xyz_" + std::to_string(t) + ".bin", mesh.coordinates());     // the mesh assemble_hcc will see
        write_raw(dir + "/rd_old_" + std::to_string(t) + ".bin", model_rds->old_local_solution.raw());
        const int its = model_rds->solve(1e-13, 5000);
        write_raw(dir + "/rd_val_" + std::to_string(t) + ".bin", model_rds->matrix->val);
        write_raw(dir + "/rd_rhs_" + std::to_string(t) + ".bin", model_rds->rhs->raw());
        check_solution(es, "HCC");
        write_raw(dir + "/rd_sol_" + std::to_string(t) + ".bin", model_rds->solution.raw());
        log << "step " << t << " rd_linear_iterations " << its << "\n";
      }
      if (loading) {
        // the very first Newton assembly of this loading step, for the parity check of J and R against the oracle
        model_sb.assembly(true, true);
        write_raw(dir + "/sb_val_" + std::to_string(t) + ".bin", model_sb.matrix->val);
        write_raw(dir + "/sb_rhs_" + std::to_string(t) + ".bin", model_sb.rhs->raw());
        write_raw(dir + "/sb_xyz_" + std::to_string(t) + ".bin", model_sb.current_local_solution.raw());
        model_sb.run_solver();
        model_sb.post_process();
        model_sb.update_data();
        log << "step " << t << " pseudo_time " << es.parameters.get<Real>("pseudo_time") << " newton_iterations " << model_sb.last.nonlinear_iterations
            << " linear_iterations " << model_sb.last.linear_iterations << " assemblies " << model_sb.last.assemblies << " first_residual "
            << model_sb.last.first_residual << " last_residual " << model_sb.last.last_residual << " converged " << (model_sb.last.converged ? 1 : 0) << "\n";
        write_raw(dir + "/sb_sol_" + std::to_string(t) + ".bin", model_sb.solution.raw());
        write_raw(dir + "/sb_disp_" + std::to_string(t) + ".bin", disp_sys.solution.raw());
        write_raw(dir + "/sb_press_" + std::to_string(t) + ".bin", press_sys.solution.raw());
        write_raw(dir + "/sb_vm_" + std::to_string(t) + ".bin", von_mises_sys.solution.raw());
        write_raw(dir + "/sb_fibre_" + std::to_string(t) + ".bin", fibre_sys.solution.raw());
      }
    }
    write_raw(dir + "/row_ptr.bin", model_sb.matrix->row_ptr);
    write_raw(dir + "/col_idx.bin", model_sb.matrix->col_idx);
    write_raw(dir + "/aux_old.bin", aux_sys.old_local_solution.raw());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "driver failed: %s\n", e.what());
    return 1;
  }
  return 0;
}
