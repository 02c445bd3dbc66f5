// Test driver of the C++ host mirror (rdcfes_amd/host/rdc_host.h): reads a case written by
// tests/test_gpu_host_mirror.py, runs the reference-shaped call sequence
//     es.add_system<TransientLinearImplicitSystem>("PIHNA"); add_variable x5;
//     model.attach_assemble_function(assemble_pihna); es.init(); model.assemble() / model.solve()
// (src/pihna.C:28-48, :77-80) and writes the assembled CSR values and rhs back.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

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
  if (argc < 4) { std::fprintf(stderr, "usage: driver <dir> <model: pihna|ripf|hcc|adpm|proteas> <elem_type> [solve]\n"); return 2; }
  const std::string dir = argv[1], model_name = argv[2];
  const int elem_type = std::atoi(argv[3]);
  const bool do_solve = argc > 4 && std::string(argv[4]) == "solve";
  try {
    Mesh mesh(elem_type, read_raw<uint32_t>(dir + "/conn.bin"), read_raw<double>(dir + "/xyz.bin"));
    EquationSystems es(mesh);
    {  // input(): GetPot key/value -> es.parameters (src/pihna.C:98-237)
      std::ifstream in(dir + "/params.txt");
      std::string key, val;
      while (in >> key >> val) {
        if (key == "RT_dose/total/max" || key == "rdc/handback_chunks") es.parameters.set<int>(key) = std::atoi(val.c_str());
        else es.parameters.set<Real>(key) = std::atof(val.c_str());
      }
    }
    TransientLinearImplicitSystem* model = nullptr;
    if (model_name == "pihna") {
      model = &es.add_system<TransientLinearImplicitSystem>("PIHNA");
      for (const char* v : {"n", "c", "h", "v", "a"}) model->add_variable(v);
      model->attach_assemble_function(assemble_pihna);
    } else if (model_name == "ripf") {
      model = &es.add_system<TransientLinearImplicitSystem>("RIPF");
      for (const char* v : {"HU", "cc", "fb"}) model->add_variable(v);
      model->attach_assemble_function(assemble_ripf);
      System& td = es.add_system<System>("RIPF-TimeDeriv");
      for (const char* v : {"HU_TimeDeriv", "cc_TimeDeriv", "fb_TimeDeriv"}) td.add_variable(v);
      System& rt = es.add_system<System>("RT");
      for (const char* v : {"RT_dose/broad", "RT_dose/focus", "RT_dose/total"}) rt.add_variable(v);
    } else if (model_name == "hcc") {
      model = &es.add_system<TransientLinearImplicitSystem>("HCC");
      for (const char* v : {"l", "c", "n"}) model->add_variable(v);
      model->attach_assemble_function(assemble_hcc);
    } else if (model_name == "adpm") {
      model = &es.add_system<TransientLinearImplicitSystem>("ADPM");
      for (const char* v : {"PrP", "A_b", "Tau"}) model->add_variable(v);
      model->attach_assemble_function(assemble_adpm);
      System& tr = es.add_system<System>("Tracts");
      tr.elemental = true;
      for (const char* v : {"tract_x", "tract_y", "tract_z"}) tr.add_variable(v);
    } else if (model_name == "proteas") {
      model = &es.add_system<TransientLinearImplicitSystem>("PROTEAS_model");
      for (const char* v : {"hos", "tum", "nec", "vsc", "oed"}) model->add_variable(v);
      model->attach_assemble_function(assemble_proteas_model);
      System& aux = es.add_system<System>("AUX");
      for (const char* v : {"HU", "RTD"}) aux.add_variable(v);
    } else {
      throw std::runtime_error("unknown model " + model_name);
    }
    es.init();
    model->current_local_solution.raw() = read_raw<double>(dir + "/u.bin");
    model->solution = model->current_local_solution;
    if (model_name == "ripf") {
      es.get_system<System>("RIPF-TimeDeriv").current_local_solution.raw() = read_raw<double>(dir + "/td.bin");
      es.get_system<System>("RT").current_local_solution.raw() = read_raw<double>(dir + "/rt.bin");
    }
    if (model_name == "adpm") {
      es.get_system<System>("Tracts").solution.raw() = read_raw<double>(dir + "/tracts.bin");
      model->time = es.parameters.get<Real>("time");
    }
    if (model_name == "proteas") es.get_system<System>("AUX").current_local_solution.raw() = read_raw<double>(dir + "/aux2.bin");
    // the time-loop prologue, src/pihna.C:77-78
    model->older_local_solution = model->old_local_solution;
    model->old_local_solution = model->current_local_solution;
    if (do_solve) {
      const int its = model->solve(1e-12, 2000);
      std::vector<double> Ax;
      model->matrix->vector_mult(Ax, model->solution.raw());
      double rn = 0, bn = 0;
      for (size_t i = 0; i < Ax.size(); i++) { const double r = Ax[i] - (*model->rhs)((int64_t)i); rn += r * r; bn += (*model->rhs)((int64_t)i) * (*model->rhs)((int64_t)i); }
      std::printf("solve: %d iterations, relative residual %.3e\n", its, std::sqrt(rn / bn));
      write_raw(dir + "/solution.bin", model->solution.raw());
    } else {
      model->assemble();
    }
    {  // check_solution(): the clamp of the solved state runs on the device through the same context
      std::vector<double> keep = model->solution.raw();
      for (size_t i = 0; i < model->solution.raw().size(); i += 7) model->solution.raw()[i] = -1.0 - (double)i;
      std::vector<double> expect = model->solution.raw();
      for (double& x : expect) if (x < 0.0) x = 0.0;
      check_solution(es, model->name());
      if (model->solution.raw() != expect) throw std::runtime_error("check_solution: clamp mismatch");
      model->solution.raw() = keep;
    }
    // save_solution(): the CSV line of the model (src/pihna.C:59, src/ripf.C:63), volume sums computed on the device
    if (model_name == "pihna" && es.parameters.have_parameter<Real>("range/active_tumor/min")) {
      std::ofstream csv(dir + "/out.csv");
      csv.precision(17);
      save_solution_pihna(csv, es);
    }
    if (model_name == "ripf" && es.parameters.have_parameter<Real>("range_cc/min")) {
      std::ofstream csv(dir + "/out.csv");
      csv.precision(17);
      save_solution_ripf(csv, es);
    }
    if (!model->handback_log.empty()) write_raw(dir + "/handback_log.bin", model->handback_log);
    write_raw(dir + "/val.bin", model->matrix->val);
    write_raw(dir + "/rhs.bin", model->rhs->raw());
    write_raw(dir + "/row_ptr.bin", model->matrix->row_ptr);
    write_raw(dir + "/col_idx.bin", model->matrix->col_idx);
    // error behaviour: a missing parameter is reported, not silently defaulted
    try { es.parameters.get<Real>("no/such/key"); return 3; } catch (const std::runtime_error&) {}
  } catch (const std::exception& e) {
    std::fprintf(stderr, "driver failed: %s\n", e.what());
    return 1;
  }
  return 0;
}
