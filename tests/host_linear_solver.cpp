// CPU-only check of the host mirror's linear-solver stand-in (bicgstab_ilu0 of rdcfes_amd/host/rdc_host.h):
// reads a CSR system written by tests/test_host_linear_solver.py, solves it, writes the solution.
#include <cstdio>
#include <fstream>

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

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string dir = argv[1];
  try {
    SparseMatrix A;
    A.row_ptr = read_raw<int64_t>(dir + "/row_ptr.bin");
    A.col_idx = read_raw<int32_t>(dir + "/col_idx.bin");
    A.val = read_raw<double>(dir + "/val.bin");
    const std::vector<double> b = read_raw<double>(dir + "/b.bin");
    std::vector<double> x(b.size(), 0.0);
    const int its = bicgstab_ilu0(A, b, x, std::atof(argv[2]), 20000);
    std::ofstream out(dir + "/x.bin", std::ios::binary);
    out.write(reinterpret_cast<const char*>(x.data()), (std::streamsize)(x.size() * sizeof(double)));
    std::printf("%d\n", its);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "failed: %s\n", e.what());
    return 1;
  }
  return 0;
}
