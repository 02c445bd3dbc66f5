// ref_eig3_wrap.cpp — TEST INFRASTRUCTURE.  C-linkage door to the reference's OWN eigen-decomposition routine
// (src/eig3.C:261-271, `eigen_decomposition`, the one function SolidSystem::post_process calls at
// src/solid_system.C:514).  src/eig3.C needs nothing but <cmath>, so it is compiled where it lies under /root/reference
// (oracle/Makefile, target _ref/libref_eig3.so; never copied into this repository) together with this file, and the CPU
// tests pin the oracle's restatement of that step (rdc_oracle.c: sym3_eigenvalues, the pressure / von Mises formulas)
// against it.  Nothing of the product links, loads or ships this library.
void eigen_decomposition(double A[3][3], double V[3][3], double d[3]);   // defined in the reference's src/eig3.C

extern "C" void ref_eigen_decomposition(const double* A9, double* V9, double* d3) {
  double A[3][3], V[3][3], d[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = A9[3 * i + j];
  eigen_decomposition(A, V, d);
  for (int i = 0; i < 3; i++) {
    d3[i] = d[i];
    for (int j = 0; j < 3; j++) V9[3 * i + j] = V[i][j];
  }
}
