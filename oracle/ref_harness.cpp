// ref_harness.cpp — builds the REAL reference (mishraiiit/MultiGridSolver) CPU path
// from the sources where they lie under $REFERENCE_DIR (default /root/reference) into
// oracle/_ref/.  TEST INFRASTRUCTURE ONLY: it pins oracle/mgs_oracle.c and generates
// tests/golden/*.  Nothing from the reference is copied into this repository; this TU
// only #includes the reference's translation units (src/common/bicg.cpp which pulls in
// src/common/MatrixIO.cpp, and src/CPU_C++/AGMG.cpp) and the vendored Eigen 3.3.4
// headers (lib/Eigen).
//
// Two build modes (see oracle/Makefile):
//   -DREF_DUMP_MAIN   → oracle/_ref/ref_dump   (golden-vector generator, run here only)
//   -DREF_EIGEN_SO    → oracle/_ref/libref_eigen.so (Eigen SpMV for bench.py's
//                        cpu_baseline kind="reference"; travels to the GPU box as a
//                        prebuilt .so, never reads /root/reference at run time)
#define main ref_bicg_main
#include "src/common/bicg.cpp"   // MultiGridPrecond, BiCGSTABiml, readMatrix, writeMatrix
#undef main
#include "src/CPU_C++/AGMG.cpp"   // AGMG::multiple_pairwise_aggregation

#include <cstdint>
#include <cstdio>

#ifdef REF_EIGEN_SO
extern "C" {
// y = A*x with the reference's own kernel (Eigen row-major sparse × dense,
// lib/Eigen/src/SparseCore/SparseDenseProduct.h:26-71) on caller-owned CSR arrays,
// mapped zero-copy.  Returns the mean seconds per product over `reps` (after one
// warm-up), mirroring how the reference uses it (`A * v`, bicg.cpp:107).
double ref_eigen_spmv(int rows, int cols, int nnz, const int *rowptr, const int *col,
                      const double *val, const double *x, double *y, int reps) {
  Eigen::Map<const SparseMatrix<double, RowMajor, int> > A(rows, cols, nnz, rowptr, col, val);
  Eigen::Map<const VectorXd> xv(x, cols);
  Eigen::Map<VectorXd> yv(y, rows);
  yv.noalias() = A * xv;
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) yv.noalias() = A * xv;
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count() / (reps > 0 ? reps : 1);
}
int ref_eigen_threads() { return Eigen::nbThreads(); }
void ref_eigen_set_threads(int n) { Eigen::setNbThreads(n); }
int ref_eigen_version() { return EIGEN_WORLD_VERSION * 10000 + EIGEN_MAJOR_VERSION * 100 + EIGEN_MINOR_VERSION; }
}
#endif

#ifdef REF_DUMP_MAIN
static void dump_f64(const std::string &p, const double *d, size_t n) {
  FILE *f = fopen(p.c_str(), "wb"); fwrite(d, 8, n, f); fclose(f);
}
static void dump_i32(const std::string &p, const int *d, size_t n) {
  FILE *f = fopen(p.c_str(), "wb"); fwrite(d, 4, n, f); fclose(f);
}
static void dump_vec(const std::string &dir, const char *name, const VectorXd &v) {
  dump_f64(dir + "/" + name + ".f64", v.data(), (size_t)v.size());
}
static void dump_csr(const std::string &dir, const char *name, SMatrix m) {
  m.makeCompressed();
  dump_i32(dir + "/" + name + ".rowptr.i32", m.outerIndexPtr(), (size_t)m.rows() + 1);
  dump_i32(dir + "/" + name + ".col.i32", m.innerIndexPtr(), (size_t)m.nonZeros());
  dump_f64(dir + "/" + name + ".val.f64", m.valuePtr(), (size_t)m.nonZeros());
  FILE *f = fopen((dir + "/" + name + ".shape.txt").c_str(), "w");
  fprintf(f, "%d %d %d\n", (int)m.rows(), (int)m.cols(), (int)m.nonZeros()); fclose(f);
}

// Derived oracle: the reference's two-grid operator with the
// smoother M2 swapped from ILUT to ωD⁻¹ (solve.m:17), built from the SAME Eigen objects
// as bicg.cpp:29-61.
struct JacobiTwoGrid {
  SMatrix A, P, Pt; SparseLU<SparseMatrix<double> > lu; VectorXd dinv; double omega;
  JacobiTwoGrid(const SMatrix &A_, const SMatrix &P_, double w) : A(A_), P(P_), omega(w) {
    Pt = P.transpose();
    const SMatrix Ac = Pt * A * P;                      // bicg.cpp:33
    lu.analyzePattern(Ac); lu.factorize(Ac);            // bicg.cpp:35-36
    dinv = A.diagonal().cwiseInverse();
  }
  bool additive = false;
  template <typename T> T solve(const T &v) const {
    T res = P * (lu.solve(Pt * v));                     // bicg.cpp:48
    if (additive) return res + omega * dinv.cwiseProduct(v);   // bicg.cpp:59 with M2 = ωD⁻¹
    T r = v - A * res;
    return res + omega * dinv.cwiseProduct(r);          // bicg.cpp:57 with M2 = ωD⁻¹
  }
};
struct IdentityPrecond { template <typename T> T solve(const T &v) const { return v; } };

static VectorXd ref_rhs(int n) {                        // bicg.cpp:139,159-162
  srand(0);
  VectorXd b(n);
  for (int i = 0; i < n; i++) b[i] = rand() / (RAND_MAX + 0.0);
  return b;
}

// usage: ref_dump case <A.mtx> <P.mtx> <outdir>     → hot-path golden vectors
//        ref_dump agmg <A.mtx> <ktg> <npass> <tou> <out_P.mtx> <outdir>
//        ref_dump small <A.mtx> <outdir>
int main(int argc, char **argv) {
  std::string mode = argc > 1 ? argv[1] : "";
  if (mode == "small" && argc == 4) {
    SMatrix A = readMatrix(argv[2]); std::string out = argv[3];
    dump_csr(out, "A", A);
    VectorXd v(A.cols()); for (int i = 0; i < A.cols(); i++) v[i] = i + 1;
    VectorXd w(A.rows()); for (int i = 0; i < A.rows(); i++) w[i] = i + 1;
    dump_vec(out, "A_times_1toN", A * v);
    SMatrix At = A.transpose();
    dump_csr(out, "At", At);
    dump_vec(out, "At_times_1toM", At * w);
    SMatrix AAt = A * At;
    dump_csr(out, "AAt", AAt);
    return 0;
  }
  if (mode == "agmg" && argc == 8) {
    SMatrix A = readMatrix(argv[2]);
    SMatrix P = AGMG::multiple_pairwise_aggregation(A, atof(argv[3]), atoi(argv[4]), atof(argv[5]), 0);
    writeMatrix(argv[6], P);
    std::string out = argv[7];
    std::vector<int> groups(A.rows(), -1);
    for (int i = 0; i < P.rows(); i++)
      for (SMatrix::InnerIterator it(P, i); it; ++it) groups[i] = it.index();
    dump_i32(out + "/groups.i32", groups.data(), groups.size());
    FILE *f = fopen((out + "/P.shape.txt").c_str(), "w");
    fprintf(f, "%d %d %d\n", (int)P.rows(), (int)P.cols(), (int)P.nonZeros()); fclose(f);
    return 0;
  }
  if (mode == "case" && argc == 5) {
    SMatrix A = readMatrix(argv[2]), P = readMatrix(argv[3]); std::string out = argv[4];
    const int n = A.rows();
    VectorXd b = ref_rhs(n);
    dump_vec(out, "b", b);
    dump_vec(out, "A_b", A * b);                                   // (ii) bicg.cpp:107
    SMatrix Pt = P.transpose();                                    // bicg.cpp:32
    VectorXd rc = Pt * b;
    dump_vec(out, "Pt_b", rc);                                     // (iii) bicg.cpp:48
    dump_vec(out, "P_Pt_b", P * rc);                               // (iv) bicg.cpp:48
    SMatrix Ac = Pt * A * P;                                       // (v) bicg.cpp:33
    dump_csr(out, "Ac", Ac);
    {
      MultiGridPrecond M(A, P);
      dump_vec(out, "mg_solve_b", M.multigrid_solve(b));           // a5, bicg.cpp:46-49
      dump_vec(out, "M_solve_b", M.solve(b));                      // (vi) bicg.cpp:51-61 (ILUT)
      VectorXd x = VectorXd::Zero(n); int it = 10000; double tol = 1e-6;
      int st = BiCGSTABiml(A, x, b, M, it, tol);                   // (vii) as shipped
      FILE *f = fopen((out + "/bicg_ilut.txt").c_str(), "w");
      fprintf(f, "%d %d %.17g\n", st, it, tol); fclose(f);
      dump_vec(out, "x_bicg_ilut", x);
    }
    for (double w : {0.5, 0.8}) {
      JacobiTwoGrid J(A, P, w);
      char nm[64];
      snprintf(nm, sizeof nm, "jac2grid_w%02d_b", (int)(w * 10 + 0.5));
      dump_vec(out, nm, J.solve(b));                               // derived oracle, eq. (3.5)
      VectorXd x = VectorXd::Zero(n); int it = 10000; double tol = 1e-10;
      int st = BiCGSTABiml(A, x, b, J, it, tol);
      snprintf(nm, sizeof nm, "/bicg_jac_w%02d.txt", (int)(w * 10 + 0.5));
      FILE *f = fopen((out + nm).c_str(), "w");
      fprintf(f, "%d %d %.17g\n", st, it, tol); fclose(f);
      snprintf(nm, sizeof nm, "x_bicg_jac_w%02d", (int)(w * 10 + 0.5));
      dump_vec(out, nm, x);
      if (w == 0.5) {                                              // the additive switch of solve(), bicg.cpp:59
        J.additive = true;
        dump_vec(out, "jac2grid_add_w05_b", J.solve(b));
      }
    }
    {
      IdentityPrecond I; VectorXd x = VectorXd::Zero(n); int it = 10000; double tol = 1e-8;
      int st = BiCGSTABiml(A, x, b, I, it, tol);
      FILE *f = fopen((out + "/bicg_identity.txt").c_str(), "w");
      fprintf(f, "%d %d %.17g\n", st, it, tol); fclose(f);
      dump_vec(out, "x_bicg_identity", x);
    }
    return 0;
  }
  fprintf(stderr, "usage: ref_dump case|agmg|small ...\n");
  return 2;
}
#endif
