// mgs_host.hpp — the C++ face of libmgs.so that keeps the reference's spelling, written on top
// of the C ABI (include/mgs.h) only.  A maintainer of mishraiiit/MultiGridSolver swaps
//     #include "../common/MatrixIO.cpp"   →   #include "mgs_host.hpp" + `using namespace mgs;`
// in src/common/bicg.cpp and keeps `readMatrix`, `SMatrix`, `MultiGridPrecond precond(A, P)`,
// `BiCGSTABiml(A, x, b, precond, max_iter, tol)` as they are.
//
// Reference interfaces mirrored (file:line relative to the reference checkout):
//   SMatrix / readMatrix / writeMatrix        src/common/MatrixIO.cpp:10,12-37,39-57
//   MultiGridPrecond(A,P), solve(v)           src/common/bicg.cpp:19-62
//   dot / norm / BiCGSTABiml                  src/common/bicg.cpp:64-136
//   TicToc / printScreen                      src/CPU_C++/TicToc.cpp:18-53
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mgs.h"

namespace mgs {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error("libmgs error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc, const mgs_ctx *ctx = nullptr) {
  if (rc != MGS_OK) { const char *m = mgs_last_error(ctx); throw Error(rc, m ? m : "?"); }
}

// one process-wide context (the reference is single-device, single-thread: SURVEY §0 G5)
inline mgs_ctx *context() {
  static mgs_ctx *ctx = [] {
    mgs_ctx *c = nullptr;
    const char *dev = std::getenv("MGS_DEVICE");
    check(mgs_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &c));
    return c;
  }();
  return ctx;
}

// ---------------------------------------------------------------- host CSR: `SMatrix`
// Eigen::SparseMatrix<double,RowMajor> stand-in (MatrixIO.cpp:10): f64 values, int32 indices,
// sorted columns.  Only the members the reference's drivers use are provided.
struct SMatrix {
  int m_rows = 0, m_cols = 0;
  std::vector<int> rowptr{0}, col;
  std::vector<double> val;
  int rows() const { return m_rows; }
  int cols() const { return m_cols; }
  int nonZeros() const { return (int)col.size(); }
};

inline SMatrix readMatrix(std::string filename) {          // MatrixIO.cpp:12-37
  SMatrix A; int nnz = 0; int *rp = nullptr, *ci = nullptr; double *v = nullptr;
  check(mgs_mtx_read(filename.c_str(), &A.m_rows, &A.m_cols, &nnz, &rp, &ci, &v));
  A.rowptr.assign(rp, rp + A.m_rows + 1); A.col.assign(ci, ci + nnz); A.val.assign(v, v + nnz);
  mgs_host_free(rp); mgs_host_free(ci); mgs_host_free(v);
  return A;
}
inline void writeMatrix(std::string filename, const SMatrix &A) {   // MatrixIO.cpp:39-57
  check(mgs_mtx_write(filename.c_str(), A.rows(), A.cols(), A.nonZeros(), A.rowptr.data(), A.col.data(), A.val.data()));
}

// ---------------------------------------------------------------- device vector: `VectorXd`
// Value semantics like Eigen::VectorXd (copies are deep); arithmetic runs on the device.
class Vector {
  std::shared_ptr<mgs_vec> v_;
  // Host staging for element access (`b[i] = …`, `x(0)`: bicg.cpp:78,159-162).  The first access downloads the vector once;
  // writes mark the mirror dirty and the next device use (handle()) uploads it.  Plain device arithmetic never touches it.
  mutable std::vector<double> host_;
  mutable bool host_valid_ = false, host_dirty_ = false;
  static std::shared_ptr<mgs_vec> make(int64_t n) {
    mgs_vec *p = nullptr; check(mgs_vec_create(context(), n, &p), context());
    return std::shared_ptr<mgs_vec>(p, [](mgs_vec *q) { mgs_vec_destroy(q); });
  }
  void flush() const {
    if (host_dirty_) { check(mgs_vec_upload(v_.get(), host_.data(), (int64_t)host_.size()), context()); host_dirty_ = false; }
  }
  double *stage() const {
    if (!host_valid_) { host_.resize((size_t)size()); if (size()) check(mgs_vec_download(v_.get(), host_.data(), size()), context()); host_valid_ = true; }
    return host_.data();
  }
 public:
  Vector() {}
  explicit Vector(int64_t n) : v_(make(n)) {}
  Vector(const Vector &o) { if (o.v_) { v_ = make(o.size()); check(mgs_vec_copy(o.handle(), v_.get()), context()); } }
  Vector(Vector &&) = default;
  Vector &operator=(const Vector &o) { if (this != &o) { Vector t(o); v_ = std::move(t.v_); host_valid_ = host_dirty_ = false; } return *this; }
  Vector &operator=(Vector &&) = default;
  int64_t size() const { return v_ ? mgs_vec_size(v_.get()) : 0; }
  int64_t rows() const { return size(); }
  // device handle for reading: pending element writes are uploaded first
  mgs_vec *handle() const { flush(); return v_.get(); }
  // device handle for writing: the host mirror no longer reflects the device
  mgs_vec *out() { flush(); host_valid_ = false; return v_.get(); }
  double &operator[](int64_t i) { double *h = stage(); host_dirty_ = true; return h[i]; }          // bicg.cpp:161
  double operator[](int64_t i) const { return stage()[i]; }
  double &operator()(int64_t i) { return (*this)[i]; }                                              // bicg.cpp:78 style access
  double operator()(int64_t i) const { return (*this)[i]; }
  void setZero() { check(mgs_vec_fill(out(), 0.0), context()); }
  void upload(const std::vector<double> &h) { check(mgs_vec_upload(out(), h.data(), (int64_t)h.size()), context()); }
  std::vector<double> download() const { std::vector<double> h((size_t)size()); check(mgs_vec_download(handle(), h.data(), size()), context()); return h; }
  double dot(const Vector &o) const { double s; check(mgs_dot(handle(), o.handle(), &s), context()); return s; }
  double norm() const { double s; check(mgs_nrm2(handle(), &s), context()); return s; }
  Vector &operator+=(const Vector &o) { check(mgs_axpby(1.0, o.handle(), 1.0, out()), context()); return *this; }
  Vector &operator-=(const Vector &o) { check(mgs_axpby(-1.0, o.handle(), 1.0, out()), context()); return *this; }
  friend Vector operator+(const Vector &a, const Vector &b) { Vector r(a.size()); check(mgs_axpbypcz(1.0, a.handle(), 1.0, b.handle(), 0.0, r.out()), context()); return r; }
  friend Vector operator-(const Vector &a, const Vector &b) { Vector r(a.size()); check(mgs_axpbypcz(1.0, a.handle(), -1.0, b.handle(), 0.0, r.out()), context()); return r; }
  friend Vector operator*(double s, const Vector &a) { Vector r(a.size()); check(mgs_axpby(s, a.handle(), 0.0, r.out()), context()); return r; }
  friend Vector operator*(const Vector &a, double s) { return s * a; }
};
typedef Vector VectorXd;

// ---------------------------------------------------------------- device operator: `A * v`
class DeviceMatrix {
  std::shared_ptr<mgs_csr> a_;
  int rows_ = 0, cols_ = 0;
 public:
  DeviceMatrix() {}
  DeviceMatrix(const SMatrix &A) : rows_(A.rows()), cols_(A.cols()) {
    mgs_csr *p = nullptr;
    check(mgs_csr_upload(context(), A.rows(), A.cols(), A.nonZeros(), A.rowptr.data(), A.col.data(), A.val.data(), &p), context());
    a_ = std::shared_ptr<mgs_csr>(p, [](mgs_csr *q) { mgs_csr_destroy(q); });
  }
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  mgs_csr *handle() const { return a_.get(); }
  Vector operator*(const Vector &x) const {            // bicg.cpp:57,82,107,117
    Vector y(rows_); check(mgs_spmv(a_.get(), x.handle(), y.out()), context()); return y;
  }
};

// ---------------------------------------------------------------- MultiGridPrecond (bicg.cpp:19-62)
// Same constructor and solve() shape.  The coarse direct solve + ILUT smoother of the reference
// are replaced by the V-cycle of this library: P gives level 1, further levels are aggregated on
// the device until the coarsest has ≤ coarse_rows rows (then inverted densely on the device);
// smoother = damped Jacobi (solve.m:17).  With levels == 2, nu1 == 0, nu2 == 1 solve() is exactly
// bicg.cpp:46-61 with M2 = ωD⁻¹.
struct PrecondOptions {
  double omega = 0.6; int nu1 = 1, nu2 = 1;
  double correction_scale = 1.0;                         // x ← x + σ·P e_c; 1 = the reference's form (bicg.cpp:48)
  double ktg = 10.0; int npass = 2; double tou = 8.0;   // src/GPU_CUDAC++/results.txt:22-24
  int coarse_rows = 2500; int max_levels = 32;   // ≤ 2500 rows: dense inverse (a GEMV beats two more latency-bound levels)
  // the two switches of the reference's solve() (bicg.cpp:42-43,53-59; both fixed to true there):
  bool multiplicative_precond = true;   // false: multigrid_solve(v) + M2(v) (:59) with M2 = ωD⁻¹
  bool use_preconditioner = true;       // false: solve(v) = v (:53-54)
};
class MultiGridPrecond {
  DeviceMatrix A_;
  std::shared_ptr<mgs_hier> h_;
  bool use_preconditioner_ = true;
 public:
  typedef PrecondOptions Options;
  MultiGridPrecond(const SMatrix &A_in, const SMatrix &P_in, Options o = Options()) : A_(A_in) { build(A_, &P_in, o); }
  MultiGridPrecond(const DeviceMatrix &A_dev, const SMatrix *P_in, Options o = Options()) : A_(A_dev) { build(A_, P_in, o); }
  template <typename T> T solve(const T &vec) const {     // bicg.cpp:51-61
    if (!use_preconditioner_) return vec;                  // :53-54
    T out(vec.size());
    check(mgs_vcycle(h_.get(), vec.handle(), out.out(), 1), context());
    return out;
  }
  bool use_preconditioner() const { return use_preconditioner_; }
  mgs_hier *handle() const { return h_.get(); }
  const DeviceMatrix &matrix() const { return A_; }
  int levels() const { return mgs_hier_nlev(h_.get()); }
 private:
  void build(const DeviceMatrix &A, const SMatrix *P, const Options &o) {
    mgs_hier *h = nullptr;
    check(mgs_hier_create(context(), A.handle(), o.omega, o.nu1, o.nu2, &h), context());
    h_ = std::shared_ptr<mgs_hier>(h, [](mgs_hier *q) { mgs_hier_destroy(q); });
    if (P) { DeviceMatrix Pd(*P); check(mgs_hier_push_P(h, Pd.handle()), context()); }
    check(mgs_hier_coarsen(h, o.ktg, o.npass, o.tou, o.coarse_rows, o.max_levels), context());
    check(mgs_hier_finalize(h), context());
    if (!o.multiplicative_precond) check(mgs_hier_set_additive(h, 1), context());
    if (o.correction_scale != 1.0) check(mgs_hier_set_correction_scale(h, o.correction_scale), context());
    use_preconditioner_ = o.use_preconditioner;
  }
};

// ---------------------------------------------------------------- bicg.cpp:64-72
template <typename V> inline double dot(const V &A, const V &B) { return A.dot(B); }
template <typename V> inline double norm(const V &A) { return A.norm(); }

// BiCGSTABiml, bicg.cpp:74-136: same signature, status codes (0 ok / 1 max_iter / 2 rho = 0 /
// 3 omega = 0) and write-back of max_iter / tol.  Generic form: works with any Matrix providing
// operator*(Vector) and any Preconditioner providing solve(Vector); scalars are plain Reals (the
// reference keeps them in 1-element Vectors, :78).
template <class Matrix, class Vec, class Preconditioner, class Real>
int BiCGSTABiml(const Matrix &A, Vec &x, const Vec &b, const Preconditioner &M, int &max_iter, Real &tol) {
  Real resid, rho_1 = 0, rho_2 = 0, alpha = 0, beta = 0, omega = 0;
  Vec p, phat, s, shat, t, v;
  Real normb = norm(b);
  Vec r = b - A * x;
  Vec rtilde = r;
  if (normb == 0.0) normb = 1;
  if ((resid = norm(r) / normb) <= tol) { tol = resid; max_iter = 0; return 0; }
  for (int i = 1; i <= max_iter; i++) {
    rho_1 = dot(rtilde, r);
    if (rho_1 == 0) { tol = norm(r) / normb; return 2; }
    if (i == 1) p = r;
    else { beta = (rho_1 / rho_2) * (alpha / omega); p = r + beta * (p - omega * v); }
    phat = M.solve(p);
    v = A * phat;
    alpha = rho_1 / dot(rtilde, v);
    s = r - alpha * v;
    if ((resid = norm(s) / normb) < tol) { x += alpha * phat; max_iter = i; tol = resid; return 0; }
    shat = M.solve(s);
    t = A * shat;
    omega = dot(t, s) / dot(t, t);
    x += alpha * phat + omega * shat;
    r = s - omega * t;
    rho_2 = rho_1;
    if ((resid = norm(r) / normb) < tol) { tol = resid; max_iter = i; return 0; }
    if (omega == 0) { tol = norm(r) / normb; return 3; }
  }
  tol = resid;
  return 1;
}
// Device-resident fast path for the library's own types (no temporaries, fused updates):
inline int BiCGSTABiml(const DeviceMatrix &A, Vector &x, const Vector &b, const MultiGridPrecond &M, int &max_iter, double &tol) {
  int status = -1;
  check(mgs_bicgstab(A.handle(), x.out(), b.handle(), M.use_preconditioner() ? M.handle() : nullptr, &max_iter, &tol, &status), context());
  return status;
}
// the call exactly as the reference's main() writes it (bicg.cpp:168): A is the host SMatrix the preconditioner was built
// from — its device copy lives in M
inline int BiCGSTABiml(const SMatrix &, Vector &x, const Vector &b, const MultiGridPrecond &M, int &max_iter, double &tol) {
  return BiCGSTABiml(M.matrix(), x, b, M, max_iter, tol);
}

// ---------------------------------------------------------------- TicToc.cpp:18-53
class TicToc {
  std::chrono::time_point<std::chrono::system_clock> start;
  std::string s; int level;
 public:
  TicToc(std::string s_, int level_) : s(s_), level(level_) {}
  void tic() { start = std::chrono::system_clock::now(); }
  double toc() {
    mgs_sync(context());
    std::chrono::duration<float> diff = std::chrono::system_clock::now() - start;
    std::string line = "\033[1;34m[time] \033[0m" + s;
    for (int i = 0; i < level; i++) fprintf(stderr, " ");
    while (line.size() < 60) line += ' ';
    fprintf(stderr, "%s : %lf.\n", line.c_str(), diff.count());
    return diff.count();
  }
};
template <typename T> void printScreen(const int level, std::string s, const T tm) {
  for (int i = 0; i < level; i++) std::cout << " ";
  std::cout << "\033[32m\033[1m[info] \033[00m";
  while (s.size() + 18 < 60) s += ' ';
  std::cout << s << " : " << tm << ".\n";
}

}  // namespace mgs
