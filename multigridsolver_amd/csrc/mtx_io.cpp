// mtx_io.cpp — Matrix-Market loader/writer with the reference's exact semantics
// (src/common/MatrixIO.cpp:12-57; tolerant GPU twin src/GPU_CUDAC++/MatrixIO.cu:182-280).
// Host C++; feeds mgs_csr_upload.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mgs.h"

int mgs_fail(struct mgs_ctx *ctx, int code, const char *fmt, ...);

extern "C" {

// readMatrix, MatrixIO.cpp:12-37: skip leading lines starting with '%' (:16), read "M N L"
// (:18), read L triples "i j v" 1-based in any order (:23-27), bucket by row, sort each row
// by (column, value) (:29).  The banner is not validated; symmetric/pattern files are not
// expanded (same as the reference).
int mgs_mtx_read(const char *path, int *rows, int *cols, int *nnz, int **rowptr, int **col, double **val) {
  if (!path || !rows || !cols || !nnz || !rowptr || !col || !val) return mgs_fail(nullptr, MGS_ERR_INVALID, "mgs_mtx_read: NULL argument");
  std::ifstream fin(path);
  if (!fin.is_open()) return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: cannot open '%s'", path);
  while (fin.peek() == '%') fin.ignore(2048, '\n');
  long long M = -1, N = -1, L = -1;
  fin >> M >> N >> L;
  if (!fin || M < 0 || N < 0 || L < 0 || M > 2147483646LL || N > 2147483646LL || L > 2147483646LL)
    return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: bad size line in '%s'", path);
  std::vector<std::vector<std::pair<int, double>>> data((size_t)M);
  for (long long l = 0; l < L; ++l) {
    long long m, n; double d;
    fin >> m >> n >> d;
    if (!fin) return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: '%s' ends after %lld of %lld entries", path, l, L);
    if (m < 1 || m > M || n < 1 || n > N)
      return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: entry %lld (%lld,%lld) outside %lld x %lld in '%s'", l + 1, m, n, M, N, path);
    data[(size_t)(m - 1)].push_back({(int)(n - 1), d});
  }
  int *rp = (int *)malloc(sizeof(int) * ((size_t)M + 1));
  int *ci = (int *)malloc(sizeof(int) * (size_t)(L ? L : 1));
  double *v = (double *)malloc(sizeof(double) * (size_t)(L ? L : 1));
  if (!rp || !ci || !v) { free(rp); free(ci); free(v); return mgs_fail(nullptr, MGS_ERR_ALLOC, "mgs_mtx_read: out of memory"); }
  size_t p = 0;
  for (long long i = 0; i < M; ++i) {
    rp[i] = (int)p;
    auto &r = data[(size_t)i];
    std::sort(r.begin(), r.end());
    for (size_t q = 0; q < r.size(); ++q) {
      if (q && r[q].first == r[q - 1].first) {  // Eigen's insert() asserts on duplicates (MatrixIO.cpp:31)
        free(rp); free(ci); free(v);
        return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: duplicate entry (%lld,%d) in '%s'", i + 1, r[q].first + 1, path);
      }
      ci[p] = r[q].first; v[p] = r[q].second; ++p;
    }
  }
  rp[M] = (int)p;
  *rows = (int)M; *cols = (int)N; *nnz = (int)L; *rowptr = rp; *col = ci; *val = v;
  fprintf(stderr, "Read matrix from file: %s\n", path);   // MatrixIO.cpp:35
  return MGS_OK;
}

// writeMatrix, MatrixIO.cpp:39-57.
int mgs_mtx_write(const char *path, int rows, int cols, int nnz, const int *rowptr, const int *col, const double *val) {
  if (!path || !rowptr || (nnz && (!col || !val))) return mgs_fail(nullptr, MGS_ERR_INVALID, "mgs_mtx_write: NULL argument");
  std::ofstream fout(path);
  if (!fout.is_open()) return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_write: cannot open '%s'", path);
  fout << "%%MatrixMarket matrix coordinate real general " << std::endl;   // :45
  fout << rows << " " << cols << " " << nnz << std::endl;                   // :46
  for (int i = 0; i < rows; ++i)
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) fout << i + 1 << " " << col[k] + 1 << " " << val[k] << std::endl;   // :52
  fout.close();
  return fout.fail() ? mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_write: write to '%s' failed", path) : MGS_OK;
}

void mgs_host_free(void *p) { free(p); }

}  // extern "C"
