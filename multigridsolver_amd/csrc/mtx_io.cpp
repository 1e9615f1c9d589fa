// mtx_io.cpp — Matrix-Market loader/writer with the reference's exact semantics
// (src/common/MatrixIO.cpp:12-57; tolerant GPU twin src/GPU_CUDAC++/MatrixIO.cu:182-280).
// Host C++; feeds mgs_csr_upload.
//
// The reference parses with `ifstream >>` into a vector of per-row vectors and writes one `<< std::endl` (= one flush) per entry:
// 0.4 µs per byte on the way in, 1 µs per entry on the way out — for a 4e6-row operator the file I/O costs more than the device
// solve.  Here the file is mapped, cut at whitespace into one piece per thread, tokenised (the contract is token based, not line
// based: "any order, any whitespace"), parsed with an exact decimal fast path (strtod for everything the fast path cannot prove
// exact), counting-sorted by row and sorted by column inside each row; the writer formats row ranges in parallel into buffers and
// emits the same bytes as the reference (`ostream <<` of an int / a double = "%d" / "%g").  MGS_IO_THREADS bounds the threads.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/mgs.h"

int mgs_fail(struct mgs_ctx *ctx, int code, const char *fmt, ...);

namespace {

// istream's separator set in the classic locale
inline bool is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

int io_threads(size_t work_bytes) {
  int t = (int)std::thread::hardware_concurrency();
  if (t < 1) t = 1;
  if (t > 16) t = 16;
  if (const char *e = getenv("MGS_IO_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) return v; }   // exact (tests cut small files too)
  const size_t by_size = work_bytes / ((size_t)1 << 20) + 1;     // a piece below 1 MiB is not worth a thread
  return (int)std::min<size_t>((size_t)t, by_size);
}

template <class F>
void parallel_for(int nthreads, F f) {
  if (nthreads <= 1) { f(0); return; }
  std::vector<std::thread> th;
  th.reserve((size_t)nthreads - 1);
  for (int t = 1; t < nthreads; ++t) th.emplace_back(f, t);
  f(0);
  for (auto &q : th) q.join();
}

// A token as `istream >> long long` accepts it: [+-]digits, nothing else.  Returns false on anything else / overflow.
bool parse_int(const char *p, const char *e, long long *out) {
  bool neg = false;
  if (p < e && (*p == '+' || *p == '-')) neg = *p++ == '-';
  if (p == e || e - p > 18) return false;
  long long v = 0;
  for (; p < e; ++p) { const unsigned d = (unsigned)(*p - '0'); if (d > 9) return false; v = v * 10 + d; }
  *out = neg ? -v : v;
  return true;
}

const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// A token as `istream >> double` accepts it (libstdc++ num_get: [+-] digits [. digits] [eE [+-] digits], at least one mantissa
// digit; no inf/nan/hex; overflow sets failbit).  Exact fast path: a mantissa below 2^53 times or divided by a power of ten that
// is itself a double (|e10| <= 22) is ONE correctly rounded operation = what strtod returns; everything else goes to strtod.
bool parse_double(const char *p, const char *e, double *out) {
  const char *s = p;
  bool neg = false;
  if (p < e && (*p == '+' || *p == '-')) neg = *p++ == '-';
  uint64_t m = 0; int nd = 0, e10 = 0; bool any = false, exactm = true;
  for (; p < e && (unsigned)(*p - '0') <= 9; ++p) { any = true; if (nd < 19) { m = m * 10 + (unsigned)(*p - '0'); if (m) ++nd; } else { exactm = false; ++e10; } }
  if (p < e && *p == '.') {
    ++p;
    for (; p < e && (unsigned)(*p - '0') <= 9; ++p) { any = true; if (nd < 19) { m = m * 10 + (unsigned)(*p - '0'); if (m) ++nd; --e10; } else exactm = false; }
  }
  if (!any) return false;
  if (p < e && (*p == 'e' || *p == 'E')) {
    ++p;
    bool eneg = false;
    if (p < e && (*p == '+' || *p == '-')) eneg = *p++ == '-';
    if (p == e) return false;
    int ex = 0;
    for (; p < e; ++p) { const unsigned d = (unsigned)(*p - '0'); if (d > 9) return false; if (ex < 100000) ex = ex * 10 + (int)d; }
    e10 += eneg ? -ex : ex;
  }
  if (p != e) return false;
  if (exactm && m < ((uint64_t)1 << 53) && e10 >= -22 && e10 <= 22) {
    const double d = (double)m;
    const double v = e10 < 0 ? d / P10[-e10] : d * P10[e10];
    *out = neg ? -v : v;
    return true;
  }
  if (m == 0 && exactm) { *out = neg ? -0.0 : 0.0; return true; }
  char tmp[64]; std::string big;
  const size_t len = (size_t)(e - s);
  const char *z;
  if (len < sizeof tmp) { memcpy(tmp, s, len); tmp[len] = 0; z = tmp; } else { big.assign(s, len); z = big.c_str(); }
  char *endp = nullptr;
  const double v = strtod(z, &endp);
  if (endp != z + len || std::isinf(v)) return false;
  *out = v;
  return true;
}

struct Mapped {
  const char *p = nullptr; size_t n = 0; int fd = -1;
  ~Mapped() { if (p && n) munmap(const_cast<char *>(p), n); if (fd >= 0) close(fd); }
};

}  // namespace

extern "C" {

// readMatrix, MatrixIO.cpp:12-37: skip leading lines starting with '%' (:16), read "M N L"
// (:18), read L triples "i j v" 1-based in any order (:23-27), bucket by row, sort each row
// by (column, value) (:29).  The banner is not validated; symmetric/pattern files are not
// expanded (same as the reference).
int mgs_mtx_read(const char *path, int *rows, int *cols, int *nnz, int **rowptr, int **col, double **val) {
  if (!path || !rows || !cols || !nnz || !rowptr || !col || !val) return mgs_fail(nullptr, MGS_ERR_INVALID, "mgs_mtx_read: NULL argument");
  Mapped f;
  f.fd = open(path, O_RDONLY);
  if (f.fd < 0) return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: cannot open '%s'", path);
  struct stat st;
  if (fstat(f.fd, &st) != 0 || !S_ISREG(st.st_mode)) return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: '%s' is not a regular file", path);
  f.n = (size_t)st.st_size;
  if (f.n) {
    void *m = mmap(nullptr, f.n, PROT_READ, MAP_PRIVATE, f.fd, 0);
    if (m == MAP_FAILED) { f.n = 0; return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: cannot map '%s'", path); }
    f.p = (const char *)m;
    madvise(m, f.n, MADV_SEQUENTIAL);
  }
  const char *buf = f.p; const size_t size = f.n;
  size_t pos = 0;
  while (pos < size && buf[pos] == '%') {                       // :16 `while (fin.peek() == '%') fin.ignore(2048, '\n')`
    size_t k = 0;
    while (pos < size && k < 2048) { const char c = buf[pos++]; ++k; if (c == '\n') break; }
  }
  auto next_token = [&](const char *&a, const char *&b) -> bool {
    while (pos < size && is_space((unsigned char)buf[pos])) ++pos;
    if (pos == size) return false;
    a = buf + pos;
    while (pos < size && !is_space((unsigned char)buf[pos])) ++pos;
    b = buf + pos;
    return true;
  };
  long long hdr[3] = {-1, -1, -1};
  for (int q = 0; q < 3; ++q) {
    const char *a, *b;
    if (!next_token(a, b) || !parse_int(a, b, &hdr[q]) || hdr[q] < 0 || hdr[q] > 2147483646LL)
      return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: bad size line in '%s'", path);
  }
  const long long M = hdr[0], N = hdr[1], L = hdr[2];
  const size_t body = pos;

  // pieces cut at token boundaries; pass 1 counts the tokens of each piece so that pass 2 knows which entry and field it starts with
  const int T = io_threads(size - body);
  std::vector<size_t> cut((size_t)T + 1);
  for (int t = 0; t <= T; ++t) {
    size_t c = body + (size_t)((double)(size - body) * t / T);
    if (t == T) c = size;
    while (c > body && c < size && !is_space((unsigned char)buf[c - 1]) && !is_space((unsigned char)buf[c])) ++c;   // inside a token: move past it
    cut[(size_t)t] = c;
  }
  std::vector<long long> ntok((size_t)T + 1, 0);
  parallel_for(T, [&](int t) {
    const char *p = buf + cut[(size_t)t], *e = buf + cut[(size_t)t + 1];
    long long c = 0; bool in = false;
    for (; p < e; ++p) { const bool sp = is_space((unsigned char)*p); c += (!sp && !in); in = !sp; }
    ntok[(size_t)t + 1] = c;
  });
  for (int t = 0; t < T; ++t) ntok[(size_t)t + 1] += ntok[(size_t)t];
  if (ntok[(size_t)T] < 3 * L)
    return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: '%s' ends after %lld of %lld entries", path, ntok[(size_t)T] / 3, L);

  const size_t La = (size_t)(L ? L : 1);
  int *ei = (int *)malloc(sizeof(int) * La), *ej = (int *)malloc(sizeof(int) * La);
  double *ev = (double *)malloc(sizeof(double) * La);
  int *rp = (int *)calloc((size_t)M + 2, sizeof(int));
  auto bail = [&](int rc) { free(ei); free(ej); free(ev); free(rp); return rc; };
  if (!ei || !ej || !ev || !rp) return bail(mgs_fail(nullptr, MGS_ERR_ALLOC, "mgs_mtx_read: out of memory"));

  // pass 2: parse.  The first offending entry (smallest index, as a sequential reader would meet it) is the one reported.
  std::atomic<long long> bad_entry{L};
  parallel_for(T, [&](int t) {
    const char *p = buf + cut[(size_t)t], *e = buf + cut[(size_t)t + 1];
    long long g = ntok[(size_t)t];
    const long long gend = 3 * L;
    while (g < gend) {
      while (p < e && is_space((unsigned char)*p)) ++p;
      if (p >= e) break;
      const char *a = p;
      while (p < e && !is_space((unsigned char)*p)) ++p;
      const long long ent = g / 3; const int fld = (int)(g % 3);
      bool ok;
      if (fld == 2) { double d = 0; ok = parse_double(a, p, &d); ev[ent] = d; }
      else {
        long long v = 0; ok = parse_int(a, p, &v) && v >= 1 && v <= (fld == 0 ? M : N);
        (fld == 0 ? ei : ej)[ent] = (int)(v - 1);
      }
      if (!ok) { long long cur = bad_entry.load(); while (ent < cur && !bad_entry.compare_exchange_weak(cur, ent)) {} break; }
      ++g;
    }
  });
  if (bad_entry.load() < L) {
    // re-read that entry sequentially for the message
    const long long be = bad_entry.load();
    pos = body; long long g = 0; const char *a = nullptr, *b = nullptr; std::string tk[3];
    while (g < 3 * be && next_token(a, b)) ++g;
    for (int q = 0; q < 3 && next_token(a, b); ++q) tk[q].assign(a, (size_t)(b - a));
    return bail(mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: entry %lld (%s %s %s) is malformed or outside %lld x %lld in '%s'", be + 1,
                         tk[0].c_str(), tk[1].c_str(), tk[2].c_str(), M, N, path));
  }

  // already row-major with ascending columns (what every writer of this format emits)?  Then the parsed arrays ARE the CSR arrays.
  const int TC = io_threads((size_t)L * 16);
  std::atomic<int> unsorted{0};
  parallel_for(TC, [&](int t) {
    const long long a = std::max<long long>(1, L * t / TC), b = L * (t + 1) / TC;
    for (long long k = a; k < b; ++k)
      if (!(ei[k - 1] < ei[k] || (ei[k - 1] == ei[k] && ej[k - 1] < ej[k]))) { unsorted.store(1); break; }
  });
  const bool sorted = unsorted.load() == 0;
  // row counts (rp[i + 1] = entries of row i), shared counters
  parallel_for(TC, [&](int t) {
    const long long a = L * t / TC, b = L * (t + 1) / TC;
    for (long long k = a; k < b; ++k) __atomic_fetch_add(&rp[ei[k] + 1], 1, __ATOMIC_RELAXED);
  });
  for (long long i = 0; i < M; ++i) rp[i + 1] += rp[i];
  int *ci = nullptr; double *v = nullptr;
  if (sorted) { ci = ej; v = ev; ej = nullptr; ev = nullptr; }
  else {
    ci = (int *)malloc(sizeof(int) * La); v = (double *)malloc(sizeof(double) * La);
    int *cur = (int *)malloc(sizeof(int) * ((size_t)M + 1));
    if (!ci || !v || !cur) { free(ci); free(v); free(cur); return bail(mgs_fail(nullptr, MGS_ERR_ALLOC, "mgs_mtx_read: out of memory")); }
    memcpy(cur, rp, sizeof(int) * ((size_t)M + 1));
    parallel_for(TC, [&](int t) {
      const long long a = L * t / TC, b = L * (t + 1) / TC;
      for (long long k = a; k < b; ++k) { const int q = __atomic_fetch_add(&cur[ei[k]], 1, __ATOMIC_RELAXED); ci[q] = ej[k]; v[q] = ev[k]; }
    });
    free(cur);
    // order inside a row: by (column, value) (:29); the arrival order above is arbitrary, the sorted order is not
    std::atomic<long long> dup_row{-1};
    parallel_for(TC, [&](int t) {
      const long long ra = M * t / TC, rb = M * (t + 1) / TC;
      std::vector<std::pair<int, double>> tmp;
      for (long long i = ra; i < rb; ++i) {
        const int a = rp[i], b = rp[i + 1];
        if (b - a <= 32) {
          for (int k = a + 1; k < b; ++k) {
            const int c = ci[k]; const double d = v[k]; int q = k - 1;
            while (q >= a && (ci[q] > c || (ci[q] == c && v[q] > d))) { ci[q + 1] = ci[q]; v[q + 1] = v[q]; --q; }
            ci[q + 1] = c; v[q + 1] = d;
          }
        } else {
          tmp.resize((size_t)(b - a));
          for (int k = a; k < b; ++k) tmp[(size_t)(k - a)] = {ci[k], v[k]};
          std::sort(tmp.begin(), tmp.end());
          for (int k = a; k < b; ++k) { ci[k] = tmp[(size_t)(k - a)].first; v[k] = tmp[(size_t)(k - a)].second; }
        }
        for (int k = a + 1; k < b; ++k)
          if (ci[k] == ci[k - 1]) { long long cur2 = dup_row.load(); while ((cur2 < 0 || i < cur2) && !dup_row.compare_exchange_weak(cur2, i)) {} break; }
      }
    });
    if (dup_row.load() >= 0) {                                   // Eigen's insert() asserts on duplicates (MatrixIO.cpp:31)
      const long long i = dup_row.load(); int c = -1;
      for (int k = rp[i] + 1; k < rp[i + 1]; ++k) if (ci[k] == ci[k - 1]) { c = ci[k]; break; }
      free(ci); free(v);
      return bail(mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_read: duplicate entry (%lld,%d) in '%s'", i + 1, c + 1, path));
    }
  }
  free(ei); free(ej); free(ev);
  // rp was allocated one int longer than the result needs; the caller frees it with mgs_host_free all the same
  *rows = (int)M; *cols = (int)N; *nnz = (int)L; *rowptr = rp; *col = ci; *val = v;
  fprintf(stderr, "Read matrix from file: %s\n", path);   // MatrixIO.cpp:35
  return MGS_OK;
}

// writeMatrix, MatrixIO.cpp:39-57: banner (:45), "rows cols nnz" (:46), then one "i j v" line per entry in row-major order, 1-based
// (:52).  `ostream << double` at its default precision is printf's "%g"; every line of the reference ends in std::endl, so the bytes
// are the same whether or not anything is flushed in between.
int mgs_mtx_write(const char *path, int rows, int cols, int nnz, const int *rowptr, const int *col, const double *val) {
  if (!path || !rowptr || (nnz && (!col || !val))) return mgs_fail(nullptr, MGS_ERR_INVALID, "mgs_mtx_write: NULL argument");
  FILE *fo = fopen(path, "wb");
  if (!fo) return mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_write: cannot open '%s'", path);
  bool ok = fprintf(fo, "%%%%MatrixMarket matrix coordinate real general \n%d %d %d\n", rows, cols, nnz) > 0;
  const long long total = rows > 0 ? (long long)rowptr[rows] - rowptr[0] : 0;
  const int T = io_threads((size_t)total * 24);
  // rounds of T row ranges of ≈ 1M entries each: formatted in parallel, written in order
  const long long per = 1 << 20;
  std::vector<std::string> out((size_t)T);
  int r = 0;
  while (r < rows && ok) {
    std::vector<int> lim((size_t)T + 1, r);
    for (int t = 0; t < T; ++t) {              // whole rows, at least one per range, about `per` entries
      const int q = lim[(size_t)t];
      if (q >= rows) { lim[(size_t)t + 1] = rows; continue; }
      const long long target = (long long)rowptr[q] + per;
      const int *hit = target > 2147483647LL ? rowptr + rows + 1 : std::lower_bound(rowptr + q + 1, rowptr + rows + 1, (int)target);
      lim[(size_t)t + 1] = (int)std::min<long long>(hit - rowptr, rows);
    }
    parallel_for(T, [&](int t) {
      std::string &s = out[(size_t)t];
      s.clear();
      const int ra = lim[(size_t)t], rb = lim[(size_t)t + 1];
      if (ra >= rb) return;
      s.reserve((size_t)(rowptr[rb] - rowptr[ra]) * 28 + 64);
      char line[96];
      for (int i = ra; i < rb; ++i) {
        char head[16]; const int hl = snprintf(head, sizeof head, "%d ", i + 1);
        for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
          memcpy(line, head, (size_t)hl);
          char *w = std::to_chars(line + hl, line + hl + 12, col[k] + 1).ptr;
          *w++ = ' ';
          w = std::to_chars(w, line + sizeof line - 1, val[k], std::chars_format::general, 6).ptr;     // = printf("%g"), the default `ostream << double`
          *w++ = '\n';
          s.append(line, (size_t)(w - line));
        }
      }
    });
    for (int t = 0; t < T && ok; ++t) if (!out[(size_t)t].empty()) ok = fwrite(out[(size_t)t].data(), 1, out[(size_t)t].size(), fo) == out[(size_t)t].size();
    r = lim[(size_t)T];
  }
  if (fclose(fo) != 0) ok = false;
  return ok ? MGS_OK : mgs_fail(nullptr, MGS_ERR_IO, "mgs_mtx_write: write to '%s' failed", path);
}

void mgs_host_free(void *p) { free(p); }

}  // extern "C"
