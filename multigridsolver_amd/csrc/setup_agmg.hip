// setup_agmg.hip — on-device hierarchy setup (needed by configs 3 and 5):
//   * own exclusive scan (the reference used cub::DeviceScan, src/GPU_CUDAC++/PrefixSum.cu:6-21)
//   * CSR transpose / aggregate member lists (reference: cusparseCsr2cscEx2,
//     src/GPU_CUDAC++/MatrixOperations.cu:388-456)
//   * Galerkin product A_c = PᵀAP specialised for 0/1 aggregation P:
//     A_c[agg i, agg j] += a_ij, sort + reduce by key per coarse row (reference: two
//     cusparseSpGEMM calls, src/GPU_CUDAC++/main.cu:251-253; CPU: bicg.cpp:33)
//   * Notay pairwise aggregation (src/CPU_C++/AGMG.cpp:101-315;
//     src/GPU_CUDAC++/Aggregation.cu:17-270) with a DETERMINISTIC matching: instead of the
//     reference GPU's race-dependent atomicCAS claim (Aggregation.cu:203) every undecided
//     node picks its best admissible neighbour from a snapshot and mutual picks pair up
//     (locally-dominant edges), so P is reproducible run to run.
// FP64 throughout (the reference GPU setup is float32, MatrixIO.cu:32-36).
#include "mgs_internal.hpp"

#include <algorithm>

namespace {
constexpr int TB = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = TB * SCAN_ITEMS;

// ------------------------------------------------------------------ exclusive scan (int32)
__device__ __forceinline__ int block_exclusive_scan(int v, int *sh /*TB/64+1*/, int *total) {
  // wave inclusive scan
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < TB / 64; ++i) { int t = sh[i]; sh[i] = run; run += t; } sh[TB / 64] = run; }
  __syncthreads();
  int res = inc - v + sh[w];
  *total = sh[TB / 64];
  __syncthreads();
  return res;
}
__global__ __launch_bounds__(TB) void scan_tile_sums(const int *__restrict__ in, int64_t n, int *__restrict__ sums) {
  __shared__ int sh[TB / 64 + 1];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int q = 0; q < SCAN_ITEMS; ++q) if (base + q < n) s += in[base + q];
  int tot; block_exclusive_scan(s, sh, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(TB) void scan_tile_apply(const int *__restrict__ in, int *__restrict__ out, int64_t n, const int *__restrict__ offs) {
  __shared__ int sh[TB / 64 + 1];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS]; int s = 0;
#pragma unroll
  for (int q = 0; q < SCAN_ITEMS; ++q) { v[q] = (base + q < n) ? in[base + q] : 0; s += v[q]; }
  int tot; int ex = block_exclusive_scan(s, sh, &tot);
  int run = ex + (offs ? offs[blockIdx.x] : 0);
#pragma unroll
  for (int q = 0; q < SCAN_ITEMS; ++q) { if (base + q < n) out[base + q] = run; run += v[q]; }
}

int scan_rec(mgs_ctx *ctx, const int *in, int *out, int64_t n) {
  if (n <= 0) return MGS_OK;
  int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 1) {
    hipLaunchKernelGGL(scan_tile_apply, dim3(1), dim3(TB), 0, ctx->stream, in, out, n, (const int *)nullptr);
    MGS_HIP(ctx, hipGetLastError());
    return MGS_OK;
  }
  int *sums = nullptr;
  MGS_TRY(mgs_dev_alloc(ctx, &sums, (size_t)nb));
  hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)nb), dim3(TB), 0, ctx->stream, in, n, sums);
  int rc = scan_rec(ctx, sums, sums, nb);
  if (rc == MGS_OK) {
    hipLaunchKernelGGL(scan_tile_apply, dim3((unsigned)nb), dim3(TB), 0, ctx->stream, in, out, n, (const int *)sums);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "scan launch: %s", hipGetErrorString(e));
  }
  hipStreamSynchronize(ctx->stream);
  mgs_hip_free(sums);
  return rc;
}

// ------------------------------------------------------------------ bucket helpers
__global__ void count_keys_kernel(int64_t n, const int *__restrict__ key, int *__restrict__ counts) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { int k = key[i]; if (k >= 0) atomicAdd(&counts[k], 1); }
}
// members of aggregate a (unordered claim), then sorted per aggregate
__global__ void fill_members_kernel(int n, const int *__restrict__ agg, const int *__restrict__ cptr, int *__restrict__ cursor, int *__restrict__ members) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int a = agg[i];
  if (a < 0) return;
  int p = cptr[a] + atomicAdd(&cursor[a], 1);
  members[p] = i;
}
__global__ void sort_segments_kernel(int nseg, const int *__restrict__ ptr, int *__restrict__ keys, double *__restrict__ vals) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nseg) return;
  int lo = ptr[c], hi = ptr[c + 1];
  for (int a = lo + 1; a < hi; ++a) {
    int k = keys[a]; double v = vals ? vals[a] : 0.0;
    int b = a - 1;
    while (b >= lo && keys[b] > k) { keys[b + 1] = keys[b]; if (vals) vals[b + 1] = vals[b]; --b; }
    keys[b + 1] = k; if (vals) vals[b + 1] = v;
  }
}
__global__ void csr_row_of_nnz_fill_T(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                                      const int *__restrict__ tptr, int *__restrict__ cursor, int *__restrict__ tcol, double *__restrict__ tval) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
    int c = col[k];
    int p = tptr[c] + atomicAdd(&cursor[c], 1);
    tcol[p] = i; tval[p] = val[k];
  }
}

// P in CSR → aggregate ids; flags rows that break the aggregation form
__global__ void p_to_agg_kernel(int n, int ncols, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                                int *__restrict__ agg, int *__restrict__ bad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int lo = rowptr[i], len = rowptr[i + 1] - lo;
  int a = -1;
  if (len == 1 && val[lo] == 1.0 && col[lo] >= 0 && col[lo] < ncols) a = col[lo];
  else if (len != 0) atomicAdd(bad, 1);
  agg[i] = a;
}

// ------------------------------------------------------------------ Galerkin (aggregation P)
// One lane per coarse row keeps the row's (coarse column, value) pairs sorted and unique while it walks its member rows (fine order
// i↑, j↑): an entry whose column is already there is ADDED to it, a new column is inserted — the same sums in the same order as
// sort-by-key + reduce, without the sort.  The pairs live in LDS ([slot][lane]: conflict-free), CAP slots per lane; a row that needs
// more continues in global memory (count pass: quadratic distinct count; fill pass: its own output segment).  Two passes (count →
// scan → fill) instead of an nnz-sized scratch copy.  cptr == NULL: identity member lists (row c alone — A·P).
__global__ void galerkin_maxub_kernel(int nc, const int *__restrict__ cptr, const int *__restrict__ members, const int *__restrict__ rowptr, int *__restrict__ mx) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  int s = 0;
  if (c < nc && !cptr) s = rowptr[c + 1] - rowptr[c];
  else if (c < nc) for (int k = cptr[c]; k < cptr[c + 1]; ++k) { int i = members[k]; s += rowptr[i + 1] - rowptr[i]; }
  for (int off = 32; off > 0; off >>= 1) s = max(s, __shfl_down(s, off));
  // (one address for the whole launch: only a wave that would raise the maximum goes to the atomic unit)
  if ((threadIdx.x & 63) == 0 && s > __atomic_load_n(mx, __ATOMIC_RELAXED)) atomicMax(mx, s);
}
template <int CAP, int TBK, bool FILL>
__global__ __launch_bounds__(TBK) void galerkin_lds_kernel(int nc, const int *__restrict__ cptr, const int *__restrict__ members, const int *__restrict__ agg /*column map, size = cols of A*/,
                                                           const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                                                           const int *__restrict__ crowptr, int *__restrict__ uniq, int *__restrict__ ccol, double *__restrict__ cval) {
  __shared__ int keys[CAP * TBK];
  __shared__ double vals[FILL ? CAP * TBK : 1];
  const int c = blockIdx.x * TBK + threadIdx.x, t = threadIdx.x;
  if (c > nc) return;
  if (c == nc) { if (!FILL) uniq[c] = 0; return; }
  const int m0 = cptr ? cptr[c] : c, m1 = cptr ? cptr[c + 1] : c + 1;
  int cnt = 0;
  bool spilled = false;      // more than CAP distinct columns: continue in global memory
  const int dst = FILL ? crowptr[c] : 0;
  int extra = 0;             // count pass, spilled: distinct columns beyond the CAP held in LDS
  for (int m = m0; m < m1; ++m) {
    const int i = cptr ? members[m] : m;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int a = agg[col[k]];
      if (a < 0) continue;
      if (!spilled) {
        int b = cnt - 1;
        while (b >= 0 && keys[b * TBK + t] > a) --b;
        if (b >= 0 && keys[b * TBK + t] == a) { if (FILL) vals[b * TBK + t] += val[k]; continue; }
        if (cnt < CAP) {
          for (int q = cnt - 1; q > b; --q) { keys[(q + 1) * TBK + t] = keys[q * TBK + t]; if (FILL) vals[(q + 1) * TBK + t] = vals[q * TBK + t]; }
          keys[(b + 1) * TBK + t] = a; if (FILL) vals[(b + 1) * TBK + t] = val[k];
          ++cnt;
          continue;
        }
        spilled = true;
        if (FILL) { for (int q = 0; q < cnt; ++q) { ccol[dst + q] = keys[q * TBK + t]; cval[dst + q] = vals[q * TBK + t]; } }
      }
      if (FILL) {            // the row's output segment (its length is the distinct count of the count pass): same insert-or-add there
        int b = cnt - 1;
        while (b >= 0 && ccol[dst + b] > a) --b;
        if (b >= 0 && ccol[dst + b] == a) { cval[dst + b] += val[k]; continue; }
        for (int q = cnt - 1; q > b; --q) { ccol[dst + q + 1] = ccol[dst + q]; cval[dst + q + 1] = cval[dst + q]; }
        ccol[dst + b + 1] = a; cval[dst + b + 1] = val[k];
        ++cnt;
      } else {               // is this column new?  not among the CAP columns in LDS and not among the earlier entries (walked again)
        bool seen = false;
        for (int q = 0; q < CAP && !seen; ++q) seen = keys[q * TBK + t] == a;
        for (int mm = m0; mm <= m && !seen; ++mm) {
          const int ii = cptr ? members[mm] : mm;
          const int ke = mm == m ? k : rowptr[ii + 1];
          for (int kk = rowptr[ii]; kk < ke && !seen; ++kk) seen = agg[col[kk]] == a;
        }
        if (!seen) ++extra;
      }
    }
  }
  if (!FILL) { uniq[c] = cnt + extra; return; }
  if (!spilled) for (int q = 0; q < cnt; ++q) { ccol[dst + q] = keys[q * TBK + t]; cval[dst + q] = vals[q * TBK + t]; }
}

// ------------------------------------------------------------------ pairwise aggregation
__device__ __forceinline__ double csr_lookup(const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, int i, int j) {
  int lo = rowptr[i], hi = rowptr[i + 1] - 1;   // reference getElementMatrixCSR, MatrixAccess.cu:28-47
  while (lo <= hi) {
    int mid = lo + ((hi - lo) >> 1); int c = col[mid];   // (lo + hi would overflow int32 near 2^31 entries)
    if (c == j) return val[mid];
    if (c < j) lo = mid + 1; else hi = mid - 1;
  }
  return 0.0;
}
__global__ void pattern_asym_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col, int *__restrict__ asym) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int bad = 0;
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
    int j = col[k];
    if (j == i || j >= n) continue;               // halo columns (row shards) are not owned rows
    int lo = rowptr[j], hi = rowptr[j + 1] - 1; bool found = false;
    while (lo <= hi) { int mid = lo + ((hi - lo) >> 1); int c = col[mid]; if (c == i) { found = true; break; } if (c < i) lo = mid + 1; else hi = mid - 1; }
    if (!found) ++bad;
  }
  if (bad) atomicAdd(asym, bad);
}
// per node: a_ii, s_i = −Σ_{j≠i}(a_ij+a_ji)/2 (AGMG.cpp:84-90, Aggregation.cu:68-90) and the G0
// test a_ii ≥ ktg/(ktg−2)·Σ_{j≠i}|a_ij+a_ji|/2 on the first pass (AGMG.cpp:118-123,
// Aggregation.cu:17-64).  With At == nullptr the pattern is symmetric and a_ji is looked up.
__global__ void agg_node_stats_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                                      const int *__restrict__ trowptr, const int *__restrict__ tcol, const double *__restrict__ tval,
                                      double ktg, int first_pass, double *__restrict__ diag, double *__restrict__ s, int *__restrict__ state) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double aii = 0.0, ssum = 0.0, asum = 0.0;
  if (trowptr == nullptr) {
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      int j = col[k]; double aij = val[k];
      if (j == i) { aii = aij; continue; }
      // off-shard neighbour (halo column of a row shard): its row lives on another GPU; the
      // coupling is taken as symmetric, a_ji := a_ij
      double aji = j < n ? csr_lookup(rowptr, col, val, j, i) : aij;
      ssum += (aij + aji) / 2; asum += fabs((aij + aji) / 2);
    }
  } else {
    int r = rowptr[i], re = rowptr[i + 1], c = trowptr[i], ce = trowptr[i + 1];
    while (r < re || c < ce) {
      int jr = r < re ? col[r] : 0x7fffffff, jc = c < ce ? tcol[c] : 0x7fffffff;
      int j = min(jr, jc);
      double aij = 0.0, aji = 0.0;
      if (jr == j) aij = val[r++];
      if (jc == j) aji = tval[c++];
      if (j >= n) aji = aij;                        // halo column: symmetric coupling assumed
      if (j == i) { aii = aij; continue; }
      ssum += (aij + aji) / 2; asum += fabs((aij + aji) / 2);
    }
  }
  diag[i] = aii; s[i] = -ssum;
  int g0 = first_pass && (aii >= (ktg / (ktg - 2)) * asum);
  state[i] = g0 ? -2 : -1;
}
// μ({i,j}) per stored entry (AGMG.cpp:92-99, Aggregation.cu:96-105), +inf if the pair is not
// admissible: j==i, a_ij==0, i or j in G0, a_ii−s_i+a_jj−s_j<0 (`okay`, Aggregation.cu:157-159),
// μ≤0 or μ>ktg (AGMG.cpp:163,171).
__global__ void agg_edge_weight_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                                       const double *__restrict__ diag, const double *__restrict__ s, const int *__restrict__ state,
                                       double ktg, const int *__restrict__ zone /*NULL: none; rows of different zones never pair*/, double *__restrict__ w) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double aii = diag[i], si = s[i];
  const bool gi = state[i] == -2;
  const int zi = zone ? zone[i] : 0;
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
    int j = col[k]; double aij = val[k];
    double wk = INFINITY;
    if (j != i && j < n && aij != 0.0 && !gi && state[j] != -2 && (!zone || zone[j] == zi)) {
      double ajj = diag[j], sj = s[j];
      if (aii - si + ajj - sj >= 0) {
        double aji = csr_lookup(rowptr, col, val, j, i);
        double num = 2 / (1 / aii + 1 / ajj);
        double den = (-(aij + aji) / 2) + 1 / (1 / (aii - si) + 1 / (ajj - sj));
        double mu = num / den;
        if (mu > 0 && mu <= ktg) wk = mu;
      }
    }
    w[k] = wk;
  }
}
struct EdgeKey { double w; int d; int par; unsigned h; int mn; };
__device__ __forceinline__ unsigned edge_hash(unsigned a, unsigned b) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
// Tie-break among equal μ (a pure heuristic, symmetric in (i,j)): nearest index first, then
// the edge whose lower end is an even multiple of the stride — on lexicographically numbered
// grids this yields aligned pairs (2m,2m+1), i.e. the regular semi-coarsening the sequential
// reference produces (AGMG.cpp:149-179 scans neighbours in ascending order) — then a hash.
// i, j are ORIGIN indices: the finest-level row a node descends from (leader of its aggregate, chained through passes
// and levels).  Node ids themselves drift against the grid wherever G0 rows or odd-sized aggregates were skipped in
// the numbering, and the parity rule then flips in patches (round 1: 13–30 % of the aggregates misaligned against
// their neighbours at 512³); in origin space the stride of a grid direction is exact on every level.
__device__ __forceinline__ EdgeKey make_key(double w, int i, int j, int hash_only) {
  EdgeKey k; int mn = min(i, j), mx = max(i, j);
  k.w = hash_only ? 0.0 : w; k.d = hash_only ? 0 : mx - mn; k.par = hash_only ? 0 : ((mn / (mx - mn)) & 1);
  k.h = edge_hash((unsigned)mn, (unsigned)mx); k.mn = mn;
  return k;
}
__device__ __forceinline__ bool key_less(const EdgeKey &a, const EdgeKey &b) {
  if (a.w != b.w) return a.w < b.w;
  if (a.d != b.d) return a.d < b.d;
  if (a.par != b.par) return a.par < b.par;
  if (a.h != b.h) return a.h < b.h;
  return a.mn < b.mn;
}
__global__ void agg_pick_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ w,
                                const int *__restrict__ state, int hash_only, const int *__restrict__ origin, int *__restrict__ pick) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (state[i] != -1) { pick[i] = -3; return; }
  int best = -1; EdgeKey bk;
  const int oi = origin ? origin[i] : i;
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
    double wk = w[k];
    if (!(wk < INFINITY)) continue;
    int j = col[k];
    if (state[j] != -1) continue;
    EdgeKey key = make_key(wk, oi, origin ? origin[j] : j, hash_only);
    if (best < 0 || key_less(key, bk)) { best = j; bk = key; }
  }
  pick[i] = best;
}
__global__ void agg_match_kernel(int n, const int *__restrict__ pick, int *__restrict__ state, int force_single, int *__restrict__ counters /*[0]=undecided left*/) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int p = pick[i];
  if (p == -3) return;
  if (p == -1) { state[i] = i; return; }            // no admissible free neighbour: singleton (AGMG.cpp:175-178)
  if (pick[p] == i) { state[i] = p; return; }       // mutual pick: pair (AGMG.cpp:171-174)
  if (force_single) { state[i] = i; return; }
  atomicAdd(&counters[0], 1);
}
__global__ void agg_leader_flag_kernel(int n, const int *__restrict__ state, int *__restrict__ flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  flag[i] = (i < n && state[i] >= i) ? 1 : 0;         // leader = lower end of a pair or a singleton (Aggregation.cu:214-225)
}
__global__ void agg_assign_kernel(int n, const int *__restrict__ state, const int *__restrict__ ids, int *__restrict__ agg) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int st = state[i];
  agg[i] = st < 0 ? -1 : ids[min(i, st)];
}
// origin of every aggregate = smallest origin among its members (atomicMin: order-independent)
__global__ void agg_origin_kernel(int n, const int *__restrict__ agg, const int *__restrict__ origin, int *__restrict__ corigin) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = agg[i];
  if (a >= 0) atomicMin(&corigin[a], origin ? origin[i] : i);
}
// zone of an aggregate = zone of its members (rows of different zones never pair, so they agree)
__global__ void agg_zone_kernel(int n, const int *__restrict__ agg, const int *__restrict__ zone, int *__restrict__ czone) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && agg[i] >= 0) czone[agg[i]] = zone[i];
}
__global__ void agg_compose_kernel(int n, int *__restrict__ agg, const int *__restrict__ agg2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int a = agg[i];
  agg[i] = a < 0 ? -1 : agg2[a];
}

struct DevBuf {  // RAII for setup temporaries
  void *p = nullptr;
  ~DevBuf() { if (p) mgs_hip_free(p); }
  template <class T> T *as() { return (T *)p; }
};
template <class T>
int dalloc(mgs_ctx *ctx, DevBuf &b, size_t count) { T *q = nullptr; MGS_TRY(mgs_dev_alloc(ctx, &q, count)); b.p = q; return MGS_OK; }

// build cptr/members from agg (device array of n_fine entries, -1 = none)
int build_member_lists(mgs_ctx *ctx, int n, int nc, const int *agg, int **cptr_out, int **members_out, int64_t *nnz_out) {
  int *cptr = nullptr, *members = nullptr; DevBuf cursor;
  MGS_TRY(mgs_dev_alloc(ctx, &cptr, (size_t)nc + 1));
  MGS_HIP(ctx, hipMemsetAsync(cptr, 0, sizeof(int) * ((size_t)nc + 1), ctx->stream));
  if (n) hipLaunchKernelGGL(count_keys_kernel, dim3(mgs_grid(n, TB)), dim3(TB), 0, ctx->stream, (int64_t)n, agg, cptr);
  MGS_TRY(scan_rec(ctx, cptr, cptr, (int64_t)nc + 1));
  int total = 0;
  MGS_HIP(ctx, hipMemcpyAsync(&total, cptr + nc, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MGS_TRY(mgs_dev_alloc(ctx, &members, (size_t)total));
  MGS_TRY(dalloc<int>(ctx, cursor, (size_t)nc + 1));
  MGS_HIP(ctx, hipMemsetAsync(cursor.p, 0, sizeof(int) * ((size_t)nc + 1), ctx->stream));
  if (n) hipLaunchKernelGGL(fill_members_kernel, dim3(mgs_grid(n, TB)), dim3(TB), 0, ctx->stream, n, agg, cptr, cursor.as<int>(), members);
  if (nc) hipLaunchKernelGGL(sort_segments_kernel, dim3(mgs_grid(nc, TB)), dim3(TB), 0, ctx->stream, nc, cptr, members, (double *)nullptr);
  MGS_HIP(ctx, hipGetLastError());
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *cptr_out = cptr; *members_out = members; *nnz_out = total;
  return MGS_OK;
}

int xfer_from_agg(mgs_ctx *ctx, int n, int nc, int *agg_owned, mgs_xfer **out) {
  mgs_xfer *T = new mgs_xfer();
  T->ctx = ctx; T->n_fine = n; T->n_coarse = nc; T->aggregation = true; T->agg = agg_owned;
  int rc = build_member_lists(ctx, n, nc, agg_owned, &T->cptr, &T->members, &T->nnz);
  if (rc != MGS_OK) { mgs_xfer_destroy(T); return rc; }
  *out = T;
  return MGS_OK;
}

}  // namespace

int k_xfer_from_agg_host(mgs_ctx *ctx, int n_fine, int n_coarse, const int *agg_host, mgs_xfer **out) {
  int *agg = nullptr;
  MGS_TRY(mgs_dev_alloc(ctx, &agg, (size_t)n_fine));
  for (int i = 0; i < n_fine; ++i)
    if (agg_host[i] < -1 || agg_host[i] >= n_coarse) { mgs_hip_free(agg); return mgs_fail(ctx, MGS_ERR_INVALID, "aggregate id %d of row %d outside [-1,%d)", agg_host[i], i, n_coarse); }
  MGS_HIP(ctx, hipMemcpyAsync(agg, agg_host, sizeof(int) * (size_t)n_fine, hipMemcpyHostToDevice, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return xfer_from_agg(ctx, n_fine, n_coarse, agg, out);
}

int k_exclusive_scan_i32(mgs_ctx *ctx, const int *in, int *out, int64_t n, int64_t *total_host) {
  MGS_TRY(scan_rec(ctx, in, out, n));
  if (total_host) {
    // caller convention: in has n entries with in[n-1] == 0 as the sentinel → out[n-1] is the total
    int t = 0;
    MGS_HIP(ctx, hipMemcpyAsync(&t, out + (n - 1), sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *total_host = t;
  }
  return MGS_OK;
}

int k_transpose(const mgs_csr *A, mgs_csr **out) {
  mgs_ctx *ctx = A->ctx;
  mgs_csr *B = nullptr;
  MGS_TRY(mgs_csr_alloc(ctx, A->cols, A->rows, A->nnz, &B));
  DevBuf cursor;
  MGS_HIP(ctx, hipMemsetAsync(B->rowptr, 0, sizeof(int) * ((size_t)B->rows + 1), ctx->stream));
  if (A->nnz) hipLaunchKernelGGL(count_keys_kernel, dim3(mgs_grid(A->nnz, TB)), dim3(TB), 0, ctx->stream, A->nnz, A->col, B->rowptr);
  MGS_TRY(scan_rec(ctx, B->rowptr, B->rowptr, (int64_t)B->rows + 1));
  MGS_TRY(dalloc<int>(ctx, cursor, (size_t)B->rows + 1));
  MGS_HIP(ctx, hipMemsetAsync(cursor.p, 0, sizeof(int) * ((size_t)B->rows + 1), ctx->stream));
  if (A->rows) hipLaunchKernelGGL(csr_row_of_nnz_fill_T, dim3(mgs_grid(A->rows, TB)), dim3(TB), 0, ctx->stream, A->rows, A->rowptr, A->col, A->val, B->rowptr, cursor.as<int>(), B->col, B->val);
  if (B->rows) hipLaunchKernelGGL(sort_segments_kernel, dim3(mgs_grid(B->rows, TB)), dim3(TB), 0, ctx->stream, B->rows, B->rowptr, B->col, B->val);
  MGS_HIP(ctx, hipGetLastError());
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MGS_TRY(mgs_plan_csr(B));
  *out = B;
  return MGS_OK;
}

int k_xfer_from_csr(const mgs_csr *P, mgs_xfer **out) {
  mgs_ctx *ctx = P->ctx;
  int *agg = nullptr; DevBuf bad;
  MGS_TRY(mgs_dev_alloc(ctx, &agg, (size_t)P->rows));
  MGS_TRY(dalloc<int>(ctx, bad, 1));
  MGS_HIP(ctx, hipMemsetAsync(bad.p, 0, sizeof(int), ctx->stream));
  if (P->rows) hipLaunchKernelGGL(p_to_agg_kernel, dim3(mgs_grid(P->rows, TB)), dim3(TB), 0, ctx->stream, P->rows, P->cols, P->rowptr, P->col, P->val, agg, bad.as<int>());
  int hbad = 0;
  MGS_HIP(ctx, hipMemcpyAsync(&hbad, bad.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (hbad == 0) return xfer_from_agg(ctx, P->rows, P->cols, agg, out);
  // general P: keep P and materialise Pᵀ (bicg.cpp:32)
  mgs_hip_free(agg);
  mgs_xfer *T = new mgs_xfer();
  T->ctx = ctx; T->n_fine = P->rows; T->n_coarse = P->cols; T->aggregation = false; T->nnz = P->nnz;
  int rc = mgs_csr_alloc(ctx, P->rows, P->cols, P->nnz, &T->P);
  if (rc == MGS_OK) {
    hipMemcpyAsync(T->P->rowptr, P->rowptr, sizeof(int) * ((size_t)P->rows + 1), hipMemcpyDeviceToDevice, ctx->stream);
    hipMemcpyAsync(T->P->col, P->col, sizeof(int) * (size_t)P->nnz, hipMemcpyDeviceToDevice, ctx->stream);
    hipMemcpyAsync(T->P->val, P->val, sizeof(double) * (size_t)P->nnz, hipMemcpyDeviceToDevice, ctx->stream);
    rc = mgs_plan_csr(T->P);
  }
  if (rc == MGS_OK) rc = k_transpose(T->P, &T->Pt);
  if (rc != MGS_OK) { mgs_xfer_destroy(T); return rc; }
  *out = T;
  return MGS_OK;
}

// A_c[I, colmap(j)] += a_ij for i in members(I).  colmap has one entry per column of A (owned
// rows first, then halo slots for row shards); ncols_out = number of coarse columns.
static int galerkin_core(const mgs_csr *A, int nc, const int *cptr, const int *members, const int *colmap, int ncols_out, mgs_csr **out) {
  mgs_ctx *ctx = A->ctx;
  DevBuf uniq, mx;
  MGS_TRY(dalloc<int>(ctx, uniq, (size_t)nc + 1));
  MGS_TRY(dalloc<int>(ctx, mx, 1));
  MGS_HIP(ctx, hipMemsetAsync(mx.p, 0, sizeof(int), ctx->stream));
  if (nc) hipLaunchKernelGGL(galerkin_maxub_kernel, dim3(mgs_grid(nc, TB)), dim3(TB), 0, ctx->stream, nc, cptr, members, A->rowptr, mx.as<int>());
  int maxub = 0;
  MGS_HIP(ctx, hipMemcpyAsync(&maxub, mx.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // slots per lane from the longest gathered row (an upper bound of its distinct columns); beyond 64 the long rows spill
  const int cap = maxub <= 16 ? 16 : (maxub <= 32 ? 32 : 64);
#define GAL_(CAPV, TBV, FILLV, CRP, CC, CV) hipLaunchKernelGGL((galerkin_lds_kernel<CAPV, TBV, FILLV>), dim3(mgs_grid(nc + 1, TBV)), dim3(TBV), 0, ctx->stream, nc, cptr, members, \
                                                              colmap, A->rowptr, A->col, A->val, CRP, uniq.as<int>(), CC, CV)
  if (cap == 16) GAL_(16, 256, false, nullptr, nullptr, nullptr); else if (cap == 32) GAL_(32, 128, false, nullptr, nullptr, nullptr); else GAL_(64, 64, false, nullptr, nullptr, nullptr);
  MGS_HIP(ctx, hipGetLastError());
  MGS_TRY(scan_rec(ctx, uniq.as<int>(), uniq.as<int>(), (int64_t)nc + 1));
  int nnzc = 0;
  MGS_HIP(ctx, hipMemcpyAsync(&nnzc, uniq.as<int>() + nc, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  mgs_csr *C = nullptr;
  MGS_TRY(mgs_csr_alloc(ctx, nc, ncols_out, nnzc, &C));
  MGS_HIP(ctx, hipMemcpyAsync(C->rowptr, uniq.p, sizeof(int) * ((size_t)nc + 1), hipMemcpyDeviceToDevice, ctx->stream));
  if (nc) { if (cap == 16) GAL_(16, 256, true, C->rowptr, C->col, C->val); else if (cap == 32) GAL_(32, 128, true, C->rowptr, C->col, C->val); else GAL_(64, 64, true, C->rowptr, C->col, C->val); }
#undef GAL_
  MGS_HIP(ctx, hipGetLastError());
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MGS_TRY(mgs_plan_csr(C));
  *out = C;
  return MGS_OK;
}

__global__ void colmap_ext_kernel(int n, int n_ext, int nc, const int *__restrict__ agg, const int *__restrict__ halo_map, int *__restrict__ colmap) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_ext) return;
  colmap[j] = j < n ? agg[j] : (halo_map ? halo_map[j - n] : nc + (j - n));   // identity on halo slots when no map is given
}

// square A: A_c = PᵀAP.  Row shard (cols > rows): halo_map gives the coarse column of every halo
// slot (device array of cols−rows ints, −1 = not aggregated), n_halo_c coarse halo slots.
int k_galerkin_agg_ext(const mgs_csr *A, const mgs_xfer *T, const int *halo_map_dev, int n_halo_c, mgs_csr **out) {
  mgs_ctx *ctx = A->ctx;
  MGS_CHECK(ctx, T->aggregation && T->n_fine == A->rows && A->rows <= A->cols, MGS_ERR_INVALID, "galerkin: shape mismatch");
  auto inherit = [&]() -> int {     // the coarse rows keep their origins (tie-break space of the next level's matching)
    if (!T->corigin || T->n_coarse <= 0) return MGS_OK;
    MGS_TRY(mgs_dev_alloc(ctx, &(*out)->origin, (size_t)T->n_coarse));
    MGS_HIP(ctx, hipMemcpyAsync((*out)->origin, T->corigin, sizeof(int) * (size_t)T->n_coarse, hipMemcpyDeviceToDevice, ctx->stream));
    MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MGS_OK;
  };
  if (A->rows == A->cols) { MGS_TRY(galerkin_core(A, T->n_coarse, T->cptr, T->members, T->agg, T->n_coarse, out)); return inherit(); }
  DevBuf cm;
  MGS_TRY(dalloc<int>(ctx, cm, (size_t)A->cols));
  const int n_halo = A->cols - A->rows;
  hipLaunchKernelGGL(colmap_ext_kernel, dim3(mgs_grid(A->cols, TB)), dim3(TB), 0, ctx->stream, A->rows, A->cols, T->n_coarse, T->agg, halo_map_dev, cm.as<int>());
  MGS_TRY(galerkin_core(A, T->n_coarse, T->cptr, T->members, cm.as<int>(), T->n_coarse + (halo_map_dev ? n_halo_c : n_halo), out));
  return inherit();
}
int k_galerkin_agg(const mgs_csr *A, const mgs_xfer *T, mgs_csr **out) { return k_galerkin_agg_ext(A, T, nullptr, 0, out); }

// A·P for an aggregation P (rows of A, one column per aggregate): the entries of a row that fall into one aggregate are summed in
// ascending column order, entries of columns outside every aggregate drop out.  Setup-time operand of the fused post pass
// (A·(P e_c) = (A·P) e_c).  On a row shard cmap_ext maps every local column — owned rows through agg, halo slots through the coarse
// halo map of the shard's Galerkin product — to the coarse level's local numbering (n_coarse + coarse halo slot), so the halo
// columns of A·P are the coarse level's own halo columns and the post pass's payload is the plain halo exchange of e_c.
int k_build_ap(const mgs_csr *A, const mgs_xfer *T, const int *cmap_ext, int ncols, mgs_csr **out) {
  mgs_ctx *ctx = A->ctx;
  MGS_CHECK(ctx, T->aggregation && T->n_fine == A->rows && A->rows <= A->cols, MGS_ERR_INVALID, "A·P: shape mismatch");
  if (!cmap_ext) {
    MGS_CHECK(ctx, A->rows == A->cols, MGS_ERR_INVALID, "A·P of a row shard needs the coarse map of its halo columns");
    return galerkin_core(A, A->rows, nullptr, nullptr, T->agg, T->n_coarse, out);
  }
  return galerkin_core(A, A->rows, nullptr, nullptr, cmap_ext, ncols, out);
}

// general P (not an aggregation): host Gustavson product, as the reference does with Eigen on
// the CPU at setup (bicg.cpp:33).  Setup only — never on the solve path.
int k_galerkin_general(const mgs_csr *A, const mgs_xfer *T, mgs_csr **out) {
  mgs_ctx *ctx = A->ctx;
  auto pull = [&](const mgs_csr *M, std::vector<int> &rp, std::vector<int> &ci, std::vector<double> &v) -> int {
    rp.resize((size_t)M->rows + 1); ci.resize((size_t)M->nnz); v.resize((size_t)M->nnz);
    return mgs_csr_download(M, rp.data(), ci.data(), v.data());
  };
  std::vector<int> arp, aci, prp, pci, trp, tci; std::vector<double> av, pv, tv;
  MGS_TRY(pull(A, arp, aci, av)); MGS_TRY(pull(T->P, prp, pci, pv)); MGS_TRY(pull(T->Pt, trp, tci, tv));
  auto spgemm = [](int n, int m, const std::vector<int> &rp1, const std::vector<int> &c1, const std::vector<double> &v1,
                   const std::vector<int> &rp2, const std::vector<int> &c2, const std::vector<double> &v2,
                   std::vector<int> &rpo, std::vector<int> &co, std::vector<double> &vo) {
    std::vector<double> acc((size_t)m, 0.0); std::vector<int> mark((size_t)m, -1);
    rpo.assign((size_t)n + 1, 0); co.clear(); vo.clear();
    for (int i = 0; i < n; ++i) {
      size_t base = co.size();
      for (int ka = rp1[i]; ka < rp1[i + 1]; ++ka) {
        int k = c1[ka]; double a = v1[ka];
        for (int kb = rp2[k]; kb < rp2[k + 1]; ++kb) {
          int j = c2[kb];
          if (mark[j] != i) { mark[j] = i; co.push_back(j); acc[j] = a * v2[kb]; } else acc[j] += a * v2[kb];
        }
      }
      std::sort(co.begin() + base, co.end());
      for (size_t q = base; q < co.size(); ++q) vo.push_back(acc[co[q]]);
      rpo[i + 1] = (int)co.size();
    }
  };
  std::vector<int> r1, c1, r2, c2; std::vector<double> v1, v2;
  spgemm(T->n_coarse, A->cols, trp, tci, tv, arp, aci, av, r1, c1, v1);
  spgemm(T->n_coarse, T->n_coarse, r1, c1, v1, prp, pci, pv, r2, c2, v2);
  return mgs_csr_upload(ctx, T->n_coarse, T->n_coarse, (int64_t)c2.size(), r2.data(), c2.data(), v2.data(), out);
}

// one pairwise pass on matrix M → agg ids (device array, caller frees) and count
static int pairwise_pass(const mgs_csr *M, double ktg, int first_pass, const int *origin, const int *zone, int **agg_out, int *nc_out) {
  mgs_ctx *ctx = M->ctx;
  const int n = M->rows;
  DevBuf diag, s, state, w, pick, cnt, flag, asym;
  MGS_TRY(dalloc<double>(ctx, diag, (size_t)n)); MGS_TRY(dalloc<double>(ctx, s, (size_t)n));
  MGS_TRY(dalloc<int>(ctx, state, (size_t)n)); MGS_TRY(dalloc<double>(ctx, w, (size_t)M->nnz));
  MGS_TRY(dalloc<int>(ctx, pick, (size_t)n)); MGS_TRY(dalloc<int>(ctx, cnt, 2)); MGS_TRY(dalloc<int>(ctx, flag, (size_t)n + 1));
  MGS_TRY(dalloc<int>(ctx, asym, 1));
  hipStream_t st = ctx->stream;
  const dim3 g(mgs_grid(n, TB)), b(TB);
  MGS_HIP(ctx, hipMemsetAsync(asym.p, 0, sizeof(int), st));
  hipLaunchKernelGGL(pattern_asym_kernel, g, b, 0, st, n, M->rowptr, M->col, asym.as<int>());
  int hasym = 0;
  MGS_HIP(ctx, hipMemcpyAsync(&hasym, asym.p, sizeof(int), hipMemcpyDeviceToHost, st));
  MGS_HIP(ctx, hipStreamSynchronize(st));
  mgs_csr *Mt = nullptr;
  if (hasym) MGS_TRY(k_transpose(M, &Mt));
  hipLaunchKernelGGL(agg_node_stats_kernel, g, b, 0, st, n, M->rowptr, M->col, M->val, Mt ? Mt->rowptr : nullptr, Mt ? Mt->col : nullptr,
                     Mt ? Mt->val : nullptr, ktg, first_pass, diag.as<double>(), s.as<double>(), state.as<int>());
  hipLaunchKernelGGL(agg_edge_weight_kernel, g, b, 0, st, n, M->rowptr, M->col, M->val, diag.as<double>(), s.as<double>(), state.as<int>(), ktg, zone, w.as<double>());
  MGS_HIP(ctx, hipGetLastError());
  const int MAX_ROUNDS = 96, MU_ROUNDS = 24;
  for (int round = 0; round < MAX_ROUNDS; ++round) {
    const int hash_only = round >= MU_ROUNDS;
    const int force = round == MAX_ROUNDS - 1;
    MGS_HIP(ctx, hipMemsetAsync(cnt.p, 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL(agg_pick_kernel, g, b, 0, st, n, M->rowptr, M->col, w.as<double>(), state.as<int>(), hash_only, origin, pick.as<int>());
    hipLaunchKernelGGL(agg_match_kernel, g, b, 0, st, n, pick.as<int>(), state.as<int>(), force, cnt.as<int>());
    int left = 0;
    MGS_HIP(ctx, hipMemcpyAsync(&left, cnt.p, sizeof(int), hipMemcpyDeviceToHost, st));
    MGS_HIP(ctx, hipStreamSynchronize(st));
    if (left == 0) break;
  }
  if (Mt) mgs_csr_destroy(Mt);
  hipLaunchKernelGGL(agg_leader_flag_kernel, dim3(mgs_grid(n + 1, TB)), b, 0, st, n, state.as<int>(), flag.as<int>());
  MGS_TRY(scan_rec(ctx, flag.as<int>(), flag.as<int>(), (int64_t)n + 1));
  int nc = 0;
  MGS_HIP(ctx, hipMemcpyAsync(&nc, flag.as<int>() + n, sizeof(int), hipMemcpyDeviceToHost, st));
  int *agg = nullptr;
  MGS_TRY(mgs_dev_alloc(ctx, &agg, (size_t)n));
  hipLaunchKernelGGL(agg_assign_kernel, g, b, 0, st, n, state.as<int>(), flag.as<int>(), agg);
  MGS_HIP(ctx, hipGetLastError());
  MGS_HIP(ctx, hipStreamSynchronize(st));
  *agg_out = agg; *nc_out = nc;
  return MGS_OK;
}

// multiple pairwise aggregation, AGMG.cpp:299-315 / main.cu:95-277
int k_pairwise_aggregate(const mgs_csr *A, double ktg, int npass, double tou, mgs_xfer **T_out, mgs_csr **Ac_out, const int *zone) {
  mgs_ctx *ctx = A->ctx;
  MGS_CHECK(ctx, A->rows <= A->cols && A->rows > 0, MGS_ERR_INVALID, "aggregate: need a non-empty operator with rows <= cols");
  MGS_CHECK(ctx, ktg > 2.0 && npass >= 1, MGS_ERR_INVALID, "aggregate: need ktg > 2 and npass >= 1");
  const int n = A->rows;
  int *agg = nullptr; int nc = 0;
  MGS_TRY(pairwise_pass(A, ktg, 1, A->origin, zone, &agg, &nc));
  mgs_xfer *T = nullptr;
  MGS_TRY(xfer_from_agg(ctx, n, nc, agg, &T));
  auto origin_of = [&](int nfine, const int *aggv, const int *org_fine, int ncoarse, int **out) -> int {
    MGS_TRY(mgs_dev_alloc(ctx, out, (size_t)std::max(ncoarse, 1)));
    MGS_HIP(ctx, hipMemsetAsync(*out, 0x7f, sizeof(int) * (size_t)std::max(ncoarse, 1), ctx->stream));
    if (nfine) hipLaunchKernelGGL(agg_origin_kernel, dim3(mgs_grid(nfine, TB)), dim3(TB), 0, ctx->stream, nfine, aggv, org_fine, *out);
    MGS_HIP(ctx, hipGetLastError());
    return MGS_OK;
  };
  { int rc0 = origin_of(n, T->agg, A->origin, nc, &T->corigin); if (rc0 != MGS_OK) { mgs_xfer_destroy(T); return rc0; } }
  mgs_csr *Abar = nullptr;
  int rc = k_galerkin_agg(A, T, &Abar);
  if (rc != MGS_OK) { mgs_xfer_destroy(T); return rc; }
  for (int s = 2; s <= npass; ++s) {
    if ((double)Abar->nnz <= (double)A->nnz / tou) break;               // AGMG.cpp:309
    if (Abar->rows <= 1) break;
    const int n_halo = A->cols - A->rows;                               // row shard: halo slots stay unaggregated here
    int *agg2 = nullptr; int nc2 = 0;
    DevBuf czone;                                                         // zones of the pairs (row shards: exported rows pair among themselves only)
    if (zone) {
      rc = dalloc<int>(ctx, czone, (size_t)std::max(Abar->rows, 1));
      if (rc != MGS_OK) break;
      hipMemsetAsync(czone.p, 0, sizeof(int) * (size_t)std::max(Abar->rows, 1), ctx->stream);
      hipLaunchKernelGGL(agg_zone_kernel, dim3(mgs_grid(n, TB)), dim3(TB), 0, ctx->stream, n, T->agg, zone, czone.as<int>());
    }
    rc = pairwise_pass(Abar, ktg, 0, T->corigin, zone ? czone.as<int>() : nullptr, &agg2, &nc2);
    if (rc != MGS_OK) break;
    int *org2 = nullptr;
    rc = origin_of(Abar->rows, agg2, T->corigin, nc2, &org2);
    if (rc != MGS_OK) { mgs_hip_free(agg2); if (org2) mgs_hip_free(org2); break; }
    // compose fine→pair→pair-of-pairs (AGMG.cpp:247-263) and rebuild member lists
    int *aggc = nullptr;
    rc = mgs_dev_alloc(ctx, &aggc, (size_t)n);
    if (rc != MGS_OK) { mgs_hip_free(agg2); mgs_hip_free(org2); break; }
    hipMemcpyAsync(aggc, T->agg, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream);
    hipLaunchKernelGGL(agg_compose_kernel, dim3(mgs_grid(n, TB)), dim3(TB), 0, ctx->stream, n, aggc, agg2);
    mgs_xfer *T2 = nullptr, *Tn = nullptr;
    rc = xfer_from_agg(ctx, Abar->rows, nc2, agg2, &T2);
    mgs_csr *Anew = nullptr;
    (void)n_halo;
    if (rc == MGS_OK) rc = k_galerkin_agg(Abar, T2, &Anew);          // (P1 P2)ᵀ A (P1 P2) = P2ᵀ A_bar P2
    if (T2) mgs_xfer_destroy(T2);
    if (rc == MGS_OK) rc = xfer_from_agg(ctx, n, nc2, aggc, &Tn); else mgs_hip_free(aggc);
    if (rc != MGS_OK) { if (Anew) mgs_csr_destroy(Anew); mgs_hip_free(org2); break; }
    Tn->corigin = org2;
    mgs_xfer_destroy(T); T = Tn;
    mgs_csr_destroy(Abar); Abar = Anew;
  }
  if (rc != MGS_OK) { mgs_xfer_destroy(T); mgs_csr_destroy(Abar); return rc; }
  // the coarse operator inherits its rows' origins (tie-breaks of the next level's matching)
  if (Abar->origin) { mgs_hip_free(Abar->origin); Abar->origin = nullptr; }
  rc = mgs_dev_alloc(ctx, &Abar->origin, (size_t)std::max(T->n_coarse, 1));
  if (rc == MGS_OK && hipMemcpyAsync(Abar->origin, T->corigin, sizeof(int) * (size_t)T->n_coarse, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "origin copy failed");
  if (rc == MGS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "origin copy failed");
  if (rc != MGS_OK) { mgs_xfer_destroy(T); mgs_csr_destroy(Abar); return rc; }
  *T_out = T;
  if (A->rows == A->cols) *Ac_out = Abar;
  else { mgs_csr_destroy(Abar); *Ac_out = nullptr; }   // shard: the caller resolves remote aggregates, then k_galerkin_agg_ext
  return MGS_OK;
}
