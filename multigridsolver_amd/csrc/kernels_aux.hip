// kernels_aux.hip — grid transfer (restriction / prolongation), BLAS-1, coarsest dense solve,
// on-device operator generators, halo pack.  gfx950 only.
#include "mgs_internal.hpp"

#include <chrono>
#include <time.h>

namespace {
constexpr int TB = 256;

// ------------------------------------------------------------------ diag / elementwise
// dinv_i = 1/a_ii (reference src/CPU_Matlab/solve.m:17).  Binary search in the sorted row,
// as src/GPU_CUDAC++/MatrixAccess.cu:28-47 does for element access.
__global__ void diag_inv_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                const double *__restrict__ val, double *__restrict__ dinv, int *__restrict__ bad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int lo = rowptr[i], hi = rowptr[i + 1] - 1;
  double d = 0.0;
  while (lo <= hi) {
    int mid = lo + ((hi - lo) >> 1);   // lo + hi would overflow int32 near 2^31 entries
    int c = col[mid];
    if (c == i) { d = val[mid]; break; }
    if (c < i) lo = mid + 1; else hi = mid - 1;
  }
  if (d == 0.0) { atomicAdd(bad, 1); dinv[i] = 0.0; }
  else dinv[i] = 1.0 / d;
}

__global__ void fill_kernel(double *__restrict__ d, int64_t n, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) d[i] = v;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void rand_kernel(double *__restrict__ d, int64_t n, uint64_t seed, int64_t off) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t h = splitmix64(seed * 0xD1342543DE82EF95ull + (uint64_t)(i + off));
    d[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0);
  }
}

__global__ void axpby_kernel(int64_t n, double a, const double *__restrict__ x, double b, double *__restrict__ y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (b == 0.0) { for (; i < n; i += stride) y[i] = a * x[i]; }
  else { for (; i < n; i += stride) y[i] = a * x[i] + b * y[i]; }
}
__global__ void axpbypcz_kernel(int64_t n, double a, const double *__restrict__ x, double b,
                                const double *__restrict__ y, double c, double *__restrict__ z) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (c == 0.0) { for (; i < n; i += stride) z[i] = a * x[i] + b * y[i]; }
  else { for (; i < n; i += stride) z[i] = a * x[i] + b * y[i] + c * z[i]; }
}

// 16-byte forms of the two update kernels (option blas1_vec, default on; used when every operand is 16-byte aligned): each lane
// moves a pair of doubles per load/store and the loop is unrolled so that several loads per stream are in flight per wave — the
// 8-byte grid-stride loops above stop at ≈ 4 TB/s on this part, a pure stream wants ≥ 16 B per lane (MI355X_MICROARCH.md, HBM:
// 6.3 TB/s for a float4 copy).  Element i is computed by the same expression as above, so the results have the same bits.
typedef double vd2 __attribute__((ext_vector_type(2)));
// streaming store (the instruction spelled out: see st_stream in kernels_spmv.hip); nothing here reads back what it stored
__device__ __forceinline__ void st2(vd2 *p, vd2 v, bool nt) {
  if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
  else *p = v;
}
template <bool HASB>
__global__ __launch_bounds__(TB) void axpby_vec_kernel(int64_t n, double a, const double *__restrict__ x, double b, double *__restrict__ y, int nts) {
  const int64_t n2 = n >> 1;
  const vd2 *__restrict__ xv = reinterpret_cast<const vd2 *>(x);
  vd2 *__restrict__ yv = reinterpret_cast<vd2 *>(y);
  const int64_t stride = (int64_t)gridDim.x * TB;
#pragma unroll 4
  for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n2; i += stride) {
    const vd2 p = xv[i]; vd2 q;
    if (HASB) { const vd2 o = yv[i]; q.x = a * p.x + b * o.x; q.y = a * p.y + b * o.y; }
    else { q.x = a * p.x; q.y = a * p.y; }
    st2(yv + i, q, nts != 0);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = HASB ? a * x[n - 1] + b * y[n - 1] : a * x[n - 1];
}
template <bool HASC>
__global__ __launch_bounds__(TB) void axpbypcz_vec_kernel(int64_t n, double a, const double *__restrict__ x, double b,
                                                          const double *__restrict__ y, double c, double *__restrict__ z, int nts) {
  const int64_t n2 = n >> 1;
  const vd2 *__restrict__ xv = reinterpret_cast<const vd2 *>(x);
  const vd2 *__restrict__ yv = reinterpret_cast<const vd2 *>(y);
  vd2 *__restrict__ zv = reinterpret_cast<vd2 *>(z);
  const int64_t stride = (int64_t)gridDim.x * TB;
#pragma unroll 4
  for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n2; i += stride) {
    const vd2 p = xv[i], o = yv[i]; vd2 q;
    if (HASC) { const vd2 w = zv[i]; q.x = a * p.x + b * o.x + c * w.x; q.y = a * p.y + b * o.y + c * w.y; }
    else { q.x = a * p.x + b * o.x; q.y = a * p.y + b * o.y; }
    st2(zv + i, q, nts != 0);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) z[n - 1] = HASC ? a * x[n - 1] + b * y[n - 1] + c * z[n - 1] : a * x[n - 1] + b * y[n - 1];
}
static inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// first damped-Jacobi sweep from x = 0: x = 0 + (ωD⁻¹)(b − A·0) = (ω·dinv_i)·b_i, bit-identical
__global__ void jacobi_zero_kernel(int n, double omega, const double *__restrict__ dinv, const double *__restrict__ b, double *__restrict__ x) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (omega * dinv[i]) * b[i];
}

// ------------------------------------------------------------------ grid transfer
// r_c = Pᵀ r for an aggregation P (unit values are not loaded): `Ptrans * vec`,
// reference bicg.cpp:48.  One lane per aggregate, members summed in ascending fine-row
// order == Eigen's row-major SpMV order with values 1.0 → bit-identical.
__global__ void restrict_agg_kernel(int nc, const int *__restrict__ cptr, const int *__restrict__ members,
                                    const double *__restrict__ r, double *__restrict__ rc) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  // aggregates of the pairwise passes hold at most 4 members: the first four member indices, then their four residuals, are loaded
  // together (predicated, no loop: a loop drains every outstanding load at its header, and member → residual is a dependent pair,
  // i.e. eight serialized memory latencies per lane in the looped form); the sum keeps the ascending order.  Longer lists continue in a loop.
  const int k0 = cptr[c], e = cptr[c + 1];
  int m[4]; double v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) m[q] = k0 + q < e ? members[k0 + q] : -1;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = m[q] >= 0 ? r[m[q]] : 0.0;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) if (k0 + q < e) s += v[q];
  for (int k = k0 + 4; k < e; ++k) s += r[members[k]];
  rc[c] = s;
}
// Small levels (launch-bound, a few µs per dispatch whatever they do): the zero-guess pre pass r = b − Â·b and the restriction
// r_c = Pᵀr in ONE kernel, aggregate-parallel — four lanes per aggregate, lane q walks the row of member q (plain CSR of Â, ascending
// columns, unfused multiply/add from 0.0: the row-block kernel's arithmetic), stores r for the post pass, and lane 0 adds the members'
// residuals in ascending member order (restrict_agg_kernel's order): same bits as the two kernels it replaces, one dispatch less per
// level.  Rows outside every aggregate (G0) get their residual from the tail of the grid.  hv: halo payload of a row shard (NULL: none).
__global__ __launch_bounds__(TB) void agg_pre_kernel(int n, int nc, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                      const double *__restrict__ valhat, const double *__restrict__ b, const double *__restrict__ hv,
                                                      const int *__restrict__ cptr, const int *__restrict__ members, const int *__restrict__ agg,
                                                      double *__restrict__ r_out, double *__restrict__ rc_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  auto row_residual = [&](int m) -> double {
    // eight entries per step, every load of a step issued before the first is waited for (a loop over single entries drains its
    // loads at every header: two serialized memory latencies per ENTRY); the sum itself stays sequential in ascending column order
    double s = 0.0;
    const int e0 = rowptr[m], ee = rowptr[m + 1];
    const double bm = b[m];
    for (int e = e0; e < ee; e += 8) {
      int c[8]; double v[8], xv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) c[q] = e + q < ee ? col[e + q] : -1;
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = e + q < ee ? valhat[e + q] : 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) xv[q] = c[q] < 0 ? 0.0 : (c[q] < n ? b[c[q]] : hv[c[q] - n]);
#pragma unroll
      for (int q = 0; q < 8; ++q) if (e + q < ee) s += v[q] * xv[q];
    }
    return bm - s;
  };
  const int a = t >> 2, q = t & 3;
  if (a < nc) {                                   // the four lanes of a group share a: same trip counts, converged shuffles
    const int k0 = cptr[a], ke = cptr[a + 1];
    double acc = 0.0;
    for (int kb = k0; kb < ke; kb += 4) {         // one round for the aggregates of the pairwise passes (≤ 4 members)
      double rm = 0.0;
      if (kb + q < ke) { const int m = members[kb + q]; rm = row_residual(m); r_out[m] = rm; }
#pragma unroll
      for (int j = 0; j < 4; ++j) { const double v = __shfl(rm, j, 4); if (kb + j < ke) acc += v; }
    }
    if (q == 0) rc_out[a] = acc;
  } else {
    const int row = t - 4 * nc;
    if (row < n && agg[row] < 0) r_out[row] = row_residual(row);
  }
}
// e = P e_c / x += P e_c: `P * (...)`, reference bicg.cpp:48; P has ≤1 unit entry per row
// (src/CPU_C++/AGMG.cpp:181-186) so e_i = e_c[agg(i)] or 0 for G0 rows.
template <int ADD>
__global__ void prolong_agg_kernel(int n, const int *__restrict__ agg, const double *__restrict__ ec, double *__restrict__ x) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int a = agg[i];
  double e = a >= 0 ? ec[a] : 0.0;
  if (ADD) x[i] += e; else x[i] = e;
}

__global__ void gather_kernel(const double *__restrict__ x, const int *__restrict__ idx, int64_t n, double *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[idx[i]];
}

// setup-time operands of the fused cycle passes (unsharded levels): scaled values and aggregate-mapped columns
// position of the diagonal entry inside its row (0..254; 255 = absent or further in): the t-form post pass reads a_ii from there
__global__ void diag_pos_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col, unsigned char *__restrict__ dpos) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = rowptr[i];
  int lo = a, hi = rowptr[i + 1] - 1, pos = 255;
  while (lo <= hi) {
    int mid = lo + ((hi - lo) >> 1);
    int c = col[mid];
    if (c == i) { pos = mid - a < 255 ? mid - a : 255; break; }
    if (c < i) lo = mid + 1; else hi = mid - 1;
  }
  dpos[i] = (unsigned char)pos;
}
__global__ void scale_vals_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                                  const double *__restrict__ wd, double *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // row shards: wd's halo part holds the owners' ω/a_jj (fetched once at setup), so a halo column is scaled like an owned one and
  // the pre pass's payload is the peers' raw right-hand side
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) out[k] = val[k] * wd[col[k]];
}
__global__ void map_cols_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const int *__restrict__ cmap, int *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) out[k] = cmap[col[k]];
}
__global__ void concat_i32_kernel(const int *__restrict__ a, int na, const int *__restrict__ b, int nb, int *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < na) out[i] = a[i]; else if (i < na + nb) out[i] = b[i - na];
}

// ------------------------------------------------------------------ reductions
// Two-stage deterministic dot: fixed grid of partials (independent of scheduling), then one
// block folds them in index order.  reference bicg.cpp:64-72 (dot/norm helpers).
// where the last stage of a reduction posts its results for the host (see fetch_results): mapped host buffer + ticket; hv = NULL: nowhere
struct Post { double *hv; unsigned long long *ht; unsigned long long ticket; };
__device__ __forceinline__ void post_to_host(const Post &pa, const double *vals, int cnt) {
  for (int q = 0; q < cnt; ++q) __hip_atomic_store(pa.hv + q, vals[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(pa.ht, pa.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr int DOT_BLOCKS = 4096;   // partial slots (a pair kernel uses half of them per sum): 8 workgroups per CU on 256 CUs
__global__ __launch_bounds__(TB) void dot_partial_kernel(int64_t n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ part) {
  __shared__ double sh[TB / 64];
  double s = 0.0;
  int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * TB;
  for (; i < n; i += stride) s += x[i] * y[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < TB / 64; ++w) t += sh[w]; part[blockIdx.x] = t; }
}
__global__ __launch_bounds__(TB) void dot_final_kernel(int nb, const double *__restrict__ part, double *__restrict__ out, Post pa) {
  __shared__ double sh[TB];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += TB) s += part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = TB / 2; w > 0; w >>= 1) { if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) { const double r = sh[0]; out[0] = r; if (pa.hv) post_to_host(pa, &r, 1); }
}

// two inner products in one pass and one host round trip: (x·y, z·w) — BiCGSTAB's (t·s, t·t) and (r·r, r̃·r)
__global__ __launch_bounds__(TB) void dot2_partial_kernel(int64_t n, const double *__restrict__ x, const double *__restrict__ y,
                                                          const double *__restrict__ z, const double *__restrict__ w, double *__restrict__ part) {
  __shared__ double sh[2][TB / 64];
  double s0 = 0.0, s1 = 0.0;
  int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * TB;
  for (; i < n; i += stride) { s0 += x[i] * y[i]; s1 += z[i] * w[i]; }
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int q = 0; q < TB / 64; ++q) { t0 += sh[0][q]; t1 += sh[1][q]; }
    part[blockIdx.x] = t0; part[gridDim.x + blockIdx.x] = t1;
  }
}
__global__ __launch_bounds__(TB) void dot2_final_kernel(int nb, const double *__restrict__ part, double *__restrict__ out, Post pa) {
  __shared__ double sh[TB];
  double res[2] = {0.0, 0.0};
  for (int q = 0; q < 2; ++q) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += TB) s += part[q * nb + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = TB / 2; w > 0; w >>= 1) { if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) { res[q] = sh[0]; out[q] = sh[0]; }
    __syncthreads();
  }
  if (threadIdx.x == 0 && pa.hv) post_to_host(pa, res, 2);
}

// z = a·x + b·y fused with (z·z, w·z): BiCGSTAB's s = r − αv with ‖s‖² and r = s − ωt with (‖r‖², r̃·r) in one pass each
__global__ __launch_bounds__(TB) void update_dot2_partial_kernel(int64_t n, double a, const double *__restrict__ x, double b, const double *__restrict__ y,
                                                                 double *__restrict__ z, const double *__restrict__ w, double *__restrict__ part) {
  __shared__ double sh[2][TB / 64];
  double s0 = 0.0, s1 = 0.0;
  int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * TB;
  for (; i < n; i += stride) {
    const double zi = a * x[i] + b * y[i];
    z[i] = zi;
    s0 += zi * zi;
    if (w) s1 += w[i] * zi;
  }
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int q = 0; q < TB / 64; ++q) { t0 += sh[0][q]; t1 += sh[1][q]; }
    part[blockIdx.x] = t0; part[gridDim.x + blockIdx.x] = t1;
  }
}

// 16-byte form of the kernel above (same per-element expression; the partial sums associate differently: lane t adds its pairs
// x then y, a fixed order that does not depend on scheduling)
template <bool HASW>
__global__ __launch_bounds__(TB) void update_dot2_partial_vec_kernel(int64_t n, double a, const double *__restrict__ x, double b, const double *__restrict__ y,
                                                                     double *__restrict__ z, const double *__restrict__ w, double *__restrict__ part, int nts) {
  __shared__ double sh[2][TB / 64];
  const int64_t n2 = n >> 1;
  const vd2 *__restrict__ xv = reinterpret_cast<const vd2 *>(x);
  const vd2 *__restrict__ yv = reinterpret_cast<const vd2 *>(y);
  const vd2 *__restrict__ wv = reinterpret_cast<const vd2 *>(w);
  vd2 *__restrict__ zv = reinterpret_cast<vd2 *>(z);
  double s0 = 0.0, s1 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * TB;
#pragma unroll 4
  for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n2; i += stride) {
    const vd2 p = xv[i], o = yv[i]; vd2 q;
    q.x = a * p.x + b * o.x; q.y = a * p.y + b * o.y;
    st2(zv + i, q, nts != 0);
    s0 += q.x * q.x; s0 += q.y * q.y;
    if (HASW) { const vd2 ww = wv[i]; s1 += ww.x * q.x; s1 += ww.y * q.y; }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double zi = a * x[n - 1] + b * y[n - 1];
    z[n - 1] = zi; s0 += zi * zi;
    if (HASW) s1 += w[n - 1] * zi;
  }
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int q = 0; q < TB / 64; ++q) { t0 += sh[0][q]; t1 += sh[1][q]; }
    part[blockIdx.x] = t0; part[gridDim.x + blockIdx.x] = t1;
  }
}

// one or two inner products, one pair of elements per lane, one-shot workgroups (see grid_vec): partial pair per workgroup ([2][gridDim.x];
// NP = 1 leaves the second sum at zero so that the staged fold of the pairs serves both)
template <int NP>
__global__ __launch_bounds__(TB) void dot_block_vec_kernel(int64_t n, const double *__restrict__ x, const double *__restrict__ y,
                                                           const double *__restrict__ z, const double *__restrict__ w, double *__restrict__ part) {
  __shared__ double sh[2][TB / 64];
  const int64_t n2 = n >> 1;
  const vd2 *__restrict__ xv = reinterpret_cast<const vd2 *>(x);
  const vd2 *__restrict__ yv = reinterpret_cast<const vd2 *>(y);
  const vd2 *__restrict__ zv = reinterpret_cast<const vd2 *>(z);
  const vd2 *__restrict__ wv = reinterpret_cast<const vd2 *>(w);
  double s0 = 0.0, s1 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * TB;
  for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n2; i += stride) {
    const vd2 p = xv[i], q = yv[i];
    s0 += p.x * q.x; s0 += p.y * q.y;
    if (NP == 2) { const vd2 a = zv[i], b = wv[i]; s1 += a.x * b.x; s1 += a.y * b.y; }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { s0 += x[n - 1] * y[n - 1]; if (NP == 2) s1 += z[n - 1] * w[n - 1]; }
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int q = 0; q < TB / 64; ++q) { t0 += sh[0][q]; t1 += sh[1][q]; }
    part[blockIdx.x] = t0; part[gridDim.x + blockIdx.x] = t1;
  }
}

// ------------------------------------------------------------------ several inner products against one vector, one pass
// s_k = Σ a_i·b_k[i], k < K ≤ MDOT_MAX: `a` is read once.  Partials [K][gridDim.x] in a fixed order (no atomics), folded by
// mdot_final_kernel — the K-cycle's (ρ1, α1) and the flexible Krylov methods' orthogonalisation coefficients.
constexpr int MDOT_MAX = 16;
struct MDotVecs { const double *b[MDOT_MAX]; };
__global__ __launch_bounds__(TB) void mdot_partial_kernel(int64_t n, int K, const double *__restrict__ a, const MDotVecs B, double *__restrict__ part) {
  __shared__ double sh[MDOT_MAX][TB / 64];
  double s[MDOT_MAX];
#pragma unroll
  for (int k = 0; k < MDOT_MAX; ++k) s[k] = 0.0;
  const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * TB;
  const vd2 *__restrict__ av = reinterpret_cast<const vd2 *>(a);
  for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n2; i += stride) {
    const vd2 ai = av[i];
#pragma unroll
    for (int k = 0; k < MDOT_MAX; ++k)
      if (k < K) { const vd2 bi = reinterpret_cast<const vd2 *>(B.b[k])[i]; s[k] += ai.x * bi.x; s[k] += ai.y * bi.y; }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < MDOT_MAX; ++k) if (k < K) s[k] += a[n - 1] * B.b[k][n - 1];
  }
#pragma unroll
  for (int k = 0; k < MDOT_MAX; ++k) {
    if (k < K) {
      double t = s[k];
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
      if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = t;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < K) {
    double t = 0.0;
    for (int q = 0; q < TB / 64; ++q) t += sh[threadIdx.x][q];
    part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = t;
  }
}
// fold of K partial arrays of nb entries each (one workgroup, fixed order); out (device) and, with pa.hv, the host's mapped buffer
__global__ __launch_bounds__(TB) void mdot_final_kernel(int nb, int K, const double *__restrict__ part, double *__restrict__ out, Post pa) {
  __shared__ double sh[TB];
  __shared__ double res[MDOT_MAX + 4];
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += TB) s += part[(size_t)k * nb + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = TB / 2; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) { res[k] = sh[0]; if (out) out[k] = sh[0]; }
    __syncthreads();
  }
  if (threadIdx.x == 0 && pa.hv) post_to_host(pa, res, K);
}

// z = x − Σ_{j<K} coef_j·w_j (coefficients by value: they come from the host, which needed them for its own bookkeeping), with
// (z·z, z·u) accumulated in the same pass (u = NULL: second sum stays 0) — the orthogonalisation update of the flexible Krylov
// methods: the new direction against the window, its norm and its product with the residual, one read of every vector.
struct MAxpyArgs { const double *w[MDOT_MAX]; double coef[MDOT_MAX]; };
template <bool VEC>
__global__ __launch_bounds__(TB) void maxpy_dot2_kernel(int64_t n, int K, const double *__restrict__ x, const MAxpyArgs W, double *__restrict__ z,
                                                        const double *__restrict__ u, const double *__restrict__ u2, double *__restrict__ part) {
  __shared__ double sh[2][TB / 64];
  double s0 = 0.0, s1 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * TB;
  if (VEC) {              // 16 bytes per lane and stream (every operand 16-byte aligned; an odd last element goes to one lane below)
    const int64_t n2 = n >> 1;
    const vd2 *__restrict__ xv = reinterpret_cast<const vd2 *>(x);
    vd2 *__restrict__ zv = reinterpret_cast<vd2 *>(z);
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n2; i += stride) {
      vd2 zi = xv[i];
#pragma unroll
      for (int k = 0; k < MDOT_MAX; ++k)
        if (k < K) { const vd2 wi = reinterpret_cast<const vd2 *>(W.w[k])[i]; zi.x -= W.coef[k] * wi.x; zi.y -= W.coef[k] * wi.y; }
      zv[i] = zi;
      if (part) {
        if (u2) { const vd2 q = reinterpret_cast<const vd2 *>(u2)[i]; s0 += zi.x * q.x; s0 += zi.y * q.y; } else { s0 += zi.x * zi.x; s0 += zi.y * zi.y; }
        if (u) { const vd2 q = reinterpret_cast<const vd2 *>(u)[i]; s1 += zi.x * q.x; s1 += zi.y * q.y; }
      }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      double zi = x[n - 1];
      for (int k = 0; k < K; ++k) zi -= W.coef[k] * W.w[k][n - 1];
      z[n - 1] = zi;
      s0 += zi * (u2 ? u2[n - 1] : zi);
      if (u) s1 += zi * u[n - 1];
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += stride) {
      double zi = x[i];
#pragma unroll
      for (int k = 0; k < MDOT_MAX; ++k) if (k < K) zi -= W.coef[k] * W.w[k][i];
      z[i] = zi;
      s0 += zi * (u2 ? u2[i] : zi);
      if (u) s1 += zi * u[i];
    }
  }
  if (!part) return;                               // plain multi-axpy
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int q = 0; q < TB / 64; ++q) { t0 += sh[0][q]; t1 += sh[1][q]; }
    part[blockIdx.x] = t0; part[gridDim.x + blockIdx.x] = t1;
  }
}

// ------------------------------------------------------------------ K-cycle helpers (device-resident scalars)
// Two Krylov steps on the coarse problem (Notay, SISC 34 (2012), K-cycle; docs/AGMG_For_Convection_Diffusion.pdf §3.1) with the second
// direction orthogonalised EXPLICITLY: scal = {ρ1 = d1·v1, α1 = d1·r, γ = d2·v1, ρ2 = d2'·v2', α2 = d2'·r'} with g = γ/ρ1,
// c2' = c2 − g·c1, v2' = v2 − g·v1 and d = v (GCR form) or d = c (energy form).  ρ2 is a sum of products of the orthogonalised vectors,
// not the difference β − γ²/ρ1 of two nearly equal numbers (round 3: lost up to five digits where c2 is almost parallel to c1).
// No host round trip, graph-capturable.
__global__ void kc_update_r_kernel(int n, const double *__restrict__ scal, const double *__restrict__ r, const double *__restrict__ v1, double *__restrict__ rp) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double rho1 = scal[0], a = rho1 != 0.0 ? scal[1] / rho1 : 0.0;
  rp[i] = r[i] - a * v1[i];
}
// (ρ2, α2) of the orthogonalised second direction: partial pair per workgroup
template <bool ENERGY>
__global__ __launch_bounds__(TB) void kc_orth_dots_kernel(int n, const double *__restrict__ scal, const double *__restrict__ c1, const double *__restrict__ c2,
                                                          const double *__restrict__ v1, const double *__restrict__ v2, const double *__restrict__ rp,
                                                          double *__restrict__ part) {
  __shared__ double sh[2][TB / 64];
  const double rho1 = scal[0], g = rho1 != 0.0 ? scal[2] / rho1 : 0.0;
  double s0 = 0.0, s1 = 0.0;
  const int stride = gridDim.x * TB;
  for (int i = blockIdx.x * TB + threadIdx.x; i < n; i += stride) {
    const double v2o = v2[i] - g * v1[i];
    const double d2o = ENERGY ? c2[i] - g * c1[i] : v2o;
    s0 += d2o * v2o;
    s1 += d2o * rp[i];
  }
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int q = 0; q < TB / 64; ++q) { t0 += sh[0][q]; t1 += sh[1][q]; }
    part[blockIdx.x] = t0; part[gridDim.x + blockIdx.x] = t1;
  }
}
// x = (α1/ρ1)·c1 + (α2/ρ2)·(c2 − g·c1)
__global__ void kc_combine_kernel(int n, const double *__restrict__ scal, const double *__restrict__ c1, const double *__restrict__ c2, double *__restrict__ x) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double rho1 = scal[0], alpha1 = scal[1], gamma = scal[2], rho2 = scal[3], alpha2 = scal[4];
  double k1 = 0.0, k2 = 0.0;
  if (rho1 != 0.0) {
    k1 = alpha1 / rho1;
    if (rho2 > 0.0) { k2 = alpha2 / rho2; k1 -= (gamma / rho1) * k2; }
  }
  x[i] = k1 * c1[i] + k2 * c2[i];
}

// ------------------------------------------------------------------ coarsest level: dense
// x = A⁻¹ b with the explicit inverse (stands in for SparseLU::solve, reference bicg.cpp:48).
// One wavefront per row, lanes stride the columns (coalesced), shuffle reduction.
__global__ __launch_bounds__(TB) void dense_gemv_kernel(int n, const double *__restrict__ M, const double *__restrict__ b, double *__restrict__ x) {
  int row = blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  const double *m = M + (size_t)row * n;
  double s = 0.0;
  for (int j = lane; j < n; j += 64) s += m[j] * b[j];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) x[row] = s;
}

// Gauss-Jordan inversion with partial pivoting on the augmented matrix [A | I] (n × 2n).
__global__ void dense_scatter_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, double *__restrict__ W) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double *w = W + (size_t)i * 2 * n;
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) w[col[k]] += val[k];
  w[n + i] = 1.0;
}
__global__ __launch_bounds__(TB) void gj_pivot_kernel(int n, int k, const double *__restrict__ W, int *__restrict__ piv, double *__restrict__ pivval) {
  __shared__ double bv[TB]; __shared__ int bi[TB];
  double best = -1.0; int idx = k;
  for (int i = k + threadIdx.x; i < n; i += TB) {
    double v = fabs(W[(size_t)i * 2 * n + k]);
    if (v > best) { best = v; idx = i; }
  }
  bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
  __syncthreads();
  for (int w = TB / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      double o = bv[threadIdx.x + w]; int oi = bi[threadIdx.x + w];
      if (o > bv[threadIdx.x] || (o == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = o; bi[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { piv[0] = bi[0]; if (bv[0] == 0.0) piv[1] = 1; pivval[0] = W[(size_t)bi[0] * 2 * n + k]; }
}
__global__ void gj_swap_scale_kernel(int n, int k, double *__restrict__ W, const int *__restrict__ piv, const double *__restrict__ pivval) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int p = piv[0];
  if (j < 2 * n) {
    double a = W[(size_t)k * 2 * n + j], b = W[(size_t)p * 2 * n + j];
    double d = pivval[0];
    if (d == 0.0) d = 1.0;
    W[(size_t)p * 2 * n + j] = a;       // (p == k: a == b, harmless)
    W[(size_t)k * 2 * n + j] = b / d;
  }
}
__global__ void gj_colsave_kernel(int n, int k, const double *__restrict__ W, double *__restrict__ colk) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) colk[i] = (i == k) ? 0.0 : W[(size_t)i * 2 * n + k];
}
__global__ void gj_eliminate_kernel(int n, int k, double *__restrict__ W, const double *__restrict__ colk) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (j >= 2 * n) return;
  double f = colk[i];
  if (f != 0.0) W[(size_t)i * 2 * n + j] -= f * W[(size_t)k * 2 * n + j];
}
__global__ void gj_extract_kernel(int n, const double *__restrict__ W, double *__restrict__ inv) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (j < n) inv[(size_t)i * n + j] = W[(size_t)i * 2 * n + n + j];
}

// ------------------------------------------------------------------ generators
// 7-point 3-D Poisson, SURVEY §8d row d2 (3-D extension of reference src/common/poisson.cpp:11-33).
// closed-form rowptr: entries before row e = 7e − (#missing neighbours in rows < e)
__device__ __forceinline__ int64_t p3_rowptr(int N, int64_t e) {
  const int64_t N2 = (int64_t)N * N;
  if (e >= N2 * N) return 7 * N2 * N - 6 * N2;
  int i = (int)(e / N2); int rem = (int)(e - (int64_t)i * N2); int j = rem / N; int k = rem - j * N;
  // missing i−: rows with i==0 before e; missing i+: rows with i==N-1 before e
  int64_t miss = 0;
  miss += (i > 0) ? N2 : rem;                     // i == 0 rows
  miss += (i == N - 1) ? rem : 0;                 // i == N-1 rows
  // j == 0 rows before e: per full plane N, in current plane: (j>0 ? N : k)
  miss += (int64_t)i * N + ((j > 0) ? N : k);
  miss += (int64_t)i * N + ((j == N - 1) ? k : 0);   // j == N-1 rows
  // k == 0 rows before e: one per line
  int64_t lines = (int64_t)i * N + j;
  miss += lines + (k > 0 ? 1 : 0);
  miss += lines;                                      // k == N-1 rows of complete lines
  return 7 * e - miss;
}
__global__ void poisson3d_kernel(int N, int plane_lo, int64_t n_loc, int local_cols, int has_lo, int has_hi,
                                 int *__restrict__ rowptr, int *__restrict__ col, double *__restrict__ val) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n_loc) return;
  const int64_t N2 = (int64_t)N * N;
  const int64_t e0 = (int64_t)plane_lo * N2;
  const int64_t base = p3_rowptr(N, e0);
  const int64_t e = e0 + t;
  const int64_t p0 = p3_rowptr(N, e) - base;
  rowptr[t] = (int)p0;
  if (t == n_loc) return;
  int i = (int)(e / N2); int rem = (int)(e - (int64_t)i * N2); int j = rem / N; int k = rem - j * N;
  int64_t p = p0;
  // local numbering: owned rows [0,n_loc), lower halo plane [n_loc, n_loc+N2), upper after it
  const int64_t lo_halo = n_loc, hi_halo = n_loc + (has_lo ? N2 : 0);
  auto cidx = [&](int64_t g) -> int {
    if (!local_cols) return (int)g;
    int64_t l = g - e0;
    if (l < 0) return (int)(lo_halo + (l + N2));
    if (l >= n_loc) return (int)(hi_halo + (l - n_loc));
    return (int)l;
  };
  // rows stay sorted by (local) column: with local numbering the halo columns are the largest,
  // so an off-shard e−N² / e+N² neighbour moves to the end of the row
  const bool lo_is_halo = local_cols && i > 0 && i == plane_lo;
  const bool hi_is_halo = local_cols && i < N - 1 && (e + N2 - e0) >= n_loc;
  if (i > 0 && !lo_is_halo)     { col[p] = cidx(e - N2); val[p++] = -1.0; }
  if (j > 0)     { col[p] = cidx(e - N);  val[p++] = -1.0; }
  if (k > 0)     { col[p] = cidx(e - 1);  val[p++] = -1.0; }
  col[p] = cidx(e); val[p++] = 6.0;
  if (k < N - 1) { col[p] = cidx(e + 1);  val[p++] = -1.0; }
  if (j < N - 1) { col[p] = cidx(e + N);  val[p++] = -1.0; }
  if (i < N - 1 && !hi_is_halo) { col[p] = cidx(e + N2); val[p++] = -1.0; }
  if (lo_is_halo) { col[p] = cidx(e - N2); val[p++] = -1.0; }
  if (hi_is_halo) { col[p] = cidx(e + N2); val[p++] = -1.0; }
}
// 2-D 5-point, exactly reference src/common/poisson.cpp:9-37.
__global__ void poisson2d_kernel(int n, int *__restrict__ rowptr, int *__restrict__ col, double *__restrict__ val) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  int N = n * n;
  if (e > N) return;
  auto rp = [&](int r) -> int {
    if (r >= N) return 5 * N - 4 * n;
    int i = r / n, j = r % n;
    // missing: i==0 rows, i==n-1 rows, j==0 rows, j==n-1 rows before r
    int miss = (i > 0 ? n : j) + (i == n - 1 ? j : 0) + (i + (j > 0 ? 1 : 0)) + i;
    return 5 * r - miss;
  };
  int p = rp(e);
  rowptr[e] = p;
  if (e == N) return;
  int i = e / n, j = e % n;
  if (i > 0)     { col[p] = e - n; val[p++] = -1.0; }
  if (j > 0)     { col[p] = e - 1; val[p++] = -1.0; }
  col[p] = e; val[p++] = 4.0;
  if (j < n - 1) { col[p] = e + 1; val[p++] = -1.0; }
  if (i < n - 1) { col[p] = e + n; val[p++] = -1.0; }
}

}  // namespace

// ============================================================ host launchers
// Grid of the 16-byte update kernels: ONE pair per lane, one-shot workgroups (the loop in the kernels then runs once).  Measured at
// 512³ (tools/studies_r1_r3/blas1_grid.py): axpby 0.695 ms with the capped persistent grid (n_cu × 8 workgroups, grid-stride, unroll 4), 0.541 ms
// one-shot (5.96 TB/s); 2 / 4 / 8 pairs per lane: 0.568 / 0.623 / 0.671 ms — on this part a stream wants many short-lived
// workgroups, not a few resident ones (the row-block kernels are one-shot already; tools/microbench/rw_mix.hip shows the same).
// Option blas1_pairs = k: k pairs per lane (0: the capped grid), for that A/B.
static inline int grid_vec(int64_t n2, const mgs_ctx *ctx) {
  const int iters = ctx->opt_blas1_pairs;
  int64_t g = (n2 + TB - 1) / TB;
  if (iters > 0) { g = (g + iters - 1) / iters; return (int)(g < 1 ? 1 : g); }
  const int64_t cap = (int64_t)ctx->n_cu * 8;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
// per-workgroup partial pairs of a one-shot reduction launch ([2][nb] doubles; grown outside any capture)
int mgs_ensure_dot_part(mgs_ctx *ctx, int64_t doubles) {
  if (ctx->dot_part_cap >= doubles) return MGS_OK;
  if (ctx->dot_part) { MGS_HIP(ctx, hipStreamSynchronize(ctx->stream)); mgs_hip_free(ctx->dot_part); ctx->dot_part = nullptr; ctx->dot_part_cap = 0; }
  MGS_TRY(mgs_dev_alloc(ctx, &ctx->dot_part, (size_t)doubles));
  ctx->dot_part_cap = doubles;
  return MGS_OK;
}
static inline int grid_cap(int64_t n, int ncu) {
  int64_t g = (n + TB - 1) / TB;
  int64_t cap = (int64_t)ncu * 8;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

int k_diag_inv(const mgs_csr *A, double *dinv, int *bad_count_host) {
  mgs_ctx *ctx = A->ctx;
  int *bad = nullptr;
  MGS_TRY(mgs_dev_alloc(ctx, &bad, 1));
  MGS_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
  if (A->rows)
    hipLaunchKernelGGL(diag_inv_kernel, dim3(mgs_grid(A->rows, TB)), dim3(TB), 0, ctx->stream, A->rows, A->rowptr, A->col, A->val, dinv, bad);
  MGS_HIP(ctx, hipMemcpyAsync(bad_count_host, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MGS_HIP(ctx, mgs_hip_free(bad));
  return MGS_OK;
}

int k_diag_pos(const mgs_csr *A, unsigned char *dpos) {
  mgs_ctx *ctx = A->ctx;
  if (A->rows) hipLaunchKernelGGL(diag_pos_kernel, dim3(mgs_grid(A->rows, TB)), dim3(TB), 0, ctx->stream, A->rows, A->rowptr, A->col, dpos);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_agg_pre(const mgs_csr *A, const double *valhat, const double *b, const double *hv, const mgs_xfer *T, double *r_out, double *rc_out) {
  mgs_ctx *ctx = A->ctx;
  const int64_t threads = 4 * (int64_t)T->n_coarse + (T->nnz < (int64_t)A->rows ? A->rows : 0);   // rows outside every aggregate only where there are any
  if (threads) hipLaunchKernelGGL(agg_pre_kernel, dim3(mgs_grid(threads, TB)), dim3(TB), 0, ctx->stream, A->rows, T->n_coarse, A->rowptr, A->col, valhat, b, hv,
                                  T->cptr, T->members, T->agg, r_out, rc_out);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_restrict_agg(mgs_ctx *ctx, int nc, const int *cptr, const int *members, const double *r, double *rc) {
  if (nc) hipLaunchKernelGGL(restrict_agg_kernel, dim3(mgs_grid(nc, TB)), dim3(TB), 0, ctx->stream, nc, cptr, members, r, rc);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_prolong_agg(mgs_ctx *ctx, int n, const int *agg, const double *ec, double *x, int add) {
  if (n) {
    if (add) hipLaunchKernelGGL(prolong_agg_kernel<1>, dim3(mgs_grid(n, TB)), dim3(TB), 0, ctx->stream, n, agg, ec, x);
    else hipLaunchKernelGGL(prolong_agg_kernel<0>, dim3(mgs_grid(n, TB)), dim3(TB), 0, ctx->stream, n, agg, ec, x);
  }
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_jacobi_zero(mgs_ctx *ctx, int n, double omega, const double *dinv, const double *b, double *x) {
  if (n) hipLaunchKernelGGL(jacobi_zero_kernel, dim3(mgs_grid(n, TB)), dim3(TB), 0, ctx->stream, n, omega, dinv, b, x);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_fill(mgs_ctx *ctx, double *d, int64_t n, double v) {
  if (n) hipLaunchKernelGGL(fill_kernel, dim3(grid_cap(n, ctx->n_cu)), dim3(TB), 0, ctx->stream, d, n, v);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_rand(mgs_ctx *ctx, double *d, int64_t n, uint64_t seed, int64_t off) {
  if (n) hipLaunchKernelGGL(rand_kernel, dim3(grid_cap(n, ctx->n_cu)), dim3(TB), 0, ctx->stream, d, n, seed, off);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_axpby(mgs_ctx *ctx, int64_t n, double a, const double *x, double b, double *y) {
  if (n && ctx->opt_blas1_vec && al16(x) && al16(y)) {
    const dim3 g(grid_vec((n + 1) / 2, ctx));
    const int nts = ctx->opt_nt_store > 0 && n >= ctx->opt_nt_store;
    if (b == 0.0) hipLaunchKernelGGL(axpby_vec_kernel<false>, g, dim3(TB), 0, ctx->stream, n, a, x, b, y, nts);
    else hipLaunchKernelGGL(axpby_vec_kernel<true>, g, dim3(TB), 0, ctx->stream, n, a, x, b, y, nts);
  } else
  if (n) hipLaunchKernelGGL(axpby_kernel, dim3(grid_cap(n, ctx->n_cu)), dim3(TB), 0, ctx->stream, n, a, x, b, y);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_axpbypcz(mgs_ctx *ctx, int64_t n, double a, const double *x, double b, const double *y, double c, double *z) {
  if (n && ctx->opt_blas1_vec && al16(x) && al16(y) && al16(z)) {
    const dim3 g(grid_vec((n + 1) / 2, ctx));
    const int nts = ctx->opt_nt_store > 0 && n >= ctx->opt_nt_store;
    if (c == 0.0) hipLaunchKernelGGL(axpbypcz_vec_kernel<false>, g, dim3(TB), 0, ctx->stream, n, a, x, b, y, c, z, nts);
    else hipLaunchKernelGGL(axpbypcz_vec_kernel<true>, g, dim3(TB), 0, ctx->stream, n, a, x, b, y, c, z, nts);
  } else
  if (n) hipLaunchKernelGGL(axpbypcz_kernel, dim3(grid_cap(n, ctx->n_cu)), dim3(TB), 0, ctx->stream, n, a, x, b, y, c, z);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_gather(mgs_ctx *ctx, const double *x, const int *idx, int64_t n, double *out) {
  if (n) hipLaunchKernelGGL(gather_kernel, dim3(mgs_grid(n, TB)), dim3(TB), 0, ctx->stream, x, idx, n, out);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}

int k_scale_vals(mgs_ctx *ctx, const mgs_csr *A, const double *wd, double *out) {
  if (A->rows) hipLaunchKernelGGL(scale_vals_kernel, dim3(mgs_grid(A->rows, TB)), dim3(TB), 0, ctx->stream, A->rows, A->rowptr, A->col, A->val, wd, out);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_map_cols(mgs_ctx *ctx, const mgs_csr *A, const int *cmap, int *out) {
  if (A->rows) hipLaunchKernelGGL(map_cols_kernel, dim3(mgs_grid(A->rows, TB)), dim3(TB), 0, ctx->stream, A->rows, A->rowptr, A->col, cmap, out);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
// last sharded level: own slice of the replicated tail's solution, and the level's halo slots from the same vector (it holds the
// neighbours' entries too) — what the copy of the slice plus a halo exchange would deliver, in one dispatch and without the wire
__global__ void tail_scatter_kernel(const double *__restrict__ xt, int my_off, int n_loc, const int *__restrict__ halo_global, int n_halo, double *__restrict__ x) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_loc) x[j] = xt[my_off + j];
  else if (j < n_loc + n_halo) x[j] = xt[halo_global[j - n_loc]];
}
int k_tail_scatter(mgs_ctx *ctx, const double *xt, int my_off, int n_loc, const int *halo_global, int n_halo, double *x) {
  if (n_loc + n_halo) hipLaunchKernelGGL(tail_scatter_kernel, dim3(mgs_grid(n_loc + n_halo, TB)), dim3(TB), 0, ctx->stream, xt, my_off, n_loc, halo_global, n_halo, x);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_concat_i32(mgs_ctx *ctx, const int *a, int na, const int *b, int nb, int *out) {
  if (na + nb) hipLaunchKernelGGL(concat_i32_kernel, dim3(mgs_grid(na + nb, TB)), dim3(TB), 0, ctx->stream, a, na, b, nb, out);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
// ---- results of a reduction to the host without a copy engine and without an interrupt
// The last stage of a reduction (one workgroup) stores the folded values into red_dev[DOT_BLOCKS ..] and, for the host, into the
// context's mapped, coherent host buffer, then a ticket (system-scope release); the host polls the ticket.  Against hipMemcpyAsync + hipStreamSynchronize this removes
// the blit/SDMA hop and the interrupt wake-up from every inner product of a Krylov loop: four per BiCGSTAB iteration — ≈ 60 of the
// 135 µs of an iteration on the bundled operators, and on a freshly started box, where the first process' wake-ups take milliseconds,
// 13 of 33 ms per iteration at 512³ (tools/studies_r1_r3/solve_repeat.py).  Option post_results = 0 restores copy + synchronize; so does any
// transport that sums over ranks in between (ncomm / all-reduce callback), and a wait that sees no ticket for 2 s.
static Post begin_post(mgs_ctx *ctx) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool posted = ctx->opt_post_results && ctx->red_host_dev && !ctx->ncomm && hipStreamIsCapturing(ctx->stream, &cs) == hipSuccess &&
                      cs == hipStreamCaptureStatusNone;
  if (!posted) return Post{nullptr, nullptr, 0ull};
  return Post{ctx->red_host_dev, reinterpret_cast<unsigned long long *>(ctx->red_host_dev + MGS_RED_VALS), ++ctx->red_ticket};
}
// the reduction's last kernel (launched with `pa`) has been enqueued: wait for its results
static int fetch_results(mgs_ctx *ctx, int cnt, double *out_host, const Post &pa) {
  if (pa.hv) {
    volatile unsigned long long *tk = reinterpret_cast<volatile unsigned long long *>(ctx->red_host + MGS_RED_VALS);
    const unsigned long long want = pa.ticket;
    MGS_HIP(ctx, hipGetLastError());
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (;;) {
      if (__atomic_load_n(const_cast<unsigned long long *>(tk), __ATOMIC_ACQUIRE) == want) { seen = true; break; }
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (waited > 2.0) break;                                   // never seen: fall back to the stream's own completion below
      if (waited < 300e-6) { for (int q = 0; q < 8; ++q) __builtin_ia32_pause(); }
      else { struct timespec ts = {0, 25000}; nanosleep(&ts, nullptr); }     // long kernels ahead of the ticket: stop burning the core
    }
    if (!seen) {
      MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (__atomic_load_n(const_cast<unsigned long long *>(tk), __ATOMIC_ACQUIRE) != want)
        return mgs_fail(ctx, MGS_ERR_HIP, "reduction results never reached the host buffer");
    }
    for (int q = 0; q < cnt; ++q) out_host[q] = ctx->red_host[q];
  } else {
    MGS_HIP(ctx, hipMemcpyAsync(ctx->red_host, ctx->red_dev + DOT_BLOCKS, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream));
    MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < cnt; ++q) out_host[q] = ctx->red_host[q];
  }
  if (ctx->allreduce && !ctx->ncomm) {
    int rc = ctx->allreduce(ctx->allreduce_user, out_host, cnt);
    if (rc) return mgs_fail(ctx, MGS_ERR_STATE, "allreduce callback failed (%d)", rc);
  }
  return MGS_OK;
}

int k_dot(mgs_ctx *ctx, int64_t n, const double *x, const double *y, double *out_host) {
  if (ctx->opt_blas1_vec && n >= 2 && al16(x) && al16(y)) {       // one-shot workgroups, staged fold (as the fused update kernels)
    const int nbv = grid_vec(n / 2, ctx);
    double *part = ctx->red_dev;
    if (nbv > DOT_BLOCKS / 2) { MGS_TRY(mgs_ensure_dot_part(ctx, 2 * (int64_t)nbv)); part = ctx->dot_part; }
    hipLaunchKernelGGL(dot_block_vec_kernel<1>, dim3(nbv), dim3(TB), 0, ctx->stream, n, x, y, x, y, part);
    MGS_HIP(ctx, hipGetLastError());
    double two[2] = {0.0, 0.0};
    MGS_TRY(k_dot2_finish(ctx, nbv, part, two));
    *out_host = two[0];
    return MGS_OK;
  }
  int nb = (int)((n + TB - 1) / TB);
  if (nb > DOT_BLOCKS) nb = DOT_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(TB), 0, ctx->stream, n, x, y, ctx->red_dev);
  const Post pa = begin_post(ctx);
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(TB), 0, ctx->stream, nb, ctx->red_dev, ctx->red_dev + DOT_BLOCKS, pa);
  if (ctx->ncomm) MGS_TRY(mgs_comm_allreduce_sum(ctx->ncomm, ctx->red_dev + DOT_BLOCKS, 1));   // sum over the row shards, on this stream
  return fetch_results(ctx, 1, out_host, pa);
}

// middle stage for long partial arrays (one pair per row block of a 512³ operator = 2 × 524 288): workgroup g sums the g-th
// contiguous chunk of each array in a fixed order
__global__ __launch_bounds__(TB) void dot2_mid_kernel(int nb, int chunk, const double *__restrict__ part, double *__restrict__ out /*[2][gridDim.x]*/) {
  __shared__ double sh[2][TB / 64];
  const int lo = blockIdx.x * chunk, hi = min(lo + chunk, nb);
  double s0 = 0.0, s1 = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += TB) { s0 += part[i]; s1 += part[nb + i]; }
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int q = 0; q < TB / 64; ++q) { t0 += sh[0][q]; t1 += sh[1][q]; }
    out[blockIdx.x] = t0; out[gridDim.x + blockIdx.x] = t1;
  }
}
int k_dot2_finish(mgs_ctx *ctx, int nb, const double *part, double *out_host2) {
  if (nb > 4096) {          // three stages: row-block partials → 256 chunk sums (in red_dev) → the pair
    const int groups = 256, chunk = (nb + groups - 1) / groups;
    hipLaunchKernelGGL(dot2_mid_kernel, dim3(groups), dim3(TB), 0, ctx->stream, nb, chunk, part, ctx->red_dev);
    part = ctx->red_dev; nb = groups;
  }
  const Post pa = begin_post(ctx);
  hipLaunchKernelGGL(dot2_final_kernel, dim3(1), dim3(TB), 0, ctx->stream, nb, part, ctx->red_dev + DOT_BLOCKS, pa);
  MGS_HIP(ctx, hipGetLastError());
  if (ctx->ncomm) MGS_TRY(mgs_comm_allreduce_sum(ctx->ncomm, ctx->red_dev + DOT_BLOCKS, 2));
  return fetch_results(ctx, 2, out_host2, pa);
}
int k_dot2(mgs_ctx *ctx, int64_t n, const double *x, const double *y, const double *z, const double *w, double *out_host2) {
  if (ctx->opt_blas1_vec && n >= 2 && al16(x) && al16(y) && al16(z) && al16(w)) {
    const int nbv = grid_vec(n / 2, ctx);
    double *part = ctx->red_dev;
    if (nbv > DOT_BLOCKS / 2) { MGS_TRY(mgs_ensure_dot_part(ctx, 2 * (int64_t)nbv)); part = ctx->dot_part; }
    hipLaunchKernelGGL(dot_block_vec_kernel<2>, dim3(nbv), dim3(TB), 0, ctx->stream, n, x, y, z, w, part);
    MGS_HIP(ctx, hipGetLastError());
    return k_dot2_finish(ctx, nbv, part, out_host2);
  }
  int nb = (int)((n + TB - 1) / TB);
  if (nb > DOT_BLOCKS / 2) nb = DOT_BLOCKS / 2;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(dot2_partial_kernel, dim3(nb), dim3(TB), 0, ctx->stream, n, x, y, z, w, ctx->red_dev);
  const Post pa = begin_post(ctx);
  hipLaunchKernelGGL(dot2_final_kernel, dim3(1), dim3(TB), 0, ctx->stream, nb, ctx->red_dev, ctx->red_dev + DOT_BLOCKS, pa);
  if (ctx->ncomm) MGS_TRY(mgs_comm_allreduce_sum(ctx->ncomm, ctx->red_dev + DOT_BLOCKS, 2));
  return fetch_results(ctx, 2, out_host2, pa);
}
int k_update_dot2(mgs_ctx *ctx, int64_t n, double a, const double *x, double b, const double *y, double *z, const double *w, double *out_host2) {
  int nb = (int)((n + TB - 1) / TB);
  if (nb > DOT_BLOCKS / 2) nb = DOT_BLOCKS / 2;
  if (nb < 1) nb = 1;
  if (ctx->opt_blas1_vec && al16(x) && al16(y) && al16(z) && al16(w)) {
    // one-shot workgroups (see grid_vec): one partial pair per workgroup, folded in fixed order by k_dot2_finish
    nb = grid_vec((n + 1) / 2, ctx);
    const int nts = ctx->opt_nt_store > 0 && n >= ctx->opt_nt_store;
    double *part = ctx->red_dev;
    if (nb > DOT_BLOCKS / 2) { MGS_TRY(mgs_ensure_dot_part(ctx, 2 * (int64_t)nb)); part = ctx->dot_part; }
    if (w) hipLaunchKernelGGL(update_dot2_partial_vec_kernel<true>, dim3(nb), dim3(TB), 0, ctx->stream, n, a, x, b, y, z, w, part, nts);
    else hipLaunchKernelGGL(update_dot2_partial_vec_kernel<false>, dim3(nb), dim3(TB), 0, ctx->stream, n, a, x, b, y, z, w, part, nts);
    MGS_HIP(ctx, hipGetLastError());
    return k_dot2_finish(ctx, nb, part, out_host2);
  }
  hipLaunchKernelGGL(update_dot2_partial_kernel, dim3(nb), dim3(TB), 0, ctx->stream, n, a, x, b, y, z, w, ctx->red_dev);
  const Post pa = begin_post(ctx);
  hipLaunchKernelGGL(dot2_final_kernel, dim3(1), dim3(TB), 0, ctx->stream, nb, ctx->red_dev, ctx->red_dev + DOT_BLOCKS, pa);
  if (ctx->ncomm) MGS_TRY(mgs_comm_allreduce_sum(ctx->ncomm, ctx->red_dev + DOT_BLOCKS, 2));
  return fetch_results(ctx, 2, out_host2, pa);
}
int k_dot_dev(mgs_ctx *ctx, int64_t n, const double *x, const double *y, double *out_dev) {
  int nb = (int)((n + TB - 1) / TB);
  if (nb > DOT_BLOCKS) nb = DOT_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(TB), 0, ctx->stream, n, x, y, ctx->red_dev);
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(TB), 0, ctx->stream, nb, ctx->red_dev, out_dev, Post{nullptr, nullptr, 0ull});
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
// workgroups of a multi-vector reduction pass: one-shot at small sizes, a few elements per lane at large ones (partials stay short)
static int mdot_grid(int64_t n) {
  int64_t nb = (n / 2 + (int64_t)TB * 4 - 1) / ((int64_t)TB * 4);
  return (int)std::min<int64_t>(std::max<int64_t>(nb, 1), 2048);
}
// s_k = a·b_k, k < K ≤ MDOT_MAX: to device scalars (out_dev) and/or, summed over the ranks, to the host (out_host)
int k_mdot(mgs_ctx *ctx, int64_t n, int K, const double *a, const double *const *b, double *out_dev, double *out_host) {
  MGS_CHECK(ctx, K >= 1 && K <= MDOT_MAX && K <= MGS_RED_VALS, MGS_ERR_INVALID, "k_mdot: %d inner products (at most %d)", K, MDOT_MAX);
  MDotVecs B;
  bool aligned = al16(a);
  for (int k = 0; k < MDOT_MAX; ++k) { B.b[k] = k < K ? b[k] : a; if (k < K && !al16(b[k])) aligned = false; }
  const int nb = mdot_grid(n);
  double *part = ctx->red_dev;        // (the K-cycle's pairs always fit the context's own scratch: nothing is allocated inside a captured cycle)
  if ((int64_t)K * nb > DOT_BLOCKS) { MGS_TRY(mgs_ensure_dot_part(ctx, (int64_t)K * nb)); part = ctx->dot_part; }
  double *res = out_dev ? out_dev : ctx->red_dev + DOT_BLOCKS;
  if (!aligned) {      // operands that do not start on 16 bytes (views into the middle of a vector): K two-vector reductions
    for (int k = 0; k < K; ++k) MGS_TRY(k_dot_dev(ctx, n, a, b[k], res + k));
    if (!out_host) return MGS_OK;
    if (ctx->ncomm) {
      if (res != ctx->red_dev + DOT_BLOCKS) MGS_HIP(ctx, hipMemcpyAsync(ctx->red_dev + DOT_BLOCKS, res, sizeof(double) * (size_t)K, hipMemcpyDeviceToDevice, ctx->stream));
      MGS_TRY(mgs_comm_allreduce_sum(ctx->ncomm, ctx->red_dev + DOT_BLOCKS, (size_t)K));
      res = ctx->red_dev + DOT_BLOCKS;
    }
    MGS_HIP(ctx, hipMemcpyAsync(ctx->red_host, res, sizeof(double) * (size_t)K, hipMemcpyDeviceToHost, ctx->stream));
    MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < K; ++k) out_host[k] = ctx->red_host[k];
    if (ctx->allreduce && !ctx->ncomm && ctx->allreduce(ctx->allreduce_user, out_host, K)) return mgs_fail(ctx, MGS_ERR_STATE, "allreduce callback failed");
    return MGS_OK;
  }
  hipLaunchKernelGGL(mdot_partial_kernel, dim3(nb), dim3(TB), 0, ctx->stream, n, K, a, B, part);
  const Post pa = out_host ? begin_post(ctx) : Post{nullptr, nullptr, 0ull};
  hipLaunchKernelGGL(mdot_final_kernel, dim3(1), dim3(TB), 0, ctx->stream, nb, K, part, res, pa);
  MGS_HIP(ctx, hipGetLastError());
  if (!out_host) return MGS_OK;
  if (res != ctx->red_dev + DOT_BLOCKS) MGS_HIP(ctx, hipMemcpyAsync(ctx->red_dev + DOT_BLOCKS, res, sizeof(double) * (size_t)K, hipMemcpyDeviceToDevice, ctx->stream));   // where fetch_results looks
  if (ctx->ncomm) MGS_TRY(mgs_comm_allreduce_sum(ctx->ncomm, ctx->red_dev + DOT_BLOCKS, (size_t)K));
  return fetch_results(ctx, K, out_host, pa);
}
// z = x − Σ_{j<K} coef_j·w_j with (z·u2 — or z·z when u2 is NULL —, z·u) to the host in the same pass (out_host2 = NULL: no sums, no round trip)
int k_maxpy_dot2(mgs_ctx *ctx, int64_t n, int K, const double *x, const double *const *w, const double *coef, double *z, const double *u, const double *u2, double *out_host2) {
  MGS_CHECK(ctx, K >= 0 && K <= MDOT_MAX, MGS_ERR_INVALID, "k_maxpy_dot2: %d terms (at most %d)", K, MDOT_MAX);
  MAxpyArgs W;
  for (int k = 0; k < MDOT_MAX; ++k) { W.w[k] = k < K ? w[k] : x; W.coef[k] = k < K ? coef[k] : 0.0; }
  bool vec = n >= 2 && al16(x) && al16(z) && (!u || al16(u)) && (!u2 || al16(u2));
  for (int k = 0; k < K; ++k) vec = vec && al16(w[k]);
  int64_t nbl = (n + (int64_t)TB * 4 - 1) / ((int64_t)TB * 4);       // two 16-byte (or four 8-byte) elements per lane
  const int nb = (int)std::min<int64_t>(std::max<int64_t>(nbl, 1), 1 << 20);
  double *part = nullptr;
  if (out_host2) { part = ctx->red_dev; if (nb > DOT_BLOCKS / 2) { MGS_TRY(mgs_ensure_dot_part(ctx, 2 * (int64_t)nb)); part = ctx->dot_part; } }
  if (vec) hipLaunchKernelGGL(maxpy_dot2_kernel<true>, dim3(nb), dim3(TB), 0, ctx->stream, n, K, x, W, z, u, u2, part);
  else hipLaunchKernelGGL(maxpy_dot2_kernel<false>, dim3(nb), dim3(TB), 0, ctx->stream, n, K, x, W, z, u, u2, part);
  MGS_HIP(ctx, hipGetLastError());
  return out_host2 ? k_dot2_finish(ctx, nb, part, out_host2) : MGS_OK;
}
int k_kc_update_r(mgs_ctx *ctx, int n, const double *scal, const double *r, const double *v1, double *rp) {
  if (n) hipLaunchKernelGGL(kc_update_r_kernel, dim3(mgs_grid(n, TB)), dim3(TB), 0, ctx->stream, n, scal, r, v1, rp);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
// scal[3] = ρ2, scal[4] = α2 of the explicitly orthogonalised second direction (see kc_orth_dots_kernel)
int k_kc_orth_dots(mgs_ctx *ctx, int n, bool energy, double *scal, const double *c1, const double *c2, const double *v1, const double *v2, const double *rp) {
  const int nb = std::max(1, std::min(mgs_grid(n, TB * 4), DOT_BLOCKS / 2));
  if (energy) hipLaunchKernelGGL(kc_orth_dots_kernel<true>, dim3(nb), dim3(TB), 0, ctx->stream, n, scal, c1, c2, v1, v2, rp, ctx->red_dev);
  else hipLaunchKernelGGL(kc_orth_dots_kernel<false>, dim3(nb), dim3(TB), 0, ctx->stream, n, scal, c1, c2, v1, v2, rp, ctx->red_dev);
  hipLaunchKernelGGL(dot2_final_kernel, dim3(1), dim3(TB), 0, ctx->stream, nb, ctx->red_dev, scal + 3, Post{nullptr, nullptr, 0ull});
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
int k_kc_combine(mgs_ctx *ctx, int n, const double *scal, const double *c1, const double *c2, double *x) {
  if (n) hipLaunchKernelGGL(kc_combine_kernel, dim3(mgs_grid(n, TB)), dim3(TB), 0, ctx->stream, n, scal, c1, c2, x);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}

int k_dense_gemv(mgs_ctx *ctx, int n, const double *M, const double *b, double *x) {
  if (n) hipLaunchKernelGGL(dense_gemv_kernel, dim3(mgs_grid(n, TB / 64)), dim3(TB), 0, ctx->stream, n, M, b, x);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}

int k_dense_inverse(mgs_ctx *ctx, const mgs_csr *A, double **inv_out) {
  const int n = A->rows;
  MGS_CHECK(ctx, A->cols == n, MGS_ERR_INVALID, "coarsest operator is not square (%d x %d)", A->rows, A->cols);
  double *W = nullptr, *colk = nullptr, *inv = nullptr; int *piv = nullptr;
  MGS_TRY(mgs_dev_alloc(ctx, &W, (size_t)n * 2 * n));
  MGS_TRY(mgs_dev_alloc(ctx, &colk, (size_t)n + 1));
  MGS_TRY(mgs_dev_alloc(ctx, &inv, (size_t)n * n));
  MGS_TRY(mgs_dev_alloc(ctx, &piv, 2));
  hipStream_t s = ctx->stream;
  MGS_HIP(ctx, hipMemsetAsync(W, 0, sizeof(double) * (size_t)n * 2 * n, s));
  MGS_HIP(ctx, hipMemsetAsync(piv, 0, 2 * sizeof(int), s));
  if (n) hipLaunchKernelGGL(dense_scatter_kernel, dim3(mgs_grid(n, TB)), dim3(TB), 0, s, n, A->rowptr, A->col, A->val, W);
  for (int k = 0; k < n; ++k) {
    hipLaunchKernelGGL(gj_pivot_kernel, dim3(1), dim3(TB), 0, s, n, k, W, piv, colk + n);
    hipLaunchKernelGGL(gj_swap_scale_kernel, dim3(mgs_grid(2 * n, TB)), dim3(TB), 0, s, n, k, W, piv, colk + n);
    hipLaunchKernelGGL(gj_colsave_kernel, dim3(mgs_grid(n, TB)), dim3(TB), 0, s, n, k, W, colk);
    hipLaunchKernelGGL(gj_eliminate_kernel, dim3(mgs_grid(2 * n, TB), n), dim3(TB), 0, s, n, k, W, colk);
  }
  if (n) hipLaunchKernelGGL(gj_extract_kernel, dim3(mgs_grid(n, TB), n), dim3(TB), 0, s, n, W, inv);
  int h[2] = {0, 0};
  MGS_HIP(ctx, hipMemcpyAsync(h, piv, sizeof h, hipMemcpyDeviceToHost, s));
  MGS_HIP(ctx, hipStreamSynchronize(s));
  MGS_HIP(ctx, mgs_hip_free(W)); MGS_HIP(ctx, mgs_hip_free(colk)); MGS_HIP(ctx, mgs_hip_free(piv));
  if (h[1]) { mgs_hip_free(inv); return mgs_fail(ctx, MGS_ERR_NUMERIC, "coarsest operator (%d rows) is singular", n); }
  *inv_out = inv;
  return MGS_OK;
}

int k_poisson3d(mgs_ctx *ctx, int N, int plane_lo, int plane_hi, int local_cols, mgs_csr **out) {
  MGS_CHECK(ctx, N >= 2 && plane_lo >= 0 && plane_hi <= N && plane_lo < plane_hi, MGS_ERR_INVALID, "poisson3d: bad N/planes");
  const int64_t N2 = (int64_t)N * N, n_loc = (int64_t)(plane_hi - plane_lo) * N2;
  // nnz of the shard = rowptr(e1) − rowptr(e0); computed on host with the same closed form
  auto rp = [&](int64_t e) -> int64_t {
    if (e >= N2 * N) return 7 * N2 * N - 6 * N2;
    int64_t i = e / N2;  // e is a plane boundary here
    // rows before plane i: 7 per row minus missing neighbours
    int64_t rows = i * N2;
    int64_t miss = (i > 0 ? N2 : 0) + 0 /*i==N-1 rows before: none unless i==N*/ + 2 * i * N /*j==0,N-1*/ + 2 * i * N /*k==0,N-1*/;
    return 7 * rows - miss;
  };
  const int64_t nnz = rp((int64_t)plane_hi * N2) - rp((int64_t)plane_lo * N2);
  MGS_CHECK(ctx, nnz < 2147483647LL && n_loc < 2147483647LL, MGS_ERR_INVALID, "poisson3d shard too large for int32 indices");
  const int has_lo = plane_lo > 0, has_hi = plane_hi < N;
  int64_t ncols = local_cols ? n_loc + (has_lo + has_hi) * N2 : N2 * N;
  MGS_CHECK(ctx, ncols < 2147483647LL, MGS_ERR_INVALID, "poisson3d: too many columns for int32");
  mgs_csr *A = nullptr;
  MGS_TRY(mgs_csr_alloc(ctx, (int)n_loc, (int)ncols, nnz, &A));
  hipLaunchKernelGGL(poisson3d_kernel, dim3(mgs_grid(n_loc + 1, TB)), dim3(TB), 0, ctx->stream, N, plane_lo, n_loc, local_cols, has_lo, has_hi, A->rowptr, A->col, A->val);
  MGS_HIP(ctx, hipGetLastError());
  MGS_TRY(mgs_plan_csr(A));
  *out = A;
  return MGS_OK;
}

int k_poisson2d(mgs_ctx *ctx, int n, mgs_csr **out) {
  MGS_CHECK(ctx, n >= 2 && (int64_t)n * n < 400000000LL, MGS_ERR_INVALID, "poisson2d: bad n");
  int N = n * n; int64_t nnz = 5LL * N - 4LL * n;
  mgs_csr *A = nullptr;
  MGS_TRY(mgs_csr_alloc(ctx, N, N, nnz, &A));
  hipLaunchKernelGGL(poisson2d_kernel, dim3(mgs_grid(N + 1, TB)), dim3(TB), 0, ctx->stream, n, A->rowptr, A->col, A->val);
  MGS_HIP(ctx, hipGetLastError());
  MGS_TRY(mgs_plan_csr(A));
  *out = A;
  return MGS_OK;
}
