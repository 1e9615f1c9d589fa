// kernels_spmv.hip — the fine-level hot path: CSR SpMV-shaped kernels for gfx950.
//
//   y = A x                       (`A * v`,           reference src/common/bicg.cpp:57,82,107,117)
//   r = b − A x                   (                   reference src/common/bicg.cpp:82)
//   x' = x + ωD⁻¹(b − A x)        (damped Jacobi,     reference src/CPU_Matlab/solve.m:17)
//
// Design: AMG operators of PDE problems have 3..30 entries per row, so a
// wavefront-per-row kernel would idle ≥57 of 64 lanes.  One 256-thread workgroup owns 256
// consecutive rows ("row block") and stages the block's contiguous slice of val/col_idx in
// LDS with fully coalesced 16-/8-byte loads (every lane busy; plain loads — the non-temporal form
// measured 3 % slower here and stays an option).  Then lane t walks
// row t: the 64 lanes of a wave gather x[col] for 64 consecutive rows at once — contiguous
// runs of x for stencil-like operators — and add the products sequentially in ascending column
// order, the same order as Eigen's scalar loop
// (lib/Eigen/src/SparseCore/SparseDenseProduct.h:64-70), so the result is bit-identical to the
// CPU path; the epilogue (residual / damped Jacobi) is fused.  Row blocks whose slice does not
// fit the LDS budget (long rows) fall back, per block, to a sub-wavefront-per-row reduction with
// shuffles.  The workgroup→row-block map is XCD-contiguous (each of the 8 XCDs sweeps one eighth
// of the rows so re-read x planes stay in that XCD's 4 MiB L2), optionally strip-major.
// The earlier "products in LDS" variants are kept for A/B runs (tools/studies_r1_r3/ab_spmv.py).
// Default path for operators whose rows repeat their shape (stencils, their Galerkin levels): the same row-block
// kernel with a PATTERN-CODED column index (csr_rowblock_coded_kernel, mgs_csr_optimize) — 8 B per entry streamed
// instead of 12, same products in the same order; irregular row blocks keep their index slice.
// No MFMA: arithmetic intensity is 0.13 flop/B; the bound is HBM (≈8 TB/s peak).
#include "mgs_internal.hpp"

#include <algorithm>

namespace {

constexpr int RB = 256;         // rows per row block == threads per workgroup
constexpr int LDS_CAP_MAX = 5120;  // products (doubles) staged per block: 40 KiB → 4 blocks/CU

template <bool NT, class T>
__device__ __forceinline__ T ld_stream(const T *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// Streaming store of one output element per lane.  The builtin's hint does not survive here: both arms of `if (nt) nontemporal
// store else plain store` write the same value to the same address, the optimiser merges them into ONE plain store and the
// kernel never contained an `nt` store (seen in the ISA; the round-1/2 "nt_store is neutral" A/B compared identical code).  The
// instruction is spelled out instead.  Nothing in these kernels reads what it stored, so no wait count is owed.
__device__ __forceinline__ void st_stream(double *p, double v, bool nt) {
  if (nt) asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
  else *p = v;
}

// Value slice of a row block straight from global memory into LDS (no VGPR in between, so nothing for the register allocator to
// keep alive or spill): piece p = 64 pairs = 1 KiB, wave w takes pieces w, w + 4, w + 8, w + 12; lane l of a piece moves pair 64p + l.
// Covers up to 1024 pairs (2048 doubles); the caller checks that.  The caller waits for its own transfers (s_waitcnt vmcnt(0), explicit)
// before the workgroup barrier that publishes the slice.
__device__ __forceinline__ void stage_pairs_dma(const double *__restrict__ src, double *lds, int npairs, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = wave + 4 * k;
    if (p * 64 + lane < npairs)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 2 * (p * 64 + lane)),
                                       (__attribute__((address_space(3))) void *)(lds + 128 * p), 16, 0, 0);
  }
}

template <int OP>
__device__ __forceinline__ void epilogue(int row, double s, const double *__restrict__ x,
                                         const double *__restrict__ b, const double *__restrict__ dinv,
                                         double omega, double *__restrict__ out) {
  if (OP == MGS_OP_SPMV) out[row] = s;
  else if (OP == MGS_OP_RESIDUAL) out[row] = b[row] - s;
  else out[row] = x[row] + (omega * dinv[row]) * (b[row] - s);
}

// vector types for 16-byte wide streaming loads
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef int int2_t __attribute__((ext_vector_type(2)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// Workgroup → row-block map.  (1) XCD-contiguous: workgroups are dealt round-robin over the 8
// XCDs, so XCD x = bid & 7 sweeps the x-th eighth of the row blocks.  (2) Optional strip-major
// sweep inside an XCD for operators with a far band at ±D row blocks (the e±N² planes of the
// 7-point stencil): instead of plane after plane, a strip of S row blocks is followed through all
// planes of the XCD's range, so x[e+N²] fetched for plane p is still in L2 when planes p+1 and
// p+2 need it as x[e] and x[e−N²].  Pure permutation: every row block is visited exactly once.
struct BlockMap { int nblocks, chunk, remap, D, S, P, base, gap_at = 0x7fffffff, gap_len = 0;
                  unsigned mps = 0, mS = 0; };   // ⌊2³²/(P·S)⌋+1 and ⌊2³²/S⌋+1: quotients by multiply-high — set by strip_map with D, S, P
// base: first row block of the launched range;
// row blocks >= gap_at are shifted by gap_len (one launch over the leading + trailing boundary blocks of a row shard)
__device__ __forceinline__ int block_of(const BlockMap &m, int vb) { const int b = m.base + vb; return b >= m.gap_at ? b + m.gap_len : b; }
__device__ __forceinline__ int map_block_xi(const BlockMap &m, int xcd, int idx);
__device__ __forceinline__ int map_block(const BlockMap &m, int bid) {
  if (!m.remap) return bid < m.nblocks ? bid : -1;
  return map_block_xi(m, bid & 7, bid >> 3);
}
__device__ __forceinline__ int map_block_xi(const BlockMap &m, int xcd, int idx) {
  int lb = idx;
  if (m.D > 0) {
    // Quotients by multiply-high (m.mps = ⌊2³²/(P·S)⌋+1, m.mS = ⌊2³²/S⌋+1; exact for idx·divisor < 2³², which the host checks — it
    // falls back to the plain XCD-contiguous map otherwise).  The two integer divisions this replaces compiled to ~65 dependent scalar
    // instructions at the head of EVERY wave, in front of its first load (and stayed there, if-converted, behind a run-time switch).
    const int ps = m.P * m.S;
    const int s = (int)__umulhi((unsigned)idx, m.mps), rem = idx - s * ps;
    const int p = (int)__umulhi((unsigned)rem, m.mS), t = rem - p * m.S;
    const int off = s * m.S + t;
    if (off >= m.D) return -1;
    lb = p * m.D + off;
  }
  if (lb >= m.chunk) return -1;
  const int vb = xcd * m.chunk + lb;
  return vb < m.nblocks ? vb : -1;
}

template <int OP, bool NT, int LANES, int CHUNK>
__global__ __launch_bounds__(RB) void csr_rowblock_kernel(
    int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, const double *__restrict__ b, const double *__restrict__ dinv, double omega,
    double *__restrict__ out, int cap, BlockMap bm) {
  extern __shared__ double prod[];
  const int vb = map_block(bm, blockIdx.x);
  if (vb < 0) return;
  const int r0 = (bm.base + vb) * RB;
  const int r1 = min(r0 + RB, n);
  const int tid = threadIdx.x;
  const int lo = rowptr[r0];
  const int hi = rowptr[r1];
  if (hi - lo <= cap) {
    const int row = r0 + tid;
    int my_lo = 0, my_hi = 0;
    if (row < r1) { my_lo = rowptr[row] - lo; my_hi = rowptr[row + 1] - lo; }
    // phase 1: coalesced stream of the block's matrix slice, products to LDS
    if (CHUNK == 1) {
      const int *__restrict__ cp = col + lo;
      const double *__restrict__ vp = val + lo;
      const int cnt = hi - lo;
#pragma unroll 8
      for (int k = tid; k < cnt; k += RB) {
        const int c = ld_stream<NT>(cp + k);
        const double v = ld_stream<NT>(vp + k);
        prod[k] = v * x[c];
      }
    } else if (CHUNK == 2) {
      // 2 entries per lane per step: val as one 16-byte load, col as one 8-byte load.  The slice is
      // widened to even element offsets (allocation is padded; out-of-slice lanes are masked).
      const int start = lo & ~1;
      const int nch = (hi - start + 1) >> 1;
#pragma unroll 4
      for (int c = tid; c < nch; c += RB) {
        const int k = start + 2 * c;
        const int2_t cc = ld_stream<NT>(reinterpret_cast<const int2_t *>(col + k));
        const double2_t vv = ld_stream<NT>(reinterpret_cast<const double2_t *>(val + k));
        if (k >= lo) prod[k - lo] = vv.x * x[cc.x];
        if (k + 1 < hi) prod[k + 1 - lo] = vv.y * x[cc.y];
      }
    } else {
      // 4 entries per lane per step: col as one 16-byte load, val as two 16-byte loads
      const int start = lo & ~3;
      const int nch = (hi - start + 3) >> 2;
#pragma unroll 2
      for (int c = tid; c < nch; c += RB) {
        const int k = start + 4 * c;
        const int4_t cc = ld_stream<NT>(reinterpret_cast<const int4_t *>(col + k));
        const double2_t v01 = ld_stream<NT>(reinterpret_cast<const double2_t *>(val + k));
        const double2_t v23 = ld_stream<NT>(reinterpret_cast<const double2_t *>(val + k + 2));
        if (k >= lo && k < hi) prod[k - lo] = v01.x * x[cc.x];
        if (k + 1 >= lo && k + 1 < hi) prod[k + 1 - lo] = v01.y * x[cc.y];
        if (k + 2 >= lo && k + 2 < hi) prod[k + 2 - lo] = v23.x * x[cc.z];
        if (k + 3 >= lo && k + 3 < hi) prod[k + 3 - lo] = v23.y * x[cc.w];
      }
    }
    __syncthreads();
    // phase 2: sequential per-row sum in ascending column order
    if (row < r1) {
      double s = 0.0;
      for (int k = my_lo; k < my_hi; ++k) s += prod[k];
      epilogue<OP>(row, s, x, b, dinv, omega, out);
    }
  } else {
    // long-row block: LANES lanes cooperate on one row, shuffle reduction
    const int sub = tid / LANES, lane = tid % LANES;
    for (int row = r0 + sub; row < r1; row += RB / LANES) {
      double s = 0.0;
      const int e = rowptr[row + 1];
      for (int k = rowptr[row] + lane; k < e; k += LANES) s += val[k] * x[col[k]];
#pragma unroll
      for (int off = LANES / 2; off > 0; off >>= 1) s += __shfl_down(s, off, LANES);
      if (lane == 0) epilogue<OP>(row, s, x, b, dinv, omega, out);
    }
  }
}

// Variant B ("slice in LDS, row-parallel gather").  Phase 1 parks the block's raw val/col slice
// in LDS (coalesced 16-byte / 8-byte loads, no dependent gather in the streaming
// phase).  Phase 2: lane t walks row t; on step q the 64 lanes of a wave gather x[col] for 64
// CONSECUTIVE rows, which for stencil-like operators is one contiguous run (4–5 cache lines per
// wave instruction instead of ~12 when 64 consecutive entries are gathered) — 2.5× less L2→L1
// traffic and TA work for the x gather.  Sum order is still the ascending column order of the
// row, so the result stays bit-identical to the scalar CPU loop.
template <int OP, bool NT, int LANES, int U>
__global__ __launch_bounds__(RB) void csr_rowblock_slice_kernel(
    int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, const double *__restrict__ b, const double *__restrict__ dinv, double omega,
    double *__restrict__ out, int cap, BlockMap bm, int seq_overflow, const int *__restrict__ blkptr, int nts) {
  extern __shared__ double lds_raw[];
  const int vb = map_block(bm, blockIdx.x);
  if (vb < 0) return;
  const int blk = block_of(bm, vb);
  const int r0 = blk * RB;
  const int r1 = min(r0 + RB, n);
  const int tid = threadIdx.x;
  // block bounds from the compact per-block copy of rowptr (consecutive workgroups share cache lines;
  // rowptr[r0] itself is 1 KiB apart from block to block)
  const int lo = blkptr ? blkptr[blk] : rowptr[r0];
  const int hi = blkptr ? blkptr[blk + 1] : rowptr[r1];
  if (hi - lo <= cap) {
    double *__restrict__ vals = lds_raw;                                   // cap + 2 doubles
    int *__restrict__ cols = reinterpret_cast<int *>(lds_raw + cap + 2);   // cap + 2 ints
    const int row = r0 + tid;
    const int start = lo & ~1;                 // even element offset → 16-byte aligned val pairs
    int my_a = 0, my_e = 0;
    double bi = 0.0, di = 0.0, xi = 0.0;
    if (row < r1) {
      my_a = rowptr[row] - start; my_e = rowptr[row + 1] - start;
      if (OP != MGS_OP_SPMV) bi = b[row];
      if (OP == MGS_OP_JACOBI) { di = dinv[row]; xi = x[row]; }
    }
    const int nch = (hi - start + 1) >> 1;
    // (a loop-free form of this staging — four predicated int2 + double2 loads per lane — measured 1.5 % SLOWER here, 2.553 against
    // 2.515 ms: 64 instead of 46 VGPRs; the coded kernels, which stage values only, gain 5–7 % from it)
#pragma unroll 4
    for (int c = tid; c < nch; c += RB) {
      const int k = start + 2 * c;
      const int2_t cc = ld_stream<NT>(reinterpret_cast<const int2_t *>(col + k));
      const double2_t vv = ld_stream<NT>(reinterpret_cast<const double2_t *>(val + k));
      *reinterpret_cast<double2_t *>(vals + 2 * c) = vv;
      *reinterpret_cast<int2_t *>(cols + 2 * c) = cc;
    }
    __syncthreads();
    if (row < r1) {
      // U entries per step, branch-free: the column reads past the row's end stay inside the staged slice
      // and are clamped to the last staged entry, so every gathered index is a real column; their products are replaced by +0.0, which
      // leaves the running sum bit-identical.  All U gathers are in flight before the first add.
      double s = 0.0;
      const int lim = hi - start - 1;                  // last staged index
      for (int k = my_a; k < my_e; k += U) {
        int cq[U]; double xv[U], vq[U];
#pragma unroll
        for (int q = 0; q < U; ++q) cq[q] = cols[min(k + q, lim)];
#pragma unroll
        for (int q = 0; q < U; ++q) xv[q] = x[cq[q]];   // past the row's end: some staged entry's column, always a valid x index
#pragma unroll
        for (int q = 0; q < U; ++q) vq[q] = vals[min(k + q, lim)];
#pragma unroll
        for (int q = 0; q < U; ++q) s += (k + q < my_e) ? vq[q] * xv[q] : 0.0;
      }
      st_stream(out + row, OP == MGS_OP_SPMV ? s : (OP == MGS_OP_RESIDUAL ? bi - s : xi + (omega * di) * (bi - s)), (nts & 1) != 0);
    }
  } else if (seq_overflow) {
    // heavier-than-budget block of short rows: lane t walks row t straight from global memory, same
    // ascending order (bit-identical to the staged path)
    const int row = r0 + tid;
    if (row < r1) {
      double s = 0.0;
      for (int k = rowptr[row], e = rowptr[row + 1]; k < e; ++k) s += val[k] * x[col[k]];
      epilogue<OP>(row, s, x, b, dinv, omega, out);
    }
  } else {
    const int sub = tid / LANES, lane = tid % LANES;
    for (int row = r0 + sub; row < r1; row += RB / LANES) {
      double s = 0.0;
      const int e = rowptr[row + 1];
      for (int k = rowptr[row] + lane; k < e; k += LANES) s += val[k] * x[col[k]];
#pragma unroll
      for (int off = LANES / 2; off > 0; off >>= 1) s += __shfl_down(s, off, LANES);
      if (lane == 0) epilogue<OP>(row, s, x, b, dinv, omega, out);
    }
  }
}

// Fused V-cycle passes on the slice kernel (square, unsharded levels only):
//   FUSE_PRE : from x = 0 with ν1 = 1:  x1 = wd∘b,  r = b − A·x1   (gathers wd[c]·b[c]; one pass instead
//              of the (ωD⁻¹)b kernel + the residual kernel; bit-identical to the two-kernel form)
//   FUSE_POST: coarse-grid correction + one Jacobi sweep:  x'' = x1 + Pe + wd∘(r − A·Pe), Pe_j = ec[agg_j], x1 = wd∘b
//              recomputed from b (so the PRE pass need not store it)
//              (r = b − Ax is the residual already computed before restriction, so b − A(x+Pe) = r − A·Pe;
//              one pass instead of prolong-add + Jacobi; equal to the two-kernel form up to rounding)
// wd = ω·dinv precomputed per level.
template <int OP, bool STAGED>
__device__ __forceinline__ double fused_row_sum(const double *__restrict__ vsrc, const int *__restrict__ csrc, int a, int e,
                                                const double *__restrict__ wd, const double *__restrict__ bvec,
                                                const int *__restrict__ agg, const double *__restrict__ ec,
                                                int n_owned, const double *__restrict__ hv /*values of the halo columns (row shards)*/) {
  double s = 0.0;
  for (int k = a; k < e; k += 8) {
    double xv[8];
    const int rem = e - k;
    if (OP == FUSE_PRE) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q < rem) { const int c = csrc[k + q]; xv[q] = wd[c] * (c < n_owned ? bvec[c] : hv[c - n_owned]); } else xv[q] = 0.0;   // row shards: wd covers the halo columns, hv = the peers' raw b
      }
    } else if (OP == FUSE_POST_MAPPED) {
      int av[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) av[q] = q < rem ? csrc[k + q] : -1;      // coarse column (−1: not aggregated)
#pragma unroll
      for (int q = 0; q < 8; ++q) xv[q] = av[q] >= 0 ? ec[av[q]] : 0.0;
    } else {
      int av[8], cv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { cv[q] = q < rem ? csrc[k + q] : -1; av[q] = cv[q] >= 0 ? agg[cv[q]] : -1; }   // agg: coarse column of every local column (row shards: halo slots too)
#pragma unroll
      for (int q = 0; q < 8; ++q) xv[q] = av[q] >= 0 ? ec[av[q]] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) if (q < rem) s += vsrc[k + q] * xv[q];
  }
  return s;
}

template <int OP>
__global__ __launch_bounds__(RB) void csr_rowblock_fused_kernel(
    int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ wd, const double *__restrict__ bvec /*PRE: b, POST: r*/, const double *__restrict__ xin /*POST: b (x1 = wd∘b)*/,
    const int *__restrict__ agg, const double *__restrict__ ec, double *__restrict__ out /*PRE: r, POST: x''*/,
    double *__restrict__ out2 /*PRE: x1*/, int cap, BlockMap bm, const double *__restrict__ hv, const int *__restrict__ blkptr) {
  extern __shared__ double lds_raw[];
  const int vb = map_block(bm, blockIdx.x);
  if (vb < 0) return;
  const int blk = block_of(bm, vb);
  const int r0 = blk * RB;
  const int r1 = min(r0 + RB, n);
  const int tid = threadIdx.x;
  const int lo = blkptr ? blkptr[blk] : rowptr[r0];
  const int hi = blkptr ? blkptr[blk + 1] : rowptr[r1];
  double *__restrict__ vals = lds_raw;
  int *__restrict__ cols = reinterpret_cast<int *>(lds_raw + cap + 2);
  const int row = r0 + tid;
  const int start = lo & ~1;
  const bool staged = hi - lo <= cap;      // block-uniform; the few heavier blocks read their rows from global memory
  int ga = 0, ge = 0;
  double bi = 0.0, wi = 0.0, xi = 0.0, pei = 0.0;
  if (row < r1) {
    ga = rowptr[row]; ge = rowptr[row + 1];
    bi = bvec[row]; wi = wd[row];
    if (OP == FUSE_POST || OP == FUSE_POST_MAPPED) { xi = xin ? wi * xin[row] : 0.0; const int a = agg[row]; pei = a >= 0 ? ec[a] : 0.0; }   // x1 = wd∘b recomputed (xin = b; NULL: bvec holds t = b + r)
  }
  if (staged) {
    const int nch = (hi - start + 1) >> 1;
#pragma unroll 4
    for (int c = tid; c < nch; c += RB) {
      const int k = start + 2 * c;
      *reinterpret_cast<int2_t *>(cols + 2 * c) = *reinterpret_cast<const int2_t *>(col + k);
      *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(val + k);
    }
    __syncthreads();
  }
  if (row < r1) {
    const double s = staged ? fused_row_sum<OP, true>(vals, cols, ga - start, ge - start, wd, bvec, agg, ec, n, hv)
                            : fused_row_sum<OP, false>(val, col, ga, ge, wd, bvec, agg, ec, n, hv);
    if (OP == FUSE_PRE) { out[row] = bi - s; if (out2) out2[row] = wi * bi; }
    else out[row] = xin ? (xi + pei) + wi * (bi - s) : pei + wi * (bi - s);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Pattern-coded rows (DESIGN.md §4 "pattern-coded index").  Operators of PDE problems repeat a handful of row
// shapes: the tuple (idx[k] − base(row))_k of a row — base(row) = row for the column array, agg[row] for the
// aggregate-mapped column array of the fused post pass — takes 3 distinct values in a 256-row block of the
// 7-point operator and a few dozen on the first Galerkin levels.  Setup dedupes the tuples per row block into a
// small table (`tab`: pstart[npat] then the tuples) and stores one byte per ROW (`pid`); the kernel then streams
// only the values (8 B per entry instead of 12) and rebuilds every index as base + tab[pstart[pid] + j] from LDS
// (lanes of a wave mostly read the same table word: LDS broadcast).  Same products in the same order → the results
// stay bit-identical to the CSR kernel.  Row blocks whose table would not pay (irregular rows) keep the index
// array (block-uniform branch), so any matrix is handled.
constexpr int CODE_NEG = (int)0x80000000;   // table word of a negative index (column not aggregated)
// Row shards: an index ≥ split addresses the halo payload hv[index − split].  Its table word is tagged and holds
// slot − row (constant along a plane boundary, where index − base is not): word = CODE_HALO + (slot − row).
constexpr int CODE_HALO = 0x60000000, CODE_HALO_LO = 0x50000000, CODE_OFF_MAX = 0x40000000;

// Inclusive prefix sum over the 64 lanes of a wave in six DPP adds (row_shr 1/2/4/8 inside the rows of 16 lanes, then row_bcast:15 into
// rows 1 and 3 and row_bcast:31 into rows 2 and 3 — the gfx9 scan; lanes without a source take the `old` operand, 0).  Every lane of
// the wave must be active.
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}
// Entry range of a row of a CODED block without reading rowptr per row (option rowptr_scan): a pattern id fixes the row's length (the
// table holds the patterns back to back, ints[0] = their count), so a row starts at rowptr[first row of its wave] — one load per wave —
// plus the wave's prefix sum of lengths.  4 B per row less to stream.  Called by all lanes, behind the barrier that publishes the table.
__device__ __forceinline__ void coded_row_range(const int *__restrict__ ints, int tb, int tlen_blk, int mypid, bool valid, int wave_first, int &ga, int &ge) {
  int len = 0;
  if (valid) {
    const int np = ints[tb];
    const int p0 = ints[tb + mypid], p1 = (mypid + 1 < np) ? ints[tb + mypid + 1] : tlen_blk;
    len = p1 - p0;
  }
  const int incl = wave_incl_scan(len);
  ga = wave_first + incl - len; ge = wave_first + incl;
}

// (round 1: the post pass took 68 VGPRs unbounded = 7 waves per SIMD, bounded to 8 waves it measured 2 % faster, the other ops 0.5–0.8 % slower;
// since its prologue was rewritten it needs 52–56, the other ops 58–64, and the bound no longer binds)
// VAL: the tuples carry the values as well (`vtab`, option valcode): a coded block then streams no matrix entry at all.
// one 256-row block of the pattern-coded kernel (the two __global__ wrappers below call it once per workgroup, or once per row block
// of a group of blocks)
template <int OP, int U, bool HALO, bool VAL>
__device__ __forceinline__ void coded_block_body(
    const int blk, int n, const int *__restrict__ rowptr, const int *__restrict__ idx, const double *__restrict__ val,
    const unsigned char *__restrict__ pid, const int *__restrict__ tptr, const int *__restrict__ tab,
    const double *__restrict__ x /*gather source: x, or e_c for the post pass*/, const double *__restrict__ b /*b, or r for the post pass*/,
    const double *__restrict__ dinv /*dinv, or wd for the post pass*/, double omega, const double *__restrict__ xin /*post pass: b*/,
    const int *__restrict__ agg /*post pass*/, double *__restrict__ out, int capv, int capi, const int *__restrict__ blkptr,
    const double *__restrict__ hv /*row shards: values of the indices >= split*/, int split, const double *__restrict__ vtab,
    const unsigned char *__restrict__ dpos /*t-form post pass: position of a_ii inside the row (NULL: read wd)*/,
    const double *__restrict__ dot_w1, double *__restrict__ dot_part /*SpMV: (y·w1, y·y) partials [2][nblocks] of this launch (NULL: none)*/,
    int dot_nb, int flags /*kernel-uniform; bit 0: no row is longer than U → the gather step runs once, without a loop; bit 1: loop-free staging (option stage_unroll); bit 2: option rowptr_scan*/) {
  extern __shared__ double lds_raw[];
  constexpr bool POST = OP == FUSE_POST_MAPPED;
  const int r0 = blk * RB;
  const int r1 = min(r0 + RB, n);
  const int tid = threadIdx.x;
  const int lo = blkptr[blk], hi = blkptr[blk + 1];
  const int t0 = tptr[blk], tlen = tptr[blk + 1] - t0;
  const int capi_abs = capi < 0 ? -capi : capi;
  double *__restrict__ vals = lds_raw;                                    // capv + 2 doubles
  int *__restrict__ ints = reinterpret_cast<int *>(lds_raw + capv + 2);   // capi ints: the block's table, or its index slice
  const int row = r0 + tid;
  const int start = lo & ~1;
  const int nent = hi - start;
  const bool coded = tlen > 0 && tlen <= capi_abs && (!VAL || tlen <= capv);
  const bool staged = hi - lo <= capv && (coded || nent + 1 <= capi_abs);     // block-uniform
  int ga = 0, ge = 0, base = row;
  double bi = 0.0, di = 0.0, xi = 0.0, pei = 0.0;
  const bool dmode = POST && dpos != nullptr && xin == nullptr;    // ωD⁻¹ from the streamed diagonal entry (same bits as wd = ω·(1/a_ii))
  unsigned dp = 255;
  int mypid = 0;      // the row's pattern id: loaded HERE, with the other per-row loads — behind the barrier it was one more full memory latency in series
  const bool scan = coded && staged && (flags & 4);      // block-uniform: row ranges from the pattern lengths (coded_row_range)
  int wave_first = 0;
  if (scan) wave_first = rowptr[min(r0 + (tid & ~63), n)];
  if (row < r1) {
    if (!scan) { ga = rowptr[row]; ge = rowptr[row + 1]; }
    if (coded) mypid = pid[row];
    if (OP == MGS_OP_SPMV && dot_part) bi = dot_w1[row];       // fused dots: w1 of this row (bi is free in this op); loaded here, not behind the store
    if (OP != MGS_OP_SPMV) bi = b[row];
    if (OP == MGS_OP_JACOBI) { di = dinv[row]; xi = x[row]; }
    if (POST) {                                                     // xin = NULL: b holds t = b + r
      // Every load of this prologue is independent of the others and none is waited for before the value slice below is in flight:
      // the ISA of the first form (dpos → branch → dinv → wait; agg → wait → e_c[agg]) held four serialized memory latencies per
      // workgroup in front of the staging loop, which is why the post pass ran slower per byte than the other ops.  The loads that
      // depend on agg (e_c of the own aggregate) and on the diagonal's position are issued behind the barrier with the gathers.
      base = agg[row];
      if (dmode) dp = dpos[row]; else di = dinv[row];               // kernel-uniform branch
      if (xin) xi = xin[row];                                       // scaled by di in the epilogue
    }
  }
  auto late_loads = [&]() {      // POST: what depends on the prologue's loads
    if (POST && row < r1) {
      pei = base >= 0 ? x[base] : 0.0;
      if (dmode && dp == 255) di = dinv[row];                       // diagonal not found in the row: the wd vector after all
    }
  };
  double s = 0.0;
  if (staged) {
    const int nch = (nent + 1) >> 1;
    if (coded) {
      if (VAL) { for (int c = tid; c < tlen; c += RB) vals[c] = vtab[t0 + c]; }     // value tuples instead of the value slice (tlen <= capv)
      else if (nch <= 4 * RB && (flags & 2)) {
        // No loop in front of the barrier: four predicated 16-byte loads per lane, all in flight together with the prologue's per-row
        // loads.  A loop here makes the compiler drain every outstanding load at its header (s_waitcnt vmcnt(0) in the ISA: wait counts
        // across a back edge are not tracked), i.e. a workgroup paid one full memory latency for rowptr/b/agg/… and a second one for
        // its value slice.
        double2_t rr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int c = tid + k * RB; if (c < nch) rr[k] = *reinterpret_cast<const double2_t *>(val + start + 2 * c); }
        int tw = 0;
        if (tid < tlen) tw = tab[t0 + tid];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int c = tid + k * RB; if (c < nch) *reinterpret_cast<double2_t *>(vals + 2 * c) = rr[k]; }
        if (tid < tlen) ints[tid] = tw;
        for (int c = tid + RB; c < tlen; c += RB) ints[c] = tab[t0 + c];
      } else {
#pragma unroll 4
        for (int c = tid; c < nch; c += RB) *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(val + start + 2 * c);
        for (int c = tid; c < tlen; c += RB) ints[c] = tab[t0 + c];
      }
      if (VAL) for (int c = tid; c < tlen; c += RB) ints[c] = tab[t0 + c];
    } else {
#pragma unroll 4
      for (int c = tid; c < nch; c += RB) {
        const int k = start + 2 * c;
        *reinterpret_cast<int2_t *>(ints + 2 * c) = *reinterpret_cast<const int2_t *>(idx + k);
        *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(val + k);
      }
    }
    __syncthreads();
    if (scan) coded_row_range(ints, 0, tlen, mypid, row < r1, wave_first, ga, ge);
    late_loads();
    if (row < r1 && ge > ga) {
      const int my_a = ga - start, my_e = ge - start, lim = nent - 1;
      if (coded) {
        const int ps = ints[mypid];
        const int last = my_e - my_a - 1;
        auto step = [&](const int k, const int j) {
          int oq[U]; double xv[U], vq[U];
#pragma unroll
          for (int q = 0; q < U; ++q) oq[q] = ints[ps + min(j + q, last)];   // past the row's end: the row's last index again
#pragma unroll
          for (int q = 0; q < U; ++q) {
            if (HALO && oq[q] >= CODE_HALO_LO) xv[q] = hv[row + (oq[q] - CODE_HALO)];
            else if (POST) {           // unconditional load (a column outside every aggregate reads entry 0 and is zeroed): no branch per gather
              const bool neg = oq[q] == CODE_NEG;
              const double g = x[neg ? 0 : base + oq[q]];
              xv[q] = neg ? 0.0 : g;
            } else xv[q] = x[base + oq[q]];
          }
#pragma unroll
          for (int q = 0; q < U; ++q) vq[q] = VAL ? vals[ps + min(j + q, last)] : vals[min(k + q, lim)];
#pragma unroll
          for (int q = 0; q < U; ++q) s += (k + q < my_e) ? vq[q] * xv[q] : 0.0;
        };
        // rows that all fit one step (7-point operators with U = 7, A·P with U = 5): straight-line code, so the loads issued behind
        // the barrier (e_c of the own aggregate) stay in flight beside the gathers instead of being drained at a loop header
        if ((flags & 3) == 3) step(my_a, 0);
        else for (int k = my_a, j = 0; k < my_e; k += U, j += U) step(k, j);
        if (POST && dmode && dp != 255) di = omega * (1.0 / (VAL ? vals[ps + (int)dp] : vals[my_a + (int)dp]));
      } else {
        for (int k = my_a; k < my_e; k += U) {
          int cq[U]; double xv[U], vq[U];
#pragma unroll
          for (int q = 0; q < U; ++q) cq[q] = ints[min(k + q, lim)];
#pragma unroll
          for (int q = 0; q < U; ++q) {
            if (HALO && cq[q] >= split) xv[q] = hv[cq[q] - split];
            else if (POST) { const bool neg = cq[q] < 0; const double g = x[neg ? 0 : cq[q]]; xv[q] = neg ? 0.0 : g; }
            else xv[q] = x[cq[q]];
          }
#pragma unroll
          for (int q = 0; q < U; ++q) vq[q] = vals[min(k + q, lim)];
#pragma unroll
          for (int q = 0; q < U; ++q) s += (k + q < my_e) ? vq[q] * xv[q] : 0.0;
        }
        if (POST && dmode && dp != 255) di = omega * (1.0 / vals[my_a + (int)dp]);
      }
    }
  } else if (row < r1) {
    // heavier-than-budget block: lane t walks row t straight from global memory, same ascending order
    late_loads();
    for (int k = ga; k < ge; ++k) {
      const int c = idx[k];
      s += val[k] * ((HALO && c >= split) ? hv[c - split] : ((POST && c < 0) ? 0.0 : x[c]));
    }
    if (POST && dmode && dp != 255) di = omega * (1.0 / val[ga + (int)dp]);
  }
  if (row < r1) {
    double v;
    if (OP == MGS_OP_SPMV) v = s;
    else if (OP == MGS_OP_RESIDUAL) v = bi - s;
    else if (OP == MGS_OP_JACOBI) v = xi + (omega * di) * (bi - s);
    else v = xin ? (di * xi + pei) + di * (bi - s) : pei + di * (bi - s);     // t-form: x = Pe + wd∘(t − A·Pe), t = b + r
    st_stream(out + row, v, capi < 0);                                              // capi < 0: streaming store (option nt_store)
    if (OP == MGS_OP_SPMV && dot_part) { bi = v * bi; di = v * v; }                // this row's terms of (y·w1, y·y); bi/di are free in this op
  }
  if (OP == MGS_OP_SPMV && dot_part) {      // launch-uniform: one partial pair per row block, summed in a fixed order
    double p1 = row < r1 ? bi : 0.0, p2 = row < r1 ? di : 0.0;
    for (int off = 32; off > 0; off >>= 1) { p1 += __shfl_down(p1, off); p2 += __shfl_down(p2, off); }
    __syncthreads();                                   // the staged values are done with: their LDS holds the 4 wave sums
    if ((tid & 63) == 0) { vals[2 * (tid >> 6)] = p1; vals[2 * (tid >> 6) + 1] = p2; }
    __syncthreads();
    if (tid == 0) {
      double t1 = 0.0, t2 = 0.0;
      for (int q = 0; q < RB / 64; ++q) { t1 += vals[2 * q]; t2 += vals[2 * q + 1]; }
      dot_part[blk] = t1; dot_part[dot_nb + blk] = t2;
    }
  }
}

#define CODED_PARAMS                                                                                                                    \
    int n, const int *__restrict__ rowptr, const int *__restrict__ idx, const double *__restrict__ val,                                  \
    const unsigned char *__restrict__ pid, const int *__restrict__ tptr, const int *__restrict__ tab, const double *__restrict__ x,      \
    const double *__restrict__ b, const double *__restrict__ dinv, double omega, const double *__restrict__ xin,                         \
    const int *__restrict__ agg, double *__restrict__ out, int capv, int capi, BlockMap bm, const int *__restrict__ blkptr,               \
    const double *__restrict__ hv, int split, const double *__restrict__ vtab, const unsigned char *__restrict__ dpos,                   \
    const double *__restrict__ dot_w1, double *__restrict__ dot_part, int dot_nb, int flags
#define CODED_ARGS n, rowptr, idx, val, pid, tptr, tab, x, b, dinv, omega, xin, agg, out, capv, capi, blkptr, hv, split, vtab, dpos, dot_w1, dot_part, dot_nb, flags

// (round 1: the post pass took 68 VGPRs unbounded = 7 waves per SIMD, bounded to 8 waves it measured 2 % faster, the other ops 0.5–0.8 % slower;
// since its prologue was rewritten it needs 52–56, the other ops 58–64, and the bound no longer binds)
template <int OP, int U, bool HALO, bool VAL>
__global__ __launch_bounds__(RB, OP == FUSE_POST_MAPPED ? 8 : 1) void csr_rowblock_coded_kernel(CODED_PARAMS) {
  const int vb = map_block(bm, blockIdx.x);
  if (vb < 0) return;
  coded_block_body<OP, U, HALO, VAL>(block_of(bm, vb), CODED_ARGS);
}
// the same row blocks swept group by group (descriptors of the grouped pre pass: up to GRP_BLOCKS blocks per workgroup, one after the
// other) — experiment: does the post pass gain what the grouped pre pass gains from fewer, fatter workgroups?
template <int OP, int U, bool HALO, bool VAL>
__global__ __launch_bounds__(RB, OP == FUSE_POST_MAPPED ? 8 : 1) void csr_rowblock_coded_group_kernel(CODED_PARAMS, const int *__restrict__ gdesc) {
  const int g = map_block(bm, blockIdx.x);
  if (g < 0) return;
  const int4_t gb = *reinterpret_cast<const int4_t *>(gdesc + (size_t)32 * g);
#pragma unroll 1
  for (int h = 0; h < 4; ++h) {
    const int blk = h == 0 ? gb.x : (h == 1 ? gb.y : (h == 2 ? gb.z : gb.w));
    if (blk < 0) break;                               // group-uniform
    if (h) __syncthreads();                           // the previous block's LDS slice is done with
    coded_block_body<OP, U, HALO, VAL>(blk, CODED_ARGS);
  }
}

// setup pass 1: per row block, elect representatives (smallest row of each distinct tuple), give every row the
// rank of its representative (deterministic: order of first appearance), and size the block's table.  A block
// whose table would exceed half of its index slice is left uncoded (size 0).
// (a word outside the representable ranges makes the block uncodable: returns CODE_BAD)
constexpr int CODE_BAD = 0x7fffffff;
__device__ __forceinline__ int code_off(int i, int base, int row, int split) {
  if (i < 0) return CODE_NEG;
  if (i >= split) { const int d = (i - split) - row; return (d > -0x10000000 && d < 0x10000000) ? CODE_HALO + d : CODE_BAD; }
  const int o = i - base;
  return (o > -CODE_OFF_MAX && o < CODE_OFF_MAX) ? o : CODE_BAD;
}
__global__ __launch_bounds__(RB) void rowcode_assign_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ idx,
                                                            const int *__restrict__ base, int split, unsigned char *__restrict__ pid,
                                                            unsigned char *__restrict__ isrep, int *__restrict__ blk_ints,
                                                            const double *__restrict__ val /*non-null: tuples include the values*/) {
  __shared__ int s_rep, s_ints, s_np, s_bad;
  const int blk = blockIdx.x, r0 = blk * RB, r1 = min(r0 + RB, n), tid = threadIdx.x, row = r0 + tid;
  const bool valid = row < r1;
  int a = 0, len = 0, bs = 0;
  if (valid) { a = rowptr[row]; len = rowptr[row + 1] - a; bs = base ? base[row] : row; }
  // table budget: half of what the block streams otherwise (4 B per entry index-only, 12 B with values; a table entry
  // costs 4 resp. 12 B) → the same entry count either way
  const int budget = (rowptr[r1] - rowptr[r0]) / 2 - 64;
  bool assigned = !valid, rep = false, ok = true;
  int mypid = 0;
  if (tid == 0) { s_ints = 0; s_np = 0; s_bad = 0; }
  for (int p = 0; p < RB; ++p) {
    if (tid == 0) s_rep = 0x7fffffff;
    __syncthreads();
    if (!assigned) atomicMin(&s_rep, tid);
    __syncthreads();
    const int r = s_rep;
    if (r == 0x7fffffff) break;
    if (!assigned) {
      bool same = true;
      if (tid != r) {
        const int rr = r0 + r, ra = rowptr[rr], rl = rowptr[rr + 1] - ra, rb = base ? base[rr] : rr;
        same = rl == len;
        for (int j = 0; same && j < len; ++j) same = code_off(idx[a + j], bs, row, split) == code_off(idx[ra + j], rb, rr, split);
        if (val) for (int j = 0; same && j < len; ++j) same = __double_as_longlong(val[a + j]) == __double_as_longlong(val[ra + j]);   // same bits
      } else {
        rep = true; s_ints += len; s_np = p + 1;
        for (int j = 0; j < len; ++j) if (code_off(idx[a + j], bs, row, split) == CODE_BAD) s_bad = 1;
      }
      if (same) { assigned = true; mypid = p; }
    }
    __syncthreads();
    if (s_np + s_ints > budget || s_bad) { ok = false; break; }
  }
  if (valid) { pid[row] = ok ? (unsigned char)mypid : 0; isrep[row] = (ok && rep) ? 1 : 0; }
  if (tid == 0) blk_ints[blk] = ok ? s_np + s_ints : 0;
}
// setup pass 2: write the tables (pstart[npat], then the tuples in representative order)
__global__ __launch_bounds__(RB) void rowcode_fill_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ idx,
                                                          const int *__restrict__ base, int split, const unsigned char *__restrict__ pid,
                                                          const unsigned char *__restrict__ isrep, const int *__restrict__ tptr,
                                                          int *__restrict__ tab, const double *__restrict__ val, double *__restrict__ vtab) {
  __shared__ int plen[RB], pstart[RB];
  const int blk = blockIdx.x, r0 = blk * RB, r1 = min(r0 + RB, n), tid = threadIdx.x, row = r0 + tid;
  const int t0 = tptr[blk];
  if (tptr[blk + 1] == t0) return;
  const bool rep = row < r1 && isrep[row];
  int a = 0, len = 0, bs = 0, p = 0;
  if (rep) { a = rowptr[row]; len = rowptr[row + 1] - a; bs = base ? base[row] : row; p = pid[row]; plen[p] = len; }
  const int np = __syncthreads_count(rep);
  if (tid == 0) { int acc = np; for (int q = 0; q < np; ++q) { pstart[q] = acc; acc += plen[q]; } }
  __syncthreads();
  if (rep) {
    tab[t0 + p] = pstart[p];
    for (int j = 0; j < len; ++j) tab[t0 + pstart[p] + j] = code_off(idx[a + j], bs, row, split);
    if (vtab) for (int j = 0; j < len; ++j) vtab[t0 + pstart[p] + j] = val[a + j];     // same indexing as tab (header slots unused)
  }
}

// Variant D: variant B with ONE WAVE per workgroup (64 rows, ~5 KB of LDS).  No workgroup barrier at
// all — a wave's LDS writes are ordered before its own reads — so every wave streams, gathers and
// stores at its own pace and the CU always has waves in each phase.  Four consecutive 64-row groups of
// one 256-row block map to the same XCD (locality as in variant B).
template <int OP>
__global__ __launch_bounds__(64) void csr_wave_slice_kernel(
    int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, const double *__restrict__ b, const double *__restrict__ dinv, double omega,
    double *__restrict__ out, int cap, BlockMap bm) {
  extern __shared__ double lds_raw[];
  const int bid = blockIdx.x;
  int vb, sub;
  if (bm.remap) { const int idx = bid >> 3; vb = map_block_xi(bm, bid & 7, idx >> 2); sub = idx & 3; }
  else { vb = (bid >> 2) < bm.nblocks ? (bid >> 2) : -1; sub = bid & 3; }
  if (vb < 0) return;
  const int r0 = (bm.base + vb) * RB + sub * 64;
  if (r0 >= n) return;
  const int r1 = min(r0 + 64, n);
  const int tid = threadIdx.x;
  const int lo = rowptr[r0];
  const int hi = rowptr[r1];
  double *__restrict__ vals = lds_raw;
  int *__restrict__ cols = reinterpret_cast<int *>(lds_raw + cap + 2);
  const int row = r0 + tid;
  const int start = lo & ~1;
  int my_a = 0, my_e = 0;
  double bi = 0.0, di = 0.0, xi = 0.0;
  if (row < r1) {
    my_a = rowptr[row] - start; my_e = rowptr[row + 1] - start;
    if (OP != MGS_OP_SPMV) bi = b[row];
    if (OP == MGS_OP_JACOBI) { di = dinv[row]; xi = x[row]; }
  }
  const int nch = (hi - start + 1) >> 1;
#pragma unroll 4
  for (int c = tid; c < nch; c += 64) {
    const int k = start + 2 * c;
    *reinterpret_cast<int2_t *>(cols + 2 * c) = *reinterpret_cast<const int2_t *>(col + k);
    *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(val + k);
  }
  __syncthreads();   // single-wave workgroup: lowers to a wait on the wave's own LDS writes
  if (row < r1) {
    double s = 0.0;
    for (int k = my_a; k < my_e; k += 8) {
      double xv[8];
      const int rem = my_e - k;
#pragma unroll
      for (int q = 0; q < 8; ++q) xv[q] = q < rem ? x[cols[k + q]] : 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) if (q < rem) s += vals[k + q] * xv[q];
    }
    if (OP == MGS_OP_SPMV) out[row] = s;
    else if (OP == MGS_OP_RESIDUAL) out[row] = bi - s;
    else out[row] = xi + (omega * di) * (bi - s);
  }
}

// Variant C: variant B software-pipelined over G consecutive row blocks per workgroup.  The slice of
// row block i+1 is already in flight (registers) while the lanes gather x and sum row block i, so the
// val/col stream never waits for the gather phase of its own workgroup.  Same arithmetic, same order.
template <int OP, bool NT, int ITER, int G>
__global__ __launch_bounds__(RB) void csr_rowblock_pipe_kernel(
    int n, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, const double *__restrict__ b, const double *__restrict__ dinv, double omega,
    double *__restrict__ out, int cap, BlockMap bm, int nrb_total /*row blocks of the launched range*/) {
  extern __shared__ double lds_raw[];
  double *__restrict__ vals = lds_raw;
  int *__restrict__ cols = reinterpret_cast<int *>(lds_raw + cap + 2);
  const int sb = map_block(bm, blockIdx.x);      // super block index (G row blocks each)
  if (sb < 0) return;
  const int tid = threadIdx.x;
  const int first = sb * G;
  const int nblk = min(G, nrb_total - first);
  int bnd[G + 1];
#pragma unroll
  for (int j = 0; j <= G; ++j) { const int r = min((bm.base + first + j) * RB, n); bnd[j] = rowptr[r]; }
  double2_t rv[ITER]; int2_t rc[ITER];
  auto issue = [&](int j) {
    const int start = bnd[j] & ~1;
    const int nch = (bnd[j + 1] - start + 1) >> 1;
#pragma unroll
    for (int q = 0; q < ITER; ++q) {
      const int c = tid + q * RB;
      if (c < nch) {
        rc[q] = ld_stream<NT>(reinterpret_cast<const int2_t *>(col + start + 2 * c));
        rv[q] = ld_stream<NT>(reinterpret_cast<const double2_t *>(val + start + 2 * c));
      }
    }
  };
  issue(0);
#pragma unroll
  for (int j = 0; j < G; ++j) {
    if (j >= nblk) break;
    const int r0 = (bm.base + first + j) * RB, r1 = min(r0 + RB, n);
    const int lo = bnd[j], hi = bnd[j + 1];
    const int start = lo & ~1;
    const int nch = (hi - start + 1) >> 1;
    const int row = r0 + tid;
    int my_a = 0, my_e = 0;
    double bi = 0.0, di = 0.0, xi = 0.0;
    if (row < r1) {
      my_a = rowptr[row] - start; my_e = rowptr[row + 1] - start;
      if (OP != MGS_OP_SPMV) bi = b[row];
      if (OP == MGS_OP_JACOBI) { di = dinv[row]; xi = x[row]; }
    }
#pragma unroll
    for (int q = 0; q < ITER; ++q) {
      const int c = tid + q * RB;
      if (c < nch) {
        *reinterpret_cast<double2_t *>(vals + 2 * c) = rv[q];
        *reinterpret_cast<int2_t *>(cols + 2 * c) = rc[q];
      }
    }
    __syncthreads();
    if (j + 1 < nblk) issue(j + 1);
    if (row < r1) {
      double s = 0.0;
      int k = my_a;
      for (; k + 8 <= my_e; k += 8) {
        double xv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) xv[q] = x[cols[k + q]];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += vals[k + q] * xv[q];
      }
      {
        double xv[8];
        const int rem = my_e - k;
#pragma unroll
        for (int q = 0; q < 8; ++q) xv[q] = q < rem ? x[cols[k + q]] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) if (q < rem) s += vals[k + q] * xv[q];
      }
      if (OP == MGS_OP_SPMV) out[row] = s;
      else if (OP == MGS_OP_RESIDUAL) out[row] = bi - s;
      else out[row] = xi + (omega * di) * (bi - s);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// Grouped pre pass (DESIGN.md §4 "restriction inside the pre pass").  The V(1,1) cycle from x = 0 needs r = b − Â·b only
// for two things: its restriction r_c = Pᵀr and the post pass, which needs b + r.  Aggregates of the pairwise matching
// are 2–4 rows that sit in one row block or in two (a row and its neighbour one grid line further), so setup pairs such
// row blocks into GROUPS; one workgroup sweeps the blocks of a group with the coded row-block kernel's body, keeps the
// ≤ 512 residuals in LDS, writes t = b + r and restricts every aggregate whose members all lie in the group — same
// members, same ascending order as restrict_agg_kernel, so r_c has the same bits.  The r vector never travels through HBM
// (−8 B per row written, −8 B read back by the restriction, −4 B of member index, −8 B in the post pass, which reads t
// instead of r and b).  Aggregates that leave their group (odd shapes at domain boundaries) are "strays": their member rows
// also store r (bit mask), and a small trailing kernel restricts them from memory.
constexpr int GRP_MAX_MEMBERS = 6;     // 4 + 10 + 5·10 = 64 bits of the member code
constexpr int GRP_BLOCKS = 4;          // row blocks per group (residual buffer: up to 4·256 doubles of LDS)
constexpr int GRP_DESC = 32;           // ints per group descriptor (128 B: blocks, entry bounds lo/hi, aggregate ranges lo/hi)

// setup 1: aggregate ids must ascend with their first member (they do for the device matching: ids = rank of the leader);
// afirst[b] = first aggregate whose first member lies in row block >= b; votes: per row block up to GRP_SLOTS distinct
// other blocks its aggregates reach, with counts (the host pairs every block with its most frequent partner)
constexpr int GRP_SLOTS = 4;
__global__ void grp_scan_kernel(int nc, const int *__restrict__ cptr, const int *__restrict__ members, int nblocks,
                                int *__restrict__ vkey, int *__restrict__ vcnt, int *__restrict__ afirst, int *__restrict__ bad) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nc) return;
  const int lo = cptr[a], hi = cptr[a + 1];
  if (hi <= lo) { atomicOr(bad, 1); return; }
  const int fm = members[lo], fb = fm / RB, lb = members[hi - 1] / RB;
  int pb = -1;
  if (a > 0) {
    const int plo = cptr[a - 1];
    if (lo <= plo) { atomicOr(bad, 1); return; }
    const int pfm = members[plo];
    if (pfm >= fm) { atomicOr(bad, 1); return; }
    pb = pfm / RB;
  }
  for (int q = pb + 1; q <= fb; ++q) afirst[q] = a;
  if (a == nc - 1) for (int q = fb + 1; q <= nblocks; ++q) afirst[q] = nc;
  if (lb != fb)
    for (int sl = 0; sl < GRP_SLOTS; ++sl) {
      const int old = atomicCAS(&vkey[fb * GRP_SLOTS + sl], -1, lb);
      if (old == -1 || old == lb) { atomicAdd(&vcnt[fb * GRP_SLOTS + sl], 1); break; }
    }
}
// setup 2: member positions inside the group's residual buffer (q-th block of the group: q·256 .. q·256+255), or stray
__global__ void grp_code_kernel(int nc, const int *__restrict__ cptr, const int *__restrict__ members, const int *__restrict__ blk2grp,
                                const int *__restrict__ gblk, unsigned long long *__restrict__ acode, unsigned *__restrict__ wmask,
                                int *__restrict__ stray, int stray_cap, int *__restrict__ nstray) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nc) return;
  const int lo = cptr[a], cnt = cptr[a + 1] - lo;
  const int g = blk2grp[members[lo] / RB];
  int gb[GRP_BLOCKS];
#pragma unroll
  for (int q = 0; q < GRP_BLOCKS; ++q) gb[q] = gblk[GRP_BLOCKS * g + q];
  bool ok = cnt <= GRP_MAX_MEMBERS;
  unsigned long long code = (unsigned long long)cnt;
  int prev = 0, sh = 14;
  for (int k = 0; ok && k < cnt; ++k) {
    const int m = members[lo + k], blk = m / RB;
    int pos = -1;
#pragma unroll
    for (int q = 0; q < GRP_BLOCKS; ++q) if (blk == gb[q]) pos = q * RB + m - blk * RB;
    if (pos < 0) { ok = false; break; }
    if (k == 0) code |= (unsigned long long)pos << 4;
    else { const int d = pos - prev; if (d <= 0 || d >= 1024) { ok = false; break; } code |= (unsigned long long)d << sh; sh += 10; }
    prev = pos;
  }
  if (ok) { acode[a] = code; return; }
  acode[a] = 0ull;
  const int q = atomicAdd(nstray, 1);
  if (q < stray_cap) stray[q] = a;
  for (int k = 0; k < cnt; ++k) { const int m = members[lo + k]; atomicOr(&wmask[m >> 5], 1u << (m & 31)); }
}
__global__ void restrict_stray_kernel(int ns, const int *__restrict__ stray, const int *__restrict__ cptr, const int *__restrict__ members,
                                      const double *__restrict__ r, double *__restrict__ rc) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= ns) return;
  const int a = stray[q];
  double s = 0.0;
  for (int k = cptr[a], e = cptr[a + 1]; k < e; ++k) s += r[members[k]];
  rc[a] = s;
}

// the grouped pass itself: body of csr_rowblock_coded_kernel<RESIDUAL> per row block (coded table, uncoded index slice, or the
// unstaged walk — block-uniform choices), then the in-LDS restriction of the group's aggregates
// (six waves per SIMD: the halo variants of U = 7 / 8 came out at 82 / 89 registers — one wave less than the 80 of the others — and ran 6 % behind
// their share on a row shard)
template <int U, bool HALO>
__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(6))) void csr_group_pre_kernel(
    int n, const int *__restrict__ rowptr, const int *__restrict__ idx, const double *__restrict__ val,
    const unsigned char *__restrict__ pid, const int *__restrict__ tptr, const int *__restrict__ tab,
    const double *__restrict__ x, const double *__restrict__ b, double *__restrict__ t_out, double *__restrict__ r_out,
    double *__restrict__ rc_out, const int *__restrict__ gdesc, const unsigned long long *__restrict__ acode,
    const unsigned *__restrict__ wmask, int capv, int capi, BlockMap bm, const double *__restrict__ hv, int split, int nts /*bit 0: streaming store of t; bit 1: slice staged by LDS-DMA, no loop; bit 2: option rowptr_scan*/,
    const int *__restrict__ gorder, int gper) {
  extern __shared__ double lds_raw[];
  int g;
  if (gorder) {        // option group_order: workgroup (xcd, idx) → group, in the order of the one-block kernels' plane sweep
    const int idx = blockIdx.x >> 3;
    if (idx >= gper) return;
    g = gorder[(blockIdx.x & 7) * gper + idx];
  } else g = map_block(bm, blockIdx.x);
  if (g < 0) return;
  const int tid = threadIdx.x;
  double *__restrict__ vals = lds_raw;                                    // capv + 2 doubles
  int *__restrict__ ints = reinterpret_cast<int *>(lds_raw + capv + 2);   // capi ints: the block's table, or its index slice
  double *__restrict__ rbuf = lds_raw + capv + 2 + (capi + 1) / 2;        // residuals of the group's row blocks
  // group descriptor (one 128-byte record, no dependent loads behind it): blocks, their entry bounds, their aggregate ranges
  const int4_t *__restrict__ gd = reinterpret_cast<const int4_t *>(gdesc + (size_t)GRP_DESC * g);
  const int4_t gb = gd[0], glo = gd[1], ghi = gd[2], galo = gd[3], gahi = gd[4];
#define GSEL(v, h) ((h) == 0 ? (v).x : ((h) == 1 ? (v).y : ((h) == 2 ? (v).z : (v).w)))
  // member codes of this lane's first aggregate per block: in flight while the blocks are swept
  unsigned long long code0[GRP_BLOCKS];
#pragma unroll
  for (int h = 0; h < GRP_BLOCKS; ++h) { const int a = GSEL(galo, h) + tid; code0[h] = (GSEL(gb, h) >= 0 && a < GSEL(gahi, h)) ? acode[a] : 0ull; }
#pragma unroll
  for (int h = 0; h < GRP_BLOCKS; ++h) {
    const int blk = GSEL(gb, h);
    if (blk < 0) break;                                                   // group-uniform
    if (h) __syncthreads();                                               // the previous block's LDS slice is done with
    const int r0 = blk * RB, r1 = min(r0 + RB, n);
    const int lo = GSEL(glo, h), hi = GSEL(ghi, h);
    const int t0 = tptr ? tptr[blk] : 0, tlen = tptr ? tptr[blk + 1] - t0 : 0;
    const int row = r0 + tid, start = lo & ~1, nent = hi - start;
    const bool coded = tlen > 0 && tlen <= capi;
    const bool staged = hi - lo <= capv && (coded || nent + 1 <= capi);   // block-uniform
    int ga = 0, ge = 0;
    double bi = 0.0;
    unsigned wm = 0u;
    int mypid = 0;                                                        // loaded with the other per-row loads, not behind the barrier
    const bool scan = coded && staged && (nts & 4);                       // block-uniform (coded_row_range)
    int wave_first = 0;
    if (scan) wave_first = rowptr[min(r0 + (tid & ~63), n)];
    if (row < r1) { if (!scan) { ga = rowptr[row]; ge = rowptr[row + 1]; } bi = b[row]; wm = wmask[row >> 5]; if (coded) mypid = pid[row]; }
    double s = 0.0;
    if (staged) {
      const int nch = (nent + 1) >> 1;
      if (coded && nch <= 4 * RB && (nts & 2)) {          // no loop and no staging registers in front of the barrier (see coded_block_body)
        stage_pairs_dma(val + start, vals, nch, tid);
        int tw = 0;
        if (tid < tlen) tw = tab[t0 + tid];
        if (tid < tlen) ints[tid] = tw;
        for (int c = tid + RB; c < tlen; c += RB) ints[c] = tab[t0 + c];
        // LDS-DMA transfers are vector-memory operations of the issuing wave: the workgroup barrier's fence does not cover them, so every
        // wave drains its own before it arrives (spelled out; the compiler happens to place the same wait in front of s_barrier today)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (coded) {
#pragma unroll 4
        for (int c = tid; c < nch; c += RB) *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(val + start + 2 * c);
        for (int c = tid; c < tlen; c += RB) ints[c] = tab[t0 + c];
      } else {
#pragma unroll 4
        for (int c = tid; c < nch; c += RB) {
          const int k = start + 2 * c;
          *reinterpret_cast<int2_t *>(ints + 2 * c) = *reinterpret_cast<const int2_t *>(idx + k);
          *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(val + k);
        }
      }
      __syncthreads();
      if (scan) coded_row_range(ints, 0, tlen, mypid, row < r1, wave_first, ga, ge);
      if (row < r1 && ge > ga) {
        const int my_a = ga - start, my_e = ge - start, lim = nent - 1;
        if (coded) {
          const int ps = ints[mypid];
          const int last = my_e - my_a - 1;
          for (int k = my_a, j = 0; k < my_e; k += U, j += U) {
            int oq[U]; double xv[U], vq[U];
#pragma unroll
            for (int q = 0; q < U; ++q) oq[q] = ints[ps + min(j + q, last)];
#pragma unroll
            for (int q = 0; q < U; ++q) {
              if (HALO && oq[q] >= CODE_HALO_LO) xv[q] = hv[row + (oq[q] - CODE_HALO)];
              else xv[q] = x[row + oq[q]];
            }
#pragma unroll
            for (int q = 0; q < U; ++q) vq[q] = vals[min(k + q, lim)];
#pragma unroll
            for (int q = 0; q < U; ++q) s += (k + q < my_e) ? vq[q] * xv[q] : 0.0;
          }
        } else {
          for (int k = my_a; k < my_e; k += U) {
            int cq[U]; double xv[U], vq[U];
#pragma unroll
            for (int q = 0; q < U; ++q) cq[q] = ints[min(k + q, lim)];
#pragma unroll
            for (int q = 0; q < U; ++q) xv[q] = (HALO && cq[q] >= split) ? hv[cq[q] - split] : x[cq[q]];
#pragma unroll
            for (int q = 0; q < U; ++q) vq[q] = vals[min(k + q, lim)];
#pragma unroll
            for (int q = 0; q < U; ++q) s += (k + q < my_e) ? vq[q] * xv[q] : 0.0;
          }
        }
      }
    } else if (row < r1) {
      for (int k = ga; k < ge; ++k) { const int c = idx[k]; s += val[k] * ((HALO && c >= split) ? hv[c - split] : x[c]); }
    }
    // members of stray aggregates also store r — the whole wave does when one of its rows must: 64 consecutive doubles are four full
    // cache lines, while lone 8-byte stores each cost a read-modify-write of an ECC word in HBM.  The vote is taken BEFORE the stores:
    // read after them, the mask (a loaded value) made the compiler wait for vmcnt(0) behind the t store, i.e. for the store itself to
    // complete, once per row block of the sweep.
    const bool stray_wave = __any((int)((wm >> (row & 31)) & 1u)) != 0;
    if (row < r1) {
      const double r = bi - s;
      st_stream(t_out + row, bi + r, (nts & 1) != 0);
      rbuf[h * RB + tid] = r;
      if (stray_wave) r_out[row] = r;
    }
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < GRP_BLOCKS; ++h) {
    if (GSEL(gb, h) < 0) break;
    const int a0 = GSEL(galo, h) + tid, ae = GSEL(gahi, h);
    for (int a = a0; a < ae; a += RB) {
      const unsigned long long code = a == a0 ? code0[h] : acode[a];
      const int cnt = (int)(code & 15ull);
      if (cnt == 0) continue;
      int pos = (int)((code >> 4) & 1023ull), sh = 14;
      double sum = 0.0;
      sum += rbuf[pos];
      for (int k = 1; k < cnt; ++k, sh += 10) { pos += (int)((code >> sh) & 1023ull); sum += rbuf[pos]; }
      rc_out[a] = sum;
    }
  }
#undef GSEL
}

// Concurrent form for groups of at most two row blocks: a 512-thread workgroup, each half sweeps one block of the pair at the same
// time (own value/table slices in LDS), one barrier later all 512 threads restrict the group's aggregates.  Same phase structure as the
// plain row-block kernel plus one barrier (the sequential form above pays a second staging phase per extra block), same occupancy
// (2 × slice + 4 KB of LDS per workgroup → 4 workgroups = 32 waves per CU).  Every barrier sits outside the per-block branches.
template <int U, bool HALO>
__global__ __launch_bounds__(2 * RB) void csr_group2_pre_kernel(
    int n, const int *__restrict__ rowptr, const int *__restrict__ idx, const double *__restrict__ val,
    const unsigned char *__restrict__ pid, const int *__restrict__ tptr, const int *__restrict__ tab,
    const double *__restrict__ x, const double *__restrict__ b, double *__restrict__ t_out, double *__restrict__ r_out,
    double *__restrict__ rc_out, const int *__restrict__ gdesc, const unsigned long long *__restrict__ acode,
    const unsigned *__restrict__ wmask, int capv, int capi, BlockMap bm, const double *__restrict__ hv, int split) {
  extern __shared__ double lds_raw[];
  const int g = map_block(bm, blockIdx.x);
  if (g < 0) return;
  const int half = threadIdx.x >> 8, tid = threadIdx.x & (RB - 1);
  const int half_doubles = capv + 2 + (capi + 1) / 2;
  double *__restrict__ vals = lds_raw + half * half_doubles;
  int *__restrict__ ints = reinterpret_cast<int *>(vals + capv + 2);
  double *__restrict__ rbuf = lds_raw + 2 * half_doubles;                 // 2·RB residuals: block 0 of the pair, then block 1
  const int4_t *__restrict__ gd = reinterpret_cast<const int4_t *>(gdesc + (size_t)GRP_DESC * g);
  const int4_t gb = gd[0], glo = gd[1], ghi = gd[2], galo = gd[3], gahi = gd[4];
  const int blk = half ? gb.y : gb.x;                                      // −1: a single-block group, this half only helps to restrict
  const int a0 = (half ? galo.y : galo.x) + tid, ae = half ? gahi.y : gahi.x;
  const unsigned long long code0 = (blk >= 0 && a0 < ae) ? acode[a0] : 0ull;
  const int r0 = blk * RB, r1 = blk >= 0 ? min(r0 + RB, n) : 0;
  const int lo = half ? glo.y : glo.x, hi = half ? ghi.y : ghi.x;
  const int t0 = (tptr && blk >= 0) ? tptr[blk] : 0, tlen = (tptr && blk >= 0) ? tptr[blk + 1] - t0 : 0;
  const int row = r0 + tid, start = lo & ~1, nent = hi - start;
  const bool coded = tlen > 0 && tlen <= capi;
  const bool staged = blk >= 0 && hi - lo <= capv && (coded || nent + 1 <= capi);   // uniform per half
  int ga = 0, ge = 0;
  double bi = 0.0;
  unsigned wm = 0u;
  if (blk >= 0 && row < r1) { ga = rowptr[row]; ge = rowptr[row + 1]; bi = b[row]; wm = wmask[row >> 5]; }
  if (staged) {
    const int nch = (nent + 1) >> 1;
    if (coded) {
#pragma unroll 4
      for (int c = tid; c < nch; c += RB) *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(val + start + 2 * c);
      for (int c = tid; c < tlen; c += RB) ints[c] = tab[t0 + c];
    } else {
#pragma unroll 4
      for (int c = tid; c < nch; c += RB) {
        const int k = start + 2 * c;
        *reinterpret_cast<int2_t *>(ints + 2 * c) = *reinterpret_cast<const int2_t *>(idx + k);
        *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(val + k);
      }
    }
  }
  __syncthreads();
  double s = 0.0;
  if (blk >= 0 && row < r1) {
    if (staged) {
      if (ge > ga) {
        const int my_a = ga - start, my_e = ge - start, lim = nent - 1;
        if (coded) {
          const int ps = ints[pid[row]];
          const int last = my_e - my_a - 1;
          for (int k = my_a, j = 0; k < my_e; k += U, j += U) {
            int oq[U]; double xv[U], vq[U];
#pragma unroll
            for (int q = 0; q < U; ++q) oq[q] = ints[ps + min(j + q, last)];
#pragma unroll
            for (int q = 0; q < U; ++q) {
              if (HALO && oq[q] >= CODE_HALO_LO) xv[q] = hv[row + (oq[q] - CODE_HALO)];
              else xv[q] = x[row + oq[q]];
            }
#pragma unroll
            for (int q = 0; q < U; ++q) vq[q] = vals[min(k + q, lim)];
#pragma unroll
            for (int q = 0; q < U; ++q) s += (k + q < my_e) ? vq[q] * xv[q] : 0.0;
          }
        } else {
          for (int k = my_a; k < my_e; k += U) {
            int cq[U]; double xv[U], vq[U];
#pragma unroll
            for (int q = 0; q < U; ++q) cq[q] = ints[min(k + q, lim)];
#pragma unroll
            for (int q = 0; q < U; ++q) xv[q] = (HALO && cq[q] >= split) ? hv[cq[q] - split] : x[cq[q]];
#pragma unroll
            for (int q = 0; q < U; ++q) vq[q] = vals[min(k + q, lim)];
#pragma unroll
            for (int q = 0; q < U; ++q) s += (k + q < my_e) ? vq[q] * xv[q] : 0.0;
          }
        }
      }
    } else {
      for (int k = ga; k < ge; ++k) { const int c = idx[k]; s += val[k] * ((HALO && c >= split) ? hv[c - split] : x[c]); }
    }
    const double r = bi - s;
    t_out[row] = bi + r;
    rbuf[half * RB + tid] = r;
  }
  if (__any((int)((wm >> (row & 31)) & 1u)) && blk >= 0 && row < r1) r_out[row] = bi - s;   // whole waves: see csr_group_pre_kernel
  __syncthreads();
  for (int a = a0; a < ae; a += RB) {                                      // this half's block's aggregates; positions span both halves
    const unsigned long long code = a == a0 ? code0 : acode[a];
    const int cnt = (int)(code & 15ull);
    if (cnt == 0) continue;
    int pos = (int)((code >> 4) & 1023ull), sh = 14;
    double sum = 0.0;
    sum += rbuf[pos];
    for (int k = 1; k < cnt; ++k, sh += 10) { pos += (int)((code >> sh) & 1023ull); sum += rbuf[pos]; }
    rc_out[a] = sum;
  }
}

// number of row blocks whose entry count exceeds each of 4 candidate LDS budgets
__global__ void plan_count_kernel(int n, const int *__restrict__ rowptr, int nblocks, int c0, int c1, int c2, int c3, int *__restrict__ out) {
  int vb = blockIdx.x * blockDim.x + threadIdx.x;
  if (vb >= nblocks) return;
  int r0 = vb * RB, r1 = min(r0 + RB, n);
  int cnt = rowptr[r1] - rowptr[r0];
  if (cnt > c0) atomicAdd(&out[0], 1);
  if (cnt > c1) atomicAdd(&out[1], 1);
  if (cnt > c2) atomicAdd(&out[2], 1);
  if (cnt > c3) atomicAdd(&out[3], 1);
}

__global__ void blkptr_kernel(int n, const int *__restrict__ rowptr, int nblocks, int *__restrict__ blkptr) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= nblocks) blkptr[b] = rowptr[min(b * RB, n)];
}

__global__ void plan_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col, int nblocks,
                            unsigned long long *__restrict__ farsum /* Σ_rows max |col−row| (owned columns) */, int *__restrict__ out /*[0]=max block nnz,[1]=max row len,[2]=max |col-row| over owned columns,
                                                    [3]=#row blocks touching halo columns,[4]=min block without halo,[5]=max block without halo*/) {
  int vb = blockIdx.x * blockDim.x + threadIdx.x;
  int mx = 0, mr = 0, far = 0;
  unsigned long long fsum = 0;
  if (vb < nblocks) {
    bool halo = false;
    int r0 = vb * RB, r1 = min(r0 + RB, n);
    mx = rowptr[r1] - rowptr[r0];
    for (int q = r0; q < r1; q += 64) atomicMax(&out[6], rowptr[min(q + 64, r1)] - rowptr[q]);
    for (int r = r0; r < r1; ++r) {
      const int a = rowptr[r], e = rowptr[r + 1];
      mr = max(mr, e - a);
      if (e > a) {   // sorted row: extremes are the first entry and the last owned entry
        int rf = max(0, r - col[a]);
        int k = e - 1;
        if (col[k] >= n) halo = true;
        while (k > a && col[k] >= n) --k;
        if (col[k] < n) rf = max(rf, col[k] - r);
        far = max(far, rf); fsum += (unsigned long long)rf;
      }
    }
    if (halo) atomicAdd(&out[3], 1);
    else { atomicMin(&out[4], vb); atomicMax(&out[5], vb); }
    atomicAdd(farsum, fsum);
  }
  for (int off = 32; off > 0; off >>= 1) { mx = max(mx, __shfl_down(mx, off)); mr = max(mr, __shfl_down(mr, off)); far = max(far, __shfl_down(far, off)); }
  if ((threadIdx.x & 63) == 0) { atomicMax(&out[0], mx); atomicMax(&out[1], mr); atomicMax(&out[2], far); }
}

template <int OP, bool NT, int CHUNK>
int launch_lanes(const mgs_csr *A, int lanes, dim3 grid, size_t lds, const double *x, const double *b,
                 const double *dinv, double omega, double *out, int cap, BlockMap bm) {
  hipStream_t s = A->ctx->stream;
#define L_(LN)                                                                                             \
  hipLaunchKernelGGL((csr_rowblock_kernel<OP, NT, LN, CHUNK>), grid, dim3(RB), lds, s, A->rows, A->rowptr, \
                     A->col, A->val, x, b, dinv, omega, out, cap, bm)
  switch (lanes) {
    case 4: L_(4); break;
    case 8: L_(8); break;
    case 16: L_(16); break;
    case 32: L_(32); break;
    default: L_(64); break;
  }
#undef L_
  return MGS_OK;
}
template <int OP, bool NT>
int launch_slice(const mgs_csr *A, int lanes, dim3 grid, const double *x, const double *b,
                 const double *dinv, double omega, double *out, int cap, BlockMap bm) {
  hipStream_t s = A->ctx->stream;
  const size_t lds = (size_t)(cap > 0 ? cap + 2 : 2) * 12 + 16 + (size_t)A->ctx->opt_lds_pad;   // opt_lds_pad: occupancy experiments
  // U = gathers per step of the row walk: the typical row length (7-point stencils: exactly one step)
  const double mean_len = A->rows ? (double)A->nnz / A->rows : 1.0;
  const int u = mean_len <= 4.5 ? 4 : (mean_len <= 7.5 && A->max_row_len <= 14 ? 7 : 8);
#define L_(LN)                                                                                                 \
  do {                                                                                                         \
    if (u == 4) hipLaunchKernelGGL((csr_rowblock_slice_kernel<OP, NT, LN, 4>), grid, dim3(RB), lds, s, A->rows, A->rowptr, A->col, A->val, x, b, dinv, omega, out, cap, bm, A->max_row_len <= 64 ? 1 : 0, A->ctx->opt_blkptr ? A->blkptr : nullptr, ((A->ctx->opt_nt_store > 0 && A->rows >= A->ctx->opt_nt_store) ? 1 : 0) | (A->ctx->opt_stage_unroll ? 2 : 0)); \
    else if (u == 7) hipLaunchKernelGGL((csr_rowblock_slice_kernel<OP, NT, LN, 7>), grid, dim3(RB), lds, s, A->rows, A->rowptr, A->col, A->val, x, b, dinv, omega, out, cap, bm, A->max_row_len <= 64 ? 1 : 0, A->ctx->opt_blkptr ? A->blkptr : nullptr, ((A->ctx->opt_nt_store > 0 && A->rows >= A->ctx->opt_nt_store) ? 1 : 0) | (A->ctx->opt_stage_unroll ? 2 : 0)); \
    else hipLaunchKernelGGL((csr_rowblock_slice_kernel<OP, NT, LN, 8>), grid, dim3(RB), lds, s, A->rows, A->rowptr, A->col, A->val, x, b, dinv, omega, out, cap, bm, A->max_row_len <= 64 ? 1 : 0, A->ctx->opt_blkptr ? A->blkptr : nullptr, ((A->ctx->opt_nt_store > 0 && A->rows >= A->ctx->opt_nt_store) ? 1 : 0) | (A->ctx->opt_stage_unroll ? 2 : 0)); \
  } while (0)
  switch (lanes) {
    case 4: L_(4); break;
    case 8: L_(8); break;
    case 16: L_(16); break;
    case 32: L_(32); break;
    default: L_(64); break;
  }
#undef L_
  return MGS_OK;
}
template <int OP, bool NT>
int launch_chunk(const mgs_csr *A, int chunk_elems, int lanes, dim3 grid, size_t lds, const double *x, const double *b,
                 const double *dinv, double omega, double *out, int cap, BlockMap bm) {
  switch (chunk_elems) {
    case 0: return launch_slice<OP, NT>(A, lanes, grid, x, b, dinv, omega, out, cap, bm);
    case 1: return launch_lanes<OP, NT, 1>(A, lanes, grid, lds, x, b, dinv, omega, out, cap, bm);
    case 2: return launch_lanes<OP, NT, 2>(A, lanes, grid, lds, x, b, dinv, omega, out, cap, bm);
    default: return launch_lanes<OP, NT, 4>(A, lanes, grid, lds, x, b, dinv, omega, out, cap, bm);
  }
}

}  // namespace

int mgs_plan_csr(mgs_csr *A) {
  mgs_ctx *ctx = A->ctx;
  A->max_row_len = 0;
  A->lds_cap = 0;
  if (A->rows == 0) return MGS_OK;
  int nblocks = (A->rows + RB - 1) / RB;
  if (A->blkptr) { mgs_hip_free(A->blkptr); A->blkptr = nullptr; }
  MGS_TRY(mgs_dev_alloc(ctx, &A->blkptr, (size_t)nblocks + 1));
  hipLaunchKernelGGL(blkptr_kernel, dim3((nblocks + 256) / 256), dim3(256), 0, ctx->stream, A->rows, A->rowptr, nblocks, A->blkptr);
  int *d = nullptr;
  MGS_TRY(mgs_dev_alloc(ctx, &d, 10));   // 7 ints + one 8-byte aligned 64-bit sum at d[8..9]
  const int init[10] = {0, 0, 0, 0, 0x7fffffff, -1, 0, 0, 0, 0};
  MGS_HIP(ctx, hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(plan_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, ctx->stream, A->rows, A->rowptr, A->col, nblocks,
                     reinterpret_cast<unsigned long long *>(d + 8), d);
  int h[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  MGS_HIP(ctx, hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MGS_HIP(ctx, mgs_hip_free(d));
  A->max_row_len = h[1];
  // typical far-band distance = mean over rows of the row's farthest owned column (the max is set by a few
  // odd-shaped boundary aggregates on coarse levels and would mis-size the strip-major sweep)
  unsigned long long fs; memcpy(&fs, &h[8], sizeof fs);
  A->far_band = A->rows ? (int)(fs / (unsigned long long)A->rows) : 0;
  A->far_band_max = h[2];
  // row blocks that read halo columns: usable for overlap when they form a prefix + suffix of the shard
  A->halo_lo_blocks = A->halo_hi_blocks = 0; A->halo_split_ok = false;
  if (A->cols > A->rows) {
    if (h[5] < 0) { A->halo_lo_blocks = nblocks; A->halo_split_ok = false; }          // every block touches the halo
    else {
      const int lo_cnt = h[4], hi_cnt = nblocks - 1 - h[5];
      A->halo_lo_blocks = lo_cnt; A->halo_hi_blocks = hi_cnt;
      A->halo_split_ok = (h[3] == lo_cnt + hi_cnt) && (nblocks - lo_cnt - hi_cnt) > 0;
    }
  }
  A->max_block_nnz = h[0];
  A->max_wave_nnz = h[6];
  A->lds_cap = h[0] < LDS_CAP_MAX ? h[0] : LDS_CAP_MAX;
  if (A->lds_cap < 64) A->lds_cap = 64;
  // LDS budget per workgroup sets the occupancy.  A handful of heavy row blocks (G0 fringes, irregular
  // aggregates on coarse levels) must not size it for everyone: take the smallest budget that still
  // stages ≥ 98.5 % of the row blocks; the rest run their rows from global memory (block-uniform branch).
  const double mean = (double)A->nnz / nblocks;
  if (nblocks >= 64 && (double)A->lds_cap > 1.12 * mean) {
    int cand[4] = {(int)(1.06 * mean) + 8, (int)(1.12 * mean) + 8, (int)(1.25 * mean) + 8, (int)(1.5 * mean) + 8};
    int *dc = nullptr;
    MGS_TRY(mgs_dev_alloc(ctx, &dc, 4));
    MGS_HIP(ctx, hipMemsetAsync(dc, 0, 4 * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(plan_count_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, ctx->stream, A->rows, A->rowptr, nblocks, cand[0], cand[1], cand[2], cand[3], dc);
    int over[4] = {0, 0, 0, 0};
    MGS_HIP(ctx, hipMemcpyAsync(over, dc, sizeof over, hipMemcpyDeviceToHost, ctx->stream));
    MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MGS_HIP(ctx, mgs_hip_free(dc));
    for (int q = 0; q < 4; ++q)
      if (cand[q] < A->lds_cap && over[q] <= 0.015 * nblocks) { A->lds_cap = cand[q]; break; }
  }
  return MGS_OK;
}

// Builds the pattern code of the index array `idx` (CSR-shaped like rowptr; base = nullptr: offsets from the row;
// indices >= split address the halo payload of a row shard and are coded as slot − row).
int mgs_build_rowcode(mgs_ctx *ctx, int n, const int *rowptr, const int *idx, const int *base, int split, mgs_rowcode **out,
                      const double *val) {
  *out = nullptr;
  if (n <= 0) return MGS_OK;
  const int nblocks = (n + RB - 1) / RB;
  mgs_rowcode *c = new mgs_rowcode();
  c->nblocks = nblocks;
  unsigned char *isrep = nullptr;
  int *ints = nullptr;
  int rc = mgs_dev_alloc(ctx, &c->pid, (size_t)n);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &isrep, (size_t)n);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &ints, (size_t)nblocks + 1);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &c->tptr, (size_t)nblocks + 1);
  std::vector<int> h((size_t)nblocks + 1, 0);
  if (rc == MGS_OK) {
    hipError_t e = hipMemsetAsync(ints, 0, sizeof(int) * ((size_t)nblocks + 1), ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(rowcode_assign_kernel, dim3(nblocks), dim3(RB), 0, ctx->stream, n, rowptr, idx, base, split, c->pid, isrep, ints, val);
      e = hipGetLastError();                       // launch errors (bad configuration) surface here, not at the later sync
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), ints, sizeof(int) * (size_t)nblocks, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "rowcode pass 1 failed: %s", hipGetErrorString(e));
  }
  if (rc == MGS_OK) {
    int64_t total = 0;
    std::vector<int> sizes;
    for (int q = 0; q < nblocks; ++q) { const int v = h[q]; h[q] = (int)total; total += v; if (v) { c->coded_blocks++; sizes.push_back(v); } }
    h[nblocks] = (int)total;
    c->tab_total = total;
    if (total >= 0x7fffffffLL) c->coded_blocks = 0;      // cannot index the table with 32 bits: leave the matrix uncoded
    if (c->coded_blocks) {
      // table budget in LDS: covers 98.5 % of the coded blocks (the rest walk their rows from global memory)
      std::sort(sizes.begin(), sizes.end());
      c->tab_max = sizes.back();
      c->tab_cap = sizes[(size_t)((sizes.size() - 1) * 0.985)];
      hipError_t e = hipMemcpyAsync(c->tptr, h.data(), sizeof(int) * ((size_t)nblocks + 1), hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "rowcode table offsets: %s", hipGetErrorString(e));
      if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &c->tab, (size_t)total + 4);
      if (rc == MGS_OK && val) rc = mgs_dev_alloc(ctx, &c->vtab, (size_t)total + 4);
      if (rc == MGS_OK) {
        hipLaunchKernelGGL(rowcode_fill_kernel, dim3(nblocks), dim3(RB), 0, ctx->stream, n, rowptr, idx, base, split, c->pid, isrep, c->tptr, c->tab, val, c->vtab);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "rowcode pass 2 failed: %s", hipGetErrorString(e));
      }
    }
  }
  if (isrep) mgs_hip_free(isrep);
  if (ints) mgs_hip_free(ints);
  if (rc != MGS_OK || !c->coded_blocks) { mgs_free_rowcode(c); c = nullptr; }
  *out = c;
  return rc;
}
void mgs_free_rowcode(mgs_rowcode *c) {
  if (!c) return;
  if (c->pid) mgs_hip_free(c->pid);
  if (c->tptr) mgs_hip_free(c->tptr);
  if (c->tab) mgs_hip_free(c->tab);
  if (c->vtab) mgs_hip_free(c->vtab);
  delete c;
}

// true when the coded kernel serves this operator (square or sharded; short rows; code present and worth it)
// (any = true: a row shard's operand passes — only this kernel knows the halo split, and on its uncoded blocks it is the
//  slice kernel: any code will do)
static bool use_rowcode(const mgs_csr *A, const mgs_rowcode *c, bool any = false) {
  return c && A->ctx->opt_rowcode && A->blkptr && A->lds_cap > 0 && A->max_row_len <= 64 &&
         (any || (double)c->coded_blocks >= 0.5 * c->nblocks);
}
// op ∈ {SPMV, RESIDUAL, JACOBI, FUSE_POST_MAPPED}; for the post pass: x = e_c, b = r, dinv = wd, xin = b, idx = agg[col]
static dim3 plan_group_map(const mgs_csr *A, const mgs_groups *G, BlockMap &bm);
static int launch_coded(const mgs_csr *A, const mgs_rowcode *c, int op, const int *idx, const double *x, const double *b, const double *dinv,
                        double omega, const double *xin, const int *agg, double *out, dim3 grid, BlockMap bm,
                        const double *hv = nullptr, int split = 0x7fffffff) {
  mgs_ctx *ctx = A->ctx;
  const int capv = A->lds_cap;
  // nearly everything coded: LDS holds values + tables only (8 B per entry → more workgroups per CU); otherwise
  // the integer region must also fit the index slice of the uncoded blocks
  const bool lean = (double)c->coded_blocks >= 0.985 * c->nblocks;
  const int capi = lean ? std::max(c->tab_cap, 64) : std::max(c->tab_cap, capv + 2);
  const size_t lds = (size_t)(capv + 2) * 8 + (size_t)capi * 4 + 16 + (size_t)ctx->opt_lds_pad;
  const double mean_len = A->rows ? (double)A->nnz / A->rows : 1.0;
  const int u = mean_len <= 4.5 ? 4 : (mean_len <= 7.5 && A->max_row_len <= 14 ? 7 : 8);
#define C_(O, UU, H, V) hipLaunchKernelGGL((csr_rowblock_coded_kernel<O, UU, H, V>), grid, dim3(RB), lds, ctx->stream, A->rows, A->rowptr, idx, A->val, \
                                           c->pid, c->tptr, c->tab, x, b, dinv, (O == FUSE_POST_MAPPED && A->dpos) ? A->dpos_omega : omega, xin, agg, out, capv, \
                                           (ctx->opt_nt_store > 0 && A->rows >= ctx->opt_nt_store) ? -capi : capi, bm, A->blkptr, hv, split, c->vtab, O == FUSE_POST_MAPPED ? A->dpos : nullptr, \
                                           O == MGS_OP_SPMV ? A->dot_w1 : nullptr, O == MGS_OP_SPMV ? A->dot_part : nullptr, (A->rows + RB - 1) / RB, (A->max_row_len <= UU ? 1 : 0) | (ctx->opt_stage_unroll ? 2 : 0) | (ctx->opt_rowptr_scan ? 4 : 0))
  // group sweep (views with A->sweep set; plain index codes, no halo): one workgroup per row-block group of the grouped pre pass
  BlockMap gbm; dim3 ggrid(1);
  const bool sweep = A->sweep && !hv && !c->vtab && op == FUSE_POST_MAPPED;
  if (sweep) ggrid = plan_group_map(A, A->sweep, gbm);
#define CG_(O, UU) hipLaunchKernelGGL((csr_rowblock_coded_group_kernel<O, UU, false, false>), ggrid, dim3(RB), lds, ctx->stream, A->rows, A->rowptr, idx, A->val, \
                                      c->pid, c->tptr, c->tab, x, b, dinv, (O == FUSE_POST_MAPPED && A->dpos) ? A->dpos_omega : omega, xin, agg, out, capv, \
                                      (ctx->opt_nt_store > 0 && A->rows >= ctx->opt_nt_store) ? -capi : capi, gbm, A->blkptr, hv, split, c->vtab, O == FUSE_POST_MAPPED ? A->dpos : nullptr, \
                                      nullptr, nullptr, (A->rows + RB - 1) / RB, (A->max_row_len <= UU ? 1 : 0) | (ctx->opt_stage_unroll ? 2 : 0) | (ctx->opt_rowptr_scan ? 4 : 0), A->sweep->gdesc)
#define CH_(O, UU) do { if (sweep && O == FUSE_POST_MAPPED) CG_(FUSE_POST_MAPPED, UU); \
                        else if (hv) { if (c->vtab) C_(O, UU, true, true); else C_(O, UU, true, false); } \
                        else { if (c->vtab) C_(O, UU, false, true); else C_(O, UU, false, false); } } while (0)
#define CU_(O) do { if (u == 4) CH_(O, 4); else if (u == 7) CH_(O, 7); else CH_(O, 8); } while (0)
  switch (op) {
    case MGS_OP_SPMV: CU_(MGS_OP_SPMV); break;
    case MGS_OP_RESIDUAL: CU_(MGS_OP_RESIDUAL); break;
    case MGS_OP_JACOBI: CU_(MGS_OP_JACOBI); break;
    default:
      // A·P of a 7-point operator with aggregates of four has 5 entries per row: one unrolled step of 5 instead of 7
      if (mean_len > 4.5 && mean_len <= 5.5 && A->max_row_len <= 10) CH_(FUSE_POST_MAPPED, 5); else CU_(FUSE_POST_MAPPED);
      break;
  }
#undef CU_
#undef CH_
#undef CG_
#undef C_
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}


// strip-major sweep of period D (row blocks or groups) with strips of S: installs D, S, P and the multiply-high constants of
// map_block_xi and returns the workgroups per XCD — or leaves the plain XCD-contiguous map (returns bm.chunk) where a quotient would
// not be exact by multiply-high (idx·divisor ≥ 2³²: never at the sizes 288 GB hold, checked all the same)
static int strip_map(BlockMap &bm, int D, int S) {
  const int P = (bm.chunk + D - 1) / D;
  const int per_xcd = ((D + S - 1) / S) * P * S;
  const unsigned long long ps = (unsigned long long)P * (unsigned long long)S;
  // (S < 2: ⌊2³²/1⌋+1 does not fit 32 bits — the multiply-high quotient would be 0 instead of the dividend; a strip of one block is
  // the plain plane-major order anyway, so the plain XCD-contiguous map serves)
  if (D <= 0 || S < 2 || ps < 2 || ((unsigned long long)per_xcd + 1) * ps >= (1ull << 32) || ps * (unsigned long long)S >= (1ull << 32)) { bm.D = bm.S = bm.P = 0; return bm.chunk; }
  bm.D = D; bm.S = S; bm.P = P;
  bm.mps = (unsigned)((1ull << 32) / ps) + 1u;
  bm.mS = (unsigned)((1ull << 32) / (unsigned long long)S) + 1u;
  return per_xcd;
}
// workgroup → group map: XCD-contiguous, strip-major for far bands (distances in groups instead of row blocks)
static dim3 plan_group_map(const mgs_csr *A, const mgs_groups *G, BlockMap &bm) {
  mgs_ctx *ctx = A->ctx;
  bm.base = 0; bm.nblocks = G->ngroups;
  bm.remap = ctx->opt_xcd_remap && bm.nblocks >= 64;
  bm.chunk = (bm.nblocks + 7) / 8;
  bm.D = 0; bm.S = 0; bm.P = 0;
  int per_xcd = bm.chunk;
  if (bm.remap && ctx->opt_strip != 0) {
    const int Db = (A->far_band + RB - 1) / RB;
    const int D = G->plane_groups;          // groups from one far-band period to the next (setup: first blocks Db apart)
    if (Db >= 512 && D >= 64 && bm.chunk >= 2 * D) {
      per_xcd = strip_map(bm, D, ctx->opt_group_strip > 0 ? ctx->opt_group_strip : (ctx->opt_strip > 0 ? std::max(1, ctx->opt_strip / 2) : 128));
      // strip of 128 groups by default: 512³, same-process sweep 5.51 / 5.46 / 5.42 / 5.40 / 5.40 / 5.51 ms per cycle for 2 / 16 / 32 / 64 / 128 / 512 groups
    }
  }
  const dim3 grid(bm.remap ? per_xcd * 8 : bm.nblocks);
  return grid;
}

void mgs_free_groups(mgs_groups *g) {
  if (!g) return;
  if (g->gblk) mgs_hip_free(g->gblk); if (g->afirst) mgs_hip_free(g->afirst); if (g->acode) mgs_hip_free(g->acode); if (g->gdesc) mgs_hip_free(g->gdesc);
  if (g->wmask) mgs_hip_free(g->wmask); if (g->stray) mgs_hip_free(g->stray); if (g->gorder) mgs_hip_free(g->gorder);
  delete g;
}
// Pairs the row blocks of A along its aggregates (see csr_group_pre_kernel).  *out stays NULL when the level does not
// qualify: general P, aggregate ids not ascending with their first member, long rows, or more than opt_group_stray_pct % strays.
int mgs_build_groups(mgs_ctx *ctx, const mgs_csr *A, const mgs_xfer *T, mgs_groups **out) {
  *out = nullptr;
  const int n = A->rows, nc = T ? T->n_coarse : 0;
  if (!T || !T->aggregation || n <= 0 || nc <= 0 || A->max_row_len > 64 || !A->blkptr || A->lds_cap <= 0) return MGS_OK;
  const int nblocks = (n + RB - 1) / RB;
  if (nblocks < ctx->opt_group_min_blocks) return MGS_OK;      // small levels: too few workgroups once blocks are grouped
  hipStream_t st = ctx->stream;
  mgs_groups *G = new mgs_groups();
  G->nblocks = nblocks;
  int *vkey = nullptr, *vcnt = nullptr, *flags = nullptr, *blk2grp = nullptr;
  const size_t nv = (size_t)nblocks * GRP_SLOTS;
  int rc = mgs_dev_alloc(ctx, &vkey, nv);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &vcnt, nv);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &flags, 2);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &G->afirst, (size_t)nblocks + 1);
  std::vector<int> hk(nv), hn(nv), hg, hb2g((size_t)nblocks, -1);
  int hflags[2] = {0, 0};
  if (rc == MGS_OK) {
    hipMemsetAsync(vkey, 0xff, sizeof(int) * nv, st);                  // −1: empty slot
    hipMemsetAsync(vcnt, 0, sizeof(int) * nv, st);
    hipMemsetAsync(flags, 0, 2 * sizeof(int), st);
    hipLaunchKernelGGL(grp_scan_kernel, dim3((nc + RB - 1) / RB), dim3(RB), 0, st, nc, T->cptr, T->members, nblocks, vkey, vcnt, G->afirst, flags);
    hipMemcpyAsync(hk.data(), vkey, sizeof(int) * nv, hipMemcpyDeviceToHost, st);
    hipMemcpyAsync(hn.data(), vcnt, sizeof(int) * nv, hipMemcpyDeviceToHost, st);
    hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "row-block grouping (scan) failed");
  }
  bool usable = rc == MGS_OK && hflags[0] == 0;
  if (usable) {
    // greedy union of row blocks along their most frequent links, groups of at most GRP_BLOCKS blocks (deterministic: links
    // sorted by count, then by block pair)
    struct Link { int cnt, a, b; };
    std::vector<Link> raw, links;
    for (int b = 0; b < nblocks; ++b) {
      Link mine[GRP_SLOTS]; int nm = 0;
      for (int sl = 0; sl < GRP_SLOTS; ++sl) {
        const int k = hk[(size_t)b * GRP_SLOTS + sl], c = hn[(size_t)b * GRP_SLOTS + sl];
        // option group_min_link: only links that carry a share of the block's aggregates (default 1 = every link; see mgs_internal.hpp)
        if (k > b && k < nblocks && c >= ctx->opt_group_min_link) mine[nm++] = {c, b, k};
      }
      std::sort(mine, mine + nm, [](const Link &p, const Link &q) { return p.b < q.b; });
      for (int q = 0; q < nm; ++q) raw.push_back(mine[q]);
    }
    {   // order: count descending, then (a, b) ascending — raw is already (a, b)-ascending, so a stable counting sort by count does it
        // (a count is at most the 256 aggregates that can start in one row block; half a million blocks at 512³: no comparison sort)
      std::vector<size_t> start((size_t)RB + 2, 0);
      for (const Link &l : raw) ++start[(size_t)(RB - std::min(l.cnt, RB)) + 1];
      for (size_t q = 1; q < start.size(); ++q) start[q] += start[q - 1];
      links.resize(raw.size());
      for (const Link &l : raw) links[start[(size_t)(RB - std::min(l.cnt, RB))]++] = l;
    }
    std::vector<int> parent((size_t)nblocks), gsize((size_t)nblocks, 1);
    for (int b = 0; b < nblocks; ++b) parent[b] = b;
    auto find = [&](int v) { while (parent[v] != v) { parent[v] = parent[parent[v]]; v = parent[v]; } return v; };
    const int max_blocks = std::min(std::max(ctx->opt_group_blocks, 1), GRP_BLOCKS);
    for (const Link &l : links) {
      int ra = find(l.a), rb = find(l.b);
      if (ra == rb || gsize[ra] + gsize[rb] > max_blocks) continue;
      if (ra > rb) std::swap(ra, rb);
      parent[rb] = ra; gsize[ra] += gsize[rb];                              // root = smallest block of the group
    }
    // groups in ascending order of their smallest block; blocks inside a group ascending
    std::vector<int> gid((size_t)nblocks, -1);
    int ng = 0;
    for (int b = 0; b < nblocks; ++b) if (find(b) == b) gid[b] = ng++;
    hg.assign((size_t)ng * GRP_BLOCKS, -1);
    std::vector<int> fill((size_t)ng, 0);
    for (int b = 0; b < nblocks; ++b) { const int g = gid[find(b)]; hb2g[b] = g; hg[(size_t)g * GRP_BLOCKS + fill[g]++] = b; }
    G->ngroups = ng;
    {   // distance, in groups, between a group and the one whose first block lies one far-band period further (middle of the matrix)
      const int Db = (A->far_band + RB - 1) / RB, gm = ng / 2, fb = hg[(size_t)gm * GRP_BLOCKS];
      int lo_g = gm, hi_g = ng;
      while (lo_g < hi_g) { const int mid = (lo_g + hi_g) / 2; if (hg[(size_t)mid * GRP_BLOCKS] < fb + Db) lo_g = mid + 1; else hi_g = mid; }
      G->plane_groups = lo_g - gm;
    }
    const int stray_cap = std::max((int)((int64_t)nc * std::min(std::max(ctx->opt_group_stray_pct, 0), 100) / 100), 1);
    rc = mgs_dev_alloc(ctx, &G->gblk, hg.size());
    if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &blk2grp, (size_t)nblocks);
    if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &G->acode, (size_t)nc);
    if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &G->wmask, (size_t)(n + 31) / 32 + 1);
    if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &G->stray, (size_t)stray_cap);
    if (rc == MGS_OK) {
      hipMemcpyAsync(G->gblk, hg.data(), sizeof(int) * hg.size(), hipMemcpyHostToDevice, st);
      hipMemcpyAsync(blk2grp, hb2g.data(), sizeof(int) * (size_t)nblocks, hipMemcpyHostToDevice, st);
      hipMemsetAsync(G->wmask, 0, sizeof(unsigned) * ((size_t)(n + 31) / 32 + 1), st);
      hipLaunchKernelGGL(grp_code_kernel, dim3((nc + RB - 1) / RB), dim3(RB), 0, st, nc, T->cptr, T->members, blk2grp, G->gblk, G->acode, G->wmask,
                         G->stray, stray_cap, flags + 1);
      hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, st);
      if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "row-block grouping (codes) failed");
    }
    G->nstray = hflags[1];
    usable = rc == MGS_OK && G->nstray <= stray_cap;
    if (usable) {      // group descriptors: everything a workgroup needs about its blocks in one record
      std::vector<int> hbp((size_t)nblocks + 1), haf((size_t)nblocks + 1), hd((size_t)ng * GRP_DESC, -1);
      hipMemcpyAsync(hbp.data(), A->blkptr, sizeof(int) * ((size_t)nblocks + 1), hipMemcpyDeviceToHost, st);
      hipMemcpyAsync(haf.data(), G->afirst, sizeof(int) * ((size_t)nblocks + 1), hipMemcpyDeviceToHost, st);
      if (hipStreamSynchronize(st) != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "row-block grouping (descriptors) failed");
      int maxb = 1;
      for (int g = 0; g < ng && rc == MGS_OK; ++g)
        for (int q = 0; q < GRP_BLOCKS; ++q) {
          const int b = hg[(size_t)g * GRP_BLOCKS + q];
          int *d = &hd[(size_t)g * GRP_DESC];
          d[q] = b; d[4 + q] = b >= 0 ? hbp[b] : 0; d[8 + q] = b >= 0 ? hbp[b + 1] : 0; d[12 + q] = b >= 0 ? haf[b] : 0; d[16 + q] = b >= 0 ? haf[b + 1] : 0;
          if (b >= 0) maxb = std::max(maxb, q + 1);
        }
      G->max_blocks = maxb;
      // option group_order: the groups in the order in which the one-block kernels sweep their FIRST blocks — XCD-contiguous ranges of row
      // blocks, inside a range strip-major: a strip of S row blocks followed through every plane (period Db blocks) of the range
      const int Db = (A->far_band + RB - 1) / RB, chunkB = (nblocks + 7) / 8, S = ctx->opt_group_order;
      if (rc == MGS_OK && S > 0 && Db >= 512 && chunkB >= 2 * Db) {
        std::vector<std::vector<std::pair<long long, int>>> per(8);
        for (int g = 0; g < ng; ++g) {
          const int fb = hg[(size_t)g * GRP_BLOCKS], xcd = std::min(fb / chunkB, 7), lb = fb - xcd * chunkB;
          const long long p = lb / Db, off = lb % Db, st = off / S, t = off % S;
          per[(size_t)xcd].push_back({(st * (long long)(chunkB / Db + 2) + p) * S + t, g});
        }
        size_t mx = 0;
        for (auto &v : per) { std::sort(v.begin(), v.end()); mx = std::max(mx, v.size()); }
        std::vector<int> ho(8 * mx, -1);
        for (int x = 0; x < 8; ++x) for (size_t q = 0; q < per[(size_t)x].size(); ++q) ho[(size_t)x * mx + q] = per[(size_t)x][q].second;
        rc = mgs_dev_alloc(ctx, &G->gorder, ho.size());
        if (rc == MGS_OK && hipMemcpy(G->gorder, ho.data(), sizeof(int) * ho.size(), hipMemcpyHostToDevice) != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "row-block grouping: order table upload failed");
        G->gorder_per_xcd = (int)mx;
      }
      if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &G->gdesc, hd.size());
      if (rc == MGS_OK && hipMemcpy(G->gdesc, hd.data(), sizeof(int) * hd.size(), hipMemcpyHostToDevice) != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "row-block grouping: descriptor upload failed");
      usable = rc == MGS_OK;
    }
  }
  if (vkey) mgs_hip_free(vkey); if (vcnt) mgs_hip_free(vcnt); if (flags) mgs_hip_free(flags); if (blk2grp) mgs_hip_free(blk2grp);
  if (getenv("MGS_DEBUG_GROUPS"))
    fprintf(stderr, "[mgs groups] n=%d nc=%d blocks=%d: ids ascending=%d groups=%d strays=%d (limit %d%%) -> %s\n", n, nc, nblocks, hflags[0] == 0,
            G->ngroups, G->nstray, ctx->opt_group_stray_pct, usable ? "grouped" : "separate kernels");
  if (!usable) { mgs_free_groups(G); return rc; }
  *out = G;
  return MGS_OK;
}

int mgs_launch_group_pre(const mgs_csr *A, const mgs_groups *G, const mgs_xfer *T, const double *x, const double *b,
                         double *t_out, double *r_out, double *rc_out, const double *hv, int split) {
  mgs_ctx *ctx = A->ctx;
  if (A->rows == 0 || G->ngroups == 0) return MGS_OK;
  const mgs_rowcode *c = (use_rowcode(A, A->code, hv != nullptr) && !A->code->vtab) ? A->code : nullptr;
  const int capv = A->lds_cap;
  const bool lean = c && (double)c->coded_blocks >= 0.985 * c->nblocks;
  const int capi = lean ? std::max(c->tab_cap, 64) : std::max(c ? c->tab_cap : 0, capv + 2);
  const size_t lds = (size_t)(capv + 2) * 8 + (size_t)((capi + 1) / 2) * 8 + (size_t)G->max_blocks * RB * 8 + 16 + (size_t)ctx->opt_lds_pad;
  BlockMap bm;
  dim3 grid = plan_group_map(A, G, bm);
  const double mean_len = A->rows ? (double)A->nnz / A->rows : 1.0;
  const int u = mean_len <= 4.5 ? 4 : (mean_len <= 7.5 && A->max_row_len <= 14 ? 7 : 8);
  const bool pairs = G->max_blocks <= 2 && ctx->opt_group_concurrent;     // 512 threads, both blocks of a pair at once
  const bool ordered = !pairs && G->gorder && ctx->opt_group_order > 0 && ctx->opt_xcd_remap;
  if (ordered) grid = dim3(8 * G->gorder_per_xcd);
  const size_t lds2 = (size_t)2 * ((size_t)(capv + 2) * 8 + (size_t)((capi + 1) / 2) * 8) + (size_t)2 * RB * 8 + 16 + (size_t)ctx->opt_lds_pad;
#define G2_(UU, H) hipLaunchKernelGGL((csr_group2_pre_kernel<UU, H>), grid, dim3(2 * RB), lds2, ctx->stream, A->rows, A->rowptr, A->col, A->val, \
                                      c ? c->pid : nullptr, c ? c->tptr : nullptr, c ? c->tab : nullptr, x, b, t_out, r_out, rc_out, G->gdesc, \
                                      G->acode, G->wmask, capv, capi, bm, hv, hv ? split : 0x7fffffff)
#define G_(UU, H) hipLaunchKernelGGL((csr_group_pre_kernel<UU, H>), grid, dim3(RB), lds, ctx->stream, A->rows, A->rowptr, A->col, A->val, \
                                     c ? c->pid : nullptr, c ? c->tptr : nullptr, c ? c->tab : nullptr, x, b, t_out, r_out, rc_out, G->gdesc, \
                                     G->acode, G->wmask, capv, capi, bm, hv, hv ? split : 0x7fffffff, ((ctx->opt_nt_store > 0 && A->rows >= ctx->opt_nt_store) ? 1 : 0) | (ctx->opt_stage_unroll ? 2 : 0) | (ctx->opt_rowptr_scan ? 4 : 0), \
                                     ordered ? G->gorder : nullptr, G->gorder_per_xcd)
#define GU_(UU) do { if (pairs) { if (hv) G2_(UU, true); else G2_(UU, false); } else { if (hv) G_(UU, true); else G_(UU, false); } } while (0)
  if (u == 4) GU_(4); else if (u == 7) GU_(7); else GU_(8);
#undef GU_
#undef G_
#undef G2_
  MGS_HIP(ctx, hipGetLastError());
  if (G->nstray) {
    hipLaunchKernelGGL(restrict_stray_kernel, dim3((G->nstray + RB - 1) / RB), dim3(RB), 0, ctx->stream, G->nstray, G->stray, T->cptr, T->members, r_out, rc_out);
    MGS_HIP(ctx, hipGetLastError());
  }
  return MGS_OK;
}

int mgs_spmv_dots(const mgs_csr *A, const double *x, double *y, const double *w1, double *out_host2) {
  mgs_ctx *ctx = A->ctx;
  const int nb = (A->rows + RB - 1) / RB;
  const bool fused = ctx->opt_fuse_dots && A->rows > 0 && ctx->opt_spmv_variant == 0 && !ctx->opt_nontemporal && use_rowcode(A, A->code) && A->lds_cap >= 8;
  if (!fused) {
    MGS_TRY(mgs_launch_csr_op(A, MGS_OP_SPMV, x, nullptr, nullptr, 0.0, y));
    return k_dot2(ctx, A->rows, y, w1, y, y, out_host2);
  }
  MGS_TRY(mgs_ensure_dot_part(ctx, 2 * (int64_t)nb));
  mgs_csr V = *A; V.owns = false; V.dot_w1 = w1; V.dot_part = ctx->dot_part;     // every row block of the launch writes its pair
  MGS_TRY(mgs_launch_csr_op(&V, MGS_OP_SPMV, x, nullptr, nullptr, 0.0, y));
  return k_dot2_finish(ctx, nb, ctx->dot_part, out_host2);
}

bool mgs_rowcode_usable(const mgs_csr *A, bool any) { return use_rowcode(A, A->code, any); }

// workgroup → row-block map of a launch over the row blocks [blk_lo, blk_hi) (XCD-contiguous, strip-major for far bands)
static dim3 plan_block_map(const mgs_csr *A, int blk_lo, int blk_hi, BlockMap &bm) {
  mgs_ctx *ctx = A->ctx;
  bm.base = blk_lo; bm.nblocks = blk_hi - blk_lo;
  bm.remap = ctx->opt_xcd_remap && bm.nblocks >= 64;
  bm.chunk = (bm.nblocks + 7) / 8;
  bm.D = 0; bm.S = 0; bm.P = 0;
  int per_xcd = bm.chunk;
  if (bm.remap && ctx->opt_strip != 0) {
    const int D = (A->far_band + RB - 1) / RB;
    if (D >= 512 && bm.chunk >= 2 * D) {
      per_xcd = strip_map(bm, D, ctx->opt_strip > 0 ? ctx->opt_strip : 64);
    }
  }
  const dim3 grid(bm.remap ? per_xcd * 8 : bm.nblocks);
  return grid;
}

// The coded kernel on the row blocks [blk_lo, blk_hi) of the view A (A->col = the coded index array, A->code its
// code); hv/split: halo payload of a row shard (nullptr / INT_MAX: none).  Caller checks mgs_rowcode_usable(A).
int mgs_launch_coded_range(const mgs_csr *A, int op, const double *x, const double *b, const double *dinv, double omega,
                           const double *xin, const int *agg, double *out, const double *hv, int split, int blk_lo, int blk_hi,
                           int gap_at, int gap_len) {
  if (A->rows == 0 || blk_hi <= blk_lo) return MGS_OK;
  if (!use_rowcode(A, A->code, hv != nullptr)) return MGS_ERR_STATE;
  BlockMap bm;
  const dim3 grid = plan_block_map(A, blk_lo, blk_hi, bm);
  bm.gap_at = gap_at; bm.gap_len = gap_len;
  return launch_coded(A, A->code, op, A->col, x, b, dinv, omega, xin, agg, out, grid, bm, hv, hv ? split : 0x7fffffff);
}

int mgs_launch_csr_op(const mgs_csr *A, int op, const double *x, const double *b, const double *dinv,
                      double omega, double *out) {
  return mgs_launch_csr_op_range(A, op, x, b, dinv, omega, out, 0, (A->rows + RB - 1) / RB);
}

// fused passes (see csr_rowblock_fused_kernel); returns MGS_ERR_STATE when the level cannot use them
int mgs_launch_fused(const mgs_csr *A, int which, const double *wd, const double *bvec, const double *xin, const int *agg,
                     const double *ec, double *out, double *out2) {
  return mgs_launch_fused_range(A, which, wd, bvec, xin, agg, ec, out, out2, nullptr, 0, (A->rows + RB - 1) / RB);
}
// row blocks [blk_lo, blk_hi); hv = values of the halo columns (nullptr for a square operator)
int mgs_launch_fused_range(const mgs_csr *A, int which, const double *wd, const double *bvec, const double *xin, const int *agg,
                           const double *ec, double *out, double *out2, const double *hv, int blk_lo, int blk_hi, int gap_at, int gap_len) {
  mgs_ctx *ctx = A->ctx;
  if (A->rows == 0 || blk_hi <= blk_lo) return MGS_OK;
  if (A->lds_cap <= 0 || (which == FUSE_PRE && A->rows != A->cols && !hv)) return MGS_ERR_STATE;   // (row shards: the pre pass reads its halo columns from the payload; the post passes gather e_c, halo room included)
  BlockMap bm;
  bm.gap_at = gap_at; bm.gap_len = gap_len;
  bm.base = blk_lo; bm.nblocks = blk_hi - blk_lo;
  bm.remap = ctx->opt_xcd_remap && bm.nblocks >= 64;
  bm.chunk = (bm.nblocks + 7) / 8;
  bm.D = 0; bm.S = 0; bm.P = 0;
  int per_xcd = bm.chunk;
  if (bm.remap && ctx->opt_strip != 0) {
    const int D = (A->far_band + RB - 1) / RB;
    if (D >= 512 && bm.chunk >= 2 * D) {
      per_xcd = strip_map(bm, D, ctx->opt_strip > 0 ? ctx->opt_strip : 64);
    }
  }
  dim3 grid(bm.remap ? per_xcd * 8 : bm.nblocks);
  const int cap = A->lds_cap;
  const size_t lds = (size_t)(cap + 2) * 12 + 16;
  if (which == FUSE_POST_MAPPED && use_rowcode(A, A->code))      // A is the view whose col/code are the aggregate-mapped ones
    return launch_coded(A, A->code, FUSE_POST_MAPPED, A->col, ec, bvec, wd, 0.0, xin, agg, out, grid, bm);
  if (which == FUSE_POST_MAPPED) hipLaunchKernelGGL((csr_rowblock_fused_kernel<FUSE_POST_MAPPED>), grid, dim3(RB), lds, ctx->stream, A->rows, A->rowptr, A->col, A->val, wd, bvec, xin, agg, ec, out, out2, cap, bm, hv, A->ctx->opt_blkptr ? A->blkptr : nullptr);
  else if (which == FUSE_PRE) hipLaunchKernelGGL((csr_rowblock_fused_kernel<FUSE_PRE>), grid, dim3(RB), lds, ctx->stream, A->rows, A->rowptr, A->col, A->val, wd, bvec, xin, agg, ec, out, out2, cap, bm, hv, A->ctx->opt_blkptr ? A->blkptr : nullptr);
  else hipLaunchKernelGGL((csr_rowblock_fused_kernel<FUSE_POST>), grid, dim3(RB), lds, ctx->stream, A->rows, A->rowptr, A->col, A->val, wd, bvec, xin, agg, ec, out, out2, cap, bm, hv, A->ctx->opt_blkptr ? A->blkptr : nullptr);
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}

// the same kernels on the row blocks [blk_lo, blk_hi) only (interior / boundary split of a row shard)
int mgs_launch_csr_op_range(const mgs_csr *A, int op, const double *x, const double *b, const double *dinv,
                            double omega, double *out, int blk_lo, int blk_hi, int gap_at, int gap_len) {
  mgs_ctx *ctx = A->ctx;
  if (A->rows == 0 || blk_hi <= blk_lo) return MGS_OK;
  if (gap_len > 0 && ctx->opt_spmv_variant != 0 && ctx->opt_spmv_variant != 5) {   // experimental variants: no gap support
    MGS_TRY(mgs_launch_csr_op_range(A, op, x, b, dinv, omega, out, blk_lo, gap_at, 0x7fffffff, 0));
    return mgs_launch_csr_op_range(A, op, x, b, dinv, omega, out, gap_at + gap_len, blk_hi + gap_len, 0x7fffffff, 0);
  }
  BlockMap bm;
  bm.gap_at = gap_at; bm.gap_len = gap_len;
  bm.base = blk_lo;
  bm.nblocks = blk_hi - blk_lo;
  bm.remap = ctx->opt_xcd_remap && bm.nblocks >= 64;
  bm.chunk = (bm.nblocks + 7) / 8;
  bm.D = 0; bm.S = 0; bm.P = 0;
  int per_xcd = bm.chunk;
  // strip-major sweep when three x planes (3·far·8 B) overflow an XCD's L2 share and the XCD's
  // range holds at least two planes
  if (bm.remap && ctx->opt_strip != 0) {
    const int D = (A->far_band + RB - 1) / RB;
    if (D >= 512 && bm.chunk >= 2 * D) {
      per_xcd = strip_map(bm, D, ctx->opt_strip > 0 ? ctx->opt_strip : 64);
    }
  }
  int cap = ctx->opt_spmv_variant == 1 ? -1 : A->lds_cap;
  if (ctx->opt_spmv_variant == 6 && A->max_block_nnz <= 2048 && A->max_block_nnz == A->lds_cap) {
    // pipelined variant: G row blocks per workgroup; block map over super blocks
    constexpr int G = 4;
    const int nrb = bm.nblocks;
    BlockMap sm = bm;
    sm.nblocks = (nrb + G - 1) / G;
    sm.remap = ctx->opt_xcd_remap && sm.nblocks >= 64;
    sm.chunk = (sm.nblocks + 7) / 8;
    sm.D = 0; sm.S = 0; sm.P = 0;
    int per = sm.chunk;
    if (sm.remap && ctx->opt_strip != 0) {
      const int D = ((A->far_band + RB - 1) / RB + G - 1) / G;
      if (D >= 64 && sm.chunk >= 2 * D) {
        per = strip_map(sm, D, ctx->opt_strip > 0 ? std::max(1, ctx->opt_strip / G) : 16);
      }
    }
    // base stays in row-block units inside the kernel: bm.base + first + j
    dim3 g(sm.remap ? per * 8 : sm.nblocks);
    const size_t lds = (size_t)(cap + 2) * 12 + 16;
    hipStream_t st = ctx->stream;
    const bool ntp = ctx->opt_nontemporal != 0;
#define P_(O, NTV) hipLaunchKernelGGL((csr_rowblock_pipe_kernel<O, NTV, 4, G>), g, dim3(RB), lds, st, A->rows, A->rowptr, A->col, A->val, x, b, dinv, omega, out, cap, sm, nrb)
    if (op == MGS_OP_SPMV) { if (ntp) P_(MGS_OP_SPMV, true); else P_(MGS_OP_SPMV, false); }
    else if (op == MGS_OP_RESIDUAL) { if (ntp) P_(MGS_OP_RESIDUAL, true); else P_(MGS_OP_RESIDUAL, false); }
    else { if (ntp) P_(MGS_OP_JACOBI, true); else P_(MGS_OP_JACOBI, false); }
#undef P_
    MGS_HIP(ctx, hipGetLastError());
    return MGS_OK;
  }
  if (ctx->opt_spmv_variant == 7 && A->max_wave_nnz <= 4096) {
    const int capw = A->max_wave_nnz;
    dim3 g((bm.remap ? per_xcd * 8 : bm.nblocks) * 4);
    const size_t ldsw = (size_t)(capw + 2) * 12 + 16;
    hipStream_t st = ctx->stream;
#define W_(O) hipLaunchKernelGGL((csr_wave_slice_kernel<O>), g, dim3(64), ldsw, st, A->rows, A->rowptr, A->col, A->val, x, b, dinv, omega, out, capw, bm)
    if (op == MGS_OP_SPMV) W_(MGS_OP_SPMV); else if (op == MGS_OP_RESIDUAL) W_(MGS_OP_RESIDUAL); else W_(MGS_OP_JACOBI);
#undef W_
    MGS_HIP(ctx, hipGetLastError());
    return MGS_OK;
  }
  dim3 grid(bm.remap ? per_xcd * 8 : bm.nblocks);
  if (ctx->opt_spmv_variant == 0 && !ctx->opt_nontemporal && use_rowcode(A, A->code))
    return launch_coded(A, A->code, op, A->col, x, b, dinv, omega, nullptr, nullptr, out, grid, bm);
  size_t lds = sizeof(double) * (size_t)(cap > 0 ? cap : 1);
  // lanes per row of the long-row path: next power of two ≥ mean row length, in [4,64]
  double mean = A->rows ? (double)A->nnz / A->rows : 1.0;
  int lanes = 4;
  while (lanes < 64 && lanes < mean) lanes <<= 1;
  const bool nt = ctx->opt_nontemporal > 0;
  // variants: 0 auto, 1 forced sub-wavefront rows, 2/3/4 = products-in-LDS with 1/2/4 entries per lane, 5 = slice-in-LDS
  int chunk_elems = 0;
  switch (ctx->opt_spmv_variant) { case 2: chunk_elems = 1; break; case 3: chunk_elems = 2; break; case 4: chunk_elems = 4; break; default: chunk_elems = 0; }
#define OP_(O)                                                                                                   \
  (nt ? launch_chunk<O, true>(A, chunk_elems, lanes, grid, lds, x, b, dinv, omega, out, cap, bm)                 \
      : launch_chunk<O, false>(A, chunk_elems, lanes, grid, lds, x, b, dinv, omega, out, cap, bm))
  switch (op) {
    case MGS_OP_SPMV: OP_(MGS_OP_SPMV); break;
    case MGS_OP_RESIDUAL: OP_(MGS_OP_RESIDUAL); break;
    case MGS_OP_JACOBI: OP_(MGS_OP_JACOBI); break;
    default: return mgs_fail(ctx, MGS_ERR_INVALID, "unknown csr op %d", op);
  }
#undef OP_
  MGS_HIP(ctx, hipGetLastError());
  return MGS_OK;
}
