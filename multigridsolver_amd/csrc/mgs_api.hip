// mgs_api.hip — C-ABI implementation (include/mgs.h): context, device containers, the
// multilevel hierarchy and the host-side V-cycle / BiCGSTAB orchestration.  The host logic
// is C++; every numeric step is a HIP kernel on the context's stream.  No CPU fallback.
#include "mgs_internal.hpp"

#include <cmath>

thread_local std::string g_mgs_last_error;

// ------------------------------------------------------------------ device memory arena (see mgs_internal.hpp)
#include <map>
#include <mutex>
namespace {
struct Arena {
  char *base = nullptr; size_t cap = 0; bool tried = false;
  int device = -1;                          // the arena lives on ONE device (the one current when it was taken) and serves only that one
  std::map<size_t, size_t> free_blocks;     // offset → size, address ordered
  std::map<size_t, size_t> used;            // offset → size
  std::mutex m;
} g_arena;
bool arena_take_locked(size_t cap) {
  void *p = nullptr;
  if (hipMalloc(&p, cap) != hipSuccess) { (void)hipGetLastError(); return false; }
  g_arena.base = (char *)p; g_arena.cap = cap; g_arena.free_blocks[0] = cap;
  if (hipGetDevice(&g_arena.device) != hipSuccess) g_arena.device = -1;
  return true;
}
void arena_init_locked() {
  if (g_arena.tried) return;
  g_arena.tried = true;
  const char *e = getenv("MGS_ARENA_GB");
  const double gb = e ? atof(e) : 0.0;
  if (gb <= 0.0) return;
  if (!arena_take_locked((size_t)(gb * 1024.0) << 20)) fprintf(stderr, "[mgs] arena of %.1f GiB not available: plain hipMalloc\n", gb);
}
}  // namespace
// One arena per process, on the device current at the reservation: a context on another device of the same process allocates with
// plain hipMalloc (its operators must not land in a peer's memory).  Reserved before the library's first device allocation (else MGS_ARENA_GB decides at that allocation).
extern "C" int mgs_arena_reserve(size_t bytes) {
  std::lock_guard<std::mutex> lk(g_arena.m);
  if (g_arena.base) return mgs_fail(nullptr, MGS_ERR_STATE, "mgs_arena_reserve: an arena of %zu bytes exists already", g_arena.cap);
  g_arena.tried = true;
  if (bytes == 0) return MGS_OK;                                   // explicit "no arena", whatever the environment says
  if (!arena_take_locked(bytes)) return mgs_fail(nullptr, MGS_ERR_ALLOC, "mgs_arena_reserve: hipMalloc(%zu bytes) failed", bytes);
  return MGS_OK;
}
extern "C" int mgs_arena_info(size_t out[3]) {                     // capacity, bytes in use, largest free block
  std::lock_guard<std::mutex> lk(g_arena.m);
  size_t used = 0, big = 0;
  for (auto &u : g_arena.used) used += u.second;
  for (auto &f : g_arena.free_blocks) big = std::max(big, f.second);
  out[0] = g_arena.cap; out[1] = used; out[2] = big;
  return MGS_OK;
}
hipError_t mgs_hip_malloc(void **p, size_t bytes) {
  {
    std::lock_guard<std::mutex> lk(g_arena.m);
    arena_init_locked();
    int cur = -1;
    if (g_arena.base && hipGetDevice(&cur) == hipSuccess && cur == g_arena.device) {
      const size_t align = bytes >= ((size_t)1 << 20) ? ((size_t)2 << 20) : (size_t)4096;
      const size_t need = (std::max(bytes, (size_t)1) + align - 1) / align * align;
      for (auto it = g_arena.free_blocks.begin(); it != g_arena.free_blocks.end(); ++it) {
        const size_t off = (it->first + align - 1) / align * align, pad = off - it->first;
        if (it->second < pad + need) continue;
        const size_t bo = it->first, bs = it->second;
        g_arena.free_blocks.erase(it);
        if (pad) g_arena.free_blocks[bo] = pad;
        if (bs > pad + need) g_arena.free_blocks[off + need] = bs - pad - need;
        g_arena.used[off] = need;
        *p = g_arena.base + off;
        return hipSuccess;
      }
    }
  }
  return hipMalloc(p, bytes);
}
hipError_t mgs_hip_free(void *p) {
  if (!p) return hipSuccess;
  {
    std::lock_guard<std::mutex> lk(g_arena.m);
    if (g_arena.base && (char *)p >= g_arena.base && (char *)p < g_arena.base + g_arena.cap) {
      const size_t off = (size_t)((char *)p - g_arena.base);
      auto u = g_arena.used.find(off);
      if (u == g_arena.used.end()) return hipErrorInvalidValue;
      size_t bo = off, bs = u->second;
      g_arena.used.erase(u);
      {   // hipFree's contract: nothing in flight touches the block afterwards — on the ARENA's device, whatever is current
        int cur = -1;
        const bool sw = hipGetDevice(&cur) == hipSuccess && g_arena.device >= 0 && cur != g_arena.device;
        if (sw) (void)hipSetDevice(g_arena.device);
        (void)hipDeviceSynchronize();
        if (sw) (void)hipSetDevice(cur);
      }
      auto nx = g_arena.free_blocks.lower_bound(bo);
      if (nx != g_arena.free_blocks.end() && nx->first == bo + bs) { bs += nx->second; nx = g_arena.free_blocks.erase(nx); }
      if (nx != g_arena.free_blocks.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == bo) { bo = pv->first; bs += pv->second; g_arena.free_blocks.erase(pv); } }
      g_arena.free_blocks[bo] = bs;
      return hipSuccess;
    }
  }
  return hipFree(p);
}

int mgs_fail(mgs_ctx *ctx, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_mgs_last_error = buf;
  if (ctx) ctx->err = buf;
  return code;
}

// Halo exchange of level l through the native transport, on the context's stream: dst[slot] = src[row the owner of that slot was
// asked for] — src is a vector of the level (its owned entries), dst the halo slots (behind the owned entries of the same vector,
// or a payload buffer).  Where every peer's rows are a few contiguous ranges (plane shards) and the receive segmentation has been
// installed (mgs_hier_set_native_recv_segments), the ranges are sent straight from src: no pack kernel.
static int native_exchange(mgs_hier *h, int l, const double *src, double *dst) {
  mgs_native_plan *P = h->lev[l].nx;
  mgs_ctx *ctx = h->ctx;
  if (!P->ns && !P->nr) return MGS_OK;
  const int world = (int)P->scnt.size();
  std::vector<mgs_xfer_op> ops;
  const bool seg = P->use_seg && P->seg_ok;
  if (P->ns && !seg) MGS_TRY(k_gather(ctx, src, P->send_idx, P->ns, P->sendbuf));
  size_t so = 0, ro = 0;
  for (int p = 0; p < world; ++p) {
    if (seg) { for (size_t q = 0; q < P->sseg_len[p].size(); ++q) ops.push_back({src + P->sseg_start[p][q], nullptr, (size_t)P->sseg_len[p][q], p}); }
    else { size_t q0 = so; for (int len : P->pseg_len[p]) { ops.push_back({P->sendbuf + q0, nullptr, (size_t)len, p}); q0 += (size_t)len; } }      // packed: one message per peer
    if (P->use_seg) { size_t r = ro; for (int len : P->rseg_len[p]) { ops.push_back({nullptr, dst + r, (size_t)len, p}); r += (size_t)len; } }
    else { size_t r = ro; for (int len : P->prseg_len[p]) { ops.push_back({nullptr, dst + r, (size_t)len, p}); r += (size_t)len; } }
    so += (size_t)P->scnt[p]; ro += (size_t)P->rcnt[p];
  }
  return mgs_comm_exchange_ops(P->comm, ops.data(), (int)ops.size());
}

// Overlapped exchange inside a captured cycle (option native_overlap).  The RCCL calls stay on the stream the capture began on: RCCL forks to
// streams of its own inside a capture, and this HIP runtime only survives that on the ORIGIN stream (a forked stream that
// forks again ends in a cycle of the capture bookkeeping — hipStreamEndCapture recursed until the stack ran out).  What
// moves to the context's second stream is the kernel over the interior row blocks, which needs no halo value:
//   origin:  producer → [fork] → pack → RCCL group → [join] → boundary row blocks → …
//   second:             interior row blocks ──────────┘
// Under capture the two cross-stream dependencies are edges of the graph.
static int fork_side(mgs_hier *h, hipEvent_t *join) {
  mgs_ctx *ctx = h->ctx;
  while (h->fork_events.size() < h->fork_used + 2) {
    hipEvent_t e; MGS_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->fork_events.push_back(e);
  }
  hipEvent_t fork = h->fork_events[h->fork_used++]; *join = h->fork_events[h->fork_used++];
  MGS_HIP(ctx, hipEventRecord(fork, ctx->stream));
  MGS_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, fork, 0));
  return MGS_OK;
}
// runs `interior` (kernel launches on ctx->stream) on the second stream, then the exchange on the origin, then joins
template <class F>
static int overlapped_exchange(mgs_hier *h, int l, const double *src, double *out, F interior) {
  mgs_ctx *ctx = h->ctx;
  hipEvent_t join;
  MGS_TRY(fork_side(h, &join));
  hipStream_t origin = ctx->stream;
  ctx->stream = ctx->comm_stream;
  int rc = interior();
  ctx->stream = origin;
  MGS_TRY(rc);
  MGS_HIP(ctx, hipEventRecord(join, ctx->comm_stream));
  MGS_TRY(native_exchange(h, l, src, out));
  MGS_HIP(ctx, hipStreamWaitEvent(ctx->stream, join, 0));
  return MGS_OK;
}

extern "C" {

const char *mgs_version(void) { return "multigridsolver_amd 0.1 (gfx950, f64)"; }

// ------------------------------------------------------------------ context
int mgs_ctx_create(int device, void *stream, mgs_ctx **out) {
  if (!out) return mgs_fail(nullptr, MGS_ERR_INVALID, "mgs_ctx_create: out is NULL");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return mgs_fail(nullptr, MGS_ERR_HIP, "no HIP device available (%s): libmgs has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev) return mgs_fail(nullptr, MGS_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
  MGS_HIP(nullptr, hipSetDevice(device));
  mgs_ctx *c = new mgs_ctx();
  c->device = device;
  if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
  else { e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); if (e != hipSuccess) { delete c; return mgs_fail(nullptr, MGS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); } c->own_stream = true; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  c->red_cap = 4096 + 64;    // DOT_BLOCKS partials + the folded results (kernels_aux.hip)
  if (mgs_hip_malloc((void **)&c->red_dev, sizeof(double) * c->red_cap) != hipSuccess || hipHostMalloc((void **)&c->red_host, sizeof(double) * 2 * MGS_RED_VALS, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    delete c; return mgs_fail(nullptr, MGS_ERR_ALLOC, "context scratch allocation failed");
  }
  for (int q = 0; q < 2 * MGS_RED_VALS; ++q) c->red_host[q] = 0.0;
  if (hipHostGetDevicePointer((void **)&c->red_host_dev, c->red_host, 0) != hipSuccess) { c->red_host_dev = nullptr; (void)hipGetLastError(); }
  // MGS_OPTIONS="key=value,key=value": initial option values of every context (A/B runs of whole test suites)
  if (const char *env = getenv("MGS_OPTIONS")) {
    std::string all(env);
    size_t pos = 0;
    while (pos < all.size()) {
      size_t end = all.find(',', pos); if (end == std::string::npos) end = all.size();
      const std::string kv = all.substr(pos, end - pos); const size_t eq = kv.find('=');
      if (eq != std::string::npos && mgs_ctx_set_option(c, kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1)) != MGS_OK) {
        const std::string msg = g_mgs_last_error; mgs_ctx_destroy(c); return mgs_fail(nullptr, MGS_ERR_INVALID, "MGS_OPTIONS: %s", msg.c_str());
      }
      pos = end + 1;
    }
  }
  *out = c;
  return MGS_OK;
}
int mgs_ctx_destroy(mgs_ctx *c) {
  if (!c) return MGS_OK;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  for (mgs_vec *v : c->ws_free) mgs_vec_destroy(v);
  c->ws_free.clear();
  if (c->red_dev) mgs_hip_free(c->red_dev);
  if (c->dot_part) mgs_hip_free(c->dot_part);
  if (c->red_host) hipHostFree(c->red_host);
  if (c->own_stream) hipStreamDestroy(c->stream);
  if (c->comm_stream) hipStreamDestroy(c->comm_stream);
  delete c;
  return MGS_OK;
}
const char *mgs_last_error(const mgs_ctx *ctx) { return ctx ? ctx->err.c_str() : g_mgs_last_error.c_str(); }
int mgs_sync(mgs_ctx *ctx) { MGS_HIP(ctx, hipStreamSynchronize(ctx->stream)); return MGS_OK; }
void *mgs_ctx_stream(mgs_ctx *ctx) { return (void *)ctx->stream; }
// Releases what the context keeps between calls: the Krylov solvers' work vectors (8 vectors of the operator's size after a
// BiCGSTAB solve) and the partial-sum scratch of the fused inner products.
int mgs_ctx_trim(mgs_ctx *ctx) {
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (mgs_vec *v : ctx->ws_free) mgs_vec_destroy(v);
  ctx->ws_free.clear();
  if (ctx->dot_part) { mgs_hip_free(ctx->dot_part); ctx->dot_part = nullptr; ctx->dot_part_cap = 0; }
  return MGS_OK;
}
int mgs_ctx_set_option(mgs_ctx *ctx, const char *key, int value) {
  std::string k(key ? key : "");
  if (k == "xcd_remap") ctx->opt_xcd_remap = value;
  else if (k == "nontemporal") ctx->opt_nontemporal = value;
  else if (k == "spmv_variant") ctx->opt_spmv_variant = value;
  else if (k == "graph") ctx->opt_graph = value;
  else if (k == "graph_split_rows") ctx->opt_graph_split_rows = value;
  else if (k == "strip") ctx->opt_strip = value;
  else if (k == "fuse") ctx->opt_fuse = value;
  else if (k == "lds_pad") ctx->opt_lds_pad = value;
  else if (k == "fuse_operands") ctx->opt_fuse_operands = value;
  else if (k == "rowcode") ctx->opt_rowcode = value;
  else if (k == "nt_store") ctx->opt_nt_store = value;
  else if (k == "valcode") ctx->opt_valcode = value;
  else if (k == "split_min_rows") ctx->opt_split_min_rows = value;
  else if (k == "blkptr") ctx->opt_blkptr = value;
  else if (k == "fuse_restrict") ctx->opt_fuse_restrict = value;
  else if (k == "diag_from_values") ctx->opt_diag_from_values = value;
  else if (k == "fuse_dots") ctx->opt_fuse_dots = value;
  else if (k == "merge_ap") ctx->opt_merge_ap = value;
  else if (k == "group_strip") ctx->opt_group_strip = value;
  else if (k == "group_order") ctx->opt_group_order = value;
  else if (k == "group_stray_pct") ctx->opt_group_stray_pct = value;
  else if (k == "group_blocks") ctx->opt_group_blocks = value;
  else if (k == "group_min_link") ctx->opt_group_min_link = value;
  else if (k == "group_concurrent") ctx->opt_group_concurrent = value;
  else if (k == "group_sweep") ctx->opt_group_sweep = value;
  else if (k == "group_min_blocks") ctx->opt_group_min_blocks = value;
  else if (k == "native_graph") ctx->opt_native_graph = value;
  else if (k == "native_overlap") ctx->opt_native_overlap = value;
  else if (k == "blas1_vec") ctx->opt_blas1_vec = value;
  else if (k == "post_results") ctx->opt_post_results = value;
  else if (k == "blas1_pairs") ctx->opt_blas1_pairs = value;
  else if (k == "stage_unroll") ctx->opt_stage_unroll = value;
  else if (k == "rowptr_scan") ctx->opt_rowptr_scan = value;
  else if (k == "kcycle_energy") ctx->opt_kcycle_energy = value;
  else if (k == "aggpre_max_rows") ctx->opt_aggpre_max_rows = value;
  else if (k == "emu_split_self") ctx->opt_emu_split_self = value;
  else return mgs_fail(ctx, MGS_ERR_INVALID, "unknown option '%s'", k.c_str());
  ++ctx->opt_epoch;      // every captured cycle was recorded under the old options: mgs_vcycle drops them
  return MGS_OK;
}
int mgs_ctx_set_native_allreduce(mgs_ctx *ctx, mgs_comm *c) { ctx->ncomm = c; return MGS_OK; }
int mgs_ctx_set_allreduce(mgs_ctx *ctx, mgs_allreduce_fn fn, void *user) { ctx->allreduce = fn; ctx->allreduce_user = user; return MGS_OK; }

}  // extern "C"

// ------------------------------------------------------------------ CSR
int mgs_csr_alloc(mgs_ctx *ctx, int rows, int cols, int64_t nnz, mgs_csr **out) {
  MGS_CHECK(ctx, rows >= 0 && cols >= 0 && nnz >= 0 && nnz < 2147483647LL, MGS_ERR_INVALID, "csr: bad shape %d x %d nnz %lld", rows, cols, (long long)nnz);
  mgs_csr *A = new mgs_csr();
  A->ctx = ctx; A->rows = rows; A->cols = cols; A->nnz = nnz;
  int rc = mgs_dev_alloc(ctx, &A->rowptr, (size_t)rows + 1);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &A->col, (size_t)nnz + 4);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &A->val, (size_t)nnz + 4);
  if (rc != MGS_OK) { mgs_csr_destroy(A); return rc; }
  *out = A;
  return MGS_OK;
}

extern "C" {

int mgs_csr_upload(mgs_ctx *ctx, int rows, int cols, int64_t nnz, const int *rowptr, const int *col, const double *val, mgs_csr **out) {
  MGS_CHECK(ctx, ctx && out && rowptr && (nnz == 0 || (col && val)), MGS_ERR_INVALID, "mgs_csr_upload: NULL argument");
  MGS_CHECK(ctx, rowptr[0] == 0 && rowptr[rows] == nnz, MGS_ERR_INVALID, "mgs_csr_upload: rowptr[0]=%d rowptr[rows]=%d nnz=%lld inconsistent", rowptr[0], rowptr[rows], (long long)nnz);
  for (int i = 0; i < rows; ++i) {
    MGS_CHECK(ctx, rowptr[i] <= rowptr[i + 1], MGS_ERR_INVALID, "mgs_csr_upload: rowptr not monotone at row %d", i);
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      MGS_CHECK(ctx, col[k] >= 0 && col[k] < cols, MGS_ERR_INVALID, "mgs_csr_upload: column %d out of range in row %d", col[k], i);
      MGS_CHECK(ctx, k == rowptr[i] || col[k - 1] < col[k], MGS_ERR_INVALID, "mgs_csr_upload: columns of row %d not strictly ascending", i);
    }
  }
  mgs_csr *A = nullptr;
  MGS_TRY(mgs_csr_alloc(ctx, rows, cols, nnz, &A));
  hipMemcpyAsync(A->rowptr, rowptr, sizeof(int) * ((size_t)rows + 1), hipMemcpyHostToDevice, ctx->stream);
  if (nnz) {
    hipMemcpyAsync(A->col, col, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
    hipMemcpyAsync(A->val, val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream);
  }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { mgs_csr_destroy(A); return mgs_fail(ctx, MGS_ERR_HIP, "csr upload: %s", hipGetErrorString(e)); }
  int rc = mgs_plan_csr(A);
  if (rc != MGS_OK) { mgs_csr_destroy(A); return rc; }
  *out = A;
  return MGS_OK;
}
int mgs_csr_download(const mgs_csr *A, int *rowptr, int *col, double *val) {
  mgs_ctx *ctx = A->ctx;
  if (rowptr) MGS_HIP(ctx, hipMemcpyAsync(rowptr, A->rowptr, sizeof(int) * ((size_t)A->rows + 1), hipMemcpyDeviceToHost, ctx->stream));
  if (col && A->nnz) MGS_HIP(ctx, hipMemcpyAsync(col, A->col, sizeof(int) * (size_t)A->nnz, hipMemcpyDeviceToHost, ctx->stream));
  if (val && A->nnz) MGS_HIP(ctx, hipMemcpyAsync(val, A->val, sizeof(double) * (size_t)A->nnz, hipMemcpyDeviceToHost, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MGS_OK;
}
int mgs_csr_shape(const mgs_csr *A, int *rows, int *cols, int64_t *nnz) {
  if (rows) *rows = A->rows; if (cols) *cols = A->cols; if (nnz) *nnz = A->nnz;
  return MGS_OK;
}
int mgs_csr_device_ptrs(const mgs_csr *A, void **rowptr, void **col, void **val) {
  if (rowptr) *rowptr = A->rowptr; if (col) *col = A->col; if (val) *val = A->val;
  return MGS_OK;
}
int mgs_csr_plan_info(const mgs_csr *A, int64_t out[8]) {
  out[0] = A->max_block_nnz; out[1] = A->max_row_len; out[2] = A->far_band; out[3] = (int64_t)(A->lds_cap + 2) * 12 + 16;
  out[4] = A->halo_lo_blocks; out[5] = A->halo_hi_blocks; out[6] = A->halo_split_ok ? 1 : 0; out[7] = A->max_block_nnz > A->lds_cap ? 1 : 0;
  return MGS_OK;
}
int mgs_csr_get_origin(const mgs_csr *A, int *origin) {
  mgs_ctx *ctx = A->ctx;
  if (!A->origin) return 1;                       // identity (no origin recorded): nothing written
  MGS_HIP(ctx, hipMemcpyAsync(origin, A->origin, sizeof(int) * (size_t)A->rows, hipMemcpyDeviceToHost, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MGS_OK;
}
int mgs_csr_set_origin(mgs_csr *A, const int *origin) {
  mgs_ctx *ctx = A->ctx;
  if (A->origin) { mgs_hip_free(A->origin); A->origin = nullptr; }
  if (!origin || A->rows == 0) return MGS_OK;
  MGS_TRY(mgs_dev_alloc(ctx, &A->origin, (size_t)A->rows));
  MGS_HIP(ctx, hipMemcpy(A->origin, origin, sizeof(int) * (size_t)A->rows, hipMemcpyHostToDevice));
  return MGS_OK;
}
int mgs_csr_destroy(mgs_csr *A) {
  if (!A) return MGS_OK;
  if (A->owns) { if (A->rowptr) mgs_hip_free(A->rowptr); if (A->col) mgs_hip_free(A->col); if (A->val) mgs_hip_free(A->val); }
  if (A->blkptr) mgs_hip_free(A->blkptr);
  if (A->origin) mgs_hip_free(A->origin);
  mgs_free_rowcode(A->code);
  delete A;
  return MGS_OK;
}
int mgs_csr_optimize(mgs_csr *A) {
  if (!A || A->code_tried || !A->ctx->opt_rowcode || A->rows == 0 || A->nnz == 0) return MGS_OK;
  A->code_tried = true;
  return mgs_build_rowcode(A->ctx, A->rows, A->rowptr, A->col, nullptr, 0x7fffffff, &A->code, A->ctx->opt_valcode ? A->val : nullptr);
}
int mgs_csr_rowcode_info(const mgs_csr *A, int64_t out[4]) {
  out[0] = A->code ? A->code->coded_blocks : 0; out[1] = A->code ? A->code->nblocks : (A->rows + 255) / 256;
  out[2] = A->code ? A->code->tab_total : 0; out[3] = A->code ? A->code->tab_cap : 0;
  return MGS_OK;
}
int mgs_csr_poisson3d(mgs_ctx *ctx, int N, int plane_lo, int plane_hi, int local_cols, mgs_csr **out) { return k_poisson3d(ctx, N, plane_lo, plane_hi, local_cols, out); }
int mgs_csr_poisson2d(mgs_ctx *ctx, int n, mgs_csr **out) { return k_poisson2d(ctx, n, out); }
int mgs_csr_transpose(const mgs_csr *A, mgs_csr **out) { return k_transpose(A, out); }
int mgs_csr_galerkin(const mgs_csr *A, const mgs_xfer *T, mgs_csr **out) {
  MGS_CHECK(A->ctx, T->n_fine == A->rows && A->rows == A->cols, MGS_ERR_INVALID, "galerkin: A is %d x %d, P has %d rows", A->rows, A->cols, T->n_fine);
  return T->aggregation ? k_galerkin_agg(A, T, out) : k_galerkin_general(A, T, out);
}

// ------------------------------------------------------------------ vectors
int mgs_vec_create(mgs_ctx *ctx, int64_t n, mgs_vec **out) {
  MGS_CHECK(ctx, n >= 0, MGS_ERR_INVALID, "mgs_vec_create: n < 0");
  mgs_vec *v = new mgs_vec(); v->ctx = ctx; v->n = n; v->owns = true;
  int rc = mgs_dev_alloc(ctx, &v->d, (size_t)n);
  if (rc != MGS_OK) { delete v; return rc; }
  hipMemsetAsync(v->d, 0, sizeof(double) * (size_t)(n ? n : 1), ctx->stream);
  *out = v;
  return MGS_OK;
}
int mgs_vec_wrap(mgs_ctx *ctx, void *p, int64_t n, mgs_vec **out) {
  MGS_CHECK(ctx, p || n == 0, MGS_ERR_INVALID, "mgs_vec_wrap: NULL pointer");
  mgs_vec *v = new mgs_vec(); v->ctx = ctx; v->n = n; v->owns = false; v->d = (double *)p;
  *out = v;
  return MGS_OK;
}
int mgs_vec_destroy(mgs_vec *v) { if (!v) return MGS_OK; if (v->owns && v->d) mgs_hip_free(v->d); delete v; return MGS_OK; }
int mgs_vec_upload(mgs_vec *v, const double *host, int64_t n) {
  MGS_CHECK(v->ctx, n <= v->n, MGS_ERR_INVALID, "mgs_vec_upload: %lld > size %lld", (long long)n, (long long)v->n);
  MGS_HIP(v->ctx, hipMemcpyAsync(v->d, host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, v->ctx->stream));
  MGS_HIP(v->ctx, hipStreamSynchronize(v->ctx->stream));
  return MGS_OK;
}
int mgs_vec_download(const mgs_vec *v, double *host, int64_t n) {
  MGS_CHECK(v->ctx, n <= v->n, MGS_ERR_INVALID, "mgs_vec_download: %lld > size %lld", (long long)n, (long long)v->n);
  MGS_HIP(v->ctx, hipMemcpyAsync(host, v->d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, v->ctx->stream));
  MGS_HIP(v->ctx, hipStreamSynchronize(v->ctx->stream));
  return MGS_OK;
}
int mgs_vec_fill(mgs_vec *v, double value) { return k_fill(v->ctx, v->d, v->n, value); }
int mgs_vec_copy(const mgs_vec *src, mgs_vec *dst) {
  MGS_CHECK(src->ctx, src->n <= dst->n, MGS_ERR_INVALID, "mgs_vec_copy: size mismatch");
  MGS_HIP(src->ctx, hipMemcpyAsync(dst->d, src->d, sizeof(double) * (size_t)src->n, hipMemcpyDeviceToDevice, src->ctx->stream));
  return MGS_OK;
}
int64_t mgs_vec_size(const mgs_vec *v) { return v->n; }
void *mgs_vec_ptr(const mgs_vec *v) { return v->d; }
int mgs_vec_rand(mgs_vec *v, uint64_t seed, int64_t off) { return k_rand(v->ctx, v->d, v->n, seed, off); }

// ------------------------------------------------------------------ primitives
int mgs_spmv(const mgs_csr *A, const mgs_vec *x, mgs_vec *y) {
  MGS_CHECK(A->ctx, x->n >= A->cols && y->n >= A->rows, MGS_ERR_INVALID, "mgs_spmv: A %d x %d, x %lld, y %lld", A->rows, A->cols, (long long)x->n, (long long)y->n);
  MGS_CHECK(A->ctx, x->d != y->d, MGS_ERR_INVALID, "mgs_spmv: y must not alias x");
  return mgs_launch_csr_op(A, MGS_OP_SPMV, x->d, nullptr, nullptr, 0.0, y->d);
}
int mgs_residual(const mgs_csr *A, const mgs_vec *x, const mgs_vec *b, mgs_vec *r) {
  MGS_CHECK(A->ctx, x->n >= A->cols && b->n >= A->rows && r->n >= A->rows, MGS_ERR_INVALID, "mgs_residual: shape mismatch");
  MGS_CHECK(A->ctx, x->d != r->d, MGS_ERR_INVALID, "mgs_residual: r must not alias x");
  return mgs_launch_csr_op(A, MGS_OP_RESIDUAL, x->d, b->d, nullptr, 0.0, r->d);
}
int mgs_diag_inv(const mgs_csr *A, mgs_vec *dinv) {
  MGS_CHECK(A->ctx, dinv->n >= A->rows, MGS_ERR_INVALID, "mgs_diag_inv: dinv too small");
  int bad = 0;
  MGS_TRY(k_diag_inv(A, dinv->d, &bad));
  MGS_CHECK(A->ctx, bad == 0, MGS_ERR_NUMERIC, "mgs_diag_inv: %d rows have a missing or zero diagonal", bad);
  return MGS_OK;
}
int mgs_jacobi(const mgs_csr *A, const mgs_vec *dinv, double omega, const mgs_vec *b, const mgs_vec *x_in, mgs_vec *x_out) {
  MGS_CHECK(A->ctx, A->rows <= A->cols && x_in->n >= A->cols && b->n >= A->rows && x_out->n >= A->rows && dinv->n >= A->rows, MGS_ERR_INVALID, "mgs_jacobi: shape mismatch");
  MGS_CHECK(A->ctx, x_in->d != x_out->d, MGS_ERR_INVALID, "mgs_jacobi: out of place only (x_out must not alias x_in)");
  return mgs_launch_csr_op(A, MGS_OP_JACOBI, x_in->d, b->d, dinv->d, omega, x_out->d);
}

int mgs_xfer_create(const mgs_csr *P, mgs_xfer **out) { return k_xfer_from_csr(P, out); }
int mgs_xfer_destroy(mgs_xfer *T) {
  if (!T) return MGS_OK;
  if (T->agg) mgs_hip_free(T->agg); if (T->cptr) mgs_hip_free(T->cptr); if (T->members) mgs_hip_free(T->members); if (T->corigin) mgs_hip_free(T->corigin); if (T->halo_cmap) mgs_hip_free(T->halo_cmap);
  if (T->P) mgs_csr_destroy(T->P); if (T->Pt) mgs_csr_destroy(T->Pt);
  delete T;
  return MGS_OK;
}
int mgs_xfer_shape(const mgs_xfer *T, int *n_fine, int *n_coarse, int *is_agg) {
  if (n_fine) *n_fine = T->n_fine; if (n_coarse) *n_coarse = T->n_coarse; if (is_agg) *is_agg = T->aggregation ? 1 : 0;
  return MGS_OK;
}
int mgs_xfer_download_agg(const mgs_xfer *T, int *agg) {
  MGS_CHECK(T->ctx, T->aggregation, MGS_ERR_STATE, "transfer is not in aggregation form");
  MGS_HIP(T->ctx, hipMemcpyAsync(agg, T->agg, sizeof(int) * (size_t)T->n_fine, hipMemcpyDeviceToHost, T->ctx->stream));
  MGS_HIP(T->ctx, hipStreamSynchronize(T->ctx->stream));
  return MGS_OK;
}
int mgs_restrict(const mgs_xfer *T, const mgs_vec *r, mgs_vec *rc) {
  MGS_CHECK(T->ctx, r->n >= T->n_fine && rc->n >= T->n_coarse, MGS_ERR_INVALID, "mgs_restrict: shape mismatch");
  if (T->aggregation) return k_restrict_agg(T->ctx, T->n_coarse, T->cptr, T->members, r->d, rc->d);
  return mgs_launch_csr_op(T->Pt, MGS_OP_SPMV, r->d, nullptr, nullptr, 0.0, rc->d);
}
int mgs_prolong(const mgs_xfer *T, const mgs_vec *ec, mgs_vec *e) {
  MGS_CHECK(T->ctx, e->n >= T->n_fine && ec->n >= T->n_coarse, MGS_ERR_INVALID, "mgs_prolong: shape mismatch");
  if (T->aggregation) return k_prolong_agg(T->ctx, T->n_fine, T->agg, ec->d, e->d, 0);
  return mgs_launch_csr_op(T->P, MGS_OP_SPMV, ec->d, nullptr, nullptr, 0.0, e->d);
}

int mgs_dot(const mgs_vec *x, const mgs_vec *y, double *out) {
  MGS_CHECK(x->ctx, x->n == y->n, MGS_ERR_INVALID, "mgs_dot: size mismatch");
  return k_dot(x->ctx, x->n, x->d, y->d, out);
}
int mgs_nrm2(const mgs_vec *x, double *out) { double s = 0; MGS_TRY(k_dot(x->ctx, x->n, x->d, x->d, &s)); *out = std::sqrt(s); return MGS_OK; }
int mgs_axpby(double a, const mgs_vec *x, double b, mgs_vec *y) {
  MGS_CHECK(x->ctx, x->n == y->n, MGS_ERR_INVALID, "mgs_axpby: size mismatch");
  return k_axpby(x->ctx, x->n, a, x->d, b, y->d);
}
int mgs_axpbypcz(double a, const mgs_vec *x, double b, const mgs_vec *y, double c, mgs_vec *z) {
  MGS_CHECK(x->ctx, x->n == y->n && x->n == z->n, MGS_ERR_INVALID, "mgs_axpbypcz: size mismatch");
  return k_axpbypcz(x->ctx, x->n, a, x->d, b, y->d, c, z->d);
}
int mgs_halo_pack(mgs_ctx *ctx, const mgs_vec *x, const int *send_idx_dev, int64_t n_send, double *send_buf_dev) {
  return k_gather(ctx, x->d, send_idx_dev, n_send, send_buf_dev);
}

}  // extern "C"

// prolong_add for the general form needs a temporary: e = P ec; x += e
static int prolong_add_impl(const mgs_xfer *T, const double *ec, double *x, double *tmp) {
  if (T->aggregation) return k_prolong_agg(T->ctx, T->n_fine, T->agg, ec, x, 1);
  MGS_TRY(mgs_launch_csr_op(T->P, MGS_OP_SPMV, ec, nullptr, nullptr, 0.0, tmp));
  return k_axpby(T->ctx, T->n_fine, 1.0, tmp, 1.0, x);
}

extern "C" int mgs_prolong_add(const mgs_xfer *T, const mgs_vec *ec, mgs_vec *x) {
  MGS_CHECK(T->ctx, x->n >= T->n_fine && ec->n >= T->n_coarse, MGS_ERR_INVALID, "mgs_prolong_add: shape mismatch");
  if (T->aggregation) return k_prolong_agg(T->ctx, T->n_fine, T->agg, ec->d, x->d, 1);
  mgs_vec *tmp = nullptr;
  MGS_TRY(mgs_vec_create(T->ctx, T->n_fine, &tmp));
  int rc = prolong_add_impl(T, ec->d, x->d, tmp->d);
  hipStreamSynchronize(T->ctx->stream);
  mgs_vec_destroy(tmp);
  return rc;
}

// ------------------------------------------------------------------ hierarchy
static int level_init(mgs_hier *h, mgs_level &L, const mgs_csr *A, bool own) {
  mgs_ctx *ctx = h->ctx;
  L.A = A; L.own_A = own; L.n = A->rows; L.n_ext = A->cols > A->rows ? A->cols : A->rows;
  MGS_TRY(mgs_vec_create(ctx, L.n_ext, &L.dinv));      // row shards: the halo part receives the owners' values once (prepare_fused)
  MGS_TRY(mgs_vec_create(ctx, L.n_ext, &L.r));
  MGS_TRY(mgs_vec_create(ctx, L.n_ext, &L.tmp));
  MGS_TRY(mgs_vec_create(ctx, L.n_ext, &L.b)); MGS_TRY(mgs_vec_create(ctx, L.n_ext, &L.x));
  return mgs_diag_inv(A, L.dinv);
}
static void level_free(mgs_level &L) {
  if (L.own_A && L.A) mgs_csr_destroy(const_cast<mgs_csr *>(L.A));
  if (L.T) mgs_xfer_destroy(L.T);
  mgs_vec_destroy(L.dinv); mgs_vec_destroy(L.r); mgs_vec_destroy(L.tmp); mgs_vec_destroy(L.b); mgs_vec_destroy(L.x); mgs_vec_destroy(L.wd);
  mgs_vec_destroy(L.hbuf);
  if (L.val_wd) mgs_hip_free(L.val_wd);
  if (L.col_agg) mgs_hip_free(L.col_agg);
  if (L.cmap_ext) mgs_hip_free(L.cmap_ext);
  mgs_free_rowcode(L.code_agg);
  if (L.AP) mgs_csr_destroy(L.AP);
  mgs_free_rowcode(L.code_ap);
  mgs_free_rowcode(L.code_pre);
  mgs_free_rowcode(L.code_hat);
  mgs_free_groups(L.grp);
  if (L.dpos) mgs_hip_free(L.dpos);
  if (L.nx) { if (L.nx->send_idx) mgs_hip_free(L.nx->send_idx); if (L.nx->sendbuf) mgs_hip_free(L.nx->sendbuf); delete L.nx; L.nx = nullptr; }
  mgs_vec_destroy(L.kc1); mgs_vec_destroy(L.kv1); mgs_vec_destroy(L.kc2); mgs_vec_destroy(L.kv2); mgs_vec_destroy(L.kr);
  if (L.kscal) mgs_hip_free(L.kscal);
  L = mgs_level();
}
static void drop_graph(mgs_hier *h) {
  for (auto &g : h->graphs) { if (g.exec) hipGraphExecDestroy(g.exec); g = mgs_hier::GraphSlot(); }
  if (h->coarse_exec) { hipGraphExecDestroy(h->coarse_exec); h->coarse_exec = nullptr; }
}

extern "C" {

int mgs_hier_create(mgs_ctx *ctx, const mgs_csr *A, double omega, int nu1, int nu2, mgs_hier **out) {
  MGS_CHECK(ctx, A && out, MGS_ERR_INVALID, "mgs_hier_create: NULL argument");
  MGS_CHECK(ctx, A->rows <= A->cols && A->rows > 0, MGS_ERR_INVALID, "mgs_hier_create: operator must have rows <= cols (square, or a row shard with halo columns); got %d x %d", A->rows, A->cols);
  MGS_CHECK(ctx, nu1 >= 0 && nu2 >= 0, MGS_ERR_INVALID, "mgs_hier_create: negative sweep count");
  mgs_hier *h = new mgs_hier();
  h->ctx = ctx; h->omega = omega; h->nu1 = nu1; h->nu2 = nu2;
  h->lev.emplace_back();
  int rc = level_init(h, h->lev.back(), A, false);
  if (rc != MGS_OK) { mgs_hier_destroy(h); return rc; }
  *out = h;
  return MGS_OK;
}
static void free_native_tail(mgs_hier *h) {
  mgs_native_tail *T = h->ntail;
  if (!T) return;
  if (T->send) mgs_hip_free(T->send); if (T->all) mgs_hip_free(T->all); if (T->gidx) mgs_hip_free(T->gidx); if (T->halo_global) mgs_hip_free(T->halo_global);
  mgs_vec_destroy(T->b); mgs_vec_destroy(T->x);
  delete T; h->ntail = nullptr;
}
int mgs_hier_destroy(mgs_hier *h) {
  if (!h) return MGS_OK;
  hipStreamSynchronize(h->ctx->stream);
  drop_graph(h);
  for (hipEvent_t e : h->fork_events) hipEventDestroy(e);
  for (auto &L : h->lev) level_free(L);
  if (h->inv) mgs_hip_free(h->inv);
  free_native_tail(h);
  delete h;
  return MGS_OK;
}
int mgs_hier_set_smoother(mgs_hier *h, double omega, int nu1, int nu2) {
  MGS_CHECK(h->ctx, nu1 >= 0 && nu2 >= 0, MGS_ERR_INVALID, "negative sweep count");
  h->omega = omega; h->nu1 = nu1; h->nu2 = nu2; drop_graph(h);
  return MGS_OK;
}
// ---- native RCCL transport of a sharded hierarchy (comm_rccl.hip) ----
int mgs_hier_set_native_exchange(mgs_hier *h, int level, mgs_comm *c, const int *send_idx, const int *send_counts, const int *recv_counts) {
  mgs_ctx *ctx = h->ctx;
  MGS_CHECK(ctx, level >= 0 && level < (int)h->lev.size(), MGS_ERR_INVALID, "mgs_hier_set_native_exchange: level %d out of range", level);
  mgs_level &L = h->lev[level];
  auto free_plan = [](mgs_native_plan *P) { if (!P) return; if (P->send_idx) mgs_hip_free(P->send_idx); if (P->sendbuf) mgs_hip_free(P->sendbuf); delete P; };
  free_plan(L.nx); L.nx = nullptr;
  drop_graph(h);
  h->native = false; for (auto &q : h->lev) h->native = h->native || q.nx;
  if (!c) return MGS_OK;
  // build and validate the plan aside; it is installed only when complete (a caller that handles the error keeps its callbacks)
  int world = 0; mgs_comm_size(c, &world, nullptr);
  mgs_native_plan *P = new mgs_native_plan();
  P->comm = c; P->scnt.assign(send_counts, send_counts + world); P->rcnt.assign(recv_counts, recv_counts + world);
  for (int p = 0; p < world; ++p) { P->ns += P->scnt[p]; P->nr += P->rcnt[p]; }
  int rc = MGS_OK;
  if (P->nr != (int64_t)L.A->cols - L.A->rows)
    rc = mgs_fail(ctx, MGS_ERR_INVALID, "mgs_hier_set_native_exchange: level %d has %d halo columns, plan delivers %lld", level, L.A->cols - L.A->rows, (long long)P->nr);
  if (rc == MGS_OK && P->ns && !send_idx) rc = mgs_fail(ctx, MGS_ERR_INVALID, "mgs_hier_set_native_exchange: send_idx is NULL");
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &P->send_idx, (size_t)P->ns);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &P->sendbuf, (size_t)P->ns);
  if (rc == MGS_OK && P->ns && hipMemcpy(P->send_idx, send_idx, sizeof(int) * (size_t)P->ns, hipMemcpyHostToDevice) != hipSuccess)
    rc = mgs_fail(ctx, MGS_ERR_HIP, "mgs_hier_set_native_exchange: upload of the send list failed");
  if (rc != MGS_OK) { free_plan(P); return rc; }
  // contiguous ranges of every peer's send list (pack-free form, see mgs_native_plan)
  P->sseg_start.assign((size_t)world, {}); P->sseg_len.assign((size_t)world, {}); P->rseg_len.assign((size_t)world, {});
  P->seg_ok = true;
  size_t o = 0;
  for (int p = 0; p < world; ++p) {
    for (int k = 0; k < P->scnt[p]; ++k) {
      const int r = send_idx[o + (size_t)k];
      if (k && r == send_idx[o + (size_t)k - 1] + 1) ++P->sseg_len[p].back();
      else { P->sseg_start[p].push_back(r); P->sseg_len[p].push_back(1); }
    }
    if ((int)P->sseg_len[p].size() > MGS_MAX_SEG) P->seg_ok = false;
    o += (size_t)P->scnt[p];
  }
  // packed mode: one message per peer.  (Rehearsal of a middle rank on one GPU, option emu_split_self: the buffer the rank sends to
  // ITSELF stands for the messages to two neighbours — it goes out as two, so packed and pack-free exchanges are timed on equal terms.)
  P->pseg_len.assign((size_t)world, {}); P->prseg_len.assign((size_t)world, {});
  for (int p = 0; p < world; ++p) {
    const bool split = ctx->opt_emu_split_self && world == 1 && P->scnt[p] == P->rcnt[p] && P->scnt[p] >= 2;
    if (P->scnt[p]) { if (split) { P->pseg_len[p] = {P->scnt[p] / 2, P->scnt[p] - P->scnt[p] / 2}; } else P->pseg_len[p] = {P->scnt[p]}; }
    if (P->rcnt[p]) { if (split) { P->prseg_len[p] = {P->rcnt[p] / 2, P->rcnt[p] - P->rcnt[p] / 2}; } else P->prseg_len[p] = {P->rcnt[p]}; }
  }
  L.nx = P;
  h->native = true;
  return MGS_OK;
}
// What this rank will send to each peer once ranges are switched on: nseg[p] ranges whose lengths follow in seglens (peer after
// peer; a rank whose lists do not split into few ranges reports one range per peer = its packed buffer).  Returns the number of
// lengths written, or a negative error; cap = room in seglens.
int mgs_hier_native_send_segments(const mgs_hier *h, int level, int *nseg, int *seglens, int cap) {
  MGS_CHECK(h->ctx, level >= 0 && level < (int)h->lev.size() && h->lev[level].nx && nseg && seglens, MGS_ERR_STATE, "mgs_hier_native_send_segments: level %d has no native plan", level);
  const mgs_native_plan *P = h->lev[level].nx;
  int w = 0;
  for (size_t p = 0; p < P->scnt.size(); ++p) {
    if (P->seg_ok) {
      nseg[p] = (int)P->sseg_len[p].size();
      for (int len : P->sseg_len[p]) { MGS_CHECK(h->ctx, w < cap, MGS_ERR_INVALID, "mgs_hier_native_send_segments: buffer too small"); seglens[w++] = len; }
    } else {
      nseg[p] = (int)P->pseg_len[p].size();
      for (int len : P->pseg_len[p]) { MGS_CHECK(h->ctx, w < cap, MGS_ERR_INVALID, "mgs_hier_native_send_segments: buffer too small"); seglens[w++] = len; }
    }
  }
  return w;
}
// The ranges the peers will send (their mgs_hier_native_send_segments, shipped by the host side).  COLLECTIVE by contract: every rank
// of the communicator calls it for the level before the next exchange — from this call on this rank, too, sends ranges.
int mgs_hier_set_native_recv_segments(mgs_hier *h, int level, const int *nseg, const int *seglens) {
  mgs_ctx *ctx = h->ctx;
  MGS_CHECK(ctx, level >= 0 && level < (int)h->lev.size() && h->lev[level].nx && nseg && seglens, MGS_ERR_STATE, "mgs_hier_set_native_recv_segments: level %d has no native plan", level);
  mgs_native_plan *P = h->lev[level].nx;
  std::vector<std::vector<int>> rs(P->rcnt.size());
  int w = 0;
  for (size_t p = 0; p < P->rcnt.size(); ++p) {
    int64_t sum = 0;
    for (int q = 0; q < nseg[p]; ++q) { const int len = seglens[w++]; MGS_CHECK(ctx, len > 0, MGS_ERR_INVALID, "mgs_hier_set_native_recv_segments: empty range"); rs[p].push_back(len); sum += len; }
    MGS_CHECK(ctx, sum == P->rcnt[p], MGS_ERR_INVALID, "mgs_hier_set_native_recv_segments: peer %d announces %lld values, the plan expects %d", (int)p, (long long)sum, P->rcnt[p]);
  }
  P->rseg_len.swap(rs);
  P->use_seg = true;
  drop_graph(h);
  return MGS_OK;
}
int mgs_hier_set_native_tail(mgs_hier *h, mgs_comm *c, mgs_hier *tail, const int *nlocs) {
  mgs_ctx *ctx = h->ctx;
  hipStreamSynchronize(ctx->stream);
  free_native_tail(h);
  drop_graph(h);
  if (!c || !tail) return MGS_OK;
  int world = 0, rank = 0; mgs_comm_size(c, &world, &rank);
  mgs_native_tail *T = new mgs_native_tail();
  h->ntail = T;                               // free_native_tail releases a half-built one on every error path below
  auto fail = [&](int rc) { free_native_tail(h); return rc; };
  T->comm = c; T->tail = tail; T->n_loc = nlocs[rank];
  std::vector<int> gi;
  for (int p = 0; p < world; ++p) { T->maxn = std::max(T->maxn, nlocs[p]); }
  for (int p = 0; p < world; ++p) { if (p == rank) T->my_off = (int)gi.size(); for (int j = 0; j < nlocs[p]; ++j) gi.push_back(p * T->maxn + j); }
  T->n_t = (int)gi.size();
  T->even = T->maxn > 0; for (int p = 0; p < world; ++p) T->even = T->even && nlocs[p] == T->maxn;
  if (!(T->n_t == tail->lev[0].n && T->n_loc == h->lev.back().n))
    return fail(mgs_fail(ctx, MGS_ERR_INVALID, "mgs_hier_set_native_tail: tail has %d rows, shards sum to %d", tail->lev[0].n, T->n_t));
  int rc = mgs_dev_alloc(ctx, &T->send, (size_t)std::max(T->maxn, 1));
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &T->all, (size_t)std::max(T->maxn, 1) * world);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &T->gidx, (size_t)std::max(T->n_t, 1));
  if (rc == MGS_OK && hipMemset(T->send, 0, sizeof(double) * (size_t)std::max(T->maxn, 1)) != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "native tail: memset failed");
  if (rc == MGS_OK && T->n_t && hipMemcpy(T->gidx, gi.data(), sizeof(int) * gi.size(), hipMemcpyHostToDevice) != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "native tail: upload failed");
  if (rc == MGS_OK) rc = mgs_vec_create(ctx, T->n_t, &T->b);
  if (rc == MGS_OK) rc = mgs_vec_create(ctx, T->n_t, &T->x);
  if (rc != MGS_OK) return fail(rc);
  return MGS_OK;
}
// Global tail row of every halo slot of the LAST sharded level (host array, one int per slot: first row of the owner in the tail's
// numbering + the owner's local row).  The replicated tail's solution holds the neighbours' entries too, so the kernel that hands this
// rank its own slice also fills the level's halo slots from it: the post pass of the level above needs no halo exchange for e_c
// (one exchange less per cycle, same values and the same bits as the exchange would deliver).  n = 0 switches it off.
int mgs_hier_set_native_tail_halo(mgs_hier *h, const int *halo_global, int n) {
  mgs_ctx *ctx = h->ctx;
  MGS_CHECK(ctx, h->ntail, MGS_ERR_STATE, "mgs_hier_set_native_tail_halo: install the native tail first");
  mgs_native_tail *T = h->ntail;
  hipStreamSynchronize(ctx->stream);
  drop_graph(h);
  if (T->halo_global) { mgs_hip_free(T->halo_global); T->halo_global = nullptr; T->n_halo_global = 0; }
  if (n <= 0 || !halo_global) return MGS_OK;
  const mgs_csr *Al = h->lev.back().A;
  MGS_CHECK(ctx, n == Al->cols - Al->rows, MGS_ERR_INVALID, "mgs_hier_set_native_tail_halo: %d rows given, the last sharded level has %d halo slots", n, Al->cols - Al->rows);
  for (int k = 0; k < n; ++k) MGS_CHECK(ctx, halo_global[k] >= 0 && halo_global[k] < T->n_t, MGS_ERR_INVALID, "mgs_hier_set_native_tail_halo: slot %d maps to row %d outside the tail (%d rows)", k, halo_global[k], T->n_t);
  MGS_TRY(mgs_dev_alloc(ctx, &T->halo_global, (size_t)n));
  MGS_HIP(ctx, hipMemcpy(T->halo_global, halo_global, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  T->n_halo_global = n;
  return MGS_OK;
}
int mgs_hier_native_halo(mgs_hier *h, int level, void *x_dev) {
  MGS_CHECK(h->ctx, level >= 0 && level < (int)h->lev.size() && h->lev[level].nx, MGS_ERR_STATE, "mgs_hier_native_halo: level %d has no native plan", level);
  return native_exchange(h, level, (const double *)x_dev, (double *)x_dev + h->lev[level].A->rows);
}
int mgs_hier_set_halo_exchange_fused(mgs_hier *h, mgs_halo_fused_fn fn, void *user) { h->halo_fused = fn; h->halo_user = user; drop_graph(h); return MGS_OK; }
int mgs_hier_set_kcycle(mgs_hier *h, int levels) {
  MGS_CHECK(h->ctx, levels >= 0, MGS_ERR_INVALID, "mgs_hier_set_kcycle: negative level count");
  h->kcycle_levels = levels; drop_graph(h);
  return MGS_OK;
}
int mgs_hier_set_kcycle_entry(mgs_hier *h, int on) { h->kcycle_entry = on != 0; drop_graph(h); return MGS_OK; }
int mgs_hier_set_correction_scale(mgs_hier *h, double sigma) {
  MGS_CHECK(h->ctx, sigma > 0.0 && sigma <= 4.0, MGS_ERR_INVALID, "mgs_hier_set_correction_scale: sigma must lie in (0, 4]");
  h->corr_scale = sigma; drop_graph(h);
  return MGS_OK;
}
int mgs_hier_set_additive(mgs_hier *h, int on) {
  MGS_CHECK(h->ctx, !on || (!h->halo && !h->halo_begin && !h->native), MGS_ERR_STATE, "mgs_hier_set_additive: not offered on row shards");
  h->additive = on != 0; drop_graph(h);
  return MGS_OK;
}
int mgs_hier_set_halo_exchange(mgs_hier *h, mgs_halo_fn fn, void *user) { h->halo = fn; h->halo_user = user; drop_graph(h); return MGS_OK; }
int mgs_hier_set_halo_exchange_split(mgs_hier *h, mgs_halo_fn begin, mgs_halo_fn end, void *user) {
  MGS_CHECK(h->ctx, (begin == nullptr) == (end == nullptr), MGS_ERR_INVALID, "split exchange needs both begin and end");
  h->halo_begin = begin; h->halo_end = end; h->halo_user = user; drop_graph(h);
  return MGS_OK;
}
int mgs_hier_nlev(const mgs_hier *h) { return (int)h->lev.size(); }
int mgs_hier_level_shape(const mgs_hier *h, int l, int *rows, int64_t *nnz) {
  MGS_CHECK(h->ctx, l >= 0 && l < (int)h->lev.size(), MGS_ERR_INVALID, "level %d out of range", l);
  if (rows) *rows = h->lev[l].A->rows; if (nnz) *nnz = h->lev[l].A->nnz;
  return MGS_OK;
}
const mgs_csr *mgs_hier_level_A(const mgs_hier *h, int l) { return (l >= 0 && l < (int)h->lev.size()) ? h->lev[l].A : nullptr; }
const mgs_xfer *mgs_hier_level_P(const mgs_hier *h, int l) { return (l >= 0 && l < (int)h->lev.size()) ? h->lev[l].T : nullptr; }

static int push_level(mgs_hier *h, mgs_xfer *T, mgs_csr *Ac) {
  h->lev.back().T = T;
  h->lev.emplace_back();
  int rc = level_init(h, h->lev.back(), Ac, true);
  h->finalized = false; drop_graph(h);
  if (h->inv) { mgs_hip_free(h->inv); h->inv = nullptr; }
  return rc;
}

int mgs_hier_push_P(mgs_hier *h, const mgs_csr *P) {
  mgs_ctx *ctx = h->ctx;
  const mgs_csr *A = h->lev.back().A;
  MGS_CHECK(ctx, A->rows == A->cols, MGS_ERR_STATE, "mgs_hier_push_P: sharded operators take their hierarchy from mgs_hier_coarsen");
  MGS_CHECK(ctx, P->rows == A->rows, MGS_ERR_INVALID, "mgs_hier_push_P: P has %d rows, coarsest operator has %d", P->rows, A->rows);
  MGS_CHECK(ctx, P->cols > 0, MGS_ERR_INVALID, "mgs_hier_push_P: P has no columns");
  mgs_xfer *T = nullptr; mgs_csr *Ac = nullptr;
  MGS_TRY(k_xfer_from_csr(P, &T));
  int rc = mgs_csr_galerkin(A, T, &Ac);
  if (rc != MGS_OK) { mgs_xfer_destroy(T); return rc; }
  return push_level(h, T, Ac);
}

int mgs_aggregate_shard(const mgs_csr *A, double ktg, int npass, double tou, mgs_xfer **T) { return mgs_aggregate_shard_zoned(A, ktg, npass, tou, nullptr, T); }
// zone (host, one int per owned row; NULL: none): rows of different zones never share an aggregate.  The host side gives the rows a peer
// sees as halo a zone of their own: the aggregates that peer will ask for at the next level are then exactly the aggregates of those
// rows — a contiguous id range on plane shards, level after level — and every halo exchange sends ranges straight from the vectors.
int mgs_aggregate_shard_zoned(const mgs_csr *A, double ktg, int npass, double tou, const int *zone, mgs_xfer **T) {
  mgs_ctx *ctx = A->ctx;
  mgs_csr *Ac = nullptr;
  int *dz = nullptr;
  if (zone && A->rows) {
    MGS_TRY(mgs_dev_alloc(ctx, &dz, (size_t)A->rows));
    if (hipMemcpy(dz, zone, sizeof(int) * (size_t)A->rows, hipMemcpyHostToDevice) != hipSuccess) { mgs_hip_free(dz); return mgs_fail(ctx, MGS_ERR_HIP, "mgs_aggregate_shard_zoned: upload failed"); }
  }
  const int rc = k_pairwise_aggregate(A, ktg, npass, tou, T, &Ac, dz);
  hipStreamSynchronize(ctx->stream);
  if (dz) mgs_hip_free(dz);
  if (Ac) mgs_csr_destroy(Ac);
  return rc;
}
int mgs_galerkin_shard(const mgs_csr *A, const mgs_xfer *T, const int *halo_coarse_col, int n_halo_coarse, mgs_csr **Ac) {
  mgs_ctx *ctx = A->ctx;
  const int n_halo = A->cols - A->rows;
  MGS_CHECK(ctx, n_halo >= 0 && (n_halo == 0 || halo_coarse_col), MGS_ERR_INVALID, "mgs_galerkin_shard: halo map missing");
  if (n_halo == 0) return k_galerkin_agg(A, T, Ac);
  for (int k = 0; k < n_halo; ++k)
    MGS_CHECK(ctx, halo_coarse_col[k] == -1 || (halo_coarse_col[k] >= T->n_coarse && halo_coarse_col[k] < T->n_coarse + n_halo_coarse), MGS_ERR_INVALID,
              "mgs_galerkin_shard: halo slot %d maps to coarse column %d outside [%d,%d)", k, halo_coarse_col[k], T->n_coarse, T->n_coarse + n_halo_coarse);
  int *d = nullptr;
  MGS_TRY(mgs_dev_alloc(ctx, &d, (size_t)n_halo));
  MGS_HIP(ctx, hipMemcpyAsync(d, halo_coarse_col, sizeof(int) * (size_t)n_halo, hipMemcpyHostToDevice, ctx->stream));
  int rc = k_galerkin_agg_ext(A, T, d, n_halo_coarse, Ac);
  hipStreamSynchronize(ctx->stream);
  if (rc != MGS_OK) { mgs_hip_free(d); return rc; }
  // the map stays with the transfer: the fused post pass runs on A·P whose halo columns are the coarse level's own halo columns
  mgs_xfer *Tm = const_cast<mgs_xfer *>(T);
  if (Tm->halo_cmap) mgs_hip_free(Tm->halo_cmap);
  Tm->halo_cmap = d; Tm->n_halo_fine = n_halo; Tm->n_halo_coarse = n_halo_coarse;
  return MGS_OK;
}
int mgs_hier_push_level(mgs_hier *h, mgs_xfer *T, mgs_csr *Ac) {
  mgs_ctx *ctx = h->ctx;
  MGS_CHECK(ctx, T && Ac && T->n_fine == h->lev.back().A->rows && T->n_coarse == Ac->rows && Ac->rows <= Ac->cols, MGS_ERR_INVALID, "mgs_hier_push_level: shape mismatch");
  return push_level(h, T, Ac);
}
int mgs_xfer_from_agg(mgs_ctx *ctx, int n_fine, int n_coarse, const int *agg, mgs_xfer **out) { return k_xfer_from_agg_host(ctx, n_fine, n_coarse, agg, out); }
int mgs_hier_set_coarse_solver(mgs_hier *h, mgs_coarse_fn fn, void *user) { h->coarse = fn; h->coarse_user = user; drop_graph(h); if (fn) h->finalized = true; return MGS_OK; }

int mgs_hier_coarsen(mgs_hier *h, double ktg, int npass, double tou, int coarse_rows, int max_levels) {
  mgs_ctx *ctx = h->ctx;
  while ((int)h->lev.size() < max_levels && h->lev.back().A->rows > coarse_rows) {
    const mgs_csr *A = h->lev.back().A;
    MGS_CHECK(ctx, A->rows == A->cols, MGS_ERR_STATE, "mgs_hier_coarsen: operator is not square");
    mgs_xfer *T = nullptr; mgs_csr *Ac = nullptr;
    MGS_TRY(k_pairwise_aggregate(A, ktg, npass, tou, &T, &Ac));
    if (Ac->rows == 0 || Ac->rows > (int)(0.9 * A->rows)) {  // coarsening stalled (or everything in G0)
      mgs_xfer_destroy(T); mgs_csr_destroy(Ac);
      break;
    }
    MGS_TRY(push_level(h, T, Ac));
  }
  return MGS_OK;
}

int mgs_hier_finalize(mgs_hier *h) {
  mgs_ctx *ctx = h->ctx;
  const mgs_csr *Ac = h->lev.back().A;
  MGS_CHECK(ctx, Ac->rows == Ac->cols, MGS_ERR_STATE, "coarsest operator is not square");
  if (h->inv) { mgs_hip_free(h->inv); h->inv = nullptr; }
  h->nc = Ac->rows;
  h->coarse_sweeps = 0;
  if (Ac->rows > 8192) {
    // Coarsening stopped far above the dense limit (e.g. every row is in G0: the operator is so diagonally
    // dominant that Jacobi alone converges, AGMG.cpp:118-123).  The coarsest level is then smoothed
    // (8 damped-Jacobi sweeps from 0) instead of solved; the cycle stays a fixed linear operator.
    h->coarse_sweeps = 8;
    h->finalized = true; drop_graph(h);
    return MGS_OK;
  }
  MGS_TRY(k_dense_inverse(ctx, Ac, &h->inv));
  h->finalized = true; drop_graph(h);
  return MGS_OK;
}

int mgs_hier_fused_info(const mgs_hier *h, int level, int64_t out[6]) {
  MGS_CHECK(h->ctx, level >= 0 && level < (int)h->lev.size(), MGS_ERR_INVALID, "mgs_hier_fused_info: level %d out of range", level);
  const mgs_level &L = h->lev[level];
  auto coded = [](const mgs_rowcode *c) -> int64_t { return c ? c->coded_blocks : 0; };
  out[0] = (L.A->rows + 255) / 256; out[1] = L.val_wd != nullptr; out[2] = L.col_agg != nullptr || L.AP != nullptr;
  out[3] = coded(L.A->code); out[4] = coded(L.code_pre); out[5] = coded(L.AP ? L.code_ap : L.code_agg);
  return MGS_OK;
}

int mgs_hier_group_info(const mgs_hier *h, int level, int64_t out[4]) {
  MGS_CHECK(h->ctx, level >= 0 && level < (int)h->lev.size(), MGS_ERR_INVALID, "mgs_hier_group_info: level %d out of range", level);
  const mgs_groups *G = h->lev[level].grp;
  out[0] = G ? G->ngroups : 0; out[1] = G ? G->nblocks - G->ngroups : 0;   /* blocks merged into another block's group */ out[2] = G ? G->nstray : 0; out[3] = (h->lev[level].A->rows + 255) / 256;
  return MGS_OK;
}
int mgs_hier_graph_info(const mgs_hier *h, int64_t out[4]) {
  int n = 0; for (auto &g : h->graphs) n += g.exec != nullptr;
  out[0] = n; out[1] = (h->native || h->ntail) ? 1 : 0; out[2] = h->native_graph_failed ? 1 : 0; out[3] = h->native_eager_runs;
  return MGS_OK;
}

int64_t mgs_hier_vcycle_bytes(const mgs_hier *h) {
  // algorithmic bytes of the kernels one zero-guess cycle actually launches.
  // Every level starts from x = 0: the first pre-sweep is the 24n-byte (ωD⁻¹)b kernel, not a
  // Jacobi pass.  Fused form (V(1,1)): pass A reads the matrix, wd, b and writes r;
  // pass B reads the matrix, r, b, wd, agg, e_c and writes x.
  int64_t tot = 0;
  const int L = (int)h->lev.size();
  for (int l = 0; l < L - 1; ++l) {
    const mgs_level &lv = h->lev[l];
    int64_t n = lv.A->rows, nnz = lv.A->nnz, nc = h->lev[l + 1].A->rows, nnzP = lv.T ? lv.T->nnz : 0;
    const int64_t jac = 12 * nnz + 36 * n + 4, res = 12 * nnz + 28 * n + 4;
    const int64_t restr = 4 * (nc + 1) + 12 * nnzP + 8 * nc;
    const bool fused = h->ctx->opt_fuse && h->nu1 == 1 && h->nu2 == 1 && !h->halo && !h->halo_begin && lv.T && lv.T->aggregation &&
                       lv.A->rows == lv.A->cols;
    if (fused) { tot += (12 * nnz + 28 * n + 4) + restr + (12 * nnz + 40 * n + 8 * nc + 4); continue; }
    if (h->nu1 > 0) tot += 24 * n + (int64_t)(h->nu1 - 1) * jac + res;   // shortcut + sweeps + residual
    // ν1 = 0: r = b, no residual pass
    tot += restr;
    tot += (h->nu1 > 0 ? 20 * n : 12 * n) + 8 * nc;                        // prolong-add / prolong
    tot += (int64_t)h->nu2 * jac;
  }
  if (h->coarse_sweeps > 0) { const mgs_csr *Ac = h->lev.back().A; tot += 24 * (int64_t)Ac->rows + (int64_t)(h->coarse_sweeps - 1) * (12 * Ac->nnz + 36 * (int64_t)Ac->rows + 4); }
  else tot += (int64_t)h->nc * h->nc * 8 + 16 * (int64_t)h->nc;
  return tot;
}

}  // extern "C"

// The cycle body: pure kernel launches on ctx->stream (capturable into a hipGraph when no
// halo callback is installed).
//   ν1 × { x ← x + ωD⁻¹(b − Ax) };  r = b − Ax;  r_c = Pᵀr (bicg.cpp:48);  e_c = cycle(l+1, r_c) from 0
//   (coarsest: A_c⁻¹, bicg.cpp:35-36,48);  x ← x + P e_c (bicg.cpp:48);  ν2 × { x ← x + ωD⁻¹(b − Ax) }
// From x = 0 the first pre-sweep is x = (ωD⁻¹)b, the residual is b and x + P e_c is P e_c —
// bit-identical shortcuts that skip one SpMV-sized pass each.
int k_jacobi_zero(mgs_ctx *ctx, int n, double omega, const double *dinv, const double *b, double *x);

// Below opt_split_min_rows owned rows a level's kernel is too short to hide an exchange behind its interior row blocks:
// exchange first, then one launch (one launch less per pass).

static int halo_x(mgs_hier *h, int l, double *x) {
  if (h->lev[l].nx) return native_exchange(h, l, x, x + h->lev[l].A->rows);
  if (!h->halo) return MGS_OK;
  int rc = h->halo(h->halo_user, l, x);
  return rc ? mgs_fail(h->ctx, MGS_ERR_STATE, "halo exchange callback failed at level %d (%d)", l, rc) : MGS_OK;
}
// one SpMV-shaped kernel on level l with x's halo refreshed first; with split-phase callbacks the
// interior row blocks run while the exchange is in flight
static int sharded_op(mgs_hier *h, int l, const mgs_csr *A, int op, double *x, const double *b, const double *dinv, double omega, double *out) {
  if (h->lev[l].nx) {    // native RCCL exchange, then one launch
    MGS_TRY(native_exchange(h, l, x, x + A->rows));
    return mgs_launch_csr_op(A, op, x, b, dinv, omega, out);
  }
  if (!h->halo && !h->halo_begin) return mgs_launch_csr_op(A, op, x, b, dinv, omega, out);
  if (h->halo_begin && A->halo_split_ok && A->rows >= h->ctx->opt_split_min_rows) {
    const int nb = (A->rows + 255) / 256, lo = A->halo_lo_blocks, hi = nb - A->halo_hi_blocks;
    int rc = h->halo_begin(h->halo_user, l, x);
    if (rc) return mgs_fail(h->ctx, MGS_ERR_STATE, "halo exchange (begin) failed at level %d (%d)", l, rc);
    MGS_TRY(mgs_launch_csr_op_range(A, op, x, b, dinv, omega, out, lo, hi));
    rc = h->halo_end(h->halo_user, l, x);
    if (rc) return mgs_fail(h->ctx, MGS_ERR_STATE, "halo exchange (end) failed at level %d (%d)", l, rc);
    return mgs_launch_csr_op_range(A, op, x, b, dinv, omega, out, 0, lo + nb - hi, lo, hi - lo);   // leading + trailing boundary blocks, one launch
  }
  if (h->halo) MGS_TRY(halo_x(h, l, x));
  else {   // split callbacks only, but this level's halo readers are scattered: exchange, then one launch
    int rc = h->halo_begin(h->halo_user, l, x);
    if (!rc) rc = h->halo_end(h->halo_user, l, x);
    if (rc) return mgs_fail(h->ctx, MGS_ERR_STATE, "halo exchange failed at level %d (%d)", l, rc);
  }
  return mgs_launch_csr_op(A, op, x, b, dinv, omega, out);
}

static int cycle_level(mgs_hier *h, int l, const double *b, double *x, bool zero_guess);

// Approximate solve of the level-l problem A_l x = rhs from x = 0 for the level above.  V-cycle: one
// recursive cycle.  K-cycle (levels 1..kcycle_levels): two GCR steps preconditioned by that cycle
// (docs/AGMG_For_Convection_Diffusion.pdf §3.1; Fortran `nlvcyc`, src/CPU_Matlab/dagtwolev_mex.f90:59-61):
//   c1 = B rhs, v1 = A c1, r' = rhs − (α1/ρ1) v1;  c2 = B r', v2 = A c2, g = γ/ρ1;
//   x = (α1/ρ1) c1 + (α2/ρ2) (c2 − g c1),  ρ2 and α2 from the explicitly orthogonalised pair (c2 − g c1, v2 − g v1) — the paper's
//   ρ2 = β − γ²/ρ1 in exact arithmetic, without the difference of two nearly equal numbers.
static bool kcycle_here(const mgs_hier *h, int l) {
  const bool sharded = h->halo || h->halo_begin || h->native;
  // row shards: the five inner products are summed over the ranks — needs a reduction transport (native RCCL or the callback)
  return (l >= 1 ? l <= h->kcycle_levels : h->kcycle_entry) && l < (int)h->lev.size() - 1 && h->lev[l].kscal &&
         (!sharded || h->ctx->ncomm || h->ctx->allreduce);
}
// sum of `cnt` device scalars over the ranks of a row-sharded run (no-op on one GPU)
static int kc_allreduce(mgs_hier *h, double *dev, int cnt) {
  mgs_ctx *ctx = h->ctx;
  if (!(h->halo || h->halo_begin || h->native)) return MGS_OK;
  if (ctx->ncomm) return mgs_comm_allreduce_sum(ctx->ncomm, dev, (size_t)cnt);
  MGS_HIP(ctx, hipMemcpyAsync(ctx->red_host, dev, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->allreduce(ctx->allreduce_user, ctx->red_host, cnt)) return mgs_fail(ctx, MGS_ERR_STATE, "all-reduce callback failed");
  MGS_HIP(ctx, hipMemcpyAsync(dev, ctx->red_host, sizeof(double) * (size_t)cnt, hipMemcpyHostToDevice, ctx->stream));
  MGS_HIP(ctx, hipStreamSynchronize(ctx->stream));     // red_host is reused by the next reduction
  return MGS_OK;
}
static int sharded_op(mgs_hier *h, int l, const mgs_csr *A, int op, double *x, const double *b, const double *dinv, double omega, double *out);
static int coarse_solve_inner(mgs_hier *h, int l, const double *rhs, double *x);
// e_c for the level above; with an over-correction factor σ ≠ 1 (mgs_hier_set_correction_scale) it is scaled here, once, on the
// coarse vector (n_c entries), so every form of the level above — fused, grouped, unfused, K-cycle — sees σ·e_c
static int coarse_solve(mgs_hier *h, int l, const double *rhs, double *x) {
  if (l == 1 && h->coarse_launch && h->coarse_exec && rhs == h->lev[1].b->d && x == h->lev[1].x->d) {     // split launch (mgs_vcycle): everything below the fine level, one replay
    MGS_HIP(h->ctx, hipGraphLaunch(h->coarse_exec, h->ctx->stream));
    return MGS_OK;
  }
  MGS_TRY(coarse_solve_inner(h, l, rhs, x));
  if (h->corr_scale != 1.0) {
    // (the last sharded level's halo slots were filled from the replicated tail's solution together with the owned entries: scale them as well)
    const bool tail_halo = h->ntail && h->ntail->halo_global && l == (int)h->lev.size() - 1 && x == h->lev[l].x->d;
    MGS_TRY(k_axpby(h->ctx, tail_halo ? h->lev[l].n_ext : h->lev[l].n, h->corr_scale, x, 0.0, x));
  }
  return MGS_OK;
}
static int coarse_solve_inner(mgs_hier *h, int l, const double *rhs, double *x) {
  if (!kcycle_here(h, l)) return cycle_level(h, l, rhs, x, true);
  mgs_ctx *ctx = h->ctx;
  mgs_level &L = h->lev[l];
  const int n = L.n;
  double *sc = L.kscal;
  MGS_TRY(cycle_level(h, l, rhs, L.kc1->d, true));
  MGS_TRY(sharded_op(h, l, L.A, MGS_OP_SPMV, L.kc1->d, nullptr, nullptr, 0.0, L.kv1->d));      // halo of c1 refreshed on a row shard
  // GCR form (paper §3.1): inner products with v = A·c.  Energy form (option kcycle_energy, SPD operators; flexible-CG coefficients):
  // the same products with c as left factor — ρ1 = c1·Ac1, α1 = c1·rhs, γ = c2·Ac1; then, with the second direction orthogonalised
  // explicitly (c2' = c2 − (γ/ρ1)c1, v2' = v2 − (γ/ρ1)v1), ρ2 = d2'·v2' and α2 = d2'·r' (kernels_aux.hip: kc_orth_dots_kernel).
  const bool energy = ctx->opt_kcycle_energy != 0;
  const double *d1 = energy ? L.kc1->d : L.kv1->d, *d2 = energy ? L.kc2->d : L.kv2->d;
  const double *two[2] = {L.kv1->d, rhs};
  MGS_TRY(k_mdot(ctx, n, 2, d1, two, sc + 0, nullptr));                      // ρ1, α1: d1 read once
  MGS_TRY(kc_allreduce(h, sc, 2));
  MGS_TRY(k_kc_update_r(ctx, n, sc, rhs, L.kv1->d, L.kr->d));
  MGS_TRY(cycle_level(h, l, L.kr->d, L.kc2->d, true));
  MGS_TRY(sharded_op(h, l, L.A, MGS_OP_SPMV, L.kc2->d, nullptr, nullptr, 0.0, L.kv2->d));
  MGS_TRY(k_dot_dev(ctx, n, d2, L.kv1->d, sc + 2));                          // γ
  MGS_TRY(kc_allreduce(h, sc + 2, 1));
  MGS_TRY(k_kc_orth_dots(ctx, n, energy, sc, L.kc1->d, L.kc2->d, L.kv1->d, L.kv2->d, L.kr->d));   // ρ2, α2
  MGS_TRY(kc_allreduce(h, sc + 3, 2));
  return k_kc_combine(ctx, n, sc, L.kc1->d, L.kc2->d, x);
}

static int cycle_level(mgs_hier *h, int l, const double *b, double *x, bool zero_guess) {
  mgs_ctx *ctx = h->ctx;
  mgs_level &L = h->lev[l];
  const int n = L.n;
  if (l == (int)h->lev.size() - 1) {
    if (h->ntail) {                                     // replicated tail, native: all-gather the rhs, cycle, own slice back
      mgs_native_tail *T = h->ntail;
      const double *tb = T->b->d;
      if (T->even) {        // equal shards: gather straight from b; the gathered buffer is already in the tail's row order (two dispatches less per cycle)
        MGS_TRY(mgs_comm_allgather(T->comm, b, T->all, (size_t)T->maxn));
        tb = T->all;
      } else {
        if (T->n_loc) MGS_HIP(ctx, hipMemcpyAsync(T->send, b, sizeof(double) * (size_t)T->n_loc, hipMemcpyDeviceToDevice, ctx->stream));
        MGS_TRY(mgs_comm_allgather(T->comm, T->send, T->all, (size_t)std::max(T->maxn, 1)));
        MGS_TRY(k_gather(ctx, T->all, T->gidx, T->n_t, T->b->d));
      }
      if (h->capturing) MGS_TRY(coarse_solve_inner(T->tail, 0, tb, T->x->d));   // part of the outer graph (prepare_fused ran before the capture); K iteration at the entry level if asked for
      else { mgs_vec bv; bv.ctx = ctx; bv.n = T->n_t; bv.d = const_cast<double *>(tb); bv.owns = false; MGS_TRY(mgs_vcycle(T->tail, &bv, T->x, 1)); }
      // own slice back; with the global rows of this level's halo slots known (mgs_hier_set_native_tail_halo) the same kernel fills the halo
      // slots from the replicated solution — the level above then needs no exchange for e_c
      if (T->halo_global && x == L.x->d) MGS_TRY(k_tail_scatter(ctx, T->x->d, T->my_off, T->n_loc, T->halo_global, T->n_halo_global, x));
      else if (T->n_loc) MGS_HIP(ctx, hipMemcpyAsync(x, T->x->d + T->my_off, sizeof(double) * (size_t)T->n_loc, hipMemcpyDeviceToDevice, ctx->stream));
      return MGS_OK;
    }
    if (h->coarse) { int rc = h->coarse(h->coarse_user, b, x); return rc ? mgs_fail(ctx, MGS_ERR_STATE, "coarse solver callback failed (%d)", rc) : MGS_OK; }
    if (h->coarse_sweeps > 0) {                        // smoothed coarsest level (see mgs_hier_finalize)
      double *cur = x, *alt = L.tmp->d;
      if (!(h->coarse_sweeps & 1)) std::swap(cur, alt);   // odd number of buffer flips ends in x
      MGS_TRY(k_jacobi_zero(ctx, n, h->omega, L.dinv->d, b, cur));
      for (int s = 1; s < h->coarse_sweeps; ++s) {
        MGS_TRY(halo_x(h, l, cur));
        MGS_TRY(mgs_launch_csr_op(L.A, MGS_OP_JACOBI, cur, b, L.dinv->d, h->omega, alt));
        std::swap(cur, alt);
      }
      if (cur != x) MGS_HIP(ctx, hipMemcpyAsync(x, cur, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
      return MGS_OK;
    }
    return k_dense_gemv(ctx, h->nc, h->inv, b, x);
  }
  mgs_level &C = h->lev[l + 1];
  if (h->additive) {
    // additive form of solve(), reference src/common/bicg.cpp:59 with M2 = ωD⁻¹, level by level (zero guess only):
    //   x = P·cycle(l+1, Pᵀ b) + ωD⁻¹ b
    if (L.T->aggregation) MGS_TRY(k_restrict_agg(ctx, L.T->n_coarse, L.T->cptr, L.T->members, b, C.b->d));
    else MGS_TRY(mgs_launch_csr_op(L.T->Pt, MGS_OP_SPMV, b, nullptr, nullptr, 0.0, C.b->d));
    MGS_TRY(coarse_solve(h, l + 1, C.b->d, C.x->d));
    if (L.T->aggregation) MGS_TRY(k_prolong_agg(ctx, n, L.T->agg, C.x->d, x, 0));
    else MGS_TRY(mgs_launch_csr_op(L.T->P, MGS_OP_SPMV, C.x->d, nullptr, nullptr, 0.0, x));
    MGS_TRY(k_jacobi_zero(ctx, n, h->omega, L.dinv->d, b, L.tmp->d));
    return k_axpby(ctx, n, 1.0, L.tmp->d, 1.0, x);
  }
  // ---- fused form (aggregation P, V(1,1) from x = 0): two matrix passes, no separate (ωD⁻¹)b / prolong-add kernels.
  // Row shards: the only data of other ranks the two passes need are plain vector values — the peers' raw right-hand side of the
  // rows this shard sees as halo (pre pass: Â's halo columns carry the owners' ωD⁻¹, fetched once at setup) and the coarse level's
  // own halo of e_c (post pass: the halo columns of A·P are merged by REMOTE aggregate, i.e. they are the coarse level's halo
  // columns).  Both are ordinary halo exchanges — contiguous ranges on plane shards, sent without a pack kernel.
  const bool halo = L.A->cols > L.A->rows;
  const bool chalo = C.A->cols > C.A->rows;
  bool transport_ok = true;       // a shard level without halo columns (single rank, isolated shard) behaves like a square level
  if (halo) transport_ok = L.hbuf && L.halo_dinv && L.cmap_ext && (L.nx || h->halo_fused);
  if (chalo) transport_ok = transport_ok && (C.nx || h->halo_fused);
  const bool can_fuse = ctx->opt_fuse && zero_guess && h->nu1 == 1 && h->nu2 == 1 && L.wd && L.wd_omega == h->omega &&
                        L.T->aggregation && L.A->lds_cap > 0 && transport_ok;
  if (can_fuse) {
    const double *hv = halo ? L.hbuf->d : nullptr;
    const int nb = (L.A->rows + 255) / 256;
    // interior row blocks run while the halo values are in flight
    const bool split = halo && L.A->halo_split_ok && L.A->rows >= h->ctx->opt_split_min_rows;
    const int lo = split ? L.A->halo_lo_blocks : 0, hi = split ? nb - L.A->halo_hi_blocks : nb;
    // one pass whose halo values come from level q's exchange src → dst; launch(blk_lo, blk_hi, gap_at, gap_len) starts the kernel
    // on a range of row blocks (boundary blocks = those of L.A that read halo columns; A·P has them in the same rows)
    auto pass_with_exchange = [&](bool need, int q, const double *src, double *dst, auto launch) -> int {
      if (!need) return launch(0, nb, 0x7fffffff, 0);
      mgs_level &Q = h->lev[q];
      if (Q.nx && h->capturing && split && ctx->comm_stream) {      // captured cycle, option native_overlap: interior row blocks beside the exchange
        MGS_TRY(overlapped_exchange(h, q, src, dst, [&]() { return launch(lo, hi, 0x7fffffff, 0); }));
        return launch(0, lo + nb - hi, lo, hi - lo);
      }
      if (Q.nx) {      // native RCCL exchange, then one launch
        MGS_TRY(native_exchange(h, q, src, dst));
        return launch(0, nb, 0x7fffffff, 0);
      }
      int rc = h->halo_fused(h->halo_user, q, 2, src, nullptr, dst, 0);
      if (rc) return mgs_fail(ctx, MGS_ERR_STATE, "halo exchange of the fused passes (begin) failed at level %d (%d)", q, rc);
      if (split) MGS_TRY(launch(lo, hi, 0x7fffffff, 0));
      rc = h->halo_fused(h->halo_user, q, 2, src, nullptr, dst, 1);
      if (rc) return mgs_fail(ctx, MGS_ERR_STATE, "halo exchange of the fused passes (end) failed at level %d (%d)", q, rc);
      if (!split) return launch(0, nb, 0x7fffffff, 0);
      return launch(0, lo + nb - hi, lo, hi - lo);   // leading + trailing boundary blocks, one launch
    };
    auto exchange_now = [&](int q, const double *src, double *dst) -> int {      // exchange, nothing launched meanwhile
      if (h->lev[q].nx) return native_exchange(h, q, src, dst);
      int rc = h->halo_fused(h->halo_user, q, 2, src, nullptr, dst, 0);
      if (!rc) rc = h->halo_fused(h->halo_user, q, 2, src, nullptr, dst, 1);
      return rc ? mgs_fail(ctx, MGS_ERR_STATE, "halo exchange of the fused passes failed at level %d (%d)", q, rc) : MGS_OK;
    };
    // Setup-time operands: Â = A·diag(wd) makes the pre pass the plain residual kernel with x = b (one gather per
    // entry); A·P (or col_agg = the coarse column of every entry) lets the post pass gather e_c directly.  On a shard the
    // halo columns of Â read the payload buffer (the pattern-coded kernel knows that split); those of A·P read e_c's own halo.
    // Â's code: its own when the tuples carry values or halo tags, A's index-only code otherwise
    mgs_csr Ahat = *L.A; Ahat.val = L.val_wd; Ahat.owns = false;
    Ahat.code = halo ? L.code_pre : ((L.A->code && L.A->code->vtab) || L.code_hat ? L.code_hat : L.A->code);
    mgs_csr Amap = *L.A; Amap.val = L.A->val; Amap.col = L.col_agg; Amap.code = L.code_agg; Amap.owns = false;
    if (ctx->opt_diag_from_values && L.dpos) { Amap.dpos = L.dpos; Amap.dpos_omega = h->omega; }   // t-form post pass: ω/a_ii from the streamed values
    if (ctx->opt_merge_ap && L.AP) { Amap = *L.AP; Amap.code = L.code_ap; Amap.owns = false; }      // A·P, merged: fewer entries, wd read per row
    const bool operands = ctx->opt_fuse_operands && L.val_wd && (L.col_agg || (ctx->opt_merge_ap && L.AP)) &&
                          (!halo || mgs_rowcode_usable(&Ahat, true));
    const int *cmap = L.cmap_ext ? L.cmap_ext : L.T->agg;        // gather forms: coarse column of every local column
    double *ec = C.x->d, *ec_halo = C.x->d + C.n;
    // e_c's halo: one exchange on the coarse level's plan — unless that level is the replicated tail's, whose solution already holds the
    // neighbours' entries (the kernel that hands over the own slice fills the halo slots too)
    const bool ec_exchange = chalo && !(l + 2 == (int)h->lev.size() && h->ntail && h->ntail->halo_global);
    // post pass on the operand (A·P / aggregate-mapped A): the pattern-coded kernel where the code serves it, the gather kernel otherwise
    auto post_operand = [&](const double *bvec, const double *xin) -> int {
      return pass_with_exchange(ec_exchange, l + 1, ec, ec_halo, [&](int b0, int b1, int ga, int gl) {
        return mgs_launch_fused_range(&Amap, FUSE_POST_MAPPED, L.wd->d, bvec, xin, L.T->agg, ec, x, nullptr, nullptr, b0, b1, ga, gl); });
    };
    // Grouped form: pre pass + restriction in one kernel (r stays in LDS; L.r receives t = b + r, L.tmp the residuals of the
    // few rows whose aggregate leaves its row-block group), post pass in its t-form.
    // On a row shard the right-hand side of the halo rows is exchanged first (no interior/boundary split in this form).
    const bool grouped = ctx->opt_fuse_restrict && operands && L.grp && !(Ahat.code && Ahat.code->vtab) &&
                         (!halo || L.nx || !split || (h->capturing));
    if (grouped) {
      if (halo) MGS_TRY(exchange_now(l, b, L.hbuf->d));
      MGS_TRY(mgs_launch_group_pre(&Ahat, L.grp, L.T, b, b, L.r->d, L.tmp->d, C.b->d, hv, L.A->rows));
      MGS_TRY(coarse_solve(h, l + 1, C.b->d, C.x->d));
      if (ctx->opt_group_sweep & 1) Amap.sweep = L.grp;
      return post_operand(L.r->d, nullptr);
    }
    // small level (a dispatch costs what it costs, whatever it does): pre pass and restriction in one aggregate-parallel kernel
    if (ctx->opt_fuse_operands && L.val_wd && L.A->rows <= ctx->opt_aggpre_max_rows && L.A->max_row_len <= 64) {
      if (halo) MGS_TRY(exchange_now(l, b, L.hbuf->d));
      MGS_TRY(k_agg_pre(L.A, L.val_wd, b, hv, L.T, L.r->d, C.b->d));
      MGS_TRY(coarse_solve(h, l + 1, C.b->d, C.x->d));
      if (operands) return post_operand(L.r->d, b);
      return pass_with_exchange(ec_exchange, l + 1, ec, ec_halo, [&](int b0, int b1, int ga, int gl) {
        return mgs_launch_fused_range(L.A, FUSE_POST, L.wd->d, L.r->d, b, cmap, ec, x, nullptr, nullptr, b0, b1, ga, gl); });
    }
    // r = b − A·x1 with x1 = wd∘b (never stored: the POST pass recomputes it from b)
    if (operands && halo)
      MGS_TRY(pass_with_exchange(true, l, b, L.hbuf->d, [&](int b0, int b1, int ga, int gl) {
        return mgs_launch_coded_range(&Ahat, MGS_OP_RESIDUAL, b, b, nullptr, 0.0, nullptr, nullptr, L.r->d, hv, L.A->rows, b0, b1, ga, gl); }));
    else if (operands) MGS_TRY(mgs_launch_csr_op(&Ahat, MGS_OP_RESIDUAL, b, b, nullptr, 0.0, L.r->d));
    else
      MGS_TRY(pass_with_exchange(halo, l, b, halo ? L.hbuf->d : nullptr, [&](int b0, int b1, int ga, int gl) {
        return mgs_launch_fused_range(L.A, FUSE_PRE, L.wd->d, b, nullptr, nullptr, nullptr, L.r->d, nullptr, hv, b0, b1, ga, gl); }));
    MGS_TRY(k_restrict_agg(ctx, L.T->n_coarse, L.T->cptr, L.T->members, L.r->d, C.b->d));
    MGS_TRY(coarse_solve(h, l + 1, C.b->d, C.x->d));
    // x = x1 + Pe + wd∘(r − A·Pe)
    if (operands) return post_operand(L.r->d, b);
    return pass_with_exchange(ec_exchange, l + 1, ec, ec_halo, [&](int b0, int b1, int ga, int gl) {
      return mgs_launch_fused_range(L.A, FUSE_POST, L.wd->d, L.r->d, b, cmap, ec, x, nullptr, nullptr, b0, b1, ga, gl); });
  }
  // number of out-of-place sweeps decides where the ping-pong ends; start so that it ends in x
  int swaps = h->nu2 + (zero_guess ? (h->nu1 > 0 ? h->nu1 - 1 : 0) : h->nu1);
  double *cur = x, *alt = L.tmp->d;
  if (zero_guess && (swaps & 1)) { cur = L.tmp->d; alt = x; }
  bool zero = zero_guess;
  for (int s = 0; s < h->nu1; ++s) {
    if (zero) { MGS_TRY(k_jacobi_zero(ctx, n, h->omega, L.dinv->d, b, cur)); zero = false; }
    else {
      MGS_TRY(sharded_op(h, l, L.A, MGS_OP_JACOBI, cur, b, L.dinv->d, h->omega, alt));
      std::swap(cur, alt);
    }
  }
  const double *r = L.r->d;
  if (zero) r = b;                                       // r = b − A·0
  else MGS_TRY(sharded_op(h, l, L.A, MGS_OP_RESIDUAL, cur, b, nullptr, 0.0, L.r->d));
  // restriction
  if (L.T->aggregation) MGS_TRY(k_restrict_agg(ctx, L.T->n_coarse, L.T->cptr, L.T->members, r, C.b->d));
  else MGS_TRY(mgs_launch_csr_op(L.T->Pt, MGS_OP_SPMV, r, nullptr, nullptr, 0.0, C.b->d));
  MGS_TRY(coarse_solve(h, l + 1, C.b->d, C.x->d));
  // prolongation / correction
  if (L.T->aggregation) MGS_TRY(k_prolong_agg(ctx, n, L.T->agg, C.x->d, cur, zero ? 0 : 1));
  else if (zero) MGS_TRY(mgs_launch_csr_op(L.T->P, MGS_OP_SPMV, C.x->d, nullptr, nullptr, 0.0, cur));
  else {
    MGS_TRY(mgs_launch_csr_op(L.T->P, MGS_OP_SPMV, C.x->d, nullptr, nullptr, 0.0, L.r->d));
    MGS_TRY(k_axpby(ctx, n, 1.0, L.r->d, 1.0, cur));
  }
  for (int s = 0; s < h->nu2; ++s) {
    MGS_TRY(sharded_op(h, l, L.A, MGS_OP_JACOBI, cur, b, L.dinv->d, h->omega, alt));
    std::swap(cur, alt);
  }
  if (cur != x) MGS_HIP(ctx, hipMemcpyAsync(x, cur, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  return MGS_OK;
}

// halo of a level-l vector x (owned entries + halo room) through whatever transport the hierarchy has
static int halo_any(mgs_hier *h, int l, double *x) {
  if (h->lev[l].nx || h->halo) return halo_x(h, l, x);
  int rc = 0;
  if (h->halo_begin) { rc = h->halo_begin(h->halo_user, l, x); if (!rc) rc = h->halo_end(h->halo_user, l, x); }
  else if (h->halo_fused) { rc = h->halo_fused(h->halo_user, l, 2, x, nullptr, x + h->lev[l].A->rows, 0); if (!rc) rc = h->halo_fused(h->halo_user, l, 2, x, nullptr, x + h->lev[l].A->rows, 1); }
  else return mgs_fail(h->ctx, MGS_ERR_STATE, "level %d is a row shard but no halo exchange is installed", l);
  return rc ? mgs_fail(h->ctx, MGS_ERR_STATE, "halo exchange failed at level %d (%d)", l, rc) : MGS_OK;
}

// wd = ω·dinv of every level that can run the fused passes; allocated and filled outside any stream capture
static int prepare_fused(mgs_hier *h) {
  mgs_ctx *ctx = h->ctx;
  if (ctx->opt_rowcode)      // pattern codes of the level operators (a cache attached to the matrices)
    for (mgs_level &L : h->lev)
      if (!L.A->code_tried) { MGS_TRY(mgs_csr_optimize(const_cast<mgs_csr *>(L.A))); drop_graph(h); }
  for (int l = h->kcycle_entry ? 0 : 1; l <= h->kcycle_levels && l < (int)h->lev.size() - 1; ++l) {
    mgs_level &L = h->lev[l];
    if (L.kscal) continue;
    for (mgs_vec **q : {&L.kc1, &L.kv1, &L.kc2, &L.kv2, &L.kr}) MGS_TRY(mgs_vec_create(ctx, L.n_ext, q));
    MGS_TRY(mgs_dev_alloc(ctx, &L.kscal, 8));
    drop_graph(h);
  }
  const bool sharded = h->halo || h->halo_begin || h->native;
  if (!ctx->opt_fuse || h->nu1 != 1 || h->nu2 != 1 || (sharded && !h->halo_fused && !h->native)) return MGS_OK;
  for (size_t l = 0; l + 1 < h->lev.size(); ++l) {
    mgs_level &L = h->lev[l];
    if (!L.T || !L.T->aggregation) continue;
    if (L.A->rows != L.A->cols && !h->halo_fused && !L.nx) continue;
    const bool shard = L.A->cols > L.A->rows;
    if (!L.wd) MGS_TRY(mgs_vec_create(ctx, L.n_ext, &L.wd));
    if (shard && !L.hbuf) MGS_TRY(mgs_vec_create(ctx, L.A->cols - L.A->rows, &L.hbuf));
    if (shard && !L.halo_dinv) {
      // one exchange at setup: the owners' D⁻¹ of the rows this shard sees as halo.  With it Â's halo columns are scaled like the
      // owned ones, and what the pre pass needs from the peers per cycle is their raw right-hand side (no packed product).
      // Collective: every rank prepares its hierarchy in its first cycle.
      MGS_TRY(halo_any(h, (int)l, L.dinv->d));
      L.halo_dinv = true; L.wd_omega = 0.0; drop_graph(h);
    }
    if (shard && !L.cmap_ext && L.T->halo_cmap && L.T->n_halo_fine == L.A->cols - L.A->rows) {      // coarse column of every local column
      MGS_TRY(mgs_dev_alloc(ctx, &L.cmap_ext, (size_t)L.A->cols));
      MGS_TRY(k_concat_i32(ctx, L.T->agg, L.A->rows, L.T->halo_cmap, L.T->n_halo_fine, L.cmap_ext));
      drop_graph(h);
    }
    if (shard && !L.cmap_ext) continue;      // shard not built by mgs_galerkin_shard: one kernel per step on this level
    const bool rescale = L.wd_omega != h->omega;
    if (rescale) { MGS_TRY(k_axpby(ctx, L.n_ext, h->omega, L.dinv->d, 0.0, L.wd->d)); L.wd_omega = h->omega; drop_graph(h); }
    if (ctx->opt_fuse_operands) {      // derived CSR operands of the fused passes (same shape as A)
      bool new_vals = false;
      if (!L.val_wd) { MGS_TRY(mgs_dev_alloc(ctx, &L.val_wd, (size_t)L.A->nnz + 4)); MGS_TRY(k_scale_vals(ctx, L.A, L.wd->d, L.val_wd)); drop_graph(h); new_vals = true; }
      else if (rescale) { MGS_TRY(k_scale_vals(ctx, L.A, L.wd->d, L.val_wd)); new_vals = true; }
      const int ncols_c = h->lev[l + 1].A->cols;      // coarse level's local columns: its rows + its halo slots
      if (ctx->opt_merge_ap && !L.AP) {
        MGS_TRY(k_build_ap(L.A, L.T, L.cmap_ext, ncols_c, &L.AP)); drop_graph(h);
        L.AP->far_band = L.A->far_band; L.AP->far_band_max = L.A->far_band_max;     // sweep order of the launch: the band of A (AP's columns are coarse ids)
        if (ctx->opt_rowcode)
          MGS_TRY(mgs_build_rowcode(ctx, L.AP->rows, L.AP->rowptr, L.AP->col, L.T->agg, 0x7fffffff, &L.code_ap, ctx->opt_valcode ? L.AP->val : nullptr));
      }
      if (!ctx->opt_merge_ap && !L.col_agg) {
        MGS_TRY(mgs_dev_alloc(ctx, &L.col_agg, (size_t)L.A->nnz + 4)); MGS_TRY(k_map_cols(ctx, L.A, L.cmap_ext ? L.cmap_ext : L.T->agg, L.col_agg)); drop_graph(h);
        if (ctx->opt_rowcode)
          MGS_TRY(mgs_build_rowcode(ctx, L.A->rows, L.A->rowptr, L.col_agg, L.T->agg, 0x7fffffff, &L.code_agg, ctx->opt_valcode ? L.A->val : nullptr));
      }
      if (ctx->opt_diag_from_values && !L.dpos && !ctx->opt_merge_ap) {
        MGS_TRY(mgs_dev_alloc(ctx, &L.dpos, (size_t)L.A->rows)); MGS_TRY(k_diag_pos(L.A, L.dpos)); drop_graph(h);
      }
      if (ctx->opt_fuse_restrict && !L.grp_tried) {   // row-block groups of the grouped pre pass (null: level does not qualify)
        L.grp_tried = true; MGS_TRY(mgs_build_groups(ctx, L.A, L.T, &L.grp)); drop_graph(h);
      }
      // codes whose tuples depend on Â's values (or, on a shard, on the halo tags) follow val_wd
      if (new_vals && (L.code_pre || L.code_hat)) {     // whatever the options say NOW: a code built from the old Â must not survive it
        mgs_free_rowcode(L.code_pre); L.code_pre = nullptr; mgs_free_rowcode(L.code_hat); L.code_hat = nullptr; drop_graph(h);
      }
      if (ctx->opt_rowcode && new_vals && (shard || ctx->opt_valcode)) {
        drop_graph(h);
        // on a shard the pre pass reads b (owned entries only) + the payload: halo columns need tagged table words
        if (shard) MGS_TRY(mgs_build_rowcode(ctx, L.A->rows, L.A->rowptr, L.A->col, nullptr, L.A->rows, &L.code_pre, ctx->opt_valcode ? L.val_wd : nullptr));
        else MGS_TRY(mgs_build_rowcode(ctx, L.A->rows, L.A->rowptr, L.A->col, nullptr, 0x7fffffff, &L.code_hat, L.val_wd));
      }
    }
  }
  return MGS_OK;
}

extern "C" {

int mgs_vcycle(mgs_hier *h, const mgs_vec *b, mgs_vec *x, int zero_guess) {
  mgs_ctx *ctx = h->ctx;
  MGS_CHECK(ctx, h->finalized, MGS_ERR_STATE, "mgs_vcycle: call mgs_hier_finalize first");
  MGS_TRY(prepare_fused(h));
  mgs_level &L0 = h->lev[0];
  MGS_CHECK(ctx, b->n >= L0.n && x->n >= L0.n, MGS_ERR_INVALID, "mgs_vcycle: vectors shorter than the operator (%d rows)", L0.n);
  MGS_CHECK(ctx, b->d != x->d, MGS_ERR_INVALID, "mgs_vcycle: x must not alias b");
  MGS_CHECK(ctx, !h->additive || zero_guess, MGS_ERR_INVALID, "mgs_vcycle: the additive form (bicg.cpp:59) is a preconditioner application from x = 0 only");
  if (h->lev.size() == 1 && !h->coarse_sweeps) return cycle_level(h, 0, b->d, x->d, true);
  auto entry = [&](const double *bb, double *xx, bool zg) -> int {      // kcycle_entry: two Krylov steps on level 0 itself (zero guess only)
    return (h->kcycle_entry && zg) ? coarse_solve_inner(h, 0, bb, xx) : cycle_level(h, 0, bb, xx, zg);
  };
  // sharded level 0 needs halo room behind the owned entries: work in the level's own buffer
  double *xw = x->d;
  const bool staged = x->n < L0.n_ext;
  if (staged) { xw = L0.x->d; if (!zero_guess) MGS_HIP(ctx, hipMemcpyAsync(xw, x->d, sizeof(double) * (size_t)L0.n, hipMemcpyDeviceToDevice, ctx->stream)); }
  // Native transport (RCCL send/recv groups and the tail all-gather are enqueued on this stream by the cycle itself):
  // capturable like any kernel launch once RCCL has set up its connections — two eager cycles first.
  const bool native = h->native || h->ntail;
  bool native_ok = false;
  if (native && ctx->opt_native_graph && !h->native_graph_failed) {
    // every exchange of the cycle must be native (installed callbacks are then never reached): each level with halo
    // columns has a plan, and the coarsest level is either square or handed to the native tail
    native_ok = h->ntail ? mgs_comm_capturable(h->ntail->comm) : (!h->coarse && h->lev.back().A->rows == h->lev.back().A->cols);
    for (auto &q : h->lev) if (q.nx ? !mgs_comm_capturable(q.nx->comm) : q.A->cols > q.A->rows) native_ok = false;
    if (native_ok && h->native_eager_runs < 2) { ++h->native_eager_runs; native_ok = false; }
  }
  const bool use_graph = ctx->opt_graph && (native ? native_ok : (!h->halo && !h->halo_begin && !h->coarse));
  if (h->graph_epoch != ctx->opt_epoch) drop_graph(h);
  h->graph_epoch = ctx->opt_epoch;
  // Split launch: a K-cycle's graph has hundreds of nodes (1.1 ms from hipGraphLaunch to its first kernel at 512³, and a cache keyed by
  // (b, x) that a flexible Krylov method with ten direction vectors overruns).  The fine level's passes go out eagerly instead, and the
  // levels below — which only touch their own buffers — replay from ONE graph whose submission hides behind the fine level's pre pass.
  const bool split_launch = use_graph && !native && zero_guess && h->kcycle_levels > 0 && !h->kcycle_entry && !h->additive && h->lev.size() > 2 &&
                            ctx->opt_graph_split_rows > 0 && L0.n >= ctx->opt_graph_split_rows && h->lev[1].b && h->lev[1].x;
  if (split_launch) {
    if (!h->coarse_exec) {
      hipGraph_t g = nullptr;
      MGS_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      h->capturing = true;
      int rc = coarse_solve(h, 1, h->lev[1].b->d, h->lev[1].x->d);
      h->capturing = false;
      hipError_t e = hipStreamEndCapture(ctx->stream, &g);
      if (e == hipSuccess && rc == MGS_OK) { e = hipGraphInstantiate(&h->coarse_exec, g, nullptr, nullptr, 0); if (e != hipSuccess) h->coarse_exec = nullptr; }
      if (g) hipGraphDestroy(g);
      if (rc != MGS_OK) return rc;
      if (e != hipSuccess) return mgs_fail(ctx, MGS_ERR_HIP, "coarse-level capture: %s", hipGetErrorString(e));
    }
    h->coarse_launch = true;
    int rc = cycle_level(h, 0, b->d, xw, true);
    h->coarse_launch = false;
    MGS_TRY(rc);
  } else if (!use_graph) {
    MGS_TRY(entry(b->d, xw, zero_guess != 0));
  } else {
    const int zg = zero_guess != 0;
    mgs_hier::GraphSlot *slot = nullptr, *victim = &h->graphs[0];
    for (auto &g : h->graphs) {
      if (g.exec && g.b == b->d && g.x == xw && g.zero == zg) { slot = &g; break; }
      if (!g.exec) { if (victim->exec) victim = &g; }
      else if (victim->exec && g.stamp < victim->stamp) victim = &g;
    }
    if (!slot) {
      if (victim->exec) { hipGraphExecDestroy(victim->exec); *victim = mgs_hier::GraphSlot(); }
      if (h->ntail) MGS_TRY(prepare_fused(h->ntail->tail));      // allocations of the tail happen outside the capture
      if (native && ctx->opt_native_overlap && !ctx->comm_stream) MGS_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
      if (!ctx->opt_native_overlap && ctx->comm_stream) { hipStreamDestroy(ctx->comm_stream); ctx->comm_stream = nullptr; }
      h->fork_used = 0;
      hipGraph_t g = nullptr;
      MGS_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      h->capturing = true;
      int rc = entry(b->d, xw, zg != 0);
      h->capturing = false;
      hipError_t e = hipStreamEndCapture(ctx->stream, &g);
      if (e == hipSuccess && rc == MGS_OK) {
        e = hipGraphInstantiate(&victim->exec, g, nullptr, nullptr, 0);
        if (e != hipSuccess) victim->exec = nullptr;
      }
      if (g) hipGraphDestroy(g);
      if (rc != MGS_OK || e != hipSuccess) {
        if (!native) return rc != MGS_OK ? rc : mgs_fail(ctx, MGS_ERR_HIP, "cycle capture: %s", hipGetErrorString(e));
        // the transport could not be captured on this platform: eager launches from here on (same arithmetic)
        (void)hipGetLastError();
        h->native_graph_failed = true;
        ctx->err = std::string("native cycle not captured (") + (rc != MGS_OK ? ctx->err.c_str() : hipGetErrorString(e)) + "): eager launches";
        MGS_TRY(entry(b->d, xw, zg != 0));
        if (staged) MGS_HIP(ctx, hipMemcpyAsync(x->d, xw, sizeof(double) * (size_t)L0.n, hipMemcpyDeviceToDevice, ctx->stream));
        return MGS_OK;
      }
      victim->b = b->d; victim->x = xw; victim->zero = zg;
      slot = victim;
    }
    slot->stamp = ++h->graph_clock;
    MGS_HIP(ctx, hipGraphLaunch(slot->exec, ctx->stream));
  }
  if (staged) MGS_HIP(ctx, hipMemcpyAsync(x->d, xw, sizeof(double) * (size_t)L0.n, hipMemcpyDeviceToDevice, ctx->stream));
  return MGS_OK;
}

// Work vectors of the solvers come from the context and go back to it: a solve no longer pays 8 × (hipMalloc + memset + hipFree,
// the last one a device synchronisation) per call, and — what matters more — a repeated solve finds its vectors at the SAME
// addresses, so the captured cycles keyed by (rhs, out) are replayed instead of captured again.  The solvers write every vector
// before they read it; only the halo slots behind the owned entries of a row shard are cleared on reuse.
static int ws_get(mgs_ctx *ctx, int64_t n, int64_t owned, mgs_vec **out) {
  for (size_t i = ctx->ws_free.size(); i-- > 0;) {
    if (ctx->ws_free[i]->n == n) {
      *out = ctx->ws_free[i]; ctx->ws_free.erase(ctx->ws_free.begin() + (long)i);
      if (n > owned) MGS_HIP(ctx, hipMemsetAsync((*out)->d + owned, 0, sizeof(double) * (size_t)(n - owned), ctx->stream));
      return MGS_OK;
    }
  }
  return mgs_vec_create(ctx, n, out);
}
static void ws_put(mgs_ctx *ctx, mgs_vec *v) {
  if (!v) return;
  if (ctx->ws_free.size() >= 24) { mgs_vec *old = ctx->ws_free.front(); ctx->ws_free.erase(ctx->ws_free.begin()); mgs_vec_destroy(old); }
  ctx->ws_free.push_back(v);
}

// BiCGSTABiml, reference src/common/bicg.cpp:74-136, statement by statement on device vectors.
int mgs_bicgstab(const mgs_csr *A, mgs_vec *x, const mgs_vec *b, mgs_hier *h, int *max_iter, double *tol, int *status) {
  mgs_ctx *ctx = A->ctx;
  MGS_CHECK(ctx, max_iter && tol && status, MGS_ERR_INVALID, "mgs_bicgstab: NULL out parameter");
  MGS_TRY(mgs_csr_optimize(const_cast<mgs_csr *>(A)));
  const int n = A->rows, next = A->cols > n ? A->cols : n;
  MGS_CHECK(ctx, x->n >= n && b->n >= n, MGS_ERR_INVALID, "mgs_bicgstab: vectors shorter than %d", n);
  mgs_vec *p = 0, *phat = 0, *s = 0, *shat = 0, *t = 0, *v = 0, *r = 0, *rt = 0, *xe = 0;
  struct Guard { mgs_ctx *c; std::vector<mgs_vec **> vs; ~Guard() { hipStreamSynchronize(c->stream); for (auto q : vs) ws_put(c, *q); } } guard{ctx, {}};
  for (mgs_vec **q : {&p, &phat, &s, &shat, &t, &v, &r, &rt}) { MGS_TRY(ws_get(ctx, q == &phat || q == &shat ? next : n, n, q)); guard.vs.push_back(q); }
  // views of the owned part so BLAS-1 sizes agree
  auto view = [&](mgs_vec *full, mgs_vec &out) { out.ctx = ctx; out.n = n; out.d = full->d; out.owns = false; };
  mgs_vec xv, bv, phv, shv; view(x, xv); view(const_cast<mgs_vec *>(b), bv); view(phat, phv); view(shat, shv);
  const mgs_vec *xin = x;
  if (x->n < next) { MGS_TRY(ws_get(ctx, next, n, &xe)); guard.vs.push_back(&xe); MGS_TRY(mgs_vec_copy(&xv, xe)); xin = xe; }
  auto halo0 = [&](mgs_vec *w) -> int {
    if (!h) return 0;
    if (h->lev[0].nx) return native_exchange(h, 0, w->d, w->d + A->rows);
    if (h->halo) return h->halo(h->halo_user, 0, w->d);
    if (h->halo_begin) { int rc = h->halo_begin(h->halo_user, 0, w->d); return rc ? rc : h->halo_end(h->halo_user, 0, w->d); }
    return 0;
  };
  auto precond = [&](const mgs_vec *in, mgs_vec *out) -> int {       // M.solve (bicg.cpp:106,116)
    if (!h) { mgs_vec ov; view(out, ov); return mgs_vec_copy(in, &ov); }
    return mgs_vcycle(h, in, out, 1);
  };
  double rho_1 = 0, rho_2 = 0, alpha = 0, beta = 0, omega = 0, resid = 0, normb = 0, tmp = 0;
  MGS_TRY(mgs_nrm2(&bv, &normb));                                                     // :80
  if (halo0(const_cast<mgs_vec *>(xin))) return mgs_fail(ctx, MGS_ERR_STATE, "halo exchange failed");
  MGS_TRY(mgs_residual(A, xin, &bv, r));                                              // :82
  MGS_TRY(mgs_vec_copy(r, rt));                                                       // :83
  if (normb == 0.0) normb = 1;                                                        // :85-86
  double d2[2];
  MGS_TRY(k_dot2(ctx, n, r->d, r->d, rt->d, r->d, d2));             // ‖r‖² and (r̃,r) in one pass / one round trip
  if ((resid = std::sqrt(d2[0]) / normb) <= *tol) { *tol = resid; *max_iter = 0; *status = 0; return MGS_OK; }   // :88-92
  for (int i = 1; i <= *max_iter; ++i) {                                              // :94
    rho_1 = d2[1];                                                                    // :95 (computed with the last ‖r‖)
    if (rho_1 == 0) { *tol = std::sqrt(d2[0]) / normb; *status = 2; return MGS_OK; }  // :96-99
    if (i == 1) MGS_TRY(mgs_vec_copy(r, p));                                          // :100-101
    else {
      beta = (rho_1 / rho_2) * (alpha / omega);                                       // :103
      MGS_TRY(mgs_axpbypcz(1.0, r, -beta * omega, v, beta, p));                       // :104
    }
    MGS_TRY(precond(p, phat));                                                        // :106
    if (halo0(phat)) return mgs_fail(ctx, MGS_ERR_STATE, "halo exchange failed");
    MGS_TRY(mgs_spmv_dots(A, phat->d, v->d, rt->d, d2)); alpha = rho_1 / d2[0];      // :107-108  v = A·p̂ with r̃·v from the same pass
    MGS_TRY(k_update_dot2(ctx, n, 1.0, r->d, -alpha, v->d, s->d, nullptr, d2));      // :109 s = r − αv, with ‖s‖² in the same pass
    tmp = std::sqrt(d2[0]);
    if ((resid = tmp / normb) < *tol) {                                               // :110-115
      MGS_TRY(mgs_axpby(alpha, &phv, 1.0, &xv));
      *max_iter = i; *tol = resid; *status = 0; return mgs_sync(ctx);
    }
    MGS_TRY(precond(s, shat));                                                        // :116
    if (halo0(shat)) return mgs_fail(ctx, MGS_ERR_STATE, "halo exchange failed");
    MGS_TRY(mgs_spmv_dots(A, shat->d, t->d, s->d, d2)); omega = d2[0] / d2[1];       // :117-118  t = A·ŝ with (t·s, t·t) from the same pass
    MGS_TRY(mgs_axpbypcz(alpha, &phv, omega, &shv, 1.0, &xv));                        // :119
    MGS_TRY(k_update_dot2(ctx, n, 1.0, s->d, -omega, t->d, r->d, rt->d, d2));       // :120 r = s − ωt with ‖r‖² (:123) and the next (r̃,r) (:95)
    rho_2 = rho_1;                                                                    // :122
    if ((resid = std::sqrt(d2[0]) / normb) < *tol) { *tol = resid; *max_iter = i; *status = 0; return mgs_sync(ctx); }   // :123-127
    if (omega == 0) { *tol = resid; *status = 3; return mgs_sync(ctx); }              // :128-131
  }
  *tol = resid; *status = 1;                                                          // :134-135
  return mgs_sync(ctx);
}

// Flexible GCR(m): c_k = B_k r (variable preconditioner), v_k = A c_k orthogonalised against the v_j of the restart window, r ← r − α_k v_k.
// Device passes per iteration beside the preconditioner and the SpMV (round 3 ran 8 vector passes and 3 host round trips PER EARLIER
// DIRECTION — 39 % of a 512³ solve outside the preconditioner):
//   * one multi-dot pass: h_j = v_j·v_k for every j < k, v_k read once (classical Gram-Schmidt; the window is ≤ 64 and restarted);
//   * one update pass:    v_k ← v_k − Σ (h_j/ρ_j) v_j with ρ_k = v_k·v_k and t = v_k·r from the same pass;
//   * one residual pass:  r ← r − (t/ρ_k) v_k with ‖r‖²;
// and the directions c_k are NOT orthogonalised at all: with U_jk = h_j/ρ_j (strictly upper triangular) the orthogonalised directions are
// Ĉ = C (I + U)⁻¹, so x − x₀ = Ĉ α = C y with (I + U) y = α — a back substitution on k ≤ 64 numbers on the host and ONE pass
// x ← x + Σ y_j c_j when the window closes (restart, convergence, or the iteration limit).
// The recursively updated residual drifts from b − A·x (finite-precision orthogonality, and a K-cycle is a different operator at every
// call), so the TRUE residual b − A·x replaces it at every restart and decides every return with status 0: the method never reports a
// tolerance it has not reached.  (A sliding window instead of the restart was measured on the CPU restatement,
// tools/kcycle_diag_cpu.py, 64³: K-cycle on all levels 25 iterations restarted, 28–40 with a window of 10 — the restart stays.)
int mgs_fgcr(const mgs_csr *A, mgs_vec *x, const mgs_vec *b, mgs_hier *h, int restart, int *max_iter, double *tol, int *status) {
  mgs_ctx *ctx = A->ctx;
  MGS_CHECK(ctx, max_iter && tol && status && restart >= 1 && restart <= 64, MGS_ERR_INVALID, "mgs_fgcr: bad arguments");
  MGS_CHECK(ctx, A->rows == A->cols && x->n >= A->rows && b->n >= A->rows, MGS_ERR_INVALID, "mgs_fgcr: square unsharded operator required");
  MGS_TRY(mgs_csr_optimize(const_cast<mgs_csr *>(A)));
  const int n = A->rows;
  constexpr int CH = 16;                             // vectors per multi-vector pass (kernels_aux.hip: MDOT_MAX)
  struct Guard { mgs_ctx *c; std::vector<mgs_vec *> vs; ~Guard() { hipStreamSynchronize(c->stream); for (auto q : vs) ws_put(c, q); } } guard{ctx, {}};
  auto mk = [&](mgs_vec **q) -> int { int rc = ws_get(ctx, n, n, q); if (rc == MGS_OK) guard.vs.push_back(*q); return rc; };
  mgs_vec *r = nullptr; MGS_TRY(mk(&r));
  std::vector<mgs_vec *> C((size_t)restart, nullptr), V((size_t)restart, nullptr);
  std::vector<double> rho((size_t)restart, 0.0), alpha((size_t)restart, 0.0), U((size_t)restart * restart, 0.0), hj((size_t)restart, 0.0), y((size_t)restart, 0.0);
  mgs_vec xv; xv.ctx = ctx; xv.n = n; xv.d = x->d; xv.owns = false;
  mgs_vec bv; bv.ctx = ctx; bv.n = n; bv.d = b->d; bv.owns = false;
  double normb = 0, nr = 0;
  MGS_TRY(mgs_nrm2(&bv, &normb));
  if (normb == 0.0) normb = 1;
  auto true_residual = [&](double *resid_out) -> int {      // r = b − A·x, ‖r‖/‖b‖
    MGS_TRY(mgs_residual(A, &xv, &bv, r));
    MGS_TRY(mgs_nrm2(r, &nr));
    *resid_out = nr / normb;
    return MGS_OK;
  };
  // x ← x + Σ_{j<m} y_j c_j, (I + U) y = α over the m directions of the open window
  auto close_window = [&](int m) -> int {
    if (m <= 0) return MGS_OK;
    for (int k = m - 1; k >= 0; --k) { double s = alpha[k]; for (int q = k + 1; q < m; ++q) s -= U[(size_t)k * restart + q] * y[q]; y[k] = s; }
    for (int c0 = 0; c0 < m; c0 += CH) {
      const int K = std::min(CH, m - c0);
      const double *w[CH]; double cf[CH];
      for (int q = 0; q < K; ++q) { w[q] = C[c0 + q]->d; cf[q] = -y[c0 + q]; }
      MGS_TRY(k_maxpy_dot2(ctx, n, K, xv.d, w, cf, xv.d, nullptr, nullptr, nullptr));
    }
    return MGS_OK;
  };
  double resid = 0.0;
  MGS_TRY(true_residual(&resid));
  if (resid <= *tol) { *tol = resid; *max_iter = 0; *status = 0; return MGS_OK; }
  int it = 0;
  while (it < *max_iter) {
    int m = 0;                                       // directions in the open window
    bool restart_now = false;
    for (int k = 0; k < restart && it < *max_iter && !restart_now; ++k) {
      if (!C[k]) { MGS_TRY(mk(&C[k])); MGS_TRY(mk(&V[k])); }
      if (h) MGS_TRY(mgs_vcycle(h, r, C[k], 1)); else MGS_TRY(mgs_vec_copy(r, C[k]));
      MGS_TRY(mgs_spmv(A, C[k], V[k]));
      double d2[2];
      if (k == 0) {
        MGS_TRY(k_dot2(ctx, n, V[0]->d, V[0]->d, V[0]->d, r->d, d2));
      } else {
        for (int c0 = 0; c0 < k; c0 += CH) {           // every h_j against the SAME (not yet updated) v_k
          const int K = std::min(CH, k - c0);
          const double *w[CH];
          for (int q = 0; q < K; ++q) w[q] = V[c0 + q]->d;
          MGS_TRY(k_mdot(ctx, n, K, V[k]->d, w, nullptr, &hj[(size_t)c0]));
        }
        for (int j = 0; j < k; ++j) U[(size_t)j * restart + k] = rho[j] != 0.0 ? hj[j] / rho[j] : 0.0;
        for (int c0 = 0; c0 < k; c0 += CH) {
          const int K = std::min(CH, k - c0);
          const bool last = c0 + CH >= k;
          const double *w[CH]; double cf[CH];
          for (int q = 0; q < K; ++q) { w[q] = V[c0 + q]->d; cf[q] = U[(size_t)(c0 + q) * restart + k]; }
          MGS_TRY(k_maxpy_dot2(ctx, n, K, V[k]->d, w, cf, V[k]->d, last ? r->d : nullptr, nullptr, last ? d2 : nullptr));
        }
      }
      rho[k] = d2[0];
      ++it; m = k + 1;
      if (rho[k] == 0.0) {                             // A·c_k lies in the span of the window: nothing more to gain from it
        alpha[k] = 0.0;
        MGS_TRY(close_window(m));
        MGS_TRY(true_residual(&resid)); *status = resid < *tol ? 0 : 2; *tol = resid; *max_iter = it; return mgs_sync(ctx);
      }
      alpha[k] = d2[1] / rho[k];
      MGS_TRY(k_update_dot2(ctx, n, 1.0, r->d, -alpha[k], V[k]->d, r->d, nullptr, d2));   // r ← r − α v with ‖r‖²
      resid = std::sqrt(d2[0]) / normb;
      if (resid < *tol) {                         // the recursion says converged: the true residual decides
        MGS_TRY(close_window(m)); m = 0;
        MGS_TRY(true_residual(&resid));
        if (resid < *tol) { *tol = resid; *max_iter = it; *status = 0; return mgs_sync(ctx); }
        restart_now = true;                       // not yet: restart from the true residual (r holds it)
      }
    }
    if (m) {                                      // window full (or out of iterations): x catches up, r ← b − A·x
      MGS_TRY(close_window(m));
      MGS_TRY(true_residual(&resid));
      if (resid < *tol) { *tol = resid; *max_iter = it; *status = 0; return mgs_sync(ctx); }
    }
  }
  *status = resid < *tol ? 0 : 1; *tol = resid; *max_iter = it;
  return mgs_sync(ctx);
}

// ------------------------------------------------------------------ instrumentation
int mgs_time_kernel(const mgs_csr *A, int op, const mgs_vec *x, const mgs_vec *b, const mgs_vec *dinv, mgs_vec *out, int reps, double *ms) {
  mgs_ctx *ctx = A->ctx;
  MGS_CHECK(ctx, reps > 0 && ms && x && out, MGS_ERR_INVALID, "mgs_time_kernel: bad arguments");
  MGS_CHECK(ctx, op == MGS_OP_SPMV || (b && (op == MGS_OP_RESIDUAL || dinv)), MGS_ERR_INVALID, "mgs_time_kernel: missing operand");
  hipEvent_t e0, e1;
  MGS_HIP(ctx, hipEventCreate(&e0)); MGS_HIP(ctx, hipEventCreate(&e1));
  MGS_HIP(ctx, hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps; ++i)
    MGS_TRY(mgs_launch_csr_op(A, op, x->d, b ? b->d : nullptr, dinv ? dinv->d : nullptr, 0.6, out->d));
  MGS_HIP(ctx, hipEventRecord(e1, ctx->stream));
  MGS_HIP(ctx, hipEventSynchronize(e1));
  float t = 0; MGS_HIP(ctx, hipEventElapsedTime(&t, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  *ms = (double)t / reps;
  return MGS_OK;
}
int mgs_time_vcycle(mgs_hier *h, const mgs_vec *b, mgs_vec *x, int reps, double *ms) {
  mgs_ctx *ctx = h->ctx;
  MGS_CHECK(ctx, reps > 0 && ms, MGS_ERR_INVALID, "mgs_time_vcycle: bad arguments");
  hipEvent_t e0, e1;
  MGS_HIP(ctx, hipEventCreate(&e0)); MGS_HIP(ctx, hipEventCreate(&e1));
  MGS_HIP(ctx, hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps; ++i) MGS_TRY(mgs_vcycle(h, b, x, 1));
  MGS_HIP(ctx, hipEventRecord(e1, ctx->stream));
  MGS_HIP(ctx, hipEventSynchronize(e1));
  float t = 0; MGS_HIP(ctx, hipEventElapsedTime(&t, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  *ms = (double)t / reps;
  return MGS_OK;
}

}  // extern "C"
