// mgs_internal.hpp — private structures of libmgs.so (gfx950 only; no CPU fallback).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mgs.h"

constexpr int MGS_RED_VALS = 32;   // results one reduction can hand to the host (mgs_ctx::red_host)
struct mgs_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // scratch for reductions (device partials + pinned host landing zone)
  double *dot_part = nullptr; int64_t dot_part_cap = 0;   // per-row-block partials of the dots fused into the SpMV epilogue
  double *red_dev = nullptr;
  std::vector<struct mgs_vec *> ws_free;   // work vectors of the Krylov solvers, kept between solves (mgs_ctx_trim releases them)
  double *red_host = nullptr;       // 2·MGS_RED_VALS doubles, mapped + coherent: [0..MGS_RED_VALS) values, [MGS_RED_VALS] the ticket of the posted-result path
  double *red_host_dev = nullptr;   // the same buffer as the device sees it (NULL: not mappable, copy + synchronize)
  unsigned long long red_ticket = 0;
  int red_cap = 0;
  int n_cu = 256;
  // options
  int opt_xcd_remap = 1;
  int opt_nontemporal = 0;  // non-temporal loads of val/col (measured: no gain with the slice kernel)
  int opt_fuse = 1;         // fused V-cycle passes on square levels
  int opt_fuse_operands = 1; // precomputed operands Â = A·diag(wd), agg[col] for the fused passes (+12 B per entry of memory)
  int opt_spmv_variant = 0;  // 0 auto (stream), 1 force vector
  int opt_graph = 1;
  int opt_graph_split_rows = 1 << 20;   // K-cycle on an unsharded operator with at least this many rows: the fine level's two passes are launched eagerly and only the
                                        // levels below replay from ONE hipGraph that does not depend on (b, x) — its submission hides behind the fine level's pre pass (0: never)
  int opt_valcode = 0;   // pattern tuples include the VALUES (rows with equal index shape and equal values share a tuple): coded blocks stream no
                         // matrix entry at all.  Pays only where coefficients repeat (constant-coefficient / piecewise-constant operators): opt-in.
  int opt_nt_store = 1000000;  // streaming (`nt`) stores of the row-block kernels' outputs on operators with at least this many rows (0: never): SpMV −3.4 %, cycle −1.3 % at 512³
  int opt_split_min_rows = 400000;   // row shards: levels with fewer owned rows exchange first and launch once (no interior/boundary split)
  int opt_rowcode = 1;   // pattern-coded index (8 B per entry streamed instead of 12 where rows repeat their shape)
  int opt_blkptr = 1;    // row-block bounds from the compact blkptr array (0: from rowptr)
  int opt_lds_pad = 0;   // extra dynamic LDS bytes per workgroup (occupancy experiments only)
  int opt_strip = -1;  // strip-major sweep: -1 auto (32 row blocks), 0 off, >0 strip size in row blocks
  int opt_group_strip = 0;       // groups per strip of the grouped pre pass's strip-major sweep (0: from opt_strip)
  int opt_group_order = 0;       // > 0: the grouped pre pass visits its groups in the order of the ONE-BLOCK kernels' strip-major plane sweep, strips of this many
                                 // row blocks, through a table built with the groups (groups hold 1–4 blocks, so a period counted in groups drifts against the planes)
  int opt_merge_ap = 1;          // fused post pass on A·P (merged entries) instead of A with aggregate-mapped columns
  int opt_fuse_dots = 1;         // BiCGSTAB: r̃·v and (t·s, t·t) in the epilogue of the SpMV that produces v resp. t
  int opt_diag_from_values = 1;  // t-form post pass: ωD⁻¹ from the streamed diagonal entry (1 B per row of position) instead of the wd vector (8 B per row)
  int opt_fuse_restrict = 1;  // grouped pre pass: restriction inside the pre-smoothing/residual pass, post pass reads t = b + r
  int opt_group_min_blocks = 1024;  // ... levels with fewer row blocks keep the separate kernels
  int opt_group_sweep = 0;       // bit 0: the t-form post pass sweeps the grouped pre pass's row-block groups (one workgroup per group) instead of single blocks
  int opt_group_concurrent = 0;  // groups of ≤ 2 blocks: 512-thread workgroups sweep both blocks at once (csr_group2_pre_kernel; same bits; measured at 512³:
                                 // 6.59 ms per cycle against 6.50 for the sequential sweep of the same pairs and 6.13 for 4-block groups — kept for A/B)
  int opt_group_min_link = 1;   // ... aggregates two row blocks must share to be grouped (8: keeps a few odd boundary aggregates from pulling blocks of
                                // another plane into the group — 8 % less HBM traffic for that kernel, yet 2 % slower: fewer, fatter workgroups win)
  int opt_group_blocks = 4;    // ... row blocks per group (1..4); same-process A/B at 512³ (tools/studies_r1_r3/ab_group2.py): 4-block groups 6.21 ms per cycle,
                               // pairs 6.34 ms, separate kernels 6.64 ms
  int opt_group_stray_pct = 6; // ... unless more than this share of a level's aggregates leaves its row-block group
  int opt_post_results = 1;   // inner products reach the host through a mapped buffer + ticket the host polls (no copy engine, no interrupt)
  int opt_rowptr_scan = 1;    // coded row blocks take a row's entry range from its pattern's length (wave prefix sum, one rowptr load per wave) instead of
                              // two rowptr loads per row: 4 B per row less to stream (kernels_spmv.hip: coded_row_range)
  int opt_stage_unroll = 1;   // row-block kernels stage their value slice without a loop in front of the barrier (predicated 16-byte loads / LDS-DMA): −4 … −7 % per cycle
  int opt_blas1_pairs = 1;    // ... pairs per lane of those kernels: 1 = one-shot workgroups (0: capped persistent grid, k: k pairs per lane)
  int opt_blas1_vec = 1;      // axpby / axpbypcz / update+dots move 16 B per lane with four loads per stream in flight (same per-element bits)
  int opt_aggpre_max_rows = 300000;   // levels with at most this many rows run pre pass + restriction as ONE aggregate-parallel kernel (launch-bound sizes; same bits)
  int opt_emu_split_self = 0;   // tools/emulate_rank.py only: a packed exchange with the rank itself goes out as two messages (a middle rank has two neighbours)
  int opt_kcycle_energy = 0;  // K-cycle coefficients from energy inner products (flexible-CG form; SPD operators) instead of the GCR form of the paper
  int opt_native_graph = 1;   // row shards on the native RCCL transport: capture the whole cycle (exchanges included) in a hipGraph
  int opt_native_overlap = 0; // ... and, inside that graph, run the interior row blocks on a second stream beside pack + exchange (measured on one GPU:
                              // a graph with cross-stream edges costs 0.48 ms of host time per launch and 1.219 vs 1.155 ms per cycle — off by default)
  hipStream_t comm_stream = nullptr;   // second stream of the captured native cycle (fork/join become graph edges)
  unsigned long long opt_epoch = 0;   // bumped by every mgs_ctx_set_option: captured cycles of an older epoch are dropped
  struct mgs_comm *ncomm = nullptr;   // native RCCL all-reduce of the inner products (takes precedence over the callback)
  mgs_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
};

// pattern code of a CSR-shaped index array (kernels_spmv.hip): one byte per row + a small table per row block
struct mgs_rowcode {
  unsigned char *pid = nullptr;  // n: the row's tuple within its row block's table
  int *tptr = nullptr;           // nblocks+1: table slice of each row block in tab (empty: block keeps its index array)
  int *tab = nullptr;            // per coded block: pstart[npat], then the offset tuples
  double *vtab = nullptr;        // option valcode: the tuples' values, indexed like tab (rows are one tuple only if index AND values repeat)
  int tab_max = 0, tab_cap = 0;  // largest table / LDS budget covering 98.5 % of the coded blocks (ints)
  int coded_blocks = 0, nblocks = 0;
  int64_t tab_total = 0;
};

struct mgs_csr {
  mgs_ctx *ctx = nullptr;
  int rows = 0, cols = 0;
  int64_t nnz = 0;
  int *rowptr = nullptr;  // rows+1
  int *blkptr = nullptr;  // rowptr sampled every 256 rows (row-block bounds), built by mgs_plan_csr
  int *col = nullptr;     // nnz (+pad)
  double *val = nullptr;  // nnz (+pad)
  bool owns = true;
  // launch plan of the row-block stream kernel (computed at upload)
  int max_row_len = 0;
  int far_band = 0;      // typical (mean over rows) farthest owned column distance
  int far_band_max = 0;  // max |col - row| over owned columns
  int halo_lo_blocks = 0, halo_hi_blocks = 0;  // leading / trailing row blocks that read halo columns
  bool halo_split_ok = false;                  // no other block does → interior rows can overlap the exchange
  int lds_cap = 0;  // entries staged per block
  int max_block_nnz = 0;
  int max_wave_nnz = 0;   // max entries of a 64-row group
  mgs_rowcode *code = nullptr;   // pattern code of col (mgs_csr_optimize; owned unless this is a view)
  bool code_tried = false;
  // views made by mgs_spmv_dots only: (y·w1, y·y) of the product y = A·x accumulated in the SpMV's epilogue (one partial pair per row block)
  const double *dot_w1 = nullptr; double *dot_part = nullptr;
  const struct mgs_groups *sweep = nullptr;   // views only: launch the coded kernel group by group over these row-block groups (option group_sweep)
  const unsigned char *dpos = nullptr;   // views of the t-form post pass only (not owned): position of the diagonal inside every row, so the
  double dpos_omega = 0.0;               // kernel takes ω/a_ii from the values it streams anyway instead of reading wd (8 B → 1 B per row)
  int *origin = nullptr;   // coarse operators built by the device setup: the finest-level row each row descends from (its aggregate's
                           // leader, chained through the levels) — the index space the matching's tie-breaks work in; NULL = identity
};

struct mgs_vec {
  mgs_ctx *ctx = nullptr;
  int64_t n = 0;
  double *d = nullptr;
  bool owns = true;
};

struct mgs_xfer {
  mgs_ctx *ctx = nullptr;
  int n_fine = 0, n_coarse = 0;
  bool aggregation = false;
  // aggregation form
  int *agg = nullptr;      // n_fine, -1 = not aggregated (G0)
  int *cptr = nullptr;     // n_coarse+1: member list offsets (Pᵀ rowptr)
  int *members = nullptr;  // nnz(P): fine rows sorted by aggregate (Pᵀ col)
  int *corigin = nullptr;  // n_coarse: origin (see mgs_csr) of every aggregate, handed to the coarse operator
  int *halo_cmap = nullptr;   // row shards (mgs_galerkin_shard): coarse column (n_coarse + coarse halo slot, −1 = not aggregated) of every fine halo slot
  int n_halo_fine = 0, n_halo_coarse = 0;
  int64_t nnz = 0;
  // general form
  mgs_csr *P = nullptr, *Pt = nullptr;
};

// native RCCL transport (comm_rccl.hip)
struct mgs_comm;
struct mgs_xfer_op { const double *sptr; double *rptr; size_t count; int peer; };   // one ncclSend (sptr) or ncclRecv (rptr)
int mgs_comm_exchange_ops(mgs_comm *c, const mgs_xfer_op *ops, int nops);             // one group; ops of a peer are matched in order
int mgs_comm_allgather(mgs_comm *c, const double *send, double *recv, size_t count);
int mgs_comm_allreduce_sum(mgs_comm *c, double *buf, size_t count);
bool mgs_comm_capturable(const mgs_comm *c);   // false for the file-based stand-in of the tests (host-synchronous calls)
// peer-to-peer transport behind the same three operations (comm_p2p.hip): IPC-mapped windows, one kernel per exchange
struct mgs_p2p;
int mgs_p2p_create(mgs_ctx *ctx, int world, int rank, size_t slot_doubles, void *handle_out, mgs_p2p **out);
int mgs_p2p_connect(mgs_p2p *c, const void *handles);
void mgs_p2p_destroy(mgs_p2p *c);
int mgs_p2p_info(const mgs_p2p *c, long long out[6]);
int mgs_p2p_selftest(mgs_p2p *c, hipStream_t s, int rounds, long long *mismatches);
int mgs_p2p_exchange_ops(mgs_p2p *c, hipStream_t s, const mgs_xfer_op *ops, int nops);
int mgs_p2p_allgather(mgs_p2p *c, hipStream_t s, const double *send, double *recv, size_t count);
int mgs_p2p_allreduce_sum(mgs_p2p *c, hipStream_t s, double *buf, size_t count);
// halo plan of one sharded level for the native exchange
struct mgs_native_plan {
  mgs_comm *comm = nullptr;
  int *send_idx = nullptr;          // device: owned rows the peers need, peer after peer
  std::vector<int> scnt, rcnt;      // per peer
  int64_t ns = 0, nr = 0;
  double *sendbuf = nullptr;        // device, ns doubles
  // Pack-free form: a peer's send list that is a few contiguous row ranges (plane shards: one) is sent straight from the source
  // vector, one ncclSend per range, and the peer posts one ncclRecv per range — no pack kernel.  The ranges a rank sends are
  // its own decision (sseg, from send_idx); what it receives it is told by its peers (rseg, shipped by the host side at setup:
  // mgs_hier_native_send_segments → mgs_hier_set_native_recv_segments, which also switches this rank's sends to ranges).
  std::vector<std::vector<int>> sseg_start, sseg_len, rseg_len;   // per peer
  std::vector<std::vector<int>> pseg_len, prseg_len;              // per peer: message lengths of the packed form (one per peer; see emu_split_self)
  bool seg_ok = false;      // every peer's send list splits into at most MGS_MAX_SEG ranges
  bool use_seg = false;     // receive segmentation installed on this rank (collective call): sends go out as ranges
};
constexpr int MGS_MAX_SEG = 8;
struct mgs_hier;
// replicated coarse tail driven natively: all-gather of the right-hand side, tail cycle, own slice back
struct mgs_native_tail {
  mgs_comm *comm = nullptr;
  mgs_hier *tail = nullptr;         // not owned
  int maxn = 0, n_t = 0, my_off = 0, n_loc = 0;
  bool even = false;                // every rank holds maxn rows: the gathered buffer is the tail's right-hand side as it stands
  double *send = nullptr, *all = nullptr;   // maxn / world·maxn doubles
  int *gidx = nullptr;              // n_t: position of global tail row i in `all`
  mgs_vec *b = nullptr, *x = nullptr;
  int *halo_global = nullptr;       // device: global tail row of every halo slot of the last sharded level (mgs_hier_set_native_tail_halo)
  int n_halo_global = 0;
};

// Aggregate-complete row-block groups of a level (kernels_spmv.hip, "grouped pre pass"): one workgroup sweeps the 1–2 row
// blocks of a group, keeps their residuals in LDS and restricts every aggregate that lies inside the group — the
// restriction kernel and the residual vector's trip through HBM disappear.  Aggregates that leave their group
// ("strays": odd shapes at domain boundaries) are restricted afterwards from the residuals their member rows also store.
struct mgs_groups {
  int ngroups = 0, nblocks = 0, nstray = 0, plane_groups = 0, max_blocks = 1;
  int *gdesc = nullptr;                 // 32 ints per group: blocks[4], entry bounds lo[4]/hi[4], aggregate ranges lo[4]/hi[4]
  int *gblk = nullptr;                  // 4 per group: its row blocks, ascending (−1: unused slot)
  int *afirst = nullptr;                // nblocks+1: first aggregate whose first member lies in row block b (ids ascend with it)
  unsigned long long *acode = nullptr;  // n_coarse: count | first position | position deltas (10 bits each) in the group's 1024-entry
                                        // residual buffer; 0 = stray
  unsigned *wmask = nullptr;            // (n+31)/32 bit mask: rows that also store r (members of stray aggregates)
  int *stray = nullptr;                 // stray aggregate ids
  int *gorder = nullptr; int gorder_per_xcd = 0;   // option group_order: group of workgroup (xcd, idx) at [xcd·per_xcd + idx], −1 = none
};
struct mgs_level {
  const mgs_csr *A = nullptr;
  bool own_A = false;
  mgs_xfer *T = nullptr;  // to next level
  int n = 0;              // owned rows
  int n_ext = 0;          // owned + halo
  mgs_vec *dinv = nullptr, *r = nullptr, *tmp = nullptr;
  mgs_vec *wd = nullptr;       // ω·dinv (fused passes); row shards: n_ext entries, the halo part holds the owners' values
  mgs_vec *hbuf = nullptr;     // halo payload of the fused pre pass (row shards): the peers' raw b of the rows this shard sees as halo
  bool halo_dinv = false;      // row shards: dinv's halo part has been fetched from the owners (one exchange at setup)
  int *cmap_ext = nullptr;     // row shards: coarse column of every local column (owned: agg, halo slot: T->halo_cmap), n_ext ints
  double *val_wd = nullptr;    // setup-time operand of the fused pre pass: a_ij·wd_j, so A·(wd∘b) = Â·b needs one gather
  int *col_agg = nullptr;      // setup-time operand of the fused post pass: agg[col_ij], so (A·Pe) gathers e_c directly
  mgs_rowcode *code_agg = nullptr;   // pattern code of col_agg (offsets from agg[row])
  mgs_csr *AP = nullptr;             // setup-time operand of the fused post pass, option merge_ap: A·P with the entries of a row that fall into one
  mgs_rowcode *code_ap = nullptr;    // aggregate summed (5 instead of 7 entries per row on the 7-point operator) and its pattern code (offsets from agg[row])
  mgs_rowcode *code_pre = nullptr;   // row shards: pattern code of col with tagged halo words (pre pass reads b + payload)
  mgs_rowcode *code_hat = nullptr;   // option valcode: pattern code of (col, val_wd) for the pre pass on Â
  unsigned char *dpos = nullptr;     // position of the diagonal entry inside each row (255: none / beyond 254), see mgs_csr::dpos
  mgs_groups *grp = nullptr;         // aggregate-complete row-block groups (grouped pre pass = pre pass + restriction in one kernel)
  bool grp_tried = false;
  mgs_vec *kc1 = nullptr, *kv1 = nullptr, *kc2 = nullptr, *kv2 = nullptr, *kr = nullptr;   // K-cycle work vectors
  double *kscal = nullptr;     // K-cycle scalars (device)
  double wd_omega = 0.0;       // ω that wd was built with
  mgs_vec *b = nullptr, *x = nullptr;  // coarse-level rhs / solution (levels >= 1)
  mgs_native_plan *nx = nullptr;       // native RCCL halo exchange of this level (row shards)
};

struct mgs_hier {
  mgs_ctx *ctx = nullptr;
  std::vector<mgs_level> lev;
  double omega = 0.5;
  int nu1 = 1, nu2 = 1;
  bool finalized = false;
  int kcycle_levels = 0;   // levels 1..kcycle_levels solve their coarse problem with 2 GCR steps (K-cycle)
  bool kcycle_entry = false;   // ... and so does level 0 when the hierarchy is entered from x = 0 (replicated tail of a row-sharded hierarchy whose
                               // K-cycle reaches the last sharded level: that level IS the tail's level 0)
  double corr_scale = 1.0; // over-correction: x ← x + σ·P e_c (σ = 1: the reference's form, bicg.cpp:48)
  bool additive = false;   // bicg.cpp:59: multigrid_solve(v) + M2(v) instead of the multiplicative form (M2 = ωD⁻¹)
  // coarsest direct solve
  int nc = 0;
  double *inv = nullptr;  // nc*nc dense inverse (row-major)
  int coarse_sweeps = 0;  // > 0: coarsening stalled above the dense limit → the coarsest level is smoothed, not solved
  mgs_halo_fn halo = nullptr;
  void *halo_user = nullptr;
  mgs_halo_fn halo_begin = nullptr, halo_end = nullptr;   // split-phase exchange (overlap with interior rows)
  mgs_halo_fused_fn halo_fused = nullptr;                  // payload exchange of the fused passes on row shards
  mgs_native_tail *ntail = nullptr;  // native form of the replicated tail (takes precedence over `coarse`)
  bool native = false;               // some level exchanges natively: sharded semantics even without callbacks
  mgs_coarse_fn coarse = nullptr;   // replaces the dense coarsest solve (replicated tail of a sharded hierarchy)
  void *coarse_user = nullptr;
  // hipGraph cache of one V-cycle
  // small cache of captured cycles keyed by (b, x, zero_guess)
  struct GraphSlot { hipGraphExec_t exec = nullptr; const double *b = nullptr; double *x = nullptr; int zero = -1; unsigned long long stamp = 0; };
  static constexpr int kGraphSlots = 16;   // BiCGSTAB alternates two (rhs, out) pairs; flexible GCR(m) cycles through m directions (restart 10: ten pairs)
  GraphSlot graphs[kGraphSlots];
  unsigned long long graph_clock = 0;
  unsigned long long graph_epoch = 0;   // ctx->opt_epoch the cached graphs were captured under
  // native transport under capture: the first cycles run eagerly (RCCL sets up its peer connections lazily, which a
  // capturing stream does not allow); a failed capture switches the hierarchy back to eager launches for good
  int native_eager_runs = 0;
  bool native_graph_failed = false;
  bool capturing = false;
  hipGraphExec_t coarse_exec = nullptr;  // split launch (opt_graph_split_rows): coarse_solve(level 1) on the level's own buffers, captured once
  bool coarse_launch = false;            // set around the eager fine-level passes: coarse_solve(h, 1, ..) replays coarse_exec
  std::vector<hipEvent_t> fork_events;   // fork/join events of the captured native cycle (two per overlapped exchange)
  size_t fork_used = 0;
};

// ------------------------------------------------------------------ device memory arena (mgs_api.hip)
// Every device allocation of the library goes through these two (never hipMalloc / hipFree directly: tests/test_abi.py greps for it).  With an arena (MGS_ARENA_GB = N, or mgs_arena_reserve) the library takes
// ONE hipMalloc of N GiB at its first allocation and places operators and vectors inside it (first fit, lowest address first, 2 MiB
// alignment for anything of 1 MiB or more, blocks coalesced on release); requests that do not fit fall through to hipMalloc.  Without
// one (default) they are hipMalloc / hipFree.  Why: DESIGN.md §5 "process to process".
hipError_t mgs_hip_malloc(void **p, size_t bytes);
hipError_t mgs_hip_free(void *p);

// ------------------------------------------------------------------ error plumbing
extern thread_local std::string g_mgs_last_error;
int mgs_fail(mgs_ctx *ctx, int code, const char *fmt, ...);

#define MGS_HIP(ctx, call)                                                              \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess)                                                               \
      return mgs_fail((ctx), MGS_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                      hipGetErrorString(e_), __FILE__, __LINE__);                       \
  } while (0)

#define MGS_CHECK(ctx, cond, code, ...)                                                 \
  do {                                                                                  \
    if (!(cond)) return mgs_fail((ctx), (code), __VA_ARGS__);                           \
  } while (0)

#define MGS_TRY(expr)                                                                   \
  do {                                                                                  \
    int rc_ = (expr);                                                                   \
    if (rc_ != MGS_OK) return rc_;                                                      \
  } while (0)

// ------------------------------------------------------------------ kernel launchers
// (kernels_spmv.hip)
enum { MGS_OP_SPMV = 0, MGS_OP_RESIDUAL = 1, MGS_OP_JACOBI = 2, FUSE_PRE = 3, FUSE_POST = 4,
       FUSE_POST_MAPPED = 5 /* FUSE_POST whose column array already holds agg[col] */ };
int mgs_launch_fused(const mgs_csr *A, int which, const double *wd, const double *bvec, const double *xin, const int *agg,
                     const double *ec, double *out, double *out2);
int mgs_launch_csr_op(const mgs_csr *A, int op, const double *x, const double *b,
                      const double *dinv, double omega, double *out);
// row blocks [blk_lo, blk_hi) of a virtual numbering in which the blocks >= gap_at are shifted by gap_len (one launch over
// the leading and trailing boundary blocks of a row shard: range [0, lo + nb − hi), gap_at = lo, gap_len = hi − lo)
int mgs_launch_csr_op_range(const mgs_csr *A, int op, const double *x, const double *b,
                            const double *dinv, double omega, double *out, int blk_lo, int blk_hi, int gap_at = 0x7fffffff, int gap_len = 0);
int mgs_launch_fused_range(const mgs_csr *A, int which, const double *wd, const double *bvec, const double *xin, const int *agg,
                           const double *ec, double *out, double *out2, const double *hv, int blk_lo, int blk_hi,
                           int gap_at = 0x7fffffff, int gap_len = 0);
int k_scale_vals(mgs_ctx *ctx, const mgs_csr *A, const double *wd, double *out);            // out_k = a_k·wd[col_k] (row shards: wd covers the halo columns too)
int k_map_cols(mgs_ctx *ctx, const mgs_csr *A, const int *cmap, int *out);                     // out_k = cmap[col_k]
int k_concat_i32(mgs_ctx *ctx, const int *a, int na, const int *b, int nb, int *out);
int k_tail_scatter(mgs_ctx *ctx, const double *xt, int my_off, int n_loc, const int *halo_global, int n_halo, double *x);   // x[0..n_loc) = xt[my_off..], x[n_loc+s] = xt[halo_global[s]]
int mgs_plan_csr(mgs_csr *A);
int mgs_build_rowcode(mgs_ctx *ctx, int n, const int *rowptr, const int *idx, const int *base, int split, mgs_rowcode **out,
                      const double *val = nullptr);
bool mgs_rowcode_usable(const mgs_csr *A, bool any = false);
int mgs_launch_coded_range(const mgs_csr *A, int op, const double *x, const double *b, const double *dinv, double omega,
                           const double *xin, const int *agg, double *out, const double *hv, int split, int blk_lo, int blk_hi,
                           int gap_at = 0x7fffffff, int gap_len = 0);
void mgs_free_rowcode(mgs_rowcode *c);
int mgs_build_groups(mgs_ctx *ctx, const mgs_csr *A, const mgs_xfer *T, mgs_groups **out);
void mgs_free_groups(mgs_groups *g);
// t = b + (b − Â·x) and r_c = Pᵀ(b − Â·x) in one pass over the groups of G; x = gather source (b itself, or with hv/split the
// halo payload of a row shard); strays' member rows also store r into r_out, restricted by the trailing small kernel
int mgs_launch_group_pre(const mgs_csr *Ahat, const mgs_groups *G, const mgs_xfer *T, const double *x, const double *b,
                         double *t_out, double *r_out, double *rc_out, const double *hv, int split);
// (kernels_aux.hip)
int k_diag_inv(const mgs_csr *A, double *dinv, int *bad_count_host);
int k_diag_pos(const mgs_csr *A, unsigned char *dpos);
int k_restrict_agg(mgs_ctx *ctx, int nc, const int *cptr, const int *members, const double *r, double *rc);
int k_agg_pre(const mgs_csr *A, const double *valhat, const double *b, const double *hv, const mgs_xfer *T, double *r_out, double *rc_out);   // small levels: pre pass + restriction, one dispatch
int k_prolong_agg(mgs_ctx *ctx, int n, const int *agg, const double *ec, double *x, int add);
int k_fill(mgs_ctx *ctx, double *d, int64_t n, double v);
int k_rand(mgs_ctx *ctx, double *d, int64_t n, uint64_t seed, int64_t off);
int k_axpby(mgs_ctx *ctx, int64_t n, double a, const double *x, double b, double *y);
int k_axpbypcz(mgs_ctx *ctx, int64_t n, double a, const double *x, double b, const double *y, double c, double *z);
int k_dot(mgs_ctx *ctx, int64_t n, const double *x, const double *y, double *out_host);
int k_dense_gemv(mgs_ctx *ctx, int n, const double *M, const double *b, double *x);
int k_dot_dev(mgs_ctx *ctx, int64_t n, const double *x, const double *y, double *out_dev);
int k_update_dot2(mgs_ctx *ctx, int64_t n, double a, const double *x, double b, const double *y, double *z, const double *w, double *out_host2);
int k_dot2(mgs_ctx *ctx, int64_t n, const double *x, const double *y, const double *z, const double *w, double *out_host2);
int mgs_ensure_dot_part(mgs_ctx *ctx, int64_t doubles);   // per-workgroup partial pairs of one-shot reduction launches
int k_dot2_finish(mgs_ctx *ctx, int nb, const double *part /*[2][nb]*/, double *out_host2);   // second stage + rank reduction + host copy
// y = A·x with (y·w1, y·y) from the same pass where the pattern-coded kernel serves A (else SpMV + k_dot2): BiCGSTAB's
// v = A·p̂ with r̃·v (bicg.cpp:107-108) and t = A·ŝ with (t·s, t·t) (bicg.cpp:117-118)
int mgs_spmv_dots(const mgs_csr *A, const double *x, double *y, const double *w1, double *out_host2);
int k_mdot(mgs_ctx *ctx, int64_t n, int K, const double *a, const double *const *b, double *out_dev, double *out_host);   // a·b_k, k < K ≤ 16, one pass
int k_maxpy_dot2(mgs_ctx *ctx, int64_t n, int K, const double *x, const double *const *w, const double *coef, double *z, const double *u, const double *u2, double *out_host2);
int k_kc_update_r(mgs_ctx *ctx, int n, const double *scal, const double *r, const double *v1, double *rp);
int k_kc_orth_dots(mgs_ctx *ctx, int n, bool energy, double *scal, const double *c1, const double *c2, const double *v1, const double *v2, const double *rp);
int k_kc_combine(mgs_ctx *ctx, int n, const double *scal, const double *c1, const double *c2, double *x);
int k_dense_inverse(mgs_ctx *ctx, const mgs_csr *A, double **inv_out);
int k_poisson3d(mgs_ctx *ctx, int N, int plane_lo, int plane_hi, int local_cols, mgs_csr **out);
int k_poisson2d(mgs_ctx *ctx, int n, mgs_csr **out);
int k_gather(mgs_ctx *ctx, const double *x, const int *idx, int64_t n, double *out);
int k_xfer_from_csr(const mgs_csr *P, mgs_xfer **out);
int k_transpose(const mgs_csr *A, mgs_csr **out);
// (setup_agmg.hip)
int k_exclusive_scan_i32(mgs_ctx *ctx, const int *in, int *out, int64_t n, int64_t *total_host);
int k_galerkin_agg(const mgs_csr *A, const mgs_xfer *T, mgs_csr **out);
int k_build_ap(const mgs_csr *A, const mgs_xfer *T, const int *cmap_ext, int ncols, mgs_csr **out);   // cmap_ext = NULL: square level, T->agg
int k_galerkin_agg_ext(const mgs_csr *A, const mgs_xfer *T, const int *halo_map_dev, int n_halo_c, mgs_csr **out);
int k_xfer_from_agg_host(mgs_ctx *ctx, int n_fine, int n_coarse, const int *agg_host, mgs_xfer **out);
int k_galerkin_general(const mgs_csr *A, const mgs_xfer *T, mgs_csr **out);
int k_pairwise_aggregate(const mgs_csr *A, double ktg, int npass, double tou, mgs_xfer **T_out, mgs_csr **Ac_out, const int *zone_dev = nullptr);

// helpers (mgs_api.hip)
int mgs_csr_alloc(mgs_ctx *ctx, int rows, int cols, int64_t nnz, mgs_csr **out);
template <class T>
static inline int mgs_dev_alloc(mgs_ctx *ctx, T **p, size_t count) {
  *p = nullptr;
  hipError_t e = mgs_hip_malloc((void **)p, sizeof(T) * (count ? count : 1));
  if (e != hipSuccess) return mgs_fail(ctx, MGS_ERR_ALLOC, "hipMalloc(%zu bytes): %s", sizeof(T) * count, hipGetErrorString(e));
  return MGS_OK;
}
static inline int mgs_grid(int64_t work, int block) { return (int)((work + block - 1) / block); }
