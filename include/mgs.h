/*
 * mgs.h — C ABI of libmgs.so: the MI355X (gfx950) solve-phase hot path of an
 * aggregation-based AMG V-cycle behind the surface of mishraiiit/MultiGridSolver.
 *
 * The reference has no FFI boundary (one translation unit per
 * program); the surface it exposes is source level.  Every entry point below names the
 * reference interface (file:line, relative to the reference checkout) it replaces or
 * serves.  The C++ face that keeps the reference's spelling (readMatrix, SMatrix,
 * MultiGridPrecond(A,P).solve(v), BiCGSTABiml) is multigridsolver_amd/cpp/mgs_host.hpp,
 * written on top of this header only.
 *
 * Conventions
 *   - plain pointers and sizes only; opaque handles; every call returns int
 *     (MGS_OK = 0, negative = error class) and records a message retrievable with
 *     mgs_last_error().
 *   - matrices: CSR, f64 values, int32 indices, columns sorted inside a row — the layout
 *     of `typedef SparseMatrix<double,RowMajor> SMatrix` (src/common/MatrixIO.cpp:10).
 *   - host arrays are caller-owned and copied on upload; device memory is owned by the
 *     library and released by the matching *_destroy.
 *   - one context = one HIP device + one stream; calls are asynchronous on that stream
 *     unless they return a host scalar or copy to host memory.  A context is not
 *     thread-safe.
 *   - there is NO CPU fallback: every compute entry point runs HIP kernels on the
 *     context's device and fails with MGS_ERR_HIP if that is impossible.
 */
#ifndef MGS_H
#define MGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGS_OK 0
#define MGS_ERR_INVALID (-1) /* bad argument / shape mismatch                        */
#define MGS_ERR_HIP (-2)     /* HIP runtime failure (no device, launch error, ...)   */
#define MGS_ERR_IO (-3)      /* file open / parse failure                            */
#define MGS_ERR_ALLOC (-4)   /* host or device allocation failed                     */
#define MGS_ERR_NUMERIC (-5) /* zero diagonal, singular coarse operator, ...         */
#define MGS_ERR_STATE (-6)   /* call sequence error (e.g. V-cycle before finalize)   */

typedef struct mgs_ctx mgs_ctx;
typedef struct mgs_csr mgs_csr;   /* device CSR matrix                               */
typedef struct mgs_vec mgs_vec;   /* device f64 vector                               */
typedef struct mgs_xfer mgs_xfer; /* prolongation P with its restriction Pᵀ          */
typedef struct mgs_hier mgs_hier; /* multilevel hierarchy = the preconditioner state */

/* ------------------------------------------------------------------ context */
/* Device memory arena (optional; no reference counterpart).  One hipMalloc of `bytes` taken before the library's first device allocation;
 * operators, vectors and setup scratch are then placed inside it (first fit from the low end, 2 MiB alignment for blocks of 1 MiB or more,
 * coalescing on release) and requests that do not fit fall through to hipMalloc.  Same allocation sequence → same addresses in every
 * process: the fine-level SpMV's process-to-process spread shrinks from ±3–4 % to ±1 %.  Environment MGS_ARENA_GB=N does
 * the same at the first allocation; mgs_arena_reserve(0) rules an arena out.  mgs_arena_info: capacity, bytes in use, largest free block. */
int mgs_arena_reserve(size_t bytes);
int mgs_arena_info(size_t out[3]);

/* device: HIP device ordinal.  stream: a hipStream_t to launch on (e.g. the caller's
 * torch stream) or NULL to let the context create its own.                          */
int mgs_ctx_create(int device, void *stream, mgs_ctx **out);
int mgs_ctx_destroy(mgs_ctx *ctx);
const char *mgs_last_error(const mgs_ctx *ctx); /* ctx may be NULL: last global error */
int mgs_sync(mgs_ctx *ctx);                      /* hipStreamSynchronize              */
/* Releases the memory the context keeps between calls (the Krylov solvers' work vectors: eight vectors of the operator's size
 * after a BiCGSTABiml solve, bicg.cpp:75 declares them per call; the scratch of the fused inner products). */
int mgs_ctx_trim(mgs_ctx *ctx);
void *mgs_ctx_stream(mgs_ctx *ctx);              /* the hipStream_t in use            */
const char *mgs_version(void);

/* ------------------------------------------------ L0 I/O: Matrix-Market loader (host) */
/* readMatrix  — src/common/MatrixIO.cpp:12-37 (twin: src/GPU_CUDAC++/MatrixIO.cu:182-280):
 * leading '%' lines skipped, "M N L", L triples "i j v" 1-based in any order, bucketed by
 * row, each row sorted by column; coordinate real general only.  Output arrays are
 * malloc'ed by the library; release each with mgs_host_free.  Unlike the reference a
 * missing/short file is an error (MGS_ERR_IO) instead of undefined behaviour.          */
int mgs_mtx_read(const char *path, int *rows, int *cols, int *nnz,
                 int **rowptr, int **col, double **val);
/* writeMatrix — src/common/MatrixIO.cpp:39-57: banner "%%MatrixMarket matrix coordinate
 * real general " + "rows cols nnz" + row-major 1-based triples, default ostream precision
 * (6 significant digits).                                                             */
int mgs_mtx_write(const char *path, int rows, int cols, int nnz,
                  const int *rowptr, const int *col, const double *val);
void mgs_host_free(void *p);

/* ------------------------------------------------------------ device CSR (row a1/a10) */
/* Replaces deepCopyMatrixCSRCPUtoGPU / deepCopyMatrixCSRGPUtoCPU
 * (src/GPU_CUDAC++/MatrixOperations.cu:121-146,162-171): sizes stay on the host, raw
 * device arrays inside.  nnz is int64 in the signatures for headroom; indices are int32
 * so nnz must be < 2^31.                                                              */
int mgs_csr_upload(mgs_ctx *ctx, int rows, int cols, int64_t nnz, const int *rowptr,
                   const int *col, const double *val, mgs_csr **out);
int mgs_csr_download(const mgs_csr *A, int *rowptr, int *col, double *val);
int mgs_csr_shape(const mgs_csr *A, int *rows, int *cols, int64_t *nnz);
int mgs_csr_destroy(mgs_csr *A);
/* device pointers (for zero-copy interop, e.g. the Eigen CPU baseline after download) */
int mgs_csr_device_ptrs(const mgs_csr *A, void **rowptr, void **col, void **val);
/* Synthetic operator generated on device: 7-point 3-D Poisson on an
 * N^3 grid, 3-D extension of src/common/poisson.cpp:11-33 (diag 6, off-diagonals −1,
 * row e=(i*N+j)*N+k, ascending columns).  Rows of planes [plane_lo, plane_hi) only
 * (row-range shard); columns are global unless local_cols != 0, in which case they are
 * renumbered to [0,n_loc) for owned and n_loc.. for the lower/upper halo planes.       */
int mgs_csr_poisson3d(mgs_ctx *ctx, int N, int plane_lo, int plane_hi, int local_cols,
                      mgs_csr **out);
/* Same stencil family in 2-D exactly as src/common/poisson.cpp:9-37 (n^2 rows, 4/−1).  */
int mgs_csr_poisson2d(mgs_ctx *ctx, int n, mgs_csr **out);
/* Bᵀ materialised row-major (bicg.cpp:32 `Ptrans = P.transpose()`).                    */
int mgs_csr_transpose(const mgs_csr *A, mgs_csr **out);
/* A_c = PᵀAP (bicg.cpp:33).  Runs on device.                                          */
int mgs_csr_galerkin(const mgs_csr *A, const mgs_xfer *P, mgs_csr **out);

/* -------------------------------------------------------------------- device vectors */
/* Eigen::VectorXd on the device (bicg.cpp:153-162).                                    */
int mgs_vec_create(mgs_ctx *ctx, int64_t n, mgs_vec **out);
/* non-owning view of caller device memory (e.g. a torch tensor's data_ptr)            */
int mgs_vec_wrap(mgs_ctx *ctx, void *device_ptr, int64_t n, mgs_vec **out);
int mgs_vec_destroy(mgs_vec *v);
int mgs_vec_upload(mgs_vec *v, const double *host, int64_t n);
int mgs_vec_download(const mgs_vec *v, double *host, int64_t n);
int mgs_vec_fill(mgs_vec *v, double value);
int mgs_vec_copy(const mgs_vec *src, mgs_vec *dst);
int64_t mgs_vec_size(const mgs_vec *v);
void *mgs_vec_ptr(const mgs_vec *v);
/* counter-based uniform [0,1): splitmix64(seed, global_index)      */
int mgs_vec_rand(mgs_vec *v, uint64_t seed, int64_t global_offset);

/* ------------------------------------------------------ L1 hot-path primitives (★)   */
/* y = A x — `A * v`, bicg.cpp:57,82,107,117; Eigen kernel
 * lib/Eigen/src/SparseCore/SparseDenseProduct.h:26-71.                                */
int mgs_spmv(const mgs_csr *A, const mgs_vec *x, mgs_vec *y);
/* r = b − A x — bicg.cpp:82.                                                          */
int mgs_residual(const mgs_csr *A, const mgs_vec *x, const mgs_vec *b, mgs_vec *r);
/* dinv_i = 1/a_ii — M2 = diag(diag(A))\x, src/CPU_Matlab/solve.m:17.  Fails with
 * MGS_ERR_NUMERIC if a diagonal entry is missing or zero.                             */
int mgs_diag_inv(const mgs_csr *A, mgs_vec *dinv);
/* x_out = x_in + ω D⁻¹ (b − A x_in), out of place (x_out must not alias x_in) —
 * SURVEY §8a row a7; spec solve.m:17 + paper eq. (3.5).                               */
int mgs_jacobi(const mgs_csr *A, const mgs_vec *dinv, double omega, const mgs_vec *b,
               const mgs_vec *x_in, mgs_vec *x_out);

/* Prolongation operator.  P is taken as CSR (what readMatrix returns for
 * <name>promatrix_<tag>.mtx, bicg.cpp:151).  If every row holds at most one entry of
 * value exactly 1 (how AGMG builds it: src/CPU_C++/AGMG.cpp:181-186,
 * src/GPU_CUDAC++/Aggregation.cu:252-270) the aggregate-id form is used; any other P
 * runs through the general CSR kernels with Pᵀ materialised (bicg.cpp:32).            */
int mgs_xfer_create(const mgs_csr *P, mgs_xfer **out);
int mgs_xfer_destroy(mgs_xfer *T);
int mgs_xfer_shape(const mgs_xfer *T, int *n_fine, int *n_coarse, int *is_aggregation);
/* r_c = Pᵀ r — `Ptrans * vec`, bicg.cpp:48.                                           */
int mgs_restrict(const mgs_xfer *T, const mgs_vec *r, mgs_vec *rc);
/* e = P e_c — `P * (...)`, bicg.cpp:48.                                               */
int mgs_prolong(const mgs_xfer *T, const mgs_vec *ec, mgs_vec *e);
/* x += P e_c (the coarse-grid correction of the V-cycle).                             */
int mgs_prolong_add(const mgs_xfer *T, const mgs_vec *ec, mgs_vec *x);

/* BLAS-1 used by BiCGSTABiml (bicg.cpp:64-72,95,101-120): free dot()/norm() and the
 * vector updates.  dot/nrm2 synchronise and return a host scalar.                     */
int mgs_dot(const mgs_vec *x, const mgs_vec *y, double *out);
int mgs_nrm2(const mgs_vec *x, double *out);
int mgs_axpby(double a, const mgs_vec *x, double b, mgs_vec *y); /* y = a x + b y      */
int mgs_axpbypcz(double a, const mgs_vec *x, double b, const mgs_vec *y, double c,
                 mgs_vec *z);                                    /* z = a x + b y + c z */

/* ------------------------------------------- L3 preconditioner: multilevel V-cycle (★) */
/* MultiGridPrecond(A, P) — bicg.cpp:19-62.  A is borrowed (must outlive the hierarchy).
 * Level l cycle: ν1 damped-Jacobi sweeps, r = b − Ax, r_c = Pᵀr, recurse from 0,
 * x += P e_c, ν2 sweeps; coarsest level solved directly (dense inverse built on device,
 * stands in for SparseLU, bicg.cpp:35-36).  Two-level with ν1=0, ν2=1 is bicg.cpp:46-61
 * with M2 = ωD⁻¹.                                                                     */
int mgs_hier_create(mgs_ctx *ctx, const mgs_csr *A, double omega, int nu1, int nu2,
                    mgs_hier **out);
/* append a level from a given prolongation (P.rows == rows of current coarsest);
 * computes A_c = PᵀAP on device (bicg.cpp:32-33).                                     */
int mgs_hier_push_P(mgs_hier *h, const mgs_csr *P);
/* append levels by on-device pairwise aggregation (Notay AGMG: src/CPU_C++/AGMG.cpp:
 * 299-315, src/GPU_CUDAC++/main.cu:95-277) until the coarsest has <= coarse_rows rows,
 * max_levels is reached or coarsening stalls.  ktg/npass/tou as the reference's argv
 * (src/CPU_C++/main.cpp:155-182; benchmarks use 10 2 8, results.txt:22-24).           */
int mgs_hier_coarsen(mgs_hier *h, double ktg, int npass, double tou, int coarse_rows,
                     int max_levels);
/* factor the coarsest operator (dense inverse on device, ≤ 8192 rows); must be called once before
 * mgs_vcycle.  If coarsening stalled above that size (e.g. all rows in G0) the coarsest level is
 * smoothed by 8 damped-Jacobi sweeps instead.                                          */
int mgs_hier_finalize(mgs_hier *h);
int mgs_hier_set_smoother(mgs_hier *h, double omega, int nu1, int nu2);
/* K-cycle (docs/AGMG_For_Convection_Diffusion.pdf §3.1, Fortran `nlvcyc`
 * src/CPU_Matlab/dagtwolev_mex.f90:59-61,72): the coarse problems of levels 1..levels are solved
 * by two GCR steps preconditioned by the cycle below instead of one recursive cycle.  0 = V-cycle
 * (default).  Scalars stay on the device; the second direction is orthogonalised explicitly against the first (ρ2 = d2'·v2' on the
 * orthogonalised pair — the paper's β − γ²/ρ1 without the difference of nearly equal numbers).  Option "kcycle_energy" (mgs_ctx_set_option):
 * coefficients from c instead of v = A·c (flexible-CG form) — SYMMETRIC POSITIVE DEFINITE operators only; the default is the paper's GCR form.
 * On a row-sharded hierarchy the inner products are summed over the ranks in three reductions per K step
 * (mgs_ctx_set_native_allreduce, else the mgs_ctx_set_allreduce callback); without either the level runs a V-cycle. */
int mgs_hier_set_kcycle(mgs_hier *h, int levels);
/* K iteration on level 0 itself when the hierarchy is applied from x = 0: for the replicated tail of a row-sharded hierarchy whose
 * K-cycle reaches the last sharded level (that level is the tail's level 0; no rank reduction needed: the tail is replicated). */
int mgs_hier_set_kcycle_entry(mgs_hier *h, int on);
/* the additive switch of MultiGridPrecond::solve (reference src/common/bicg.cpp:59, `multiplicative_precond = false`; dead in the
 * reference — the constructor fixes it to true, :42): with on != 0 a zero-guess cycle returns, level by level,
 * P·cycle(Pᵀ v) + M2(v) with M2 = ωD⁻¹ instead of the multiplicative form.  The other switch, `use_preconditioner = false`
 * (:53-54, solve(v) = v), is mgs_bicgstab / mgs_fgcr with hier = NULL.  Not offered on row shards. */
int mgs_hier_set_additive(mgs_hier *h, int on);
/* over-correction of unsmoothed aggregation (new knob, default 1 = the reference's form `P * (…)`, bicg.cpp:48): the coarse-grid correction of
 * every level is scaled, x ← x + σ·P e_c.  Piecewise-constant prolongation under-corrects smooth error; σ ≈ 1.5–2 cuts the Krylov iterations of
 * the V-cycle preconditioner on Poisson-like problems at the cost of one axpy on each coarse vector.  Parity fixtures use σ = 1. */
int mgs_hier_set_correction_scale(mgs_hier *h, double sigma);
int mgs_hier_destroy(mgs_hier *h);
int mgs_hier_nlev(const mgs_hier *h);
int mgs_hier_level_shape(const mgs_hier *h, int level, int *rows, int64_t *nnz);
/* borrowed handles of level operators / transfer (valid while h lives)                */
const mgs_csr *mgs_hier_level_A(const mgs_hier *h, int level);
const mgs_xfer *mgs_hier_level_P(const mgs_hier *h, int level);
/* aggregate id of every fine row of `level` (−1 = not aggregated, G0) to host         */
int mgs_xfer_download_agg(const mgs_xfer *T, int *agg);
/* algorithmic HBM bytes of one V-cycle application (DESIGN.md §5 formula)             */
int64_t mgs_hier_vcycle_bytes(const mgs_hier *h);

/* One V-cycle: x ← cycle(b) from x=0 if zero_guess else improving the x passed in.
 * MultiGridPrecond::solve (bicg.cpp:51-61) == mgs_vcycle(h, v, out, 1).              */
int mgs_vcycle(mgs_hier *h, const mgs_vec *b, mgs_vec *x, int zero_guess);

/* ------------------------------------------------------ L4 Krylov: BiCGSTABiml (f-2) */
/* bicg.cpp:74-136, device resident.  h may be NULL (identity preconditioner).  On
 * return *max_iter = iterations done, *tol = achieved relative residual; the int result
 * of the reference (0 ok / 1 max_iter / 2 rho breakdown / 3 omega breakdown) is written
 * to *status; the function's own return value is the MGS_* error class.               */
int mgs_bicgstab(const mgs_csr *A, mgs_vec *x, const mgs_vec *b, mgs_hier *h,
                 int *max_iter, double *tol, int *status);

/* Flexible GCR(m) — the outer Krylov method AGMG pairs with the K-cycle, whose preconditioner is
 * not a fixed linear operator (BiCGSTAB may break down with it).  Restarted every `restart`
 * directions; same in/out convention as mgs_bicgstab (status 0 converged / 1 max_iter).  Not in the
 * reference's C++ (its Matlab driver calls pcg/bicgstab, src/CPU_Matlab/solve.m:28-33); provided
 * with the K-cycle.  Per iteration beside preconditioner and SpMV: one multi-dot pass (new direction against the
 * window), one fused update pass, one residual pass; x is updated once per window through the triangular coefficient system.  Status 0 is
 * reported only with the TRUE residual b − A·x below *tol (recomputed at every restart and before every return).  restart 1..64. */
int mgs_fgcr(const mgs_csr *A, mgs_vec *x, const mgs_vec *b, mgs_hier *h, int restart,
             int *max_iter, double *tol, int *status);

/* ------------------------------------------------------------- multi-GPU row shards   */
/* Halo plan of a row-range shard whose CSR uses LOCAL column numbering: columns
 * [0,n_loc) are owned rows, columns n_loc+k are halo slot k.  send_idx lists the owned
 * rows this rank must pack for its peers (concatenated in peer order).  The pack kernel
 * gathers x[send_idx] into send_buf; the exchange itself is done by the caller (RCCL via
 * torch.distributed) between mgs_halo_pack and the next kernel that reads x's halo.   */
int mgs_halo_pack(mgs_ctx *ctx, const mgs_vec *x, const int *send_idx_dev, int64_t n_send,
                  double *send_buf_dev);
/* exchange callback installed on a hierarchy: called with (user, level, x_dev) before
 * every kernel that gathers off-shard entries of x on that level; x_dev has n_loc +
 * n_halo entries and the callee must fill x_dev[n_loc..] on ctx's stream.             */
typedef int (*mgs_halo_fn)(void *user, int level, void *x_dev);
int mgs_hier_set_halo_exchange(mgs_hier *h, mgs_halo_fn fn, void *user);
/* Split-phase form: begin(user, level, x) packs and STARTS the exchange, end(...) waits for it.
 * Between the two the library launches the kernel on the shard's interior row blocks (those that
 * read no halo column), afterwards on the boundary row blocks — the exchange hides behind ~97 %
 * of the rows of a plane-sharded stencil operator.  Falls back to fn for levels whose halo
 * readers are not a prefix + suffix of the row range.                                          */
int mgs_hier_set_halo_exchange_split(mgs_hier *h, mgs_halo_fn begin, mgs_halo_fn end, void *user);
/* Halo exchange of the fused cycle passes on a row shard (kind is always 2: plain values).  The callee delivers, for every halo
 * slot of `level`, the entry of the owner's vector `a` (owned entries of that level on every rank) that the slot stands for, into
 * halo_out — a payload buffer (pre pass: a = the right-hand side; the owners' ωD⁻¹ sits in Â's halo columns since setup) or the halo
 * room of the vector itself (post pass: a = e_c of the coarse level, level = that coarse level).  b is NULL.  phase 0 packs
 * (mgs_halo_pack) and starts the exchange, phase 1 waits for it; the library launches the interior row blocks in between.
 * Without this callback (and without a native plan) a sharded hierarchy runs the one-kernel-per-step form.                */
typedef int (*mgs_halo_fused_fn)(void *user, int level, int kind, const void *a, const void *b, void *halo_out, int phase);
int mgs_hier_set_halo_exchange_fused(mgs_hier *h, mgs_halo_fused_fn fn, void *user);

/* Building blocks of a row-sharded hierarchy (one process per GPU; orchestration in
 * multigridsolver_amd/dist.py).  mgs_aggregate_shard: pairwise aggregation of the OWNED
 * rows only (aggregates never straddle a shard; couplings to halo columns enter s_i and the
 * G0 test as symmetric).  The caller then learns the remote aggregate of every halo slot from
 * its peers and passes the coarse column of each halo slot (host array, n_halo ints, values
 * in [n_coarse, n_coarse+n_halo_coarse) or −1) to mgs_galerkin_shard, which returns the coarse
 * shard (n_coarse rows, n_coarse+n_halo_coarse local columns).  mgs_hier_push_level appends
 * the pair to a hierarchy and takes ownership of both.                                  */
int mgs_aggregate_shard(const mgs_csr *A, double ktg, int npass, double tou, mgs_xfer **T);
/* ... with zones (host array, one int per owned row; NULL = none): rows of different zones never share an aggregate.  Giving the rows each
 * peer sees as halo a zone of their own keeps what that peer asks for at the next level a contiguous id range (plane shards), so every halo
 * exchange of the hierarchy sends ranges straight from the vectors. */
int mgs_aggregate_shard_zoned(const mgs_csr *A, double ktg, int npass, double tou, const int *zone, mgs_xfer **T);
int mgs_galerkin_shard(const mgs_csr *A, const mgs_xfer *T, const int *halo_coarse_col,
                       int n_halo_coarse, mgs_csr **Ac);
int mgs_hier_push_level(mgs_hier *h, mgs_xfer *T, mgs_csr *Ac);
/* aggregation transfer from a host array of aggregate ids (−1 = none)                   */
int mgs_xfer_from_agg(mgs_ctx *ctx, int n_fine, int n_coarse, const int *agg, mgs_xfer **out);
/* coarsest-level solver callback (x_dev = solve(b_dev), both of the coarsest level's owned
 * size) — used for the replicated tail of a sharded hierarchy; replaces the dense inverse. */
typedef int (*mgs_coarse_fn)(void *user, const void *b_dev, void *x_dev);
int mgs_hier_set_coarse_solver(mgs_hier *h, mgs_coarse_fn fn, void *user);
/* reduction callback for dot products of sharded vectors (sum over ranks)             */
typedef int (*mgs_allreduce_fn)(void *user, double *host_scalars, int count);
int mgs_ctx_set_allreduce(mgs_ctx *ctx, mgs_allreduce_fn fn, void *user);

/* ------------------------------------------------------------------ instrumentation  */
/* Time `reps` back-to-back launches of one hot-path kernel with hipEvents on the
 * context's stream (what bench.py's roofline.achieved uses).  op: 0 spmv, 1 residual,
 * 2 jacobi.  Returns mean milliseconds per launch in *ms.                              */
int mgs_time_kernel(const mgs_csr *A, int op, const mgs_vec *x, const mgs_vec *b,
                    const mgs_vec *dinv, mgs_vec *out, int reps, double *ms);
/* Time `reps` V-cycles with hipEvents on the context's stream.                        */
int mgs_time_vcycle(mgs_hier *h, const mgs_vec *b, mgs_vec *x, int reps, double *ms);
/* launch plan of a matrix (diagnostics): out[0]=max entries of a 256-row block, [1]=max row
 * length, [2]=far band (max |col−row| over owned columns), [3]=LDS bytes per workgroup of the
 * row-block kernel, [4]=leading and [5]=trailing row blocks that read halo columns, [6]=1 if the
 * interior/boundary split is usable, [7]=1 if some block takes the long-row path.            */
int mgs_csr_plan_info(const mgs_csr *A, int64_t out[8]);
/* "origin" of the rows of a coarse operator built by the device setup: the finest-level row each row descends from (leader of its
 * aggregate, chained through the levels).  The pairwise matching breaks ties between equally strong neighbours in that index space
 * (nearest first, then even multiples of the stride), which keeps aggregates aligned on grid-like problems on every level.  A row-sharded
 * hierarchy hands the origins of its last sharded level (shifted to global finest-level rows) to the replicated tail with these two
 * calls.  get: returns 1 and writes nothing when the operator carries none (= identity).  No reference counterpart (the reference's
 * sequential matching scans neighbours in index order, AGMG.cpp:149-179). */
int mgs_csr_get_origin(const mgs_csr *A, int *origin_host);
int mgs_csr_set_origin(mgs_csr *A, const int *origin_host);
/* ---- native RCCL transport of a row-sharded hierarchy (no reference counterpart: the reference is single-process) ----
 * One communicator per process/GPU.  librccl is resolved at run time from `librccl_path` — pass the copy the launcher
 * already loaded (for torch.distributed: <torch>/lib/librccl.so) so the process holds a single RCCL instance.  Rank 0
 * calls mgs_comm_unique_id and ships the MGS_COMM_ID_BYTES bytes to the other ranks by any means; every rank then
 * calls mgs_comm_create.  All communication is enqueued on the context's stream.
 *   mgs_hier_set_native_exchange: halo plan of `level` — send_idx (host): owned rows the peers need, peer after peer;
 *     send_counts/recv_counts: per peer; the received values fill the level's halo slots in peer order.  With a plan
 *     installed the cycle packs and exchanges by itself (ncclSend/ncclRecv group) and ignores the exchange callbacks
 *     for that level.  comm = NULL removes the plan.
 *   mgs_hier_set_native_tail: the coarsest sharded level is all-gathered (ncclAllGather) and solved by `tail`, an
 *     unsharded hierarchy on the globally assembled operator replicated on every rank; nlocs = rows per rank.
 *   mgs_hier_native_halo: one plain halo exchange of the level-`level` vector x_dev (owned entries + halo room).
 *   mgs_hier_native_send_segments / mgs_hier_set_native_recv_segments: pack-free exchanges.  Where a peer's rows are a few
 *     contiguous ranges (plane shards: one), they are sent straight from the source vector, one ncclSend per range, and the
 *     peer posts one ncclRecv per range.  Each rank reads what it will send (ranges per peer + their lengths; returns the number
 *     of lengths), the host side ships the lengths to the peers, and every rank installs what it will receive — a COLLECTIVE
 *     step: from that call on the rank itself sends ranges.  Without it every exchange goes through the pack kernel.  */
#define MGS_COMM_ID_BYTES 128
typedef struct mgs_comm mgs_comm;
int mgs_comm_unique_id(mgs_ctx *ctx, const char *librccl_path, void *id_out);
int mgs_comm_create(mgs_ctx *ctx, const char *librccl_path, const void *id, int world, int rank, mgs_comm **out);
int mgs_comm_destroy(mgs_comm *c);
int mgs_comm_size(const mgs_comm *c, int *world, int *rank);
int mgs_hier_set_native_exchange(mgs_hier *h, int level, mgs_comm *c, const int *send_idx, const int *send_counts, const int *recv_counts);
int mgs_hier_set_native_tail(mgs_hier *h, mgs_comm *c, mgs_hier *tail, const int *nlocs);
/* Global tail row of every halo slot of the last sharded level (host array: first tail row of the owner rank + the owner's local
 * row).  The replicated tail's solution holds the neighbours' entries too: the kernel that hands this rank its own slice then fills
 * the level's halo slots from it, and the post pass of the level above needs no halo exchange for e_c.  n = 0 switches it off. */
int mgs_hier_set_native_tail_halo(mgs_hier *h, const int *halo_global, int n);
int mgs_hier_native_halo(mgs_hier *h, int level, void *x_dev);
int mgs_hier_native_send_segments(const mgs_hier *h, int level, int *nseg_per_peer, int *seglens, int cap);
int mgs_hier_set_native_recv_segments(mgs_hier *h, int level, const int *nseg_per_peer, const int *seglens);
/* ---- peer-to-peer transport behind the same mgs_comm handle (no reference counterpart) ----
 * The neighbours store their halo values straight into this rank's memory (IPC-mapped device window, xGMI), one kernel per exchange,
 * sequence-numbered flag words for ordering — no RCCL kernel, no host round trip; captured in the cycle's hipGraph like any kernel.
 * Every rank calls mgs_comm_p2p_create (slot_doubles = the most doubles one peer ever sends it in one exchange, all-gather of the tail
 * included; world <= 8) and gets MGS_P2P_HANDLE_BYTES bytes describing its window; the host side ships those records to all ranks
 * (rank order) and every rank calls mgs_comm_p2p_connect.  From then on the communicator serves mgs_hier_set_native_exchange /
 * _tail / mgs_ctx_set_native_allreduce exactly like an RCCL one.  mgs_comm_p2p_info: out[0] memory kind of the window (1 uncached,
 * 2 fine-grained, 3 coarse-grained), [1] window bytes, [2] slot doubles, [3] exchange kernels launched, [4] error word (0 = none;
 * 1 + k: the wait for the k-th participating peer of some exchange timed out, MGS_P2P_TIMEOUT_S), [5] connected. */
#define MGS_P2P_HANDLE_BYTES 96
int mgs_comm_p2p_create(mgs_ctx *ctx, int world, int rank, size_t slot_doubles, void *handle_out, mgs_comm **out);
int mgs_comm_p2p_connect(mgs_comm *c, const void *handles);
int mgs_comm_p2p_info(const mgs_comm *c, long long out[6]);
/* COLLECTIVE self-test of a connected peer-to-peer communicator: `rounds` exchanges with every peer (sizes from the window slot down to a few
 * doubles, both slots reused many times), every value a function of (round, sender, receiver, position) and verified on the device.
 * *mismatches = wrong values this rank saw (0 = clean).  The launcher runs it before it trusts the transport on hardware it has not seen. */
int mgs_comm_p2p_selftest(mgs_comm *c, int rounds, long long *mismatches);
/* raw operations of a communicator (either transport) on the context's stream — transport tests and microbenchmarks; the cycle uses
 * them internally.  exchange: one group of sends (send_dev[q] != NULL) and receives (recv_dev[q] != NULL) of count[q] doubles with
 * rank peer[q]; the ops addressed to one peer are matched with that peer's ops in posting order.  allgather: count doubles per rank,
 * rank-major into recv_dev.  allreduce: in-place sum of count doubles (p2p: count <= 64, summed in rank order). */
int mgs_comm_exchange_raw(mgs_comm *c, int nops, const int *peer, const size_t *count, const void *const *send_dev, void *const *recv_dev);
int mgs_comm_allgather_raw(mgs_comm *c, const void *send_dev, void *recv_dev, size_t count);
int mgs_comm_allreduce_raw(mgs_comm *c, void *buf_dev, size_t count);
/* inner products of the Krylov solvers summed over the ranks with ncclAllReduce on the context's stream (NULL: off;
 * takes precedence over the mgs_ctx_set_allreduce callback) */
int mgs_ctx_set_native_allreduce(mgs_ctx *ctx, mgs_comm *c);

/* Builds the pattern code of A's column array (one byte per row + a small table per 256-row block) so the
 * SpMV-shaped kernels stream 8 instead of 12 bytes per entry wherever rows repeat their shape (stencil-like
 * operators); results stay bit-identical.  Hierarchies and the Krylov solvers call it for their operators; call
 * it yourself before timing a bare mgs_spmv.  No reference counterpart (device-side layout choice).
 * info: out[0] coded row blocks, out[1] row blocks, out[2] table ints, out[3] LDS table budget (ints). */
int mgs_csr_optimize(mgs_csr *A);
/* which form of the fused passes level `level` runs (filled in by the first cycle): out[0] row blocks, out[1]/out[2]
 * setup-time operands present (scaled values / aggregate-mapped columns), out[3..5] coded row blocks of the column
 * array, of its halo-tagged copy (row shards) and of the aggregate-mapped array. */
int mgs_hier_fused_info(const mgs_hier *h, int level, int64_t out[6]);
int mgs_csr_rowcode_info(const mgs_csr *A, int64_t out[4]);
/* grouped pre pass of level `level` (filled in by the first cycle; option "fuse_restrict", default 1): out[0] row-block groups (0: the
 * level runs the separate pre pass + restriction kernels), out[1] groups made of two row blocks, out[2] stray aggregates (restricted by
 * the trailing kernel), out[3] row blocks. */
int mgs_hier_group_info(const mgs_hier *h, int level, int64_t out[4]);
/* hipGraph state of the cycle (diagnostics): out[0] captured cycles cached, out[1] = 1 on the native RCCL transport, out[2] = 1 if
 * capturing the native cycle failed (eager launches since), out[3] eager native cycles run before the first capture.
 * Option "native_graph" (default 1): capture the row-sharded cycle including its RCCL exchanges (two eager cycles first).
 * Option "native_overlap" (default 0): inside that graph, interior row blocks of the big levels on a second stream beside the exchange.
 * Option "graph_split_rows" (default 1048576; 0 = never): with a K-cycle on an unsharded operator of at least this many rows the fine level's two passes are launched
 * eagerly and the levels below replay from ONE graph that does not depend on (b, x): hipGraphLaunch of a K-cycle's hundreds of nodes takes ≈1.1 ms before its first
 * kernel starts, which then hides behind the fine level's pre pass, and a flexible Krylov method's ten direction vectors no longer overrun the (b, x) cache.  Same kernels, same bits. */
int mgs_hier_graph_info(const mgs_hier *h, int64_t out[4]);

/* kernel-variant knobs for A/B measurements (initial values of every context: environment MGS_OPTIONS="key=value,...").  key: "spmv_variant", "xcd_remap", "nontemporal",
 * "graph", "strip", "fuse", "nt_store" (smallest operator, in rows, whose row-block kernels store their outputs with the `nt` hint; default 1000000,
 * 0 = never), "stage_unroll" (row-block kernels stage their value slice without a loop in front of the barrier; default 1), "rowptr_scan" (pattern-coded row blocks take a row's
 * entry range from its pattern's length — one rowptr load per wave + a wave prefix sum — instead of two rowptr loads per row; default 1, same bits), "blas1_vec" / "blas1_pairs" (16-byte update
 * kernels, pairs per lane: 1 = one-shot workgroups; defaults 1 / 1), "post_results" (inner products reach the host through a mapped buffer and a polled ticket; default 1), "valcode" (opt-in: pattern tuples carry the values too, set before
 * mgs_csr_optimize / the hierarchy is built; pays only where coefficients repeat), "rowcode" (pattern-coded index, default 1), "split_min_rows" (row shards: smallest level that
 * overlaps its halo exchange with interior row blocks, default 400000), "fuse_operands" (setup-time operands of the fused cycle passes,
 * +12 B of HBM per matrix entry; default 1), "merge_ap" (fused post pass on A·P with the entries of one aggregate summed at setup instead of A with
 * aggregate-mapped columns; default 1; equal to rounding, not bit for bit), "fuse_restrict" / "group_blocks" / "group_stray_pct" / "group_min_blocks" / "group_strip"
 * (grouped pre pass: restriction inside the pre-smoothing pass, see mgs_hier_group_info), "diag_from_values", "fuse_dots", "lds_pad", "blkptr".
 * Unknown key: MGS_ERR_INVALID. */
int mgs_ctx_set_option(mgs_ctx *ctx, const char *key, int value);

#ifdef __cplusplus
}
#endif
#endif /* MGS_H */
