/*
 * mgs_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference's CPU/Eigen algorithms on the hot path
 *  plus the documented Jacobi V-cycle definition
 * (SURVEY.md §0 G2, §7 "Hard parts").  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product library (libmgs.so)
 * never links, loads or calls it.
 *
 * Parity status: PINNED — checked against golden vectors produced by the
 * reference's own code compiled in oracle/_ref (see oracle/ref_harness.cpp,
 * oracle/make_golden.py, tests/golden/manifest.json) and against the
 * known-answer values of the reference's test program
 * (src/GPU_CUDAC++/test_matrix_operations.cu:329-347).
 *
 * Every function cites the reference file:line (relative to the reference
 * checkout) whose behaviour it restates.
 */
#ifndef MGS_ORACLE_H
#define MGS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* CSR: f64 values, int32 indices, sorted columns — Eigen
 * SparseMatrix<double,RowMajor> (src/common/MatrixIO.cpp:10). */
typedef struct {
  int rows, cols, nnz;
  int *rowptr; /* rows+1 */
  int *col;    /* nnz */
  double *val; /* nnz */
} orc_csr;

/* ---- L0 I/O: src/common/MatrixIO.cpp:12-57 ---- */
int orc_mtx_read(const char *path, orc_csr *out);            /* :12-37 */
int orc_mtx_write(const char *path, const orc_csr *m);       /* :39-57 */
void orc_csr_free(orc_csr *m);
int orc_csr_from_arrays(int rows, int cols, int nnz, const int *rowptr,
                        const int *col, const double *val, orc_csr *out);

/* ---- L1 primitives ---- */
/* y = A x.  lib/Eigen/src/SparseCore/SparseDenseProduct.h:64-70 (sequential
 * accumulate over the row in ascending column order). bicg.cpp:57,82,107,117 */
void orc_spmv(const orc_csr *A, const double *x, double *y);
/* B = Aᵀ as row-major CSR.  bicg.cpp:32 (Ptrans = P.transpose()) */
int orc_transpose(const orc_csr *A, orc_csr *B);
/* C = A·B (row-major CSR, sorted, structural entries kept). Used for
 * Ac = Ptrans*A*P evaluated left to right, bicg.cpp:33 */
int orc_spgemm(const orc_csr *A, const orc_csr *B, orc_csr *C);
int orc_galerkin(const orc_csr *A, const orc_csr *P, orc_csr *Ac);
/* dinv_i = 1 / a_ii (Jacobi alternative, src/CPU_Matlab/solve.m:17) */
void orc_diag_inv(const orc_csr *A, double *dinv);
/* r = b − A x (bicg.cpp:82) */
void orc_residual(const orc_csr *A, const double *x, const double *b, double *r);
/* x_out = x_in + ω D⁻¹ (b − A x_in), out of place */
void orc_jacobi(const orc_csr *A, const double *dinv, double omega,
                const double *b, const double *x_in, double *x_out);
double orc_dot(int n, const double *a, const double *b);
double orc_nrm2(int n, const double *a);

/* ---- dense LU with partial pivoting (stands in for SparseLU, bicg.cpp:35-36,48) */
typedef struct { int n; double *lu; int *piv; } orc_dense_lu;
int orc_dense_lu_factor(const orc_csr *A, orc_dense_lu *f);
void orc_dense_lu_solve(const orc_dense_lu *f, const double *b, double *x);
void orc_dense_lu_free(orc_dense_lu *f);

/* ---- Two-grid operator with damped Jacobi as M2 (bicg.cpp:46-61 with
 * M2 = ωD⁻¹; paper eq. (3.5)):  x1 = P·(Ac⁻¹·(Pᵀ·v)); x = x1 + ωD⁻¹(v − A·x1) */
int orc_twogrid_jacobi(const orc_csr *A, const orc_csr *P, double omega,
                       const double *v, double *x);

/* ---- multilevel V-cycle (definition: SURVEY §7 "Hard parts") ---- */
typedef struct orc_hier orc_hier;
/* A[0..nlev-1], P[0..nlev-2] (P[l]: rows(A[l]) × rows(A[l+1])).  Matrices are
 * copied.  Coarsest level is solved by dense LU.  */
orc_hier *orc_hier_create(int nlev, const orc_csr *const *A,
                          const orc_csr *const *P, double omega, int nu1, int nu2);
/* build coarse operators by Galerkin from A0 and P[0..nlev-2] */
orc_hier *orc_hier_create_from_P(const orc_csr *A0, int nP, const orc_csr *const *P,
                                 double omega, int nu1, int nu2);
void orc_hier_destroy(orc_hier *h);
int orc_hier_nlev(const orc_hier *h);
void orc_hier_set_smoother(orc_hier *h, double omega, int nu1, int nu2);
/* K-cycle on levels 1..levels (two GCR steps per coarse solve); 0 = V-cycle.  Derived from the
 * paper (docs/AGMG_For_Convection_Diffusion.pdf §3.1): no executable reference exists for it. */
void orc_hier_set_kcycle(orc_hier *h, int levels);
void orc_hier_set_kcycle_energy(orc_hier *h, int on);   /* K-cycle coefficients: 0 GCR form (default), 1 energy / flexible-CG form (SPD operators) */
void orc_hier_set_additive(orc_hier *h, int on);   /* bicg.cpp:59 */
void orc_hier_set_correction_scale(orc_hier *h, double sigma);   /* x += sigma * P e_c; derived knob, no reference counterpart */
const orc_csr *orc_hier_A(const orc_hier *h, int l);
/* x = Vcycle(b) from x = 0 (zero_guess != 0) or from the x passed in */
void orc_vcycle(const orc_hier *h, const double *b, double *x, int zero_guess);

/* ---- L4 Krylov: BiCGSTABiml, bicg.cpp:74-136. precond==NULL → identity.
 * Returns 0/1/2/3 and writes iterations / achieved residual like the reference. */
typedef void (*orc_precond_fn)(void *user, const double *v, double *out);
int orc_bicgstab(const orc_csr *A, double *x, const double *b,
                 orc_precond_fn M, void *user, int *max_iter, double *tol);
/* convenience preconditioners */
void orc_precond_vcycle(void *hier, const double *v, double *out);

/* ---- reference RHS convention: srand(seed); b[i] = rand()/(RAND_MAX+0.0)
 * (bicg.cpp:139,161) */
void orc_rand_rhs(unsigned seed, int n, double *b);

/* ---- synthetic operators ---- */
/* 2-D 5-pt Poisson exactly as src/common/poisson.cpp:9-37 (n from stdin) */
int orc_poisson2d(int n, orc_csr *out);
/* 3-D 7-pt analogue: row e=(i*N+j)*N+k, diag 6, off −1 */
int orc_poisson3d(int N, orc_csr *out);

/* ---- L2 setup: CPU AGMG restatement, src/CPU_C++/AGMG.cpp:14-315 ---- */
int orc_agmg(const orc_csr *A, double ktg, int npass, double tou,
             int max_restriction, orc_csr *P);

#ifdef __cplusplus
}
#endif
/* host threads of the row loops (bench.py's all-cores CPU baseline; default 1 — bits do not depend on it) */
void orc_set_threads(int n);
int orc_get_threads(void);

#endif
