/*
 * mgs_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See mgs_oracle.h for scope, parity status and usage rules.
 * Citations are relative to the reference checkout (mishraiiit/MultiGridSolver).
 */
#include "mgs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ utils */

static int csr_alloc(orc_csr *m, int rows, int cols, int nnz) {
  m->rows = rows; m->cols = cols; m->nnz = nnz;
  m->rowptr = (int *)calloc((size_t)rows + 1, sizeof(int));
  m->col = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  m->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  if (!m->rowptr || !m->col || !m->val) return -1;
  return 0;
}

void orc_csr_free(orc_csr *m) {
  if (!m) return;
  free(m->rowptr); free(m->col); free(m->val);
  m->rowptr = NULL; m->col = NULL; m->val = NULL; m->rows = m->cols = m->nnz = 0;
}

int orc_csr_from_arrays(int rows, int cols, int nnz, const int *rowptr,
                        const int *col, const double *val, orc_csr *out) {
  if (csr_alloc(out, rows, cols, nnz)) return -1;
  memcpy(out->rowptr, rowptr, sizeof(int) * ((size_t)rows + 1));
  if (nnz) { memcpy(out->col, col, sizeof(int) * (size_t)nnz);
             memcpy(out->val, val, sizeof(double) * (size_t)nnz); }
  return 0;
}

static int csr_copy(const orc_csr *a, orc_csr *b) {
  return orc_csr_from_arrays(a->rows, a->cols, a->nnz, a->rowptr, a->col, a->val, b);
}

/* --------------------------------------------------------------- mtx I/O */

typedef struct { int r, c; double v; } trip;

/* std::sort on pair<int,double>: by column, then by value (MatrixIO.cpp:29) */
static int trip_cmp(const void *pa, const void *pb) {
  const trip *a = (const trip *)pa, *b = (const trip *)pb;
  if (a->r != b->r) return a->r < b->r ? -1 : 1;
  if (a->c != b->c) return a->c < b->c ? -1 : 1;
  if (a->v != b->v) return a->v < b->v ? -1 : 1;
  return 0;
}

/* src/common/MatrixIO.cpp:12-37.  Leading lines whose first character is '%'
 * are skipped (:16, banner not validated), then "M N L" (:18), then L triples
 * "i j v", 1-based, any order, any whitespace (:23-27); entries are bucketed
 * by row and each row is sorted by column (:29).  No symmetric / pattern /
 * array support.  Unlike the reference (which does not check, :13) a missing
 * file is reported as an error. */
int orc_mtx_read(const char *path, orc_csr *out) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  int ch;
  while ((ch = fgetc(f)) == '%') { /* fin.peek()=='%' → ignore(2048,'\n') */
    int n = 1;
    while ((ch = fgetc(f)) != EOF && ch != '\n' && n < 2048) n++;
  }
  if (ch != EOF) ungetc(ch, f);
  int M, N, L;
  if (fscanf(f, "%d %d %d", &M, &N, &L) != 3 || M < 0 || N < 0 || L < 0) { fclose(f); return -2; }
  trip *t = (trip *)malloc(sizeof(trip) * (size_t)(L > 0 ? L : 1));
  for (int l = 0; l < L; l++) {
    int m, n; double d;
    if (fscanf(f, "%d %d %lf", &m, &n, &d) != 3) { free(t); fclose(f); return -3; }
    if (m < 1 || m > M || n < 1 || n > N) { free(t); fclose(f); return -4; }
    t[l].r = m - 1; t[l].c = n - 1; t[l].v = d;
  }
  fclose(f);
  qsort(t, (size_t)L, sizeof(trip), trip_cmp);
  if (csr_alloc(out, M, N, L)) { free(t); return -5; }
  for (int l = 0; l < L; l++) out->rowptr[t[l].r + 1]++;
  for (int i = 0; i < M; i++) out->rowptr[i + 1] += out->rowptr[i];
  for (int l = 0; l < L; l++) { out->col[l] = t[l].c; out->val[l] = t[l].v; }
  free(t);
  return 0;
}

/* src/common/MatrixIO.cpp:39-57: banner with trailing space (:45), "rows cols
 * nnz" (:46), then row-major triples, 1-based (:52), value printed with the
 * default ostream precision (6 significant digits == "%g"). */
int orc_mtx_write(const char *path, const orc_csr *m) {
  FILE *f = fopen(path, "w");
  if (!f) return -1;
  fprintf(f, "%%%%MatrixMarket matrix coordinate real general \n");
  fprintf(f, "%d %d %d\n", m->rows, m->cols, m->nnz);
  for (int i = 0; i < m->rows; i++)
    for (int k = m->rowptr[i]; k < m->rowptr[i + 1]; k++)
      fprintf(f, "%d %d %g\n", i + 1, m->col[k] + 1, m->val[k]);
  fclose(f);
  return 0;
}

/* ------------------------------------------------------------ primitives */

/* Row loops may run on several host threads (bench.py's "all host cores" CPU baseline; default ONE thread, which is what every test
 * uses): rows are independent and each row's sum keeps its sequential order, so the bits do not depend on the thread count.  Inner
 * products stay sequential. */
#ifdef _OPENMP
#include <omp.h>
static int g_orc_threads = 1;
void orc_set_threads(int n) { g_orc_threads = n > 0 ? n : 1; }
int orc_get_threads(void) { return g_orc_threads; }
#define ORC_PAR _Pragma("omp parallel for schedule(static) num_threads(g_orc_threads)")
#else
void orc_set_threads(int n) { (void)n; }
int orc_get_threads(void) { return 1; }
#define ORC_PAR
#endif

/* lib/Eigen/src/SparseCore/SparseDenseProduct.h:64-70 (processRow): tmp = 0;
 * for it in row: tmp += it.value() * rhs(it.index()); res(i) += alpha*tmp
 * with alpha = 1 on a zeroed res. */
void orc_spmv(const orc_csr *A, const double *x, double *y) {
  ORC_PAR
  for (int i = 0; i < A->rows; i++) {
    double tmp = 0.0;
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) tmp += A->val[k] * x[A->col[k]];
    y[i] = 0.0 + 1.0 * tmp;
  }
}

/* bicg.cpp:32: Ptrans = P.transpose() materialised row-major, columns sorted */
int orc_transpose(const orc_csr *A, orc_csr *B) {
  if (csr_alloc(B, A->cols, A->rows, A->nnz)) return -1;
  for (int k = 0; k < A->nnz; k++) B->rowptr[A->col[k] + 1]++;
  for (int i = 0; i < B->rows; i++) B->rowptr[i + 1] += B->rowptr[i];
  int *next = (int *)malloc(sizeof(int) * ((size_t)B->rows + 1));
  memcpy(next, B->rowptr, sizeof(int) * ((size_t)B->rows + 1));
  for (int i = 0; i < A->rows; i++)
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
      int p = next[A->col[k]]++;
      B->col[p] = i; B->val[p] = A->val[k];
    }
  free(next);
  return 0;
}

static int int_cmp(const void *a, const void *b) {
  int x = *(const int *)a, y = *(const int *)b; return x < y ? -1 : x > y;
}

/* Row-wise (Gustavson) product with the same accumulation order as Eigen's
 * conservative_sparse_sparse_product on row-major operands
 * (lib/Eigen/src/SparseCore/ConservativeSparseSparseProduct.h): for every row
 * i, for k over A.row(i) ascending, for j over B.row(k): acc[j] += a_ik*b_kj.
 * Structural entries are kept even when they cancel to 0. */
int orc_spgemm(const orc_csr *A, const orc_csr *B, orc_csr *C) {
  if (A->cols != B->rows) return -1;
  int n = A->rows, m = B->cols;
  double *acc = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
  int *mark = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
  for (int j = 0; j < m; j++) mark[j] = -1;
  int *rp = (int *)calloc((size_t)n + 1, sizeof(int));
  /* pass 1: count */
  for (int i = 0; i < n; i++) {
    int cnt = 0;
    for (int ka = A->rowptr[i]; ka < A->rowptr[i + 1]; ka++) {
      int k = A->col[ka];
      for (int kb = B->rowptr[k]; kb < B->rowptr[k + 1]; kb++) {
        int j = B->col[kb];
        if (mark[j] != i) { mark[j] = i; cnt++; }
      }
    }
    rp[i + 1] = rp[i] + cnt;
  }
  if (csr_alloc(C, n, m, rp[n])) return -2;
  memcpy(C->rowptr, rp, sizeof(int) * ((size_t)n + 1));
  for (int j = 0; j < m; j++) mark[j] = -1;
  for (int i = 0; i < n; i++) {
    int base = rp[i], cnt = 0;
    for (int ka = A->rowptr[i]; ka < A->rowptr[i + 1]; ka++) {
      int k = A->col[ka]; double a = A->val[ka];
      for (int kb = B->rowptr[k]; kb < B->rowptr[k + 1]; kb++) {
        int j = B->col[kb];
        if (mark[j] != i) { mark[j] = i; C->col[base + cnt++] = j; acc[j] = a * B->val[kb]; }
        else acc[j] += a * B->val[kb];
      }
    }
    qsort(C->col + base, (size_t)cnt, sizeof(int), int_cmp);
    for (int q = 0; q < cnt; q++) C->val[base + q] = acc[C->col[base + q]];
  }
  free(acc); free(mark); free(rp);
  return 0;
}

/* bicg.cpp:32-33: Ptrans = P.transpose(); Ac = Ptrans * A * P (left to right) */
int orc_galerkin(const orc_csr *A, const orc_csr *P, orc_csr *Ac) {
  orc_csr Pt, T; int rc;
  if ((rc = orc_transpose(P, &Pt))) return rc;
  if ((rc = orc_spgemm(&Pt, A, &T))) { orc_csr_free(&Pt); return rc; }
  rc = orc_spgemm(&T, P, Ac);
  orc_csr_free(&Pt); orc_csr_free(&T);
  return rc;
}

static double csr_coeff(const orc_csr *A, int i, int j) { /* Eigen coeff(): 0 if absent */
  int lo = A->rowptr[i], hi = A->rowptr[i + 1] - 1;
  while (lo <= hi) {
    int mid = lo + ((hi - lo) >> 1);
    if (A->col[mid] == j) return A->val[mid];
    if (A->col[mid] < j) lo = mid + 1; else hi = mid - 1;
  }
  return 0.0;
}

/* src/CPU_Matlab/solve.m:17 (M2 = diag(diag(A))\x) */
void orc_diag_inv(const orc_csr *A, double *dinv) {
  for (int i = 0; i < A->rows; i++) dinv[i] = 1.0 / csr_coeff(A, i, i);
}

/* bicg.cpp:82: r = b − A*x */
void orc_residual(const orc_csr *A, const double *x, const double *b, double *r) {
  ORC_PAR
  for (int i = 0; i < A->rows; i++) {
    double tmp = 0.0;
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) tmp += A->val[k] * x[A->col[k]];
    r[i] = b[i] - tmp;
  }
}

/* r = b − A x; x += ω D⁻¹ r, out of place.
 * Evaluation order fixed as  x_out = x_in + (ω·dinv_i)·(b_i − (Ax)_i). */
void orc_jacobi(const orc_csr *A, const double *dinv, double omega,
                const double *b, const double *x_in, double *x_out) {
  ORC_PAR
  for (int i = 0; i < A->rows; i++) {
    double tmp = 0.0;
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) tmp += A->val[k] * x_in[A->col[k]];
    x_out[i] = x_in[i] + (omega * dinv[i]) * (b[i] - tmp);
  }
}

double orc_dot(int n, const double *a, const double *b) {
  double s = 0.0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s;
}
double orc_nrm2(int n, const double *a) { return sqrt(orc_dot(n, a, a)); }

/* ---------------------------------------------------------------- dense LU */

int orc_dense_lu_factor(const orc_csr *A, orc_dense_lu *f) {
  int n = A->rows;
  if (A->cols != n) return -1;
  f->n = n;
  f->lu = (double *)calloc((size_t)n * (size_t)n + 1, sizeof(double));
  f->piv = (int *)malloc(sizeof(int) * (size_t)(3 * n + 1));
  if (!f->lu || !f->piv) return -2;
  /* profile bookkeeping (first/last structural column per row) keeps the work
   * proportional to the band of the coarse operator instead of n^3 */
  int *first = f->piv + n, *last = f->piv + 2 * n;
  for (int i = 0; i < n; i++) {
    first[i] = i; last[i] = i;
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
      int c = A->col[k];
      f->lu[(size_t)i * n + c] += A->val[k];
      if (c < first[i]) first[i] = c;
      if (c > last[i]) last[i] = c;
    }
  }
  for (int k = 0; k < n; k++) {
    int p = k; double best = fabs(f->lu[(size_t)k * n + k]);
    for (int i = k + 1; i < n; i++) {
      if (first[i] > k) continue;
      double v = fabs(f->lu[(size_t)i * n + k]); if (v > best) { best = v; p = i; }
    }
    f->piv[k] = p;
    if (best == 0.0) return -3;
    if (p != k) {
      int lo = first[k] < first[p] ? first[k] : first[p], hi = last[k] > last[p] ? last[k] : last[p];
      for (int j = lo; j <= hi; j++) { double t = f->lu[(size_t)k * n + j]; f->lu[(size_t)k * n + j] = f->lu[(size_t)p * n + j]; f->lu[(size_t)p * n + j] = t; }
      int t1 = first[k]; first[k] = first[p]; first[p] = t1;
      t1 = last[k]; last[k] = last[p]; last[p] = t1;
    }
    double piv = f->lu[(size_t)k * n + k];
    int lk = last[k];
    for (int i = k + 1; i < n; i++) {
      if (first[i] > k) continue;
      double *ri = f->lu + (size_t)i * n; const double *rk = f->lu + (size_t)k * n;
      if (ri[k] == 0.0) continue;
      double l = ri[k] / piv; ri[k] = l;
      for (int j = k + 1; j <= lk; j++) ri[j] -= l * rk[j];
      if (lk > last[i]) last[i] = lk;
    }
  }
  /* rows were swapped physically: first[] now bounds L's profile per final row,
   * but earlier swaps may have moved multipliers; recompute conservative bounds */
  for (int i = 0; i < n; i++) {
    const double *ri = f->lu + (size_t)i * n;
    int a = 0; while (a < i && ri[a] == 0.0) a++;
    int b = n - 1; while (b > i && ri[b] == 0.0) b--;
    first[i] = a; last[i] = b;
  }
  return 0;
}

void orc_dense_lu_solve(const orc_dense_lu *f, const double *b, double *x) {
  int n = f->n;
  const int *first = f->piv + n, *last = f->piv + 2 * n;
  for (int i = 0; i < n; i++) x[i] = b[i];
  for (int k = 0; k < n; k++) { int p = f->piv[k]; if (p != k) { double t = x[k]; x[k] = x[p]; x[p] = t; } }
  for (int i = 0; i < n; i++) { double s = x[i]; const double *ri = f->lu + (size_t)i * n; for (int j = first[i]; j < i; j++) s -= ri[j] * x[j]; x[i] = s; }
  for (int i = n - 1; i >= 0; i--) { double s = x[i]; const double *ri = f->lu + (size_t)i * n; for (int j = i + 1; j <= last[i]; j++) s -= ri[j] * x[j]; x[i] = s / ri[i]; }
}

void orc_dense_lu_free(orc_dense_lu *f) { free(f->lu); free(f->piv); f->lu = NULL; f->piv = NULL; f->n = 0; }

/* ------------------------------------------------------------- hierarchy */

struct orc_hier {
  int nlev;
  orc_csr *A;   /* nlev */
  orc_csr *P;   /* nlev-1 */
  orc_csr *Pt;  /* nlev-1 */
  double **dinv, **r, **tmp, **bc, **xc;
  orc_dense_lu lu;
  double omega; int nu1, nu2;
  int kcycle_levels;
  int kcycle_energy;   /* K-cycle coefficients from energy inner products (flexible CG form, SPD operators) instead of the GCR form */
  double corr_scale;   /* over-correction x += sigma * P e_c (0 or 1: the reference's form) */
  int additive;   /* bicg.cpp:59: multigrid_solve(v) + M2(v) instead of the multiplicative form */
  double **kc1, **kv1, **kc2, **kv2, **kr;
};

orc_hier *orc_hier_create(int nlev, const orc_csr *const *A, const orc_csr *const *P,
                          double omega, int nu1, int nu2) {
  orc_hier *h = (orc_hier *)calloc(1, sizeof(orc_hier));
  h->nlev = nlev; h->omega = omega; h->nu1 = nu1; h->nu2 = nu2;
  h->A = (orc_csr *)calloc((size_t)nlev, sizeof(orc_csr));
  h->P = (orc_csr *)calloc((size_t)nlev, sizeof(orc_csr));
  h->Pt = (orc_csr *)calloc((size_t)nlev, sizeof(orc_csr));
  h->dinv = (double **)calloc((size_t)nlev, sizeof(double *));
  h->r = (double **)calloc((size_t)nlev, sizeof(double *));
  h->tmp = (double **)calloc((size_t)nlev, sizeof(double *));
  h->bc = (double **)calloc((size_t)nlev, sizeof(double *));
  h->xc = (double **)calloc((size_t)nlev, sizeof(double *));
  h->kc1 = (double **)calloc((size_t)nlev, sizeof(double *)); h->kv1 = (double **)calloc((size_t)nlev, sizeof(double *));
  h->kc2 = (double **)calloc((size_t)nlev, sizeof(double *)); h->kv2 = (double **)calloc((size_t)nlev, sizeof(double *));
  h->kr = (double **)calloc((size_t)nlev, sizeof(double *));
  for (int l = 0; l < nlev; l++) {
    csr_copy(A[l], &h->A[l]);
    int n = A[l]->rows; size_t sz = sizeof(double) * (size_t)(n > 0 ? n : 1);
    h->dinv[l] = (double *)malloc(sz); h->r[l] = (double *)malloc(sz); h->tmp[l] = (double *)malloc(sz);
    h->bc[l] = (double *)malloc(sz); h->xc[l] = (double *)malloc(sz);
    h->kc1[l] = (double *)malloc(sz); h->kv1[l] = (double *)malloc(sz); h->kc2[l] = (double *)malloc(sz);
    h->kv2[l] = (double *)malloc(sz); h->kr[l] = (double *)malloc(sz);
    orc_diag_inv(&h->A[l], h->dinv[l]);
    if (l < nlev - 1) { csr_copy(P[l], &h->P[l]); orc_transpose(&h->P[l], &h->Pt[l]); }
  }
  if (orc_dense_lu_factor(&h->A[nlev - 1], &h->lu)) { orc_hier_destroy(h); return NULL; }
  return h;
}

orc_hier *orc_hier_create_from_P(const orc_csr *A0, int nP, const orc_csr *const *P,
                                 double omega, int nu1, int nu2) {
  int nlev = nP + 1;
  orc_csr *A = (orc_csr *)calloc((size_t)nlev, sizeof(orc_csr));
  const orc_csr **Ap = (const orc_csr **)calloc((size_t)nlev, sizeof(orc_csr *));
  csr_copy(A0, &A[0]); Ap[0] = &A[0];
  for (int l = 0; l < nP; l++) { orc_galerkin(&A[l], P[l], &A[l + 1]); Ap[l + 1] = &A[l + 1]; }
  orc_hier *h = orc_hier_create(nlev, Ap, P, omega, nu1, nu2);
  for (int l = 0; l < nlev; l++) orc_csr_free(&A[l]);
  free(A); free(Ap);
  return h;
}

void orc_hier_destroy(orc_hier *h) {
  if (!h) return;
  for (int l = 0; l < h->nlev; l++) {
    orc_csr_free(&h->A[l]); orc_csr_free(&h->P[l]); orc_csr_free(&h->Pt[l]);
    free(h->dinv[l]); free(h->r[l]); free(h->tmp[l]); free(h->bc[l]); free(h->xc[l]);
    free(h->kc1[l]); free(h->kv1[l]); free(h->kc2[l]); free(h->kv2[l]); free(h->kr[l]);
  }
  free(h->kc1); free(h->kv1); free(h->kc2); free(h->kv2); free(h->kr);
  if (h->lu.lu) orc_dense_lu_free(&h->lu);
  free(h->A); free(h->P); free(h->Pt); free(h->dinv); free(h->r); free(h->tmp); free(h->bc); free(h->xc);
  free(h);
}

int orc_hier_nlev(const orc_hier *h) { return h->nlev; }
void orc_hier_set_smoother(orc_hier *h, double omega, int nu1, int nu2) { h->omega = omega; h->nu1 = nu1; h->nu2 = nu2; }
void orc_hier_set_kcycle(orc_hier *h, int levels) { h->kcycle_levels = levels; }
void orc_hier_set_kcycle_energy(orc_hier *h, int on) { h->kcycle_energy = on; }
void orc_hier_set_additive(orc_hier *h, int on) { h->additive = on; }
void orc_hier_set_correction_scale(orc_hier *h, double sigma) { h->corr_scale = sigma; }
const orc_csr *orc_hier_A(const orc_hier *h, int l) { return &h->A[l]; }

/* V-cycle definition (SURVEY §7 "Hard parts"; two-level ν1=0, ν2=1 from x=0
 * equals bicg.cpp:46-61 with M2 = ωD⁻¹, i.e. paper eq. (3.5)):
 *   ν1 × { x ← x + ωD⁻¹(b − Ax) };  r = b − Ax;  r_c = Pᵀ r   (bicg.cpp:48)
 *   e_c = cycle(l+1, r_c) from 0 (coarsest: direct solve, bicg.cpp:35-36,48)
 *   x ← x + P e_c (bicg.cpp:48);  ν2 × { x ← x + ωD⁻¹(b − Ax) }            */
static void vcycle_rec(const orc_hier *h, int l, const double *b, double *x, int zero_guess);

/* Coarse solve for the level above: one cycle (V), or two GCR steps preconditioned by the cycle
 * (K-cycle; docs/AGMG_For_Convection_Diffusion.pdf §3.1 — derived from the paper, the reference's
 * C++ has no K-cycle):  x = (α1/ρ1) c1 + (α2/ρ2)(c2 − (γ/ρ1) c1), second direction orthogonalised explicitly (see below). */
static void coarse_solve_inner(const orc_hier *h, int l, const double *rhs, double *x);
static void coarse_solve_rec(const orc_hier *h, int l, const double *rhs, double *x) {
  coarse_solve_inner(h, l, rhs, x);
  if (h->corr_scale != 0.0 && h->corr_scale != 1.0) for (int i = 0; i < h->A[l].rows; i++) x[i] = h->corr_scale * x[i];
}
static void coarse_solve_inner(const orc_hier *h, int l, const double *rhs, double *x) {
  if (!(l >= 1 && l <= h->kcycle_levels && l < h->nlev - 1)) { vcycle_rec(h, l, rhs, x, 1); return; }
  const orc_csr *A = &h->A[l];
  int n = A->rows;
  double *c1 = h->kc1[l], *v1 = h->kv1[l], *c2 = h->kc2[l], *v2 = h->kv2[l], *rp = h->kr[l];
  vcycle_rec(h, l, rhs, c1, 1);
  orc_spmv(A, c1, v1);
  /* GCR form (paper §3.1, any operator): inner products with v = A c (minimal residual).  Energy form (flexible CG, SPD
   * operators): the same five products with c in place of the left factor — ρ1 = c1·Ac1, α1 = c1·rhs, γ = c2·Ac1, β = c2·Ac2,
   * α2 = c2·r' — i.e. the combination that minimises the A-norm of the error; the update formulas are the same. */
  const double *d1 = h->kcycle_energy ? c1 : v1, *d2 = h->kcycle_energy ? c2 : v2;
  double rho1 = orc_dot(n, d1, v1), alpha1 = orc_dot(n, d1, rhs);
  double a = rho1 != 0.0 ? alpha1 / rho1 : 0.0;
  for (int i = 0; i < n; i++) rp[i] = rhs[i] - a * v1[i];
  vcycle_rec(h, l, rp, c2, 1);
  orc_spmv(A, c2, v2);
  /* second direction orthogonalised explicitly: c2' = c2 - g c1, v2' = v2 - g v1 with g = gamma/rho1; rho2 = d2'.v2', alpha2 = d2'.r'.
   * In exact arithmetic rho2 = beta - gamma^2/rho1 (the paper's formula); formed this way it is not the difference of two nearly equal
   * numbers when c2 is almost parallel to c1.  x = (alpha1/rho1) c1 + (alpha2/rho2) c2'. */
  double gamma = orc_dot(n, d2, v1);
  double g = rho1 != 0.0 ? gamma / rho1 : 0.0;
  double rho2 = 0.0, alpha2 = 0.0;
  for (int i = 0; i < n; i++) {
    double v2o = v2[i] - g * v1[i];
    double d2o = h->kcycle_energy ? c2[i] - g * c1[i] : v2o;
    rho2 += d2o * v2o;
    alpha2 += d2o * rp[i];
  }
  double k1 = 0.0, k2 = 0.0;
  if (rho1 != 0.0) {
    k1 = alpha1 / rho1;
    if (rho2 > 0.0) { k2 = alpha2 / rho2; k1 -= (gamma / rho1) * k2; }
  }
  for (int i = 0; i < n; i++) x[i] = k1 * c1[i] + k2 * c2[i];
}

static void vcycle_rec(const orc_hier *h, int l, const double *b, double *x, int zero_guess) {
  const orc_csr *A = &h->A[l];
  int n = A->rows;
  if (l == h->nlev - 1) { orc_dense_lu_solve(&h->lu, b, x); return; }
  double *tmp = h->tmp[l], *r = h->r[l];
  if (h->additive) {   /* bicg.cpp:59 with M2 = ωD⁻¹, level by level:  x = P·cycle(Pᵀ b) + ωD⁻¹ b  (zero guess only) */
    orc_spmv(&h->Pt[l], b, h->bc[l + 1]);
    coarse_solve_rec(h, l + 1, h->bc[l + 1], h->xc[l + 1]);
    orc_spmv(&h->P[l], h->xc[l + 1], tmp);
    for (int i = 0; i < n; i++) x[i] = tmp[i] + (h->omega * h->dinv[l][i]) * b[i];
    return;
  }
  if (zero_guess) { ORC_PAR for (int i = 0; i < n; i++) x[i] = 0.0; }
  for (int s = 0; s < h->nu1; s++) {
    orc_jacobi(A, h->dinv[l], h->omega, b, x, tmp);
    memcpy(x, tmp, sizeof(double) * (size_t)n);
  }
  orc_residual(A, x, b, r);
  orc_spmv(&h->Pt[l], r, h->bc[l + 1]);
  coarse_solve_rec(h, l + 1, h->bc[l + 1], h->xc[l + 1]);
  orc_spmv(&h->P[l], h->xc[l + 1], tmp);
  ORC_PAR
  for (int i = 0; i < n; i++) x[i] += tmp[i];
  for (int s = 0; s < h->nu2; s++) {
    orc_jacobi(A, h->dinv[l], h->omega, b, x, tmp);
    memcpy(x, tmp, sizeof(double) * (size_t)n);
  }
}

void orc_vcycle(const orc_hier *h, const double *b, double *x, int zero_guess) {
  vcycle_rec(h, 0, b, x, zero_guess);
}

void orc_precond_vcycle(void *hier, const double *v, double *out) {
  orc_vcycle((const orc_hier *)hier, v, out, 1);
}

/* bicg.cpp:46-61 with M2 = ωD⁻¹ (solve.m:17): res = P·(LU(PᵀAP)⁻¹·(Pᵀ·v));
 * return res + M2(v) − M2(A·res)  ==  res + ωD⁻¹(v − A·res). */
int orc_twogrid_jacobi(const orc_csr *A, const orc_csr *P, double omega,
                       const double *v, double *x) {
  const orc_csr *Ps[1] = { P };
  orc_hier *h = orc_hier_create_from_P(A, 1, Ps, omega, 0, 1);
  if (!h) return -1;
  orc_vcycle(h, v, x, 1);
  orc_hier_destroy(h);
  return 0;
}

/* --------------------------------------------------------------- BiCGSTAB */

/* bicg.cpp:74-136, statement by statement. */
int orc_bicgstab(const orc_csr *A, double *x, const double *b,
                 orc_precond_fn M, void *user, int *max_iter, double *tol) {
  int n = A->rows;
  size_t sz = sizeof(double) * (size_t)(n > 0 ? n : 1);
  double *p = (double *)calloc(1, sz), *phat = (double *)malloc(sz), *s = (double *)malloc(sz),
         *shat = (double *)malloc(sz), *t = (double *)malloc(sz), *v = (double *)calloc(1, sz),
         *r = (double *)malloc(sz), *rtilde = (double *)malloc(sz);
  double rho_1 = 0, rho_2 = 0, alpha = 0, beta = 0, omega = 0, resid = 0;
  int status = 1, i;
  double normb = orc_nrm2(n, b);                                  /* :80 */
  orc_residual(A, x, b, r);                                       /* :82 */
  memcpy(rtilde, r, sz);                                          /* :83 */
  if (normb == 0.0) normb = 1;                                    /* :85-86 */
  if ((resid = orc_nrm2(n, r) / normb) <= *tol) {                 /* :88-92 */
    *tol = resid; *max_iter = 0; status = 0; goto done;
  }
  for (i = 1; i <= *max_iter; i++) {                              /* :94 */
    rho_1 = orc_dot(n, rtilde, r);                                /* :95 */
    if (rho_1 == 0) { *tol = orc_nrm2(n, r) / normb; status = 2; goto done; } /* :96-99 */
    if (i == 1) memcpy(p, r, sz);                                 /* :100-101 */
    else {
      beta = (rho_1 / rho_2) * (alpha / omega);                   /* :103 */
      for (int k = 0; k < n; k++) p[k] = r[k] + beta * (p[k] - omega * v[k]); /* :104 */
    }
    if (M) M(user, p, phat); else memcpy(phat, p, sz);            /* :106 */
    orc_spmv(A, phat, v);                                         /* :107 */
    alpha = rho_1 / orc_dot(n, rtilde, v);                        /* :108 */
    for (int k = 0; k < n; k++) s[k] = r[k] - alpha * v[k];       /* :109 */
    if ((resid = orc_nrm2(n, s) / normb) < *tol) {                /* :110-115 */
      for (int k = 0; k < n; k++) x[k] += alpha * phat[k];
      *max_iter = i; *tol = resid; status = 0; goto done;
    }
    if (M) M(user, s, shat); else memcpy(shat, s, sz);            /* :116 */
    orc_spmv(A, shat, t);                                         /* :117 */
    omega = orc_dot(n, t, s) / orc_dot(n, t, t);                  /* :118 */
    for (int k = 0; k < n; k++) x[k] += alpha * phat[k] + omega * shat[k]; /* :119 */
    for (int k = 0; k < n; k++) r[k] = s[k] - omega * t[k];       /* :120 */
    rho_2 = rho_1;                                                /* :122 */
    if ((resid = orc_nrm2(n, r) / normb) < *tol) {                /* :123-127 */
      *tol = resid; *max_iter = i; status = 0; goto done;
    }
    if (omega == 0) { *tol = orc_nrm2(n, r) / normb; status = 3; goto done; } /* :128-131 */
  }
  *tol = resid;                                                   /* :134 */
  status = 1;                                                     /* :135 */
done:
  free(p); free(phat); free(s); free(shat); free(t); free(v); free(r); free(rtilde);
  return status;
}

/* bicg.cpp:139,161: srand(0); b[i] = rand() / (RAND_MAX + 0.0) */
void orc_rand_rhs(unsigned seed, int n, double *b) {
  srand(seed);
  for (int i = 0; i < n; i++) b[i] = rand() / (RAND_MAX + 0.0);
}

/* -------------------------------------------------------------- generators */

/* src/common/poisson.cpp:9-37: n²×n² 5-point stencil 4/−1, rows truncated at
 * the boundary, elem = i*n + j (+1 in the file), entries in ascending column
 * order elem−n, elem−1, elem, elem+1, elem+n. */
int orc_poisson2d(int n, orc_csr *out) {
  int N = n * n, nnz = 5 * N - 4 * n;
  if (csr_alloc(out, N, N, nnz)) return -1;
  int p = 0;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
    int e = i * n + j;
    out->rowptr[e] = p;
    if (i > 0)     { out->col[p] = e - n; out->val[p++] = -1; }
    if (j > 0)     { out->col[p] = e - 1; out->val[p++] = -1; }
    out->col[p] = e; out->val[p++] = 4;
    if (j < n - 1) { out->col[p] = e + 1; out->val[p++] = -1; }
    if (i < n - 1) { out->col[p] = e + n; out->val[p++] = -1; }
  }
  out->rowptr[N] = p;
  return p == nnz ? 0 : -2;
}

/* 3-D extension of poisson.cpp:11-33. */
int orc_poisson3d(int N, orc_csr *out) {
  long long n = (long long)N * N * N, nnz = 7 * n - 6LL * N * N;
  if (nnz > 2147483647LL) return -1;
  if (csr_alloc(out, (int)n, (int)n, (int)nnz)) return -1;
  int p = 0, N2 = N * N;
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) for (int k = 0; k < N; k++) {
    int e = (i * N + j) * N + k;
    out->rowptr[e] = p;
    if (i > 0)     { out->col[p] = e - N2; out->val[p++] = -1; }
    if (j > 0)     { out->col[p] = e - N;  out->val[p++] = -1; }
    if (k > 0)     { out->col[p] = e - 1;  out->val[p++] = -1; }
    out->col[p] = e; out->val[p++] = 6;
    if (k < N - 1) { out->col[p] = e + 1;  out->val[p++] = -1; }
    if (j < N - 1) { out->col[p] = e + N;  out->val[p++] = -1; }
    if (i < N - 1) { out->col[p] = e + N2; out->val[p++] = -1; }
  }
  out->rowptr[n] = p;
  return 0;
}

/* ------------------------------------------------------------ AGMG (setup) */

/* Eigen dense sum of a contiguous double array as compiled for SSE2 (packet of
 * 2 doubles, 16-byte aligned start): lib/Eigen/src/Core/Redux.h:209-256, reached
 * from SparseVector::sum() (lib/Eigen/src/SparseCore/SparseRedux.h:41-45). */
static double eigen_sum(const double *v, int size) {
  if (size == 0) return 0.0;
  const int ps = 2;
  int alignedSize2 = (size / (2 * ps)) * (2 * ps);
  int alignedSize = (size / ps) * ps;
  double res;
  if (alignedSize) {
    double p0[2] = { v[0], v[1] };
    if (alignedSize > ps) {
      double p1[2] = { v[2], v[3] };
      for (int idx = 2 * ps; idx < alignedSize2; idx += 2 * ps) {
        p0[0] += v[idx]; p0[1] += v[idx + 1];
        p1[0] += v[idx + 2]; p1[1] += v[idx + 3];
      }
      p0[0] += p1[0]; p0[1] += p1[1];
      if (alignedSize > alignedSize2) { p0[0] += v[alignedSize2]; p0[1] += v[alignedSize2 + 1]; }
    }
    res = p0[0] + p0[1];
    for (int idx = alignedSize; idx < size; idx++) res += v[idx];
  } else {
    res = v[0];
    for (int idx = 1; idx < size; idx++) res += v[idx];
  }
  return res;
}

/* AGMG.cpp:14-46: plain BFS from node 0 over the row pattern; -1 if the graph
 * is not connected (reference: assert(used == n), :42). */
static int *agmg_bfs_order(const orc_csr *A) {
  int n = A->rows;
  int *order = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  char *vis = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
  int added = 0, used = 0;
  vis[0] = 1; order[added++] = 0;
  while (used != added) {
    int u = order[used++];
    for (int k = A->rowptr[u]; k < A->rowptr[u + 1]; k++) {
      int v = A->col[k];
      if (!vis[v]) { vis[v] = 1; order[added++] = v; }
    }
  }
  free(vis);
  if (used != n) { free(order); return NULL; }
  return order;
}

/* AGMG.cpp:48-82 */
static double agmg_abs_row_col_sum(const orc_csr *A, const orc_csr *At, int i) {
  int r = A->rowptr[i], re = A->rowptr[i + 1], c = At->rowptr[i], ce = At->rowptr[i + 1];
  double ans = 0.0;
  while (r < re && c < ce) {
    if (A->col[r] == At->col[c]) { ans = ans + fabs((A->val[r] + At->val[c]) / 2); r++; c++; }
    else if (A->col[r] < At->col[c]) { ans = ans + fabs(A->val[r] / 2); r++; }
    else { ans = ans + fabs(At->val[c] / 2); c++; }
  }
  while (r < re) { ans = ans + fabs(A->val[r] / 2); r++; }
  while (c < ce) { ans = ans + fabs(At->val[c] / 2); c++; }
  ans = ans - fabs(csr_coeff(A, i, i));
  return ans;
}

/* AGMG.cpp:84-90 */
static double agmg_row_col_sum(const orc_csr *A, const orc_csr *At, int i) {
  double rs = eigen_sum(A->val + A->rowptr[i], A->rowptr[i + 1] - A->rowptr[i]);
  double cs = eigen_sum(At->val + At->rowptr[i], At->rowptr[i + 1] - At->rowptr[i]);
  double ans = -(rs + cs) / 2;
  ans += csr_coeff(A, i, i);
  return ans;
}

/* AGMG.cpp:92-99 */
static double agmg_mu(const orc_csr *A, const double *s, int i, int j) {
  const double si = s[i], sj = s[j];
  const double num = 2 / (1 / csr_coeff(A, i, i) + 1 / csr_coeff(A, j, j));
  const double den = (-(csr_coeff(A, i, j) + csr_coeff(A, j, i)) / 2) +
                     1 / (1 / (csr_coeff(A, i, i) - si) + 1 / (csr_coeff(A, j, j) - sj));
  return num / den;
}

/* shared pair search: AGMG.cpp:149-169 / :226-244 */
static int agmg_best_partner(const orc_csr *A, const double *s, const char *in_u, int i,
                             double *best_mu_out, int check_order, int *order_violation) {
  int best_j = -1; double best_mu = 0.0;
  for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
    int j = A->col[k];
    if (!in_u[j]) continue;
    if (j != i && A->val[k] != 0) {
      if (check_order && !(i < j)) *order_violation = 1;          /* assert(i < j), :156 */
      if (csr_coeff(A, i, i) - s[i] + csr_coeff(A, j, j) - s[j] >= 0) {
        double m = agmg_mu(A, s, i, j);
        if ((best_j == -1 && m > 0) || (m > 0 && m < best_mu)) { best_j = j; best_mu = m; }
      }
    }
  }
  *best_mu_out = best_mu;
  return best_j;
}

static void groups_to_P(int n, int nc, const int *groups, orc_csr *P) {
  int nnz = 0;
  for (int i = 0; i < n; i++) if (groups[i] != -1) nnz++;
  csr_alloc(P, n, nc, nnz);
  int p = 0;
  for (int i = 0; i < n; i++) {
    P->rowptr[i] = p;
    if (groups[i] != -1) { P->col[p] = groups[i]; P->val[p++] = 1.0; }
  }
  P->rowptr[n] = p;
}

/* AGMG.cpp:101-194 */
static int agmg_initial(const orc_csr *A, double ktg, orc_csr *P) {
  int n = A->rows;
  int *cmk = agmg_bfs_order(A);
  if (!cmk) return -1;
  orc_csr At; orc_transpose(A, &At);
  int *groups = (int *)malloc(sizeof(int) * (size_t)n);
  char *in_u = (char *)malloc((size_t)n);
  double *s = (double *)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; i++) in_u[i] = 1;
  for (int i = 0; i < n; i++)                                              /* :118-123 (G0) */
    if (csr_coeff(A, i, i) >= (ktg / (ktg - 2)) * agmg_abs_row_col_sum(A, &At, i)) in_u[i] = 0;
  for (int i = 0; i < n; i++) { groups[i] = -1; s[i] = agmg_row_col_sum(A, &At, i); } /* :130-134 */
  int nc = 0, viol = 0;
  for (int idx = 0; idx < n; idx++) {                                      /* :138-179 */
    int i = cmk[idx];
    if (!in_u[i]) continue;
    double bm; int bj = agmg_best_partner(A, s, in_u, i, &bm, 1, &viol);
    nc = nc + 1;
    if (bj != -1 && bm <= ktg) { groups[i] = nc - 1; groups[bj] = nc - 1; in_u[i] = 0; in_u[bj] = 0; }
    else { groups[i] = nc - 1; in_u[i] = 0; }
  }
  groups_to_P(n, nc, groups, P);
  free(cmk); free(groups); free(in_u); free(s); orc_csr_free(&At);
  return viol ? -2 : 0;
}

/* AGMG.cpp:196-280 */
static int agmg_further(const orc_csr *A, double ktg, const orc_csr *Pbar_t,
                        const orc_csr *Abar, orc_csr *P) {
  int n = A->rows, ncb = Abar->rows;
  orc_csr Abt; orc_transpose(Abar, &Abt);
  int *groups = (int *)malloc(sizeof(int) * (size_t)n);
  char *in_u = (char *)malloc((size_t)(ncb > 0 ? ncb : 1));
  double *s = (double *)malloc(sizeof(double) * (size_t)(ncb > 0 ? ncb : 1));
  for (int i = 0; i < n; i++) groups[i] = -1;
  for (int i = 0; i < ncb; i++) in_u[i] = 1;
  for (int i = 0; i < ncb; i++) s[i] = agmg_row_col_sum(Abar, &Abt, i);
  int nc = 0, dummy = 0;
  for (int i = 0; i < ncb; i++) {
    if (!in_u[i]) continue;
    double bm; int bj = agmg_best_partner(Abar, s, in_u, i, &bm, 0, &dummy);
    nc = nc + 1;
    for (int k = Pbar_t->rowptr[i]; k < Pbar_t->rowptr[i + 1]; k++) groups[Pbar_t->col[k]] = nc - 1;
    in_u[i] = 0;
    if (bj != -1 && bm <= ktg) {
      for (int k = Pbar_t->rowptr[bj]; k < Pbar_t->rowptr[bj + 1]; k++) groups[Pbar_t->col[k]] = nc - 1;
      in_u[bj] = 0;
    }
  }
  groups_to_P(n, nc, groups, P);
  free(groups); free(in_u); free(s); orc_csr_free(&Abt);
  return 0;
}

/* AGMG.cpp:299-315 */
int orc_agmg(const orc_csr *A, double ktg, int npass, double tou, int max_restriction, orc_csr *P) {
  int rc = agmg_initial(A, ktg, P);
  if (rc && rc != -2) return rc;   /* -2: P is complete but the reference's assert(i<j) would fire */
  for (int s = 2; s <= npass; s++) {
    orc_csr Pt, Abar, Pn;
    orc_transpose(P, &Pt);
    orc_galerkin(A, P, &Abar);
    int stop = (Abar.nnz <= A->nnz / tou) || (Abar.rows < max_restriction);   /* :309-310 */
    if (!stop) {
      agmg_further(A, ktg, &Pt, &Abar, &Pn);
      orc_csr_free(P); *P = Pn;
    }
    orc_csr_free(&Pt); orc_csr_free(&Abar);
    if (stop) break;
  }
  return rc;
}
