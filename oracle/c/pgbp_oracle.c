/*
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the reference's canonical-form belief propagation, executing messages in the reference's
 * exact sequential order on one thread (orc_calibrate: the checker and the primary CPU baseline) or, for the
 * secondary all-core baseline only, level by level on every core (orc_calibrate_levels, OpenMP).  Used (i) as the checker at sizes
 * the numpy restatement cannot reach, (ii) as bench.py's `cpu_baseline` ("port": 1 core, the
 * reference is single-threaded).  Never linked into, or called by, the product.
 *
 * Follows (paths relative to the reference repository):
 *   marginalize            src/beliefupdates.jl:55-83   (Cholesky of Symmetric(J_I): upper triangle)
 *   divide!                src/beliefupdates.jl:579-587
 *   mult!                  src/beliefupdates.jl:483-488
 *   propagate_belief!      src/beliefupdates.jl:634-665
 *   iscalibrated_residnorm! src/beliefs.jl:994-1003
 *   traversals, calibrate! src/calibration.jl:35-161
 *   integratebelief        src/beliefupdates.jl:187-200
 * Storage: packed (J column-major m*m, h m, g) per belief, clusters first (include/pgbp.h).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LOG2PI 1.8378770664093454835606594728112
#define EPS 2.220446049250313e-16

typedef struct {
  int nc, ns;
  int32_t* dims;     /* [nc+ns] */
  int32_t* sepcl;    /* [2*ns] */
  int64_t* scope_off;/* [2*ns+1] */
  int32_t* scope_idx;
  int64_t* off;      /* [nc+ns+1] packed offsets */
  int64_t* roff;     /* [2*ns+1] residual offsets (dJ s*s, dh s) */
  double* pool;
  double* res;
  int32_t* flags;    /* [2*ns] iscalibrated_resid */
  double* work;      /* scratch */
  int maxdim;
  int fail_edge, fail_dir, fail_info;
} orc_t;

static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) abort(); return p; }

orc_t* orc_create(int nc, int ns, const int32_t* dims, const int32_t* sepcl, const int64_t* scope_off,
                  const int32_t* scope_idx, const double* packed) {
  orc_t* o = (orc_t*)xmalloc(sizeof(orc_t));
  int nb = nc + ns;
  o->nc = nc; o->ns = ns;
  o->dims = (int32_t*)xmalloc(sizeof(int32_t) * nb); memcpy(o->dims, dims, sizeof(int32_t) * nb);
  o->sepcl = (int32_t*)xmalloc(sizeof(int32_t) * 2 * ns); memcpy(o->sepcl, sepcl, sizeof(int32_t) * 2 * ns);
  o->scope_off = (int64_t*)xmalloc(sizeof(int64_t) * (2 * ns + 1)); memcpy(o->scope_off, scope_off, sizeof(int64_t) * (2 * ns + 1));
  int64_t nidx = ns ? scope_off[2 * ns] : 0;
  o->scope_idx = (int32_t*)xmalloc(sizeof(int32_t) * nidx); memcpy(o->scope_idx, scope_idx, sizeof(int32_t) * nidx);
  o->off = (int64_t*)xmalloc(sizeof(int64_t) * (nb + 1));
  o->off[0] = 0; o->maxdim = 0;
  for (int b = 0; b < nb; ++b) {
    int64_t m = dims[b];
    o->off[b + 1] = o->off[b] + m * m + m + 1;
    if (m > o->maxdim) o->maxdim = (int)m;
  }
  o->roff = (int64_t*)xmalloc(sizeof(int64_t) * (2 * ns + 1));
  o->roff[0] = 0;
  for (int d = 0; d < 2 * ns; ++d) { int64_t s = dims[nc + d / 2]; o->roff[d + 1] = o->roff[d] + s * s + s; }
  o->pool = (double*)xmalloc(sizeof(double) * o->off[nb]);
  memcpy(o->pool, packed, sizeof(double) * o->off[nb]);
  o->res = (double*)calloc((size_t)(o->roff[2 * ns] ? o->roff[2 * ns] : 1), sizeof(double));
  o->flags = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)(ns > 0 ? 2 * ns : 1));
  for (int d = 0; d < 2 * ns; ++d) o->flags[d] = dims[nc + d / 2] == 0;  /* empty messages born calibrated */
  int M = o->maxdim;
  o->work = (double*)xmalloc(sizeof(double) * (size_t)(4 * M * M + 8 * M + 8));
  o->fail_edge = -1; o->fail_dir = 0; o->fail_info = 0;
  return o;
}

void orc_destroy(orc_t* o) {
  if (!o) return;
  free(o->dims); free(o->sepcl); free(o->scope_off); free(o->scope_idx); free(o->off); free(o->roff);
  free(o->pool); free(o->res); free(o->flags); free(o->work); free(o);
}

void orc_set(orc_t* o, const double* packed) { memcpy(o->pool, packed, sizeof(double) * o->off[o->nc + o->ns]); }
void orc_get(const orc_t* o, double* packed) { memcpy(packed, o->pool, sizeof(double) * o->off[o->nc + o->ns]); }
void orc_reset_flags(orc_t* o) { for (int d = 0; d < 2 * o->ns; ++d) o->flags[d] = o->dims[o->nc + d / 2] == 0; }
void orc_get_residuals(const orc_t* o, double* res, int32_t* flags) {
  if (res) memcpy(res, o->res, sizeof(double) * o->roff[2 * o->ns]);
  if (flags) memcpy(flags, o->flags, sizeof(int32_t) * 2 * o->ns);
}
int64_t orc_packed_size(const orc_t* o) { return o->off[o->nc + o->ns]; }
void orc_last_failure(const orc_t* o, int* edge, int* dir, int* info) { *edge = o->fail_edge; *dir = o->fail_dir; *info = o->fail_info; }

/* Cholesky A = U'U reading only the upper triangle of the n x n column-major A (lda n); U overwrites
 * the upper triangle. Returns 0 or the 1-based order of the first non-positive leading minor
 * (LAPACK dpotrf/dpotf2 semantics: PosDefException.info). */
static int chol_upper(double* A, int n) {
  for (int j = 0; j < n; ++j) {
    double s = A[j + (size_t)j * n];
    for (int k = 0; k < j; ++k) s -= A[k + (size_t)j * n] * A[k + (size_t)j * n];
    if (!(s > 0.0)) return j + 1;
    double ujj = sqrt(s);
    A[j + (size_t)j * n] = ujj;
    for (int i = j + 1; i < n; ++i) {
      double t = A[j + (size_t)i * n];
      for (int k = 0; k < j; ++k) t -= A[k + (size_t)j * n] * A[k + (size_t)i * n];
      A[j + (size_t)i * n] = t / ujj;
    }
  }
  return 0;
}

/* marginalize: sender (J mf x mf, h, g), keep indices (s of them, increasing).
 * Outputs message (mJ s x s col-major, mh, *mg). Returns 0, or info > 0 if J_I is not PD. */
static int marginalize(orc_t* o, const double* J, const double* h, double g, int mf, const int32_t* keep, int s,
                       double* mJ, double* mh, double* mg) {
  int ni = mf - s;
  if (ni == 0) {  /* :56 */
    memcpy(mJ, J, sizeof(double) * (size_t)mf * mf);
    memcpy(mh, h, sizeof(double) * mf);
    *mg = g;
    return 0;
  }
  double* w = o->work;
  int32_t* integ = (int32_t*)(w);            /* ni ints (fits in the first doubles) */
  double* Ji = w + o->maxdim;                /* ni x ni */
  double* Z = Ji + (size_t)ni * ni;          /* s x ni : Jki then Jki U^{-1} */
  double* mu = Z + (size_t)s * ni;           /* ni */
  int t = 0, q = 0;
  for (int v = 0; v < mf; ++v) { if (q < s && keep[q] == v) ++q; else integ[t++] = v; }  /* setdiff, ascending (:52) */
  int allzero = 1;
  for (int b = 0; b < ni && allzero; ++b) {
    for (int a = 0; a < ni; ++a) if (fabs(J[integ[a] + (size_t)integ[b] * mf]) > EPS) { allzero = 0; break; }
    if (allzero) for (int a = 0; a < s; ++a) if (fabs(J[keep[a] + (size_t)integ[b] * mf]) > EPS) { allzero = 0; break; }
    if (allzero && fabs(h[integ[b]]) > EPS) allzero = 0;
  }
  for (int a = 0; a < s; ++a) {
    mh[a] = h[keep[a]];
    for (int b = 0; b < s; ++b) mJ[a + (size_t)b * s] = J[keep[a] + (size_t)keep[b] * mf];
  }
  if (allzero) { *mg = g; return 0; }        /* :62-66 */
  for (int b = 0; b < ni; ++b)
    for (int a = 0; a <= b; ++a) Ji[a + (size_t)b * ni] = J[integ[a] + (size_t)integ[b] * mf];  /* upper triangle */
  int info = chol_upper(Ji, ni);             /* :68 */
  if (info) return info;
  /* :77 X_invA_Xt: Z = Jki U^{-1}  (row a: solve z U = Jki[a,:]) */
  for (int a = 0; a < s; ++a) {
    for (int b = 0; b < ni; ++b) {
      double v = J[keep[a] + (size_t)integ[b] * mf];
      for (int k = 0; k < b; ++k) v -= Z[a + (size_t)k * s] * Ji[k + (size_t)b * ni];
      Z[a + (size_t)b * s] = v / Ji[b + (size_t)b * ni];
    }
  }
  for (int b = 0; b < s; ++b)
    for (int a = 0; a < s; ++a) {
      double acc = 0.0;
      for (int k = 0; k < ni; ++k) acc += Z[a + (size_t)k * s] * Z[b + (size_t)k * s];
      mJ[a + (size_t)b * s] -= acc;
    }
  /* :78 mu_i = Ji \ hi : U'y = hi, U mu = y */
  double logdet = 0.0, quad = 0.0;
  for (int b = 0; b < ni; ++b) {
    double v = h[integ[b]];
    for (int k = 0; k < b; ++k) v -= Ji[k + (size_t)b * ni] * mu[k];
    mu[b] = v / Ji[b + (size_t)b * ni];
  }
  for (int b = ni - 1; b >= 0; --b) {
    double v = mu[b];
    for (int k = b + 1; k < ni; ++k) v -= Ji[b + (size_t)k * ni] * mu[k];
    mu[b] = v / Ji[b + (size_t)b * ni];
  }
  for (int b = 0; b < ni; ++b) { logdet += log(Ji[b + (size_t)b * ni]); quad += h[integ[b]] * mu[b]; }
  logdet *= 2.0;
  for (int a = 0; a < s; ++a) {              /* :79 */
    double acc = 0.0;
    for (int b = 0; b < ni; ++b) acc += J[keep[a] + (size_t)integ[b] * mf] * mu[b];
    mh[a] -= acc;
  }
  *mg = g + ((double)ni * LOG2PI - logdet + quad) / 2.0;  /* :81 */
  return 0;
}

/* propagate_belief!(to, sepset k, from, residual) with the caller's scratch (work: 4 M^2 + 8 M + 8 doubles, M = the
 * largest belief dimension).  Returns 0 or info. */
static int propagate_w(orc_t* o, int to, int k, int from, double* work) {
  orc_t view = *o;   /* marginalize() takes its scratch from the handle: a per-thread view with its own */
  view.work = work;
  orc_t* const oo = o;
  o = &view;
  const int a = o->sepcl[2 * k], b = o->sepcl[2 * k + 1];
  const int dir = (to == a && from == b) ? 0 : ((to == b && from == a) ? 1 : -1);
  if (dir < 0) return -1;
  const int sfrom = dir == 0 ? 1 : 0, sto = dir == 0 ? 0 : 1;
  const int sb = o->nc + k, s = o->dims[sb], mf = o->dims[from], mt = o->dims[to];
  const int32_t* keep = o->scope_idx + o->scope_off[2 * k + sfrom];
  const int32_t* up = o->scope_idx + o->scope_off[2 * k + sto];
  double* F = o->pool + o->off[from];
  double* T = o->pool + o->off[to];
  double* S = o->pool + o->off[sb];
  double* R = o->res + o->roff[2 * k + dir];
  int M = o->maxdim;
  double* mJ = o->work + (size_t)(3 * M * M + 4 * M + 4);
  double* mh = mJ + (size_t)M * M;
  double mg;
  int info = marginalize(o, F, F + (size_t)mf * mf, F[(size_t)mf * mf + mf], mf, keep, s, mJ, mh, &mg);
  if (info) return info;
  double maxJ = 0.0, maxh = 0.0;
  int nanflag = 0;
  for (int c = 0; c < s; ++c)
    for (int r = 0; r < s; ++r) {
      double dJ = mJ[r + (size_t)c * s] - S[r + (size_t)c * s];   /* divide! */
      S[r + (size_t)c * s] = mJ[r + (size_t)c * s];
      R[r + (size_t)c * s] = dJ;
      T[up[r] + (size_t)up[c] * mt] += dJ;                          /* mult! */
      if (dJ != dJ) nanflag = 1; else if (fabs(dJ) > maxJ) maxJ = fabs(dJ);
    }
  for (int r = 0; r < s; ++r) {
    double dh = mh[r] - S[(size_t)s * s + r];
    S[(size_t)s * s + r] = mh[r];
    R[(size_t)s * s + r] = dh;
    T[(size_t)mt * mt + up[r]] += dh;
    if (dh != dh) nanflag = 1; else if (fabs(dh) > maxh) maxh = fabs(dh);
  }
  double dg = mg - S[(size_t)s * s + s];
  S[(size_t)s * s + s] = mg;
  T[(size_t)mt * mt + mt] += dg;
  /* iscalibrated_residnorm! */
  oo->flags[2 * k + dir] = (s == 0) || (!nanflag && maxh / sqrt((double)s) <= 1e-5 && maxJ / sqrt((double)s * (double)s) <= 1e-5);
  return 0;
}

int orc_propagate(orc_t* o, int to, int k, int from) { return propagate_w(o, to, k, from, o->work); }

/* ALL-CORE baseline (labelled "secondary"): the same messages, level-synchronous instead of
 * sequential.  tasks of one level touch disjoint receivers / sepsets and read no belief written in that level (the
 * caller -- oracle/cengine.py:levels_of_tree -- groups a postorder level's messages by receiver and gives every preorder
 * message its own task), so they run on all cores (OpenMP); the entries of a task keep the reference's order.
 * One (postorder, preorder) pass per iteration.  Returns succ; on a failure fail_* names ONE failing message (not
 * necessarily the first of the reference's order: this entry point exists for timing). */
#ifdef _OPENMP
#include <omp.h>
#endif
int orc_calibrate_levels(orc_t* o, int n_levels, const int32_t* level_off, const int32_t* task_off, const int32_t* ent_to,
                         const int32_t* ent_k, const int32_t* ent_from, int niter, int nthreads, int* iscal) {
  const size_t wlen = (size_t)(4 * o->maxdim * o->maxdim + 8 * o->maxdim + 8);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const int nt = omp_get_max_threads();
#else
  const int nt = 1;
  (void)nthreads;
#endif
  double* works = (double*)xmalloc(sizeof(double) * wlen * (size_t)nt);
  int failed = 0, cal = 0;
  o->fail_edge = -1;
  for (int it = 0; it < niter && !failed; ++it) {
    for (int L = 0; L < n_levels && !failed; ++L) {
      const int t0 = level_off[L], t1 = level_off[L + 1];
#pragma omp parallel for schedule(dynamic, 16)
      for (int t = t0; t < t1; ++t) {
#ifdef _OPENMP
        double* w = works + wlen * (size_t)omp_get_thread_num();
#else
        double* w = works;
#endif
        for (int e = task_off[t]; e < task_off[t + 1]; ++e) {
          const int info = propagate_w(o, ent_to[e], ent_k[e], ent_from[e], w);
          if (info) {
#pragma omp critical
            { failed = 1; o->fail_edge = e; o->fail_dir = 0; o->fail_info = info; }
            break;
          }
        }
      }
    }
    cal = 1;
    for (int d = 0; d < 2 * o->ns; ++d) if (!o->flags[d]) { cal = 0; break; }
  }
  free(works);
  if (iscal) *iscal = failed ? 0 : cal;
  return !failed;
}

/* calibrate!(beliefs, [tree], niter): edges (pa, ch, sepset k) in preorder. Returns succ; *iscal. */
int orc_calibrate(orc_t* o, int nedges, const int32_t* pa, const int32_t* ch, const int32_t* sepk, int niter,
                  int post_only, int* iscal) {
  int cal = 0;
  o->fail_edge = -1;
  for (int it = 0; it < niter; ++it) {
    for (int i = nedges - 1; i >= 0; --i) {       /* postorder: src/calibration.jl:121 */
      int info = orc_propagate(o, pa[i], sepk[i], ch[i]);
      if (info) { o->fail_edge = i; o->fail_dir = 0; o->fail_info = info; if (iscal) *iscal = 0; return 0; }
    }
    if (post_only) continue;
    for (int i = 0; i < nedges; ++i) {            /* preorder: :147 */
      int info = orc_propagate(o, ch[i], sepk[i], pa[i]);
      if (info) { o->fail_edge = i; o->fail_dir = 1; o->fail_info = info; if (iscal) *iscal = 0; return 0; }
    }
    cal = 1;
    for (int d = 0; d < 2 * o->ns; ++d) if (!o->flags[d]) { cal = 0; break; }
  }
  if (iscal) *iscal = cal;
  return 1;
}

/* integratebelief: returns info (0 ok); mu (m), *norm. */
int orc_integrate(orc_t* o, int b, double* mu, double* norm) {
  const int m = o->dims[b];
  const double* J = o->pool + o->off[b];
  const double* h = J + (size_t)m * m;
  const double g = h[m];
  int nz = 0;
  for (int i = 0; i < m * m && !nz; ++i) if (J[i] != 0.0) nz = 1;
  for (int i = 0; i < m && !nz; ++i) if (h[i] != 0.0) nz = 1;
  if (!nz) { for (int i = 0; i < m; ++i) mu[i] = INFINITY; *norm = g; return 0; }
  double* U = (double*)xmalloc(sizeof(double) * (size_t)(m > 0 ? m * m : 1));
  for (int c = 0; c < m; ++c) for (int r = 0; r <= c; ++r) U[r + (size_t)c * m] = J[r + (size_t)c * m];
  int info = chol_upper(U, m);
  if (info) { free(U); *norm = NAN; return info; }
  double logdet = 0.0, quad = 0.0;
  for (int c = 0; c < m; ++c) {
    double v = h[c];
    for (int k = 0; k < c; ++k) v -= U[k + (size_t)c * m] * mu[k];
    mu[c] = v / U[c + (size_t)c * m];
  }
  for (int c = m - 1; c >= 0; --c) {
    double v = mu[c];
    for (int k = c + 1; k < m; ++k) v -= U[c + (size_t)k * m] * mu[k];
    mu[c] = v / U[c + (size_t)c * m];
  }
  for (int c = 0; c < m; ++c) { logdet += log(U[c + (size_t)c * m]); quad += h[c] * mu[c]; }
  *norm = g + ((double)m * LOG2PI - 2.0 * logdet + quad) / 2.0;
  free(U);
  return 0;
}
