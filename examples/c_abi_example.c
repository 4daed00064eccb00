/* Minimal C host of the engine: a two-cluster cluster graph, two calibration iterations (the second one only confirms: iscal), the log-likelihood.
 * Shows that include/pgbp.h is a plain C header (no C++, no torch types) and what a foreign-function binding has
 * to pass.  Build:  gcc -std=c99 -I include examples/c_abi_example.c -L phylogaussianbeliefprop.jl_amd/csrc -lpgbp
 * (run with LD_LIBRARY_PATH pointing at the library, on a machine with a GPU).
 *
 * Model: x2 | x1 ~ N(x1, 1) with x1 ~ N(0, 1) and x2 observed = 0.5:  clusters {x1} (prior) and {x1} (likelihood
 * of the observation as a function of x1), sepset {x1}; log-likelihood = log N(0.5; 0, 2). */
#include <math.h>
#include <stdio.h>

#include "pgbp.h"

int main(void) {
  /* beliefs: cluster 0 (dim 1), cluster 1 (dim 1), sepset (dim 1); packed = [J | h | g] each */
  const int32_t dims[3] = {1, 1, 1};
  const int32_t sepset_clusters[2] = {0, 1};
  const int64_t scope_off[3] = {0, 1, 2};
  const int32_t scope_idx[2] = {0, 0};
  const double log2pi = 1.8378770664093453;
  double packed[9] = {
      1.0, 0.0, -0.5 * log2pi,              /* prior N(0,1):            J = 1, h = 0,   g = -log(2 pi)/2        */
      1.0, 0.5, -0.5 * log2pi - 0.125,      /* N(0.5; x1, 1) as f(x1):  J = 1, h = 0.5, g = -log(2 pi)/2 - y^2/2 */
      0.0, 0.0, 0.0};                       /* sepset = 1 */
  pgbp_desc d;
  d.n_clusters = 2; d.n_sepsets = 1; d.dims = dims; d.sepset_clusters = sepset_clusters;
  d.scope_off = scope_off; d.scope_idx = scope_idx; d.n_sites = 1; d.device = 0;
  pgbp_engine* e = NULL;
  int rc = pgbp_create(&d, &e);
  if (rc != PGBP_OK) { fprintf(stderr, "pgbp_create: %s\n", pgbp_last_error(NULL)); return 1; }
  const int32_t tree_off[2] = {0, 1}, pa[1] = {0}, ch[1] = {1};
  pgbp_opts o; o.auto_stop = 0; o.update_residualnorm = 1; o.update_residualkldiv = 0; o.reserved = 0; o.atol = 1e-5;
  pgbp_result r;
  double mu = 0.0, norm = 0.0;
  int32_t info = 0;
  if ((rc = pgbp_set_beliefs(e, packed, 1)) || (rc = pgbp_set_schedule(e, 1, tree_off, pa, ch)) ||
      (rc = pgbp_calibrate(e, 2, &o, &r)) || (rc = pgbp_integrate(e, 0, &mu, &norm, &info))) {
    fprintf(stderr, "pgbp: %s\n", pgbp_last_error(e));
    pgbp_destroy(e);
    return 1;
  }
  const double expect = -0.5 * (log2pi + log(2.0)) - 0.25 * 0.25;   /* log N(0.5; 0, 2) */
  printf("succ %d iscal %d  posterior mean of x1 %.6f (0.25)  loglik %.12f (expected %.12f)\n", r.succ, r.iscal, mu, norm,
         expect);
  pgbp_destroy(e);
  return fabs(norm - expect) < 1e-12 ? 0 : 2;
}
