// assignfactors! (src/beliefs.jl:786-861) on the device for every linear-Gaussian model of the reference, on trees
// and networks; missing tip values through scope masks (include/pgbp.h: pgbp_lg_families / pgbp_lg_params).
//
// One factor per node family [child, parent_1 .. parent_K]:
//   X_child | parents ~ N( sum_k q_k X_k + w , V ),   q_k = qc_k * I,  V = sum_k vc_k R[color_k],  w = sum_k wc_k theta
//     Brownian motion (homogeneous / heterogeneous: src/evomodels/homogeneousbrownianmotion.jl:222-351,
//       heterogeneousmodels.jl:110-150):        qc = gamma, vc = gamma^2 t, wc = 0
//     Ornstein-Uhlenbeck (homogeneousornsteinuhlenbeck.jl:51-66), a = exp(-alpha t):
//                                                qc = gamma a, vc = gamma^2 (1 - a^2), wc = gamma (1 - a)
//     hybrid node: the weighted combination of its parent edges (evomodels.jl:314-330); K = 0: root prior
//       (factor_root, evomodels.jl:377-396): V = R[color], w = mu.
// With c_0 = 1 (child), c_k = -q_k and j = V^-1, the factor is exp(-1/2 |sum_a c_a x_a - w|^2_j) / sqrt(det 2 pi V)
// (evomodels.jl:214-245).  Absorbing the evidence (src/beliefupdates.jl:210-274) of the nodes with a fixed value
// (tip: data row y; fixed root: mu) is plugging those values in: z = w - sum_{fixed a} c_a y_a, and on the
// in-scope nodes  J_ab = c_a c_b j,  h_a = c_a j z,  g = -(p log 2pi + log det V + z'jz) / 2.
// mult!(be, factorind, ...) (src/beliefs.jl:859) adds this at the positions of the family's in-scope nodes in its
// cluster; families of one cluster are added in the reference's loop order (one workgroup per cluster: no atomics).
// Missing data: the factor keeps the components O (child_mask) of its residual -- V_OO is inverted instead of V --,
// what absorbleaf! and the partial-scope marginalisation of assignfactors! (src/beliefs.jl:829-857) leave.
#include <hip/hip_runtime.h>

#include "pgbp_bs16.hpp"
#include "pgbp_kernels.hpp"

namespace pgbp {

#define PGBP_LOG2PI 1.8378770664093454835606594728112

extern __shared__ double lg_lds[];

// edge coefficients of parent edge (length t, inheritance gam) under the model
__device__ __forceinline__ void lg_coefs(int model, double alpha, double t, double gam, double& qc, double& vc, double& wc) {
  if (model == PGBP_LG_OU) {
    const double a = exp(-alpha * t);
    qc = gam * a;
    vc = gam * gam * (1.0 - a * a);
    wc = gam * (1.0 - a);
  } else {
    qc = gam;
    vc = gam * gam * t;
    wc = 0.0;
  }
}

// Gauss-Jordan on the p x nc system [V | I | z] (row stride ld) in LDS by one wavefront: the right part becomes
// [V^-1 | V^-1 z]; false when a pivot is not positive.
__device__ __forceinline__ bool lg_gauss_jordan(double* W, int p, int nc, int ld, int lane, double& logdet) {
  double mant = 1.0;
  int expo = 0;
  for (int k = 0; k < p; ++k) {
    const double d = W[k * ld + k];
    if (!(d > 0.0)) return false;
    int ex;
    mant *= frexp(d, &ex);
    expo += ex;
    const double rd = 1.0 / d;
    __syncthreads();
    for (int j = k + 1 + lane; j < nc; j += kWave) W[k * ld + j] *= rd;
    __syncthreads();
    const int ncol = nc - (k + 1);
    for (int idx = lane; idx < (p - 1) * ncol; idx += kWave) {
      int i = idx / ncol;
      const int j = k + 1 + (idx - i * ncol);
      if (i >= k) ++i;
      W[i * ld + j] -= W[i * ld + k] * W[k * ld + j];
    }
    __syncthreads();
  }
  logdet = log(mant) + (double)expo * 0.69314718055994530941723212145818;
  return true;
}

// rank of trait t inside a node's block = number of in-scope traits before it
__device__ __forceinline__ int lg_rank(unsigned long long mask, int t) {
  return __popcll(mask & ((1ull << t) - 1ull));
}

__global__ __launch_bounds__(64) void lg_fill_kernel(LgStatic F, LgParams M, double* __restrict__ pool, int64_t pool_stride,
                                                     double* __restrict__ fpool, int64_t fpool_stride,
                                                     const int64_t* __restrict__ boff, const int32_t* __restrict__ dim,
                                                     int bs, int fp, int rec_cap) {
  const int lane = threadIdx.x, c = blockIdx.x, site = blockIdx.y;
  const int p = F.p, K = F.K;
  const int m = dim[c];
  const int nrec = m * m + m + 1;
  double* __restrict__ out = pool + (int64_t)site * pool_stride + boff[c];
  double* __restrict__ fout = fpool ? fpool + (int64_t)site * fpool_stride + boff[c] : nullptr;
  // [m*m | m | 1] plain column-major accumulator of the cluster: in LDS, or -- a cluster of more than kLdsMaxDim variables,
  // whose record exceeds the LDS -- the cluster's own record in the pool (never packed: accumulated in place; only this
  // wavefront touches it, the barriers between its steps order the accesses)
  const bool in_place = m > kLdsMaxDim;
  double* rec = in_place ? out : lg_lds;
  double* W = lg_lds + rec_cap;       // p x ld
  const int ldmax = (2 * p + 1) | 1;
  double* cz = W + p * ldmax;         // c_a (K+1)
  int* ipos = reinterpret_cast<int*>(cz + (K + 1));   // positions (K+1), then the kept traits o_idx (p)
  int* oidx = ipos + (K + 1);
  for (int t = lane; t < nrec; t += kWave) rec[t] = 0.0;
  const int64_t ps = M.per_site ? site : 0;
  const double* __restrict__ R = M.R + ps * F.n_rates * p * p;
  const double* __restrict__ mu = M.mu + ps * p;
  const double* __restrict__ theta = M.theta ? M.theta + ps * p : nullptr;
  const double alpha = (M.model == PGBP_LG_OU) ? M.alpha[ps] : 0.0;
  const unsigned long long full = p >= 64 ? ~0ull : ((1ull << p) - 1ull);
  bool bad = false;
  for (int fi = F.cl_off[c]; fi < F.cl_off[c + 1]; ++fi) {
    const int f = F.cl_fam[fi];
    const int np = F.n_parents[f];
    const int cpos = F.child_pos[f];
    // O: the components of the residual that the factor keeps (all p without missing data)
    const unsigned long long O = F.child_mask ? (F.child_mask[f] & full) : full;
    const int mo = __popcll(O);
    if (mo == 0) continue;            // nothing observed / in scope below this node: the factor integrates to 1
    const int nc = 2 * mo + 1, ld = nc | 1;
    __syncthreads();
    // coefficients c_a, positions, kept traits (lane 0 .. publish)
    if (lane <= np) {
      if (lane == 0) {
        cz[0] = 1.0;
        ipos[0] = cpos;
      } else {
        double qc, vc, wc;
        lg_coefs(M.model, alpha, F.length[(int64_t)f * K + lane - 1], F.gamma[(int64_t)f * K + lane - 1], qc, vc, wc);
        cz[lane] = -qc;
        ipos[lane] = F.parent_pos[(int64_t)f * K + lane - 1];
      }
    }
    for (int t = lane; t < p; t += kWave)
      if ((O >> t) & 1ull) oidx[lg_rank(O, t)] = t;
    __syncthreads();
    // V_OO, identity, z_O
    for (int idx = lane; idx < mo * mo; idx += kWave) {
      const int j = idx / mo, i = idx - j * mo;
      const int e = oidx[i] + oidx[j] * p;
      double v = 0.0;
      if (np == 0) {
        v = R[(int64_t)F.color[(int64_t)f * K] * p * p + e];
      } else {
        for (int k = 0; k < np; ++k) {
          double qc, vc, wc;
          lg_coefs(M.model, alpha, F.length[(int64_t)f * K + k], F.gamma[(int64_t)f * K + k], qc, vc, wc);
          v += vc * R[(int64_t)F.color[(int64_t)f * K + k] * p * p + e];
        }
      }
      W[i * ld + j] = v;
      W[i * ld + mo + j] = (i == j) ? 1.0 : 0.0;
    }
    for (int i = lane; i < mo; i += kWave) {
      const int t = oidx[i];
      double z;
      if (np == 0) {
        z = mu[t];
      } else {
        z = 0.0;
        for (int k = 0; k < np; ++k) {
          double qc, vc, wc;
          lg_coefs(M.model, alpha, F.length[(int64_t)f * K + k], F.gamma[(int64_t)f * K + k], qc, vc, wc);
          if (theta) z += wc * theta[t];
          if (F.parent_pos[(int64_t)f * K + k] < 0) z += qc * mu[t];   // - c_k mu, c_k = -q_k: fixed root
        }
        if (cpos < 0) z -= F.data[((int64_t)site * F.n_rows + F.data_row[f]) * p + t];  // - c_0 y: tip
      }
      W[i * ld + 2 * mo] = z;
    }
    __syncthreads();
    // keep z: the elimination overwrites the last column with j z
    double zi = (lane < mo) ? W[lane * ld + 2 * mo] : 0.0;
    double logdet;
    if (!lg_gauss_jordan(W, mo, nc, ld, lane, logdet)) { bad = true; break; }
    // quadratic term z' j z
    double q = (lane < mo) ? zi * W[lane * ld + 2 * mo] : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    // J_ab += c_a c_b j (upper triangle of j mirrored: exactly symmetric), h_a += c_a j z, at the positions of the kept
    // traits inside each in-scope node's block
    const int nn = np + 1;
    for (int idx = lane; idx < nn * nn * mo * mo; idx += kWave) {
      const int ab = idx / (mo * mo), ik = idx - ab * mo * mo;
      const int a = ab / nn, b = ab - a * nn;
      const int k = ik / mo, i = ik - k * mo;
      const int pa = ipos[a], pb = ipos[b];
      if (pa < 0 || pb < 0) continue;
      const unsigned long long ma = (a == 0 || !F.parent_mask) ? (a == 0 ? O : full) : F.parent_mask[(int64_t)f * K + a - 1];
      const unsigned long long mb = (b == 0 || !F.parent_mask) ? (b == 0 ? O : full) : F.parent_mask[(int64_t)f * K + b - 1];
      const double jv = (i <= k) ? W[i * ld + mo + k] : W[k * ld + mo + i];
      rec[(pa + lg_rank(ma, oidx[i])) + (int64_t)(pb + lg_rank(mb, oidx[k])) * m] += cz[a] * cz[b] * jv;
    }
    for (int idx = lane; idx < nn * mo; idx += kWave) {
      const int a = idx / mo, i = idx - a * mo;
      if (ipos[a] < 0) continue;
      const unsigned long long ma = (a == 0 || !F.parent_mask) ? (a == 0 ? O : full) : F.parent_mask[(int64_t)f * K + a - 1];
      rec[m * m + ipos[a] + lg_rank(ma, oidx[i])] += cz[a] * W[i * ld + 2 * mo];
    }
    if (lane == 0) rec[m * m + m] += -0.5 * ((double)mo * PGBP_LOG2PI + logdet + q);
  }
  __syncthreads();
  if (bad && lane == 0) rec[m * m + m] = NAN;  // a variance that is not positive definite: the reference throws here
  __syncthreads();
  const bool packed = bs && bs16::applies(m, fp);
  if (!packed) {
    for (int t = lane; t < nrec; t += kWave) {
      const double v = rec[t];
      if (!in_place) out[t] = v;
      if (fout) fout[t] = v;
    }
  } else {
    for (int idx = lane; idx < m * m; idx += kWave) {
      const int cc = idx / m, rr = idx - cc * m;
      if (!bs16::canonical(m, rr, cc, fp)) continue;
      const int o = bs16::J_off(m, rr, cc, fp);
      out[o] = rec[idx];
      if (fout) fout[o] = rec[idx];
    }
    for (int t = lane; t < m; t += kWave) {
      const int o = bs16::h_off(m, t, fp);
      out[o] = rec[m * m + t];
      if (fout) fout[o] = rec[m * m + t];
    }
    if (lane == 0) {
      const int o = bs16::g_off(m, fp);
      out[o] = rec[m * m + m];
      if (fout) fout[o] = rec[m * m + m];
    }
  }
}

// The same for univariate batches in the site-minor layout (every belief dimension <= 2): thread = (cluster, site).
__global__ __launch_bounds__(256) void lg_fill_uni_sm_kernel(LgStatic F, LgParams M, double* __restrict__ pool,
                                                             double* __restrict__ fpool, const int64_t* __restrict__ poff,
                                                             const int32_t* __restrict__ dim, int n_clusters, int n_sites) {
  const int site = blockIdx.y * blockDim.x + threadIdx.x;
  if (site >= n_sites) return;
  const int64_t ns = sm_row(n_sites), ps = M.per_site ? site : 0;
  const int K = F.K;
  const double* __restrict__ R = M.R + ps * F.n_rates;
  const double mu = M.mu[ps];
  const double theta = M.theta ? M.theta[ps] : 0.0;
  const double alpha = (M.model == PGBP_LG_OU) ? M.alpha[ps] : 0.0;
  for (int c = blockIdx.x; c < n_clusters; c += gridDim.x) {
    const int m = dim[c];
    double J00 = 0, J10 = 0, J01 = 0, J11 = 0, h0 = 0, h1 = 0, g = 0;
    for (int fi = F.cl_off[c]; fi < F.cl_off[c + 1]; ++fi) {
      const int f = F.cl_fam[fi];
      const int np = F.n_parents[f], cpos = F.child_pos[f];
      if (F.child_mask && !(F.child_mask[f] & 1ull)) continue;   // the trait is missing / out of scope below: factor = 1
      double V = 0.0, z = 0.0;
      // q_k of the first parents, kept from the one evaluation of the edge coefficients (an exp() each under the OU model:
      // they were worked out four times per tree-edge family, and the fill took three times what its stores need)
      double qk0 = 0.0, qk1 = 0.0, qk2 = 0.0, qk3 = 0.0;
      if (np == 0) {
        V = R[F.color[(int64_t)f * K]];
        z = mu;
      } else {
        for (int k = 0; k < np; ++k) {
          double qc, vc, wc;
          lg_coefs(M.model, alpha, F.length[(int64_t)f * K + k], F.gamma[(int64_t)f * K + k], qc, vc, wc);
          V += vc * R[F.color[(int64_t)f * K + k]];
          z += wc * theta;
          if (F.parent_pos[(int64_t)f * K + k] < 0) z += qc * mu;
          if (k == 0) qk0 = qc; else if (k == 1) qk1 = qc; else if (k == 2) qk2 = qc; else if (k == 3) qk3 = qc;
        }
        if (cpos < 0) z -= F.data_sm[(int64_t)F.data_row[f] * ns + site];   // ([row][site]: one line per wavefront, not 64)
      }
      auto q_of = [&](int k) -> double {   // (the same value lg_coefs returns: same inputs, same expression)
        if (k == 0) return qk0;
        if (k == 1) return qk1;
        if (k == 2) return qk2;
        if (k == 3) return qk3;
        double qc, vc, wc;
        lg_coefs(M.model, alpha, F.length[(int64_t)f * K + k], F.gamma[(int64_t)f * K + k], qc, vc, wc);
        return qc;
      };
      const double j = 1.0 / V;
      g += -0.5 * (PGBP_LOG2PI + log(V) + z * j * z);
      if (!(V > 0.0)) g = NAN;
      // in-scope nodes: at most two (dimension <= 2)
      for (int a = 0; a <= np; ++a) {
        const int pa = a == 0 ? cpos : F.parent_pos[(int64_t)f * K + a - 1];
        if (pa < 0) continue;
        const double ca = a > 0 ? -q_of(a - 1) : 1.0;
        if (pa == 0) h0 += ca * j * z; else h1 += ca * j * z;
        for (int b = 0; b <= np; ++b) {
          const int pb = b == 0 ? cpos : F.parent_pos[(int64_t)f * K + b - 1];
          if (pb < 0) continue;
          const double cb = b > 0 ? -q_of(b - 1) : 1.0;
          const double v = ca * cb * j;
          if (pa == 0 && pb == 0) J00 += v;
          else if (pa == 1 && pb == 0) J10 += v;
          else if (pa == 0 && pb == 1) J01 += v;
          else J11 += v;
        }
      }
    }
    const int64_t p0 = poff[c];
    double v[7];
    int len;
    if (m == 2) { v[0] = J00; v[1] = J10; v[2] = J01; v[3] = J11; v[4] = h0; v[5] = h1; v[6] = g; len = 7; }
    else if (m == 1) { v[0] = J00; v[1] = h0; v[2] = g; len = 3; }
    else { v[0] = g; len = 1; }
    for (int t = 0; t < len; ++t) {
      pool[(p0 + t) * ns + site] = v[t];
      if (fpool) fpool[(p0 + t) * ns + site] = v[t];
    }
  }
}

// The same when every cluster holds exactly one family with at most one parent (LgStatic::simple): the family comes as one
// record, the loops over families, parents and pairs of in-scope nodes are gone -- the SAME operations in the SAME order as
// lg_fill_uni_sm_kernel performs for such a family (sums that start from 0.0 included), so the records are bit for bit the
// same; what is saved is a third of the instructions of a kernel bound by instruction issue (40 M threads of an exp, a log,
// a division and the bookkeeping of three nested loops each).
__global__ __launch_bounds__(256) void lg_fill_uni_simple_sm_kernel(LgStatic F, LgParams M, double* __restrict__ pool,
                                                                    double* __restrict__ fpool,
                                                                    const int64_t* __restrict__ poff,
                                                                    const int32_t* __restrict__ dim, int n_clusters,
                                                                    int n_sites) {
  const int site = blockIdx.y * blockDim.x + threadIdx.x;
  if (site >= n_sites) return;
  const int64_t ns = sm_row(n_sites), ps = M.per_site ? site : 0;
  const double* __restrict__ R = M.R + ps * F.n_rates;
  const double mu = M.mu[ps];
  const double theta = M.theta ? M.theta[ps] : 0.0;
  const double alpha = (M.model == PGBP_LG_OU) ? M.alpha[ps] : 0.0;
  for (int c = blockIdx.x; c < n_clusters; c += gridDim.x) {
    const LgSimpleFam f = F.simple[c];
    const int m = dim[c];
    double J00 = 0, J10 = 0, J01 = 0, J11 = 0, h0 = 0, h1 = 0, g = 0;
    double V = 0.0, z = 0.0, q = 0.0;
    if (f.np == 0) {
      V = R[f.color];
      z = mu;
    } else {
      double qc, vc, wc;
      lg_coefs(M.model, alpha, f.length, f.gamma, qc, vc, wc);
      V += vc * R[f.color];
      z += wc * theta;
      if (f.ppos < 0) z += qc * mu;
      q = qc;
      if (f.cpos < 0) z -= F.data_sm[(int64_t)f.row * ns + site];
    }
    const double j = 1.0 / V;
    g += -0.5 * (PGBP_LOG2PI + log(V) + z * j * z);
    if (!(V > 0.0)) g = NAN;
    // in-scope nodes: the child (coefficient 1), then the parent (coefficient -q)
    if (f.cpos >= 0) {
      const double ca = 1.0;
      if (f.cpos == 0) h0 += ca * j * z; else h1 += ca * j * z;
      {
        const double v = ca * 1.0 * j;
        if (f.cpos == 0) J00 += v; else J11 += v;
      }
      if (f.np > 0 && f.ppos >= 0) {
        const double v = ca * (-q) * j;
        if (f.cpos == 0 && f.ppos == 0) J00 += v;
        else if (f.cpos == 1 && f.ppos == 0) J10 += v;
        else if (f.cpos == 0 && f.ppos == 1) J01 += v;
        else J11 += v;
      }
    }
    if (f.np > 0 && f.ppos >= 0) {
      const double ca = -q;
      if (f.ppos == 0) h0 += ca * j * z; else h1 += ca * j * z;
      if (f.cpos >= 0) {
        const double v = ca * 1.0 * j;
        if (f.ppos == 0 && f.cpos == 0) J00 += v;
        else if (f.ppos == 1 && f.cpos == 0) J10 += v;
        else if (f.ppos == 0 && f.cpos == 1) J01 += v;
        else J11 += v;
      }
      {
        const double v = ca * (-q) * j;
        if (f.ppos == 0) J00 += v; else J11 += v;
      }
    }
    const int64_t p0 = poff[c];
    if (m == 2) {
      const double v[7] = {J00, J10, J01, J11, h0, h1, g};
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        pool[(p0 + t) * ns + site] = v[t];
        if (fpool) fpool[(p0 + t) * ns + site] = v[t];
      }
    } else if (m == 1) {
      const double v[3] = {J00, h0, g};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        pool[(p0 + t) * ns + site] = v[t];
        if (fpool) fpool[(p0 + t) * ns + site] = v[t];
      }
    } else {
      pool[p0 * ns + site] = g;
      if (fpool) fpool[p0 * ns + site] = g;
    }
  }
}

void launch_lg_fill(const LgStatic& F, const LgParams& M, double* pool, int64_t pool_stride, double* fpool,
                    int64_t fpool_stride, const int64_t* d_boff, const int32_t* d_dim, int bs16, int fast_p, int max_dim,
                    int n_clusters, int n_sites, hipStream_t st) {
  if (n_clusters <= 0) return;
  const int mm = max_dim < 1 ? 1 : (max_dim > kLdsMaxDim ? kLdsMaxDim : max_dim);   // (larger clusters accumulate in place)
  const int rec_cap = (mm * mm + mm + 1 + 1) & ~1;
  const int ld = (2 * F.p + 1) | 1;
  const size_t doubles = (size_t)rec_cap + (size_t)F.p * ld + (size_t)(F.K + 1) + (size_t)(F.K + 2 + F.p) / 2 + 2;
  if (doubles * sizeof(double) > 64 * 1024) {  // a 128-dimensional cluster record is 132 KB of the CU's 160 KB
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lg_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(doubles * sizeof(double)));
    (void)hipGetLastError();
  }
  hipLaunchKernelGGL(lg_fill_kernel, dim3(n_clusters, n_sites), dim3(kWave), doubles * sizeof(double), st, F, M, pool,
                     pool_stride, fpool, fpool_stride, d_boff, d_dim, bs16, fast_p, rec_cap);
}

void launch_lg_fill_uni_sm(const LgStatic& F, const LgParams& M, double* pool_sm, double* fpool_sm, const int64_t* d_poff,
                           const int32_t* d_dim, int n_clusters, int n_sites, hipStream_t st) {
  if (n_clusters <= 0) return;
  const int bs = n_sites >= 256 ? 256 : 64;
  // (one cluster per workgroup up to 2^20 of them: with a grid-stride loop over a capped grid the workgroups that take one
  // cluster more than the others set the kernel's time -- 40 000 clusters over 16 384 workgroups: three rounds for 2.44 of work)
  const int gx = n_clusters < (1 << 20) ? n_clusters : (1 << 20);
  if (F.simple) {
    hipLaunchKernelGGL(lg_fill_uni_simple_sm_kernel, dim3(gx, (n_sites + bs - 1) / bs), dim3(bs), 0, st, F, M, pool_sm, fpool_sm,
                       d_poff, d_dim, n_clusters, n_sites);
    return;
  }
  hipLaunchKernelGGL(lg_fill_uni_sm_kernel, dim3(gx, (n_sites + bs - 1) / bs), dim3(bs), 0, st, F, M, pool_sm, fpool_sm,
                     d_poff, d_dim, n_clusters, n_sites);
}

}  // namespace pgbp
