// Register-resident message kernel for gfx950 (wave64), sepsets of dimension P = 16.
//
// One wavefront = one task (ordered list of messages sharing a receiver or a sender), no LDS
// allocation, every HBM access a 16-byte-per-lane coalesced vector access issued up front.
//
// Lane geometry: lane = 8*b + a, a = row group, b = column group (0..7 each).
// A 32 x 32 sender precision, re-indexed so that the 16 integrated variables come first
// ("logical" index), is spread as 4 x 4 register blocks:
//     lane (a, b) holds W[R_a(i)][C_b(j)],  R_a = {2a, 2a+1, 2a+16, 2a+17}, C_b likewise with b.
// Rows 2a, 2a+1 are consecutive in memory (column-major J) -> one double2 load per (lane, column);
// the 8 lanes of one b cover one full 128-byte line.  The integrated block is rows/cols i, j < 2,
// the kept block i, j >= 2, which is exactly the sepset's 16 x 16 layout (2 x 2 block per lane) and the
// receiver's sub-block layout: divide!/mult! need no data movement at all.
//
// marginalize (src/beliefupdates.jl:55-83) = 16 symmetric rank-1 eliminations
//     W <- W - x x' / d,   x = column k of W (rows > k), d = W[k][k]
// which reads exactly what the reference reads: upper(J_I) (the integrated block is symmetrised from its
// upper triangle at load, like PDMat(Symmetric(J_I)) at :68), J_SI (Jki at :59) and J_S.  J_IS is never
// loaded.  Column k is broadcast with ds_bpermute (no LDS allocation), pivots with v_readlane.
// log det J_I is accumulated as mantissa product + exponent sum (one log per message).
#include <hip/hip_runtime.h>

#include "pgbp_kernels.hpp"

namespace pgbp {

#define PGBP_LOG2PI 1.8378770664093454835606594728112
#define PGBP_LN2 0.69314718055994530941723212145818
#define PGBP_EPS 2.220446049250313e-16

namespace {

constexpr int P = 16;

struct Frag {
  double w[4][4];  // w[i][j] = W[R(i)][C(j)]; w[0..1][2..3] (integrated rows x kept cols) is never used
  double h[4];     // h[i] = h[R(i)], replicated over b
};

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

// Elimination steps K .. 15 (pivot K lives in lane row/col K>>1, register K&1). Returns 0 or the 1-based
// index of the first non-positive pivot.
template <int K>
__device__ __forceinline__ int eliminate(Frag& f, const int a, const int b, double& mant, int& expo, double& quad) {
  if constexpr (K == P) {
    return 0;
  } else {
    constexpr int kk = K >> 1, ik = K & 1;
    const int src_r = kk * 8 + a;  // lane (a, kk): column K entries for my rows R_a
    const int src_c = kk * 8 + b;  // lane (b, kk): column K entries for the rows C_b (my columns)
    double xr[4], xc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xr[i] = __shfl(f.w[i][ik], src_r);
      xc[i] = __shfl(f.w[i][ik], src_c);
    }
    const double d = readlane_f64(f.w[ik][ik], kk * 8 + kk);
    const double hk = readlane_f64(f.h[ik], kk);
    if (!(d > 0.0)) return K + 1;
    const double rd = 1.0 / d;
    int e;
    mant *= frexp(d, &e);
    expo += e;
    quad = fma(hk * hk, rd, quad);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(i < 2 && j >= 2)) f.w[i][j] = fma(-(xr[i] * xc[j]), rd, f.w[i][j]);  // (x_r x_c) first: exactly symmetric
      f.h[i] = fma(-(xr[i] * hk), rd, f.h[i]);
    }
    return eliminate<K + 1>(f, a, b, mant, expo, quad);
  }
}

}  // namespace

__global__ __launch_bounds__(64) void bp_level_fast16(DevState S, const int32_t* __restrict__ task_off,
                                                      const Entry* __restrict__ entries, int task0,
                                                      unsigned long long seq_base, unsigned long long stop_below) {
  const int lane = threadIdx.x;
  const int site = blockIdx.y;
  if ((S.fail[site] >> kInfoBits) < stop_below) return;  // an earlier traversal failed (see bp_level_generic)
  const int task = task0 + blockIdx.x;
  const int a = lane & 7, b = lane >> 3;
  double* __restrict__ pool = S.pool + (int64_t)site * S.pool_stride;
  double* __restrict__ rpool = S.rpool + (int64_t)site * S.rpool_stride;

  double mJ[2][2] = {{0, 0}, {0, 0}}, mh[2] = {0, 0}, gmsg = 0.0;  // the message: rows 2a+ii, cols 2b+jj
  double tJ[2][2] = {{0, 0}, {0, 0}}, th[2] = {0, 0}, tg = 0.0;    // receiver block accumulators
  const int e0 = task_off[task], e1 = task_off[task + 1];
  for (int e = e0; e < e1; ++e) {
    const Entry en = entries[e];
    const MsgDesc m = S.msgs[en.msg];
    if (S.poison[(int64_t)site * S.n_clusters + m.from_b]) {
      if (lane == 0) S.poison[(int64_t)site * S.n_clusters + m.to_b] = 1;
      return;
    }
    // ---- issue every load of this entry up front: sepset, receiver block, sender
    double* __restrict__ sep = pool + m.sep_off;
    double* __restrict__ to = pool + m.to_off;
    double* __restrict__ res = rpool + m.res_off;
    const int mt = m.mt, up0 = m.up0;
    const int64_t so = 2 * a + P * (2 * b);
    const int64_t tO = (up0 + 2 * a) + (int64_t)mt * (up0 + 2 * b);
    const double2 s0 = *reinterpret_cast<const double2*>(sep + so);
    const double2 s1 = *reinterpret_cast<const double2*>(sep + so + P);
    double2 sh = make_double2(0.0, 0.0);
    if (b == 0) sh = *reinterpret_cast<const double2*>(sep + P * P + 2 * a);
    const double sg = sep[P * P + P];
    if (en.tflags & kTLoad) {
      const double2 t0 = *reinterpret_cast<const double2*>(to + tO);
      const double2 t1 = *reinterpret_cast<const double2*>(to + tO + mt);
      tJ[0][0] = t0.x; tJ[1][0] = t0.y; tJ[0][1] = t1.x; tJ[1][1] = t1.y;
      if (b == 0) {
        const double2 t2 = *reinterpret_cast<const double2*>(to + (int64_t)mt * mt + up0 + 2 * a);
        th[0] = t2.x; th[1] = t2.y;
      }
      tg = to[(int64_t)mt * mt + mt];
    }
    if (!en.reuse) {
      const double* __restrict__ from = pool + m.from_off;
      if (m.ni == 0) {
        // nothing to integrate: the message is the sender's belief (src/beliefupdates.jl:56)
        const double2 c0 = *reinterpret_cast<const double2*>(from + so);
        const double2 c1 = *reinterpret_cast<const double2*>(from + so + P);
        const double2 ch = *reinterpret_cast<const double2*>(from + P * P + 2 * a);
        mJ[0][0] = c0.x; mJ[1][0] = c0.y; mJ[0][1] = c1.x; mJ[1][1] = c1.y;
        mh[0] = ch.x; mh[1] = ch.y;
        gmsg = from[P * P + P];
      } else {
        // logical index = original index rotated so that the integrated block comes first
        const int rot = (m.keep0 == 0) ? P : 0;
        const int r0 = (2 * a + rot) & 31, r1 = (2 * a + P + rot) & 31;
        Frag f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cl = 2 * b + (j & 1) + (j >> 1) * P;  // logical column C_b(j)
          const int64_t co = (int64_t)((cl + rot) & 31) * 32;
          if (j < 2) {
            const double2 v = *reinterpret_cast<const double2*>(from + r0 + co);
            f.w[0][j] = v.x; f.w[1][j] = v.y;
          } else {
            f.w[0][j] = 0.0; f.w[1][j] = 0.0;
          }
          const double2 u = *reinterpret_cast<const double2*>(from + r1 + co);
          f.w[2][j] = u.x; f.w[3][j] = u.y;
        }
        {
          const double2 v = *reinterpret_cast<const double2*>(from + 32 * 32 + r0);
          const double2 u = *reinterpret_cast<const double2*>(from + 32 * 32 + r1);
          f.h[0] = v.x; f.h[1] = v.y; f.h[2] = u.x; f.h[3] = u.y;
        }
        gmsg = from[32 * 32 + 32];
        // "fake" message: J_I, J_SI, h_I all ~ 0 (src/beliefupdates.jl:62-66), tested on the raw values
        bool nz = fabs(f.h[0]) > PGBP_EPS || fabs(f.h[1]) > PGBP_EPS;
#pragma unroll
        for (int i = 0; i < 4; ++i) nz |= fabs(f.w[i][0]) > PGBP_EPS || fabs(f.w[i][1]) > PGBP_EPS;
        if (__any(nz)) {
          // Symmetric(J_I): entries below the diagonal take the value of their transpose (:68)
          const int tl = a * 8 + b;  // lane holding the transposed 2 x 2 block
          const double t00 = __shfl(f.w[0][0], tl), t01 = __shfl(f.w[1][0], tl);
          const double t10 = __shfl(f.w[0][1], tl), t11 = __shfl(f.w[1][1], tl);
          if (2 * a + 0 > 2 * b + 0) f.w[0][0] = t00;
          if (2 * a + 0 > 2 * b + 1) f.w[0][1] = t01;
          if (2 * a + 1 > 2 * b + 0) f.w[1][0] = t10;
          if (2 * a + 1 > 2 * b + 1) f.w[1][1] = t11;
          double mant = 1.0, quad = 0.0;
          int expo = 0;
          const int info = eliminate<0>(f, a, b, mant, expo, quad);
          if (info != 0) {
            if (lane == 0) {
              S.status[(int64_t)site * S.n_msgs + en.msg] = info;
              S.poison[(int64_t)site * S.n_clusters + m.to_b] = 1;
              atomicMin(&S.fail[site], ((seq_base + (unsigned long long)en.seq) << kInfoBits) |
                                           (unsigned long long)info);
            }
            return;
          }
          const double logdet = log(mant) + (double)expo * PGBP_LN2;
          gmsg += 0.5 * ((double)P * PGBP_LOG2PI - logdet + quad);  // :81
        }
        mJ[0][0] = f.w[2][2]; mJ[1][0] = f.w[3][2]; mJ[0][1] = f.w[2][3]; mJ[1][1] = f.w[3][3];
        mh[0] = f.h[2]; mh[1] = f.h[3];
      }
    }
    // ---- divide! (src/beliefupdates.jl:579-587) and mult! (:483-488)
    const double d00 = mJ[0][0] - s0.x, d10 = mJ[1][0] - s0.y, d01 = mJ[0][1] - s1.x, d11 = mJ[1][1] - s1.y;
    *reinterpret_cast<double2*>(sep + so) = make_double2(mJ[0][0], mJ[1][0]);
    *reinterpret_cast<double2*>(sep + so + P) = make_double2(mJ[0][1], mJ[1][1]);
    *reinterpret_cast<double2*>(res + so) = make_double2(d00, d10);
    *reinterpret_cast<double2*>(res + so + P) = make_double2(d01, d11);
    tJ[0][0] += d00; tJ[1][0] += d10; tJ[0][1] += d01; tJ[1][1] += d11;
    double maxJ = fmax(fmax(fabs(d00), fabs(d10)), fmax(fabs(d01), fabs(d11)));
    if (d00 != d00 || d10 != d10 || d01 != d01 || d11 != d11) maxJ = INFINITY;
    double maxh = 0.0;
    if (b == 0) {
      const double dh0 = mh[0] - sh.x, dh1 = mh[1] - sh.y;
      *reinterpret_cast<double2*>(sep + P * P + 2 * a) = make_double2(mh[0], mh[1]);
      *reinterpret_cast<double2*>(res + P * P + 2 * a) = make_double2(dh0, dh1);
      th[0] += dh0; th[1] += dh1;
      maxh = (dh0 != dh0 || dh1 != dh1) ? INFINITY : fmax(fabs(dh0), fabs(dh1));
    }
    if (lane == 0) {
      sep[P * P + P] = gmsg;
      tg += gmsg - sg;
      S.status[(int64_t)site * S.n_msgs + en.msg] = 0;
    }
    if (en.tflags & kTStore) {
      *reinterpret_cast<double2*>(to + tO) = make_double2(tJ[0][0], tJ[1][0]);
      *reinterpret_cast<double2*>(to + tO + mt) = make_double2(tJ[0][1], tJ[1][1]);
      if (b == 0) *reinterpret_cast<double2*>(to + (int64_t)mt * mt + up0 + 2 * a) = make_double2(th[0], th[1]);
      if (lane == 0) to[(int64_t)mt * mt + mt] = tg;
    }
    if (S.update_resnorm) {
      // iscalibrated_residnorm! (src/beliefs.jl:994-1003)
      maxJ = wave_max_f64(maxJ);
      maxh = wave_max_f64(maxh);
      if (lane == 0)
        S.flags[(int64_t)site * S.n_msgs + en.msg] =
            (maxh / sqrt((double)P) <= S.atol && maxJ / sqrt((double)P * (double)P) <= S.atol) ? 1 : 0;
    }
    if (e + 1 < e1) __threadfence_block();  // a later entry may read-modify-write the same receiver through memory
  }
}

void launch_level_fast16(const DevState& S, const int32_t* d_task_off, const Entry* d_entries, int task0,
                         int ntasks, int n_sites, unsigned long long seq_base, unsigned long long stop_below,
                         hipStream_t st) {
  if (ntasks <= 0) return;
  hipLaunchKernelGGL(bp_level_fast16, dim3(ntasks, n_sites), dim3(kWave), 0, st, S, d_task_off, d_entries, task0,
                     seq_base, stop_below);
}

}  // namespace pgbp
