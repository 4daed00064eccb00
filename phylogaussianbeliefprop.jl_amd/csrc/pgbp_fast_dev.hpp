// Device helpers shared by the register-resident message kernels (pgbp_fast.hip: one wavefront per message;
// pgbp_task.hip: one wavefront per task): lane geometry, the blocked elimination, tile loads / stores, record fetch.
#pragma once
#include <hip/hip_runtime.h>

#include "pgbp_bs16.hpp"
#include "pgbp_kernels.hpp"

namespace pgbp {

#define PGBP_LOG2PI 1.8378770664093454835606594728112
#define PGBP_LN2 0.69314718055994530941723212145818
#define PGBP_EPS 2.220446049250313e-16

namespace {


struct Frag {
  double w[4][4];  // w[i][j] = W[R(i)][C(j)]; w[0..1][2..3] (integrated rows x kept cols) is never used
  double h[4];     // h[i] = h[R(i)], replicated over b
};

// Wave-synchronous exchange through LDS: the hardware executes a wave's DS instructions in order, but the
// compiler reasons per thread; this fence pair + wave barrier stops it from forwarding a lane's own stale
// store to its later load or moving loads across the other lanes' stores.  Emits no instruction.
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Blocked elimination of the P integrated variables, two pivots per round (P / 2 rounds).
// Round R: the lanes with b == R own columns 2R, 2R+1 of W; they have published them (rows R_a) in the wave's private LDS
// strip; every lane reads back the entries of the rows R_a (xr) and C_b (xc) plus the 2 x 2 pivot block D and
// h_2R, h_2R+1, and applies  W <- W - X D^-1 X',  h <- h - X D^-1 h_K.  Mathematically identical to two successive
// rank-1 eliminations (src/beliefupdates.jl:68-81); D^-1 by v_rcp_f64 + 2 Newton steps.
// The chain of dependent latencies of a round is  strip -> D^-1 -> the two columns of the NEXT pivot -> strip, so a round
//   1. updates the integrated columns (where the next pivot's are) and h_I,
//   2. publishes the next pivot's columns into the OTHER of two strips and requests its pivot block back at once (the rest
//      of the columns at the top of the next round: they arrive while D^-1 is worked out),
//   3. updates the kept columns and h_S while that round trip is in flight;
// and it never branches: a non-positive pivot is remembered (the first one) and the arithmetic carries on on whatever it
// has -- the caller discards everything when the result is not 0.
// Returns 0, or the 1-based index of the first non-positive pivot (LAPACK potrf info).
constexpr int kColStride = 10;                       // doubles per owner-lane slot (80 B: conflict-free b128 reads)
constexpr int kColStrip = 8 * kColStride + 4;        // + h_2R, h_2R+1
constexpr int kColDoubles = 2 * kColStrip;           // two strips: a round reads one while the next pivot goes into the other

struct PivotOps {   // the pivot block and h_K: what the chain of a round starts from
  double2 p0, p1, hk;
};

template <int R>
__device__ __forceinline__ void publish_pivot(const Frag& f, const int a, const int b, const bool act, double* __restrict__ col) {
  double* strip = col + (R & 1) * kColStrip;
  if (act && b == R) {
    double* dst = strip + a * kColStride;
    *reinterpret_cast<double4*>(dst) = make_double4(f.w[0][0], f.w[1][0], f.w[2][0], f.w[3][0]);
    *reinterpret_cast<double4*>(dst + 4) = make_double4(f.w[0][1], f.w[1][1], f.w[2][1], f.w[3][1]);
    if (a == R) *reinterpret_cast<double2*>(strip + 8 * kColStride) = make_double2(f.h[0], f.h[1]);
  }
}
template <int R>
__device__ __forceinline__ PivotOps fetch_pivot(const double* __restrict__ col) {
  const double* strip = col + (R & 1) * kColStrip;
  PivotOps o;
  o.p0 = *reinterpret_cast<const double2*>(strip + R * kColStride);      // W[2R][2R], W[2R+1][2R]
  o.p1 = *reinterpret_cast<const double2*>(strip + R * kColStride + 4);  // W[2R][2R+1], W[2R+1][2R+1]
  o.hk = *reinterpret_cast<const double2*>(strip + 8 * kColStride);
  return o;
}

template <int P, int R>
__device__ __forceinline__ int eliminate_round(Frag& f, const int a, const int b, const bool act, double* __restrict__ col,
                                               const PivotOps& o, int bad, double& mant, int& expo, double& quad) {
  if constexpr (R == P / 2) {
    return bad;
  } else {
    // the rows R_a (xr) and C_b (xc) of the pivot's columns: they arrive while D^-1 is worked out
    const double* strip = col + (R & 1) * kColStrip;
    const double4 xr0 = *reinterpret_cast<const double4*>(strip + a * kColStride);
    const double4 xr1 = *reinterpret_cast<const double4*>(strip + a * kColStride + 4);
    const double4 xc0 = *reinterpret_cast<const double4*>(strip + b * kColStride);
    const double4 xc1 = *reinterpret_cast<const double4*>(strip + b * kColStride + 4);
    const double d00 = o.p0.x, d01 = o.p1.x, d11 = o.p1.y;  // upper triangle of the pivot block
    const double det = fma(d00, d11, -(d01 * d01));
    const int bad_here = __builtin_amdgcn_readfirstlane(!(d00 > 0.0) ? 2 * R + 1 : (!(det > 0.0) ? 2 * R + 2 : 0));
    bad = bad ? bad : bad_here;
    double rdet = __builtin_amdgcn_rcp(det);
    rdet = fma(fma(-det, rdet, 1.0), rdet, rdet);
    rdet = fma(fma(-det, rdet, 1.0), rdet, rdet);
    const double e00 = d11 * rdet, e01 = -(d01 * rdet), e11 = d00 * rdet;
    const double xc0v[4] = {xc0.x, xc0.y, xc0.z, xc0.w}, xc1v[4] = {xc1.x, xc1.y, xc1.z, xc1.w};
    const double xr0v[4] = {xr0.x, xr0.y, xr0.z, xr0.w}, xr1v[4] = {xr1.x, xr1.y, xr1.z, xr1.w};
    // y = D^-1 x_c for my 4 columns; g = D^-1 h_K
    const double g0 = fma(e00, o.hk.x, e01 * o.hk.y), g1 = fma(e01, o.hk.x, e11 * o.hk.y);
    // 1. the integrated columns and h_I
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double y0 = fma(e00, xc0v[j], e01 * xc1v[j]);
      const double y1 = fma(e01, xc0v[j], e11 * xc1v[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) f.w[i][j] = fma(-xr0v[i], y0, fma(-xr1v[i], y1, f.w[i][j]));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) f.h[i] = fma(-xr0v[i], g0, fma(-xr1v[i], g1, f.h[i]));
    // (what step 3 needs of this round's operands, so that nothing else of them is live beside the next round's)
    const double ys0[2] = {fma(e00, xc0v[2], e01 * xc1v[2]), fma(e00, xc0v[3], e01 * xc1v[3])};
    const double ys1[2] = {fma(e01, xc0v[2], e11 * xc1v[2]), fma(e01, xc0v[3], e11 * xc1v[3])};
    quad = fma(o.hk.x, g0, fma(o.hk.y, g1, quad));
    int e;
    mant *= frexp(det, &e);
    expo += e;
    // 2. the next pivot on its way
    PivotOps nx{};
    if constexpr (R + 1 < P / 2) {
      publish_pivot<R + 1>(f, a, b, act, col);
      wave_sync_lds();  // other LANES wrote what this lane reads: not visible to per-thread alias analysis
      nx = fetch_pivot<R + 1>(col);
    }
    // 3. the kept columns and h_S
#pragma unroll
    for (int j = 2; j < 4; ++j)
#pragma unroll
      for (int i = 2; i < 4; ++i) f.w[i][j] = fma(-xr0v[i], ys0[j - 2], fma(-xr1v[i], ys1[j - 2], f.w[i][j]));
#pragma unroll
    for (int i = 2; i < 4; ++i) f.h[i] = fma(-xr0v[i], g0, fma(-xr1v[i], g1, f.h[i]));
    // (the kept block, the quadratic form and the determinant are chains of their own -- no later pivot needs them -- and
    // instruction selection would let all of their updates sink to the end of the elimination, every round's operands live
    // until then: the empty statement below is ordered against the next round's strip accesses and wants them computed)
    asm volatile("; round done"
                 : "+v"(f.w[2][2]), "+v"(f.w[3][2]), "+v"(f.w[2][3]), "+v"(f.w[3][3]), "+v"(f.h[2]), "+v"(f.h[3]), "+v"(quad),
                   "+v"(mant), "+v"(expo));
    return eliminate_round<P, R + 1>(f, a, b, act, col, nx, bad, mant, expo, quad);
  }
}

template <int P, int R>
__device__ __forceinline__ int eliminate2(Frag& f, const int a, const int b, const bool act, double* __restrict__ col,
                                          double& mant, int& expo, double& quad) {
  static_assert(R == 0, "the elimination starts at the first pivot");
  publish_pivot<0>(f, a, b, act, col);
  wave_sync_lds();
  const PivotOps o = fetch_pivot<0>(col);
  const int bad = eliminate_round<P, 0>(f, a, b, act, col, o, 0, mant, expo, quad);
  wave_sync_lds();  // (whatever comes next in this strip)
  return bad;
}

// ---- The same elimination on TWO wavefronts (pgbp_loop.hip): the PIVOT wavefront keeps the integrated columns -- the
// chain strip -> D^-1 -> next pivot's columns -> strip -- and publishes every pivot in a strip of its own (P / 2 strips per
// record, nothing is ever overwritten inside a pass) followed by a progress word; the KEPT wavefront follows it round by
// round on the kept block, h_S, the quadratic form and the determinant.  Same operands, same operations in the same order
// as eliminate_round above on each entry: bit-identical.
// progress word: `base + r` = the strips of rounds 0 .. r - 1 are published (r = 1 .. P / 2); the LDS unit serves a
// wavefront's accesses in order, so a wavefront that reads the word after it was written finds the strips behind it.
__device__ __forceinline__ void progress_store(int* word, int value, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  if (lane == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int progress_wait(int* word, int at_least) {
  int v;
  do {
    v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  } while (v < at_least);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  return v;
}

template <int R>
__device__ __forceinline__ void publish_pivot_strip(const Frag& f, const int a, const int b, const bool act,
                                                    double* __restrict__ strips) {
  double* strip = strips + R * kColStrip;
  if (act && b == R) {
    double* dst = strip + a * kColStride;
    *reinterpret_cast<double4*>(dst) = make_double4(f.w[0][0], f.w[1][0], f.w[2][0], f.w[3][0]);
    *reinterpret_cast<double4*>(dst + 4) = make_double4(f.w[0][1], f.w[1][1], f.w[2][1], f.w[3][1]);
    if (a == R) *reinterpret_cast<double2*>(strip + 8 * kColStride) = make_double2(f.h[0], f.h[1]);
  }
}

// the pivot wavefront's rounds: f.w[0..3][0..1], f.h[0..1] only
template <int P, int R>
__device__ __forceinline__ void pivot_rounds(Frag& f, const int a, const int b, const bool act, const int lane,
                                             double* __restrict__ strips, int* word, const int base) {
  if constexpr (R == 0) {
    publish_pivot_strip<0>(f, a, b, act, strips);
    progress_store(word, base + 1, lane);
  }
  if constexpr (R < P / 2 - 1) {   // (the last pivot's round has nothing left to update on this side)
    wave_sync_lds();
    const double* strip = strips + R * kColStrip;
    const double4 xr0 = *reinterpret_cast<const double4*>(strip + a * kColStride);
    const double4 xr1 = *reinterpret_cast<const double4*>(strip + a * kColStride + 4);
    const double2 xc0 = *reinterpret_cast<const double2*>(strip + b * kColStride);
    const double2 xc1 = *reinterpret_cast<const double2*>(strip + b * kColStride + 4);
    const double2 p0 = *reinterpret_cast<const double2*>(strip + R * kColStride);
    const double2 p1 = *reinterpret_cast<const double2*>(strip + R * kColStride + 4);
    const double2 hk = *reinterpret_cast<const double2*>(strip + 8 * kColStride);
    const double d00 = p0.x, d01 = p1.x, d11 = p1.y;
    const double det = fma(d00, d11, -(d01 * d01));
    double rdet = __builtin_amdgcn_rcp(det);
    rdet = fma(fma(-det, rdet, 1.0), rdet, rdet);
    rdet = fma(fma(-det, rdet, 1.0), rdet, rdet);
    const double e00 = d11 * rdet, e01 = -(d01 * rdet), e11 = d00 * rdet;
    const double xc0v[2] = {xc0.x, xc0.y}, xc1v[2] = {xc1.x, xc1.y};
    const double xr0v[4] = {xr0.x, xr0.y, xr0.z, xr0.w}, xr1v[4] = {xr1.x, xr1.y, xr1.z, xr1.w};
    const double g0 = fma(e00, hk.x, e01 * hk.y), g1 = fma(e01, hk.x, e11 * hk.y);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double y0 = fma(e00, xc0v[j], e01 * xc1v[j]);
      const double y1 = fma(e01, xc0v[j], e11 * xc1v[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) f.w[i][j] = fma(-xr0v[i], y0, fma(-xr1v[i], y1, f.w[i][j]));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) f.h[i] = fma(-xr0v[i], g0, fma(-xr1v[i], g1, f.h[i]));
    publish_pivot_strip<R + 1>(f, a, b, act, strips);
    progress_store(word, base + R + 2, lane);
    pivot_rounds<P, R + 1>(f, a, b, act, lane, strips, word, base);
  }
}

// the kept wavefront's rounds: w22 = (W[S_a][S_b]) block of the lane, hs = h_S entries; returns potrf's info
template <int P, int R>
__device__ __forceinline__ int kept_rounds(double (&w22)[2][2], double (&hs)[2], const int a, const int b,
                                           const double* __restrict__ strips, int* word, const int base, int bad, double& mant,
                                           int& expo, double& quad) {
  if constexpr (R == P / 2) {
    return bad;
  } else {
    progress_wait(word, base + R + 1);
    const double* strip = strips + R * kColStrip;
    const double2 xr0 = *reinterpret_cast<const double2*>(strip + a * kColStride + 2);   // rows S_a of the pivot's columns
    const double2 xr1 = *reinterpret_cast<const double2*>(strip + a * kColStride + 6);
    const double2 xc0 = *reinterpret_cast<const double2*>(strip + b * kColStride + 2);   // rows S_b: the columns of my block
    const double2 xc1 = *reinterpret_cast<const double2*>(strip + b * kColStride + 6);
    const double2 p0 = *reinterpret_cast<const double2*>(strip + R * kColStride);
    const double2 p1 = *reinterpret_cast<const double2*>(strip + R * kColStride + 4);
    const double2 hk = *reinterpret_cast<const double2*>(strip + 8 * kColStride);
    const double d00 = p0.x, d01 = p1.x, d11 = p1.y;
    const double det = fma(d00, d11, -(d01 * d01));
    const int bad_here = __builtin_amdgcn_readfirstlane(!(d00 > 0.0) ? 2 * R + 1 : (!(det > 0.0) ? 2 * R + 2 : 0));
    bad = bad ? bad : bad_here;
    double rdet = __builtin_amdgcn_rcp(det);
    rdet = fma(fma(-det, rdet, 1.0), rdet, rdet);
    rdet = fma(fma(-det, rdet, 1.0), rdet, rdet);
    const double e00 = d11 * rdet, e01 = -(d01 * rdet), e11 = d00 * rdet;
    const double xc0v[2] = {xc0.x, xc0.y}, xc1v[2] = {xc1.x, xc1.y};
    const double xr0v[2] = {xr0.x, xr0.y}, xr1v[2] = {xr1.x, xr1.y};
    const double g0 = fma(e00, hk.x, e01 * hk.y), g1 = fma(e01, hk.x, e11 * hk.y);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double y0 = fma(e00, xc0v[j], e01 * xc1v[j]);
      const double y1 = fma(e01, xc0v[j], e11 * xc1v[j]);
#pragma unroll
      for (int i = 0; i < 2; ++i) w22[i][j] = fma(-xr0v[i], y0, fma(-xr1v[i], y1, w22[i][j]));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) hs[i] = fma(-xr0v[i], g0, fma(-xr1v[i], g1, hs[i]));
    quad = fma(hk.x, g0, fma(hk.y, g1, quad));
    int e;
    mant *= frexp(det, &e);
    expo += e;
    // (as in eliminate_round: these chains would otherwise sink to where they are used, every round's det live until then)
    asm volatile("; round done" : "+v"(quad), "+v"(mant), "+v"(expo));
    return kept_rounds<P, R + 1>(w22, hs, a, b, strips, word, base, bad, mant, expo, quad);
  }
}

}  // namespace

// 2 x 2 block of a lane as (x, y, z, w) = (T(2a,2b), T(2a+1,2b), T(2a,2b+1), T(2a+1,2b+1))
struct Blk { double x, y, z, w; };

// Load / store the lane's block of a 16 x 16 symmetric quantity (sepset J, residual dJ, receiver sub-block,
// a 16-dim sender).  Plain layout: column-major with leading dimension ld, all 64 lanes (two double2).
// BS16: packed upper blocks, lanes a <= b only (one double4), pgbp_bs16.hpp.
// ODD (plain layout only): the quantity really is n x n with n = P - 1 odd; the lane grid is the one of P, index n is
// a phantom (reads 0, is never stored), and the accesses are element-wise because rows no longer pair up on 16 bytes.
typedef double pgbp_d4v __attribute__((ext_vector_type(4)));
// NT: a streaming load (an operand this calibrate reads once: a sepset, a leaf's belief)
template <bool BS, bool ODD = false, bool NT = false>
__device__ __forceinline__ Blk load_blk(const double* __restrict__ base, int ld, int a, int b, bool up, int kidx,
                                        int n = 0) {
  Blk r{0.0, 0.0, 0.0, 0.0};
  if constexpr (BS) {
    if (up) {
      if constexpr (NT) {
        const pgbp_d4v v = __builtin_nontemporal_load(reinterpret_cast<const pgbp_d4v*>(base + kidx));
        r = Blk{v.x, v.y, v.z, v.w};
      } else {
        const double4 v = *reinterpret_cast<const double4*>(base + kidx);
        r = Blk{v.x, v.y, v.z, v.w};
      }
    }
  } else if constexpr (ODD) {
    const int r0 = 2 * a, r1 = 2 * a + 1, c0 = 2 * b, c1 = 2 * b + 1;
    if (r0 < n && c0 < n) r.x = base[r0 + (int64_t)ld * c0];
    if (r1 < n && c0 < n) r.y = base[r1 + (int64_t)ld * c0];
    if (r0 < n && c1 < n) r.z = base[r0 + (int64_t)ld * c1];
    if (r1 < n && c1 < n) r.w = base[r1 + (int64_t)ld * c1];
  } else {
    const double2 c0 = *reinterpret_cast<const double2*>(base + 2 * a + (int64_t)ld * (2 * b));
    const double2 c1 = *reinterpret_cast<const double2*>(base + 2 * a + (int64_t)ld * (2 * b + 1));
    r = Blk{c0.x, c0.y, c1.x, c1.y};
  }
  return r;
}
// NT: a streaming store (the residuals: written by every message, read by nobody before the next reset)
template <bool BS, bool ODD = false, bool NT = false>
__device__ __forceinline__ void store_blk(double* __restrict__ base, int ld, int a, int b, bool up, bool act, int kidx,
                                          const Blk& v, int n = 0) {
  if constexpr (BS) {
    if constexpr (NT) {
      if (up) __builtin_nontemporal_store(pgbp_d4v{v.x, v.y, v.z, v.w}, reinterpret_cast<pgbp_d4v*>(base + kidx));
    } else if (up) *reinterpret_cast<double4*>(base + kidx) = make_double4(v.x, v.y, v.z, v.w);
  } else if constexpr (ODD) {
    if (act) {
      const int r0 = 2 * a, r1 = 2 * a + 1, c0 = 2 * b, c1 = 2 * b + 1;
      if (r0 < n && c0 < n) base[r0 + (int64_t)ld * c0] = v.x;
      if (r1 < n && c0 < n) base[r1 + (int64_t)ld * c0] = v.y;
      if (r0 < n && c1 < n) base[r0 + (int64_t)ld * c1] = v.z;
      if (r1 < n && c1 < n) base[r1 + (int64_t)ld * c1] = v.w;
    }
  } else if (act) {
    *reinterpret_cast<double2*>(base + 2 * a + (int64_t)ld * (2 * b)) = make_double2(v.x, v.y);
    *reinterpret_cast<double2*>(base + 2 * a + (int64_t)ld * (2 * b + 1)) = make_double2(v.z, v.w);
  }
}
// entries 2a, 2a+1 of a vector (h, dh)
template <bool ODD>
__device__ __forceinline__ double2 load_pair(const double* __restrict__ v, int a, int n) {
  if constexpr (ODD) return make_double2(2 * a < n ? v[2 * a] : 0.0, 2 * a + 1 < n ? v[2 * a + 1] : 0.0);
  else return *reinterpret_cast<const double2*>(v + 2 * a);
}
typedef double pgbp_d2v __attribute__((ext_vector_type(2)));
template <bool ODD, bool NT = false>
__device__ __forceinline__ void store_pair(double* __restrict__ v, int a, double x, double y, int n) {
  if constexpr (ODD) {
    if (2 * a < n) v[2 * a] = x;
    if (2 * a + 1 < n) v[2 * a + 1] = y;
  } else if constexpr (NT) {
    __builtin_nontemporal_store(pgbp_d2v{x, y}, reinterpret_cast<pgbp_d2v*>(v + 2 * a));
  } else {
    *reinterpret_cast<double2*>(v + 2 * a) = make_double2(x, y);
  }
}

// One 64-byte record by scalar loads, pinned in SGPRs: left to itself the compiler sinks the field loads into the
// branches that use them, and the wave then pays a dependent memory round trip per field group.
__device__ __forceinline__ FEntry load_record(const FEntry* __restrict__ r) {
  const uint4* __restrict__ rq = reinterpret_cast<const uint4*>(r);
  uint4 q0 = rq[0], q1 = rq[1], q2 = rq[2], q3 = rq[3];
  asm("; record resident"
      : "+s"(q0.x), "+s"(q0.y), "+s"(q0.z), "+s"(q0.w), "+s"(q1.x), "+s"(q1.y), "+s"(q1.z), "+s"(q1.w), "+s"(q2.x),
        "+s"(q2.y), "+s"(q2.z), "+s"(q2.w), "+s"(q3.x), "+s"(q3.y), "+s"(q3.z), "+s"(q3.w));
  const uint4 q[4] = {q0, q1, q2, q3};
  FEntry en;
  __builtin_memcpy(&en, q, sizeof(FEntry));
  return en;
}

// the 32-byte prologue of a record (FPro), likewise
__device__ __forceinline__ FPro load_pro(const FPro* __restrict__ r) {
  const uint4* __restrict__ rq = reinterpret_cast<const uint4*>(r);
  uint4 q0 = rq[0], q1 = rq[1];
  asm("; prologue resident" : "+s"(q0.x), "+s"(q0.y), "+s"(q0.z), "+s"(q0.w), "+s"(q1.x), "+s"(q1.y), "+s"(q1.z), "+s"(q1.w));
  const uint4 q[2] = {q0, q1};
  FPro pr;
  __builtin_memcpy(&pr, q, sizeof(FPro));
  return pr;
}

}  // namespace pgbp
