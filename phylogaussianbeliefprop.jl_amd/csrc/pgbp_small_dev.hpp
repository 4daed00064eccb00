// Device helpers shared by the wave-per-task kernels (pgbp_kernels.hip) and the two-wavefront loop kernel (pgbp_pair.hip):
// the record of a message as a wavefront holds it, the register-resident elimination of a small frame, element accesses
// with 32-bit byte offsets.  Device code only.
#pragma once
#include <hip/hip_runtime.h>

#include "pgbp_internal.hpp"
#include "pgbp_kernels.hpp"

namespace pgbp {

#define PGBP_LOG2PI 1.8378770664093454835606594728112
#define PGBP_EPS 2.220446049250313e-16

// ---- the record of a message (GRec, pgbp_internal.hpp) as a wavefront holds it: one dword per lane (lanes 0 .. 31 =
// the 128 bytes of the record) and this lane's byte of the two inline index maps.  Three vector loads with addresses
// that depend on the record index alone, so the record of the NEXT message (of the task, or of the workgroup's next step
// in the loop mode) is in flight beside the current message at the cost of three registers.
struct GLoad {
  unsigned int rv;
  int pb, ub;
};
__device__ __forceinline__ GLoad load_grec(const GRec* __restrict__ recs, int ri, int lane) {
  // (an opaque copy of the lane id: the three per-lane offsets below are then formed here, two instructions each, instead of
  // being hoisted out of the caller's loops and held in registers -- or in scratch -- across them)
  asm volatile("" : "+v"(lane));
  const unsigned char* b = reinterpret_cast<const unsigned char*>(recs + ri);
  GLoad l;
  l.rv = reinterpret_cast<const unsigned int*>(b)[lane & 31];
  l.pb = b[offsetof(GRec, perm) + (lane < kGInlPerm ? lane : 0)];
  l.ub = b[offsetof(GRec, up) + (lane & (kGInlUp - 1))];
  return l;
}
// A record requested ahead (the next message of the task, the next pass's task) is waited for HERE -- a point where it has
// long arrived and no store of the current message has been issued yet.  Loads and stores share one in-order counter: a wait
// the compiler places for these registers where they are next touched (the copy at a loop's latch, the first readlane of the
// next message) sits behind the current message's stores and is a wait for their acknowledgement by the L2 -- measured on
// bp_chunk_pair: 2 800 clocks between a consumer's last store and the next pass's barrier.
__device__ __forceinline__ void settle(GLoad& l) { asm volatile("" : "+v"(l.rv), "+v"(l.pb), "+v"(l.ub)); }
__device__ __forceinline__ int grec_dw(unsigned int rv, int k) { return __builtin_amdgcn_readlane((int)rv, k); }
__device__ __forceinline__ int64_t grec_i64(unsigned int rv, int k) {
  return (int64_t)(((unsigned long long)(unsigned int)grec_dw(rv, k + 1) << 32) | (unsigned int)grec_dw(rv, k));
}

// ---- SMALL messages in registers: at most kSmallI integrated and kSmallK kept variables (a level-3 network's cluster
// graphs: clusters of up to three nodes, sepsets of one or two; a handful of traits).  The augmented sender sits in a
// fixed 16 x 17 frame, one ROW PER LANE: integrated variable k in lane k / column k, kept variable a in lane 8 + a /
// column 8 + a, h in column 16 (unused rows and columns are zero).  Every operand of the message -- the sender's rows,
// the sepset's and the receiver's entries this lane will update, the failure mark of the sender -- is requested in one
// batch at the top; the elimination is straight-line code (the pivot row travels by a DPP row broadcast, no LDS, no
// synchronisation); divide! and mult! are done by the kept lanes on their own rows.  Arithmetic and its order are those
// of the LDS path below (eliminate_leading): W[i][j] -= (W[i][k] / d_k) * W[k][j], log det as a mantissa product.
// (kSmallI = kSmallK = 8: pgbp_internal.hpp -- the planner tells the launches whose messages all fit)
struct SmallFrame {
  double row[kSmallI + kSmallK + 1];
};
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, l);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// the pivot row of a frame held one row per lane: lane K of each ROW OF 16 LANES to that row's lanes, one instruction
// (v_mov_b64_dpp row_newbcast:K) where two v_readlane serve a single row and go through scalar registers
template <int K>
__device__ __forceinline__ double row_bcast(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true);   // row_newbcast:K
}
// the KI pivots of a frame (columns 0 .. KI - 1 integrated, KI .. KI + KK - 1 kept, KI + KK = h), straight-line: every row of
// 16 lanes that enters eliminates its own frame (bp_level_small4: four tasks; small_message: the task sits in lanes 0 .. 15)
template <int KI, int KK>
struct Small4 {
  // SPREAD (the instances where registers are plentiful: the loop launches): the broadcasts of a pivot row first, each into
  // a register of its own, then the updates -- with ONE temporary the compiler emits  mov, fma, nop, mov, fma, ...  and a
  // wavefront alone on its SIMD pays each pair's latency in turn
  template <int k, class Row, bool SPREAD = false>
  static __device__ __forceinline__ void pivot(Row& row, const int ni, int& info, double& mant, int& expo, double& quad) {
    if (k < ni && info == 0) {
      const double d = row_bcast<k>(row[k]);
      const double hk = row_bcast<k>(row[KI + KK]);
      if (!(d > 0.0)) {
        info = k + 1;
      } else {
        double rd = __builtin_amdgcn_rcp(d);
        rd = fma(fma(-d, rd, 1.0), rd, rd);
        rd = fma(fma(-d, rd, 1.0), rd, rd);
        int ex;
        mant *= frexp(d, &ex);
        expo += ex;
        quad += hk * hk * rd;
        const double f = row[k] * rd;
        if constexpr (SPREAD) {
          double pk[KI + KK + 1];
#pragma unroll
          for (int j = k + 1; j <= KI + KK; ++j) pk[j] = row_bcast<k>(row[j]);
#pragma unroll
          for (int j = k + 1; j <= KI + KK; ++j) asm volatile("" : "+v"(pk[j]));   // (materialised before the first update)
#pragma unroll
          for (int j = k + 1; j <= KI + KK; ++j) row[j] -= f * pk[j];
        } else {
#pragma unroll
        for (int j = k + 1; j <= KI + KK; ++j) {
          const double pkj = row_bcast<k>(row[j]);
          row[j] -= f * pkj;
        }
        }
      }
    }
    if constexpr (k + 1 < KI) pivot<k + 1, Row, SPREAD>(row, ni, info, mant, expo, quad);
  }
};

// element `idx` of a record whose base is wave-uniform (a scalar register pair): the byte offset formed in 32 bits, so that
// the access is  global_load / store  v, v_offset, s[base]  -- one shift in front of it instead of a sign or zero extension
// and a 64-bit shift-add (a wavefront alone on its SIMD pays every dependent step of an address in full)
__device__ __forceinline__ double ld8(const double* __restrict__ base, int idx) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + ((unsigned int)idx << 3));
}
__device__ __forceinline__ void st8(double* __restrict__ base, int idx, double v) {
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + ((unsigned int)idx << 3)) = v;
}
// ... with the byte offset formed beforehand (small_message: in front of the synchronisation, and once for load and store)
__device__ __forceinline__ double ld8o(const double* __restrict__ base, unsigned int off) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ void st8o(double* __restrict__ base, unsigned int off, double v) {
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off) = v;
}

}  // namespace pgbp
