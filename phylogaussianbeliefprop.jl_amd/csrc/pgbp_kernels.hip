// HIP kernels for gfx950 (MI355X, CDNA4; wave64).
//
// bp_level_generic: one wavefront = one task of a BP level = an ordered list of
// canonical-form messages (src/beliefupdates.jl:650-665) that share a target
// (postorder) or a sender (preorder).  For each message:
//   marginalize (src/beliefupdates.jl:55-83)  -> Schur complement by in-LDS elimination
//   divide!     (src/beliefupdates.jl:579-587) -> residual + sepset overwrite
//   mult!       (src/beliefupdates.jl:483-488) -> scatter-add into the receiver
//   iscalibrated_residnorm! (src/beliefs.jl:994-1003)
// Handles any belief dimension <= PGBP_MAX_DIM and arbitrary (ragged) scope index maps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdlib>

#include "pgbp_bs16.hpp"
#include "pgbp_kernels.hpp"
#include "pgbp_small_dev.hpp"

namespace pgbp {

#ifndef PGBP_SMALL_DPP
#define PGBP_SMALL_DPP 1   // small_message: pivot rows by DPP row broadcast (0: by v_readlane, the form before)
#endif
#ifndef PGBP_SMALL_SETTLE
#define PGBP_SMALL_SETTLE 1   // small_message: the operands' arrival is stated once, in front of the stores (0: the form before)
#endif

static constexpr int kPermDoubles = PGBP_MAX_DIM / 2;  // PGBP_MAX_DIM int32 at the front of LDS

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ int pow2_at_least(int n) {  // n in [1, 64] -> smallest power of two >= n
  return n <= 1 ? 1 : 1 << (32 - __clz(n - 1));
}

extern __shared__ double lds[];

#ifdef PGBP_GSTAMP  // experiment builds only (tools/stamp_generic.py): clock stamps of the phases of a message
constexpr int kGStampSlots = 1 << 16, kGStampN = 8;
__device__ unsigned int g_gstamp[kGStampSlots][kGStampN + 4];
__device__ unsigned int g_gstamp_n;
#define PGBP_GST(i) do { gst[i] = (unsigned int)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PGBP_GST(i) do { } while (0)
#endif

size_t generic_lds_bytes(int max_mf);

// lane -> (lane & (L - 1), lane >> lg) grids with L = 2^lg >= n: index arithmetic without integer division
__device__ __forceinline__ int log2_ceil(int n) {  // n in [1, 64]
  return n <= 1 ? 0 : 32 - __clz(n - 1);
}

// Eliminate the leading `ni` variables of the (mf x (mf+1)) augmented system held row-major in W
// (leading dimension ld): W[i][j] -= (W[i][k] / W[k][k]) * W[k][j].  Returns 0 or the 1-based index of
// the first non-positive pivot (LAPACK potrf `info`, src/beliefupdates.jl:68-76). Accumulates
// sum log(d_k) and sum h~_k^2 / d_k.  All 64 lanes take part; results are wave-uniform.
// WAVE: the working matrix belongs to this wavefront alone (several tasks per workgroup): the barrier between pivots is
// the wave-local ordering of LDS accesses; else the workgroup is the one wavefront and __syncthreads() says the same.
__device__ __forceinline__ void wave_local_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool WAVE>
__device__ __forceinline__ void task_sync() {
  if constexpr (WAVE) wave_local_sync();
  else __syncthreads();
}

template <bool WAVE = false>
__device__ __forceinline__ int eliminate_leading(double* W, int ld, int mf, int ni, int lane, double& logdet,
                                                 double& quad, const double2* __restrict__ logtab = nullptr) {
  logdet = 0.0;
  quad = 0.0;
  double mant = 1.0;
  int expo = 0;
  // fixed lane grid over the columns 0 .. mf (the last one is h; 65 columns for mf = 64: two passes) and a stripe of rows
  const int ncol = mf + 1;
  const int lg = log2_ceil(ncol < kWave ? ncol : kWave);
  const int L = 1 << lg;
  const int jj = lane & (L - 1);
  const int i0 = lane >> lg, R = kWave >> lg;   // R >= 1
  for (int k = 0; k < ni; ++k) {
    const double d = W[k * ld + k];
    const double hk = W[k * ld + mf];
    if (!(d > 0.0)) return k + 1;
    double rd = __builtin_amdgcn_rcp(d);  // v_rcp_f64 + 2 Newton steps: ~1 ulp, half the latency of a division
    rd = fma(fma(-d, rd, 1.0), rd, rd);
    rd = fma(fma(-d, rd, 1.0), rd, rd);
    int ex;
    mant *= frexp(d, &ex);  // log det as mantissa product + exponent sum: one log per message
    expo += ex;
    if ((k & 15) == 15) { mant = frexp(mant, &ex); expo += ex; }
    quad += hk * hk * rd;
    for (int j = jj; j < ncol; j += L) {
      if (j > k) {
        const double pkj = W[k * ld + j];
        for (int i = k + 1 + i0; i < mf; i += R) W[i * ld + j] -= (W[i * ld + k] * rd) * pkj;
      }
    }
    task_sync<WAVE>();
  }
  logdet = (logtab ? log_by_table(logtab, mant) : log(mant)) + (double)expo * 0.69314718055994530941723212145818;
  return 0;
}

// returns 0: message applied; 1: the task ends here (the sender is downstream of a failure, or J_I is not positive definite)
// KI, KK: the frame of this instance (KI integrated + KK kept rows / columns + h): 8 + 8 covers every small message; the
// loop launches, where occupancy does not matter, also have 4 + 4, 4 + 8 and 8 + 4 (half the straight-line code of a message
// of a 4-trait network: clusters of one to three nodes)
template <bool WAVE, int KI, int KK, bool DENSE = false>
__device__ __forceinline__ int small_message(const DevState& S, const GRec* __restrict__ recs, const GLoad& cur, const int site,
                                             const int lane, unsigned long long seq_base, double* __restrict__ pool,
                                             double* __restrict__ rpool, SmallFrame& F, double& gmsg_io, const int pend
#ifdef PGBP_GSTAMP
                                             , unsigned int* gst
#endif
) {
  // ---- DECODE: everything that depends on the record alone (static plan data, resident since the previous message) -- the
  // record's words into scalar registers, the three or four pointers, this lane's role, rows and columns from the inline
  // maps.  No access to a belief yet: the synchronisation that makes the previous message's (task: fence) or the previous
  // level's (loop mode: workgroup barrier) stores visible comes BEHIND the decode (`pend`), so these ~ 150 instructions run
  // while those stores are still on their way to the L2 instead of after their acknowledgement.
  const unsigned int rv = cur.rv;
  const int next = grec_dw(rv, 15);
  const int en_msg = grec_dw(rv, 8), en_seq = grec_dw(rv, 9), from_b = grec_dw(rv, 10);
  const int dims = grec_dw(rv, 16), fl = grec_dw(rv, 17);
  const int mf = dims & 255, mt = (dims >> 8) & 255, s = (dims >> 16) & 255, ni = (dims >> 24) & 255;
  const int k0 = (fl & 255) == 255 ? -1 : (fl & 255), u0 = ((fl >> 8) & 255) == 255 ? -1 : ((fl >> 8) & 255);
  const bool en_reuse = ((fl >> 16) & 255) != 0;
  double* __restrict__ sep = pool + grec_i64(rv, 4);
  double* __restrict__ to = pool + grec_i64(rv, 2);
  double* __restrict__ res = rpool + grec_i64(rv, 6);
  const double* __restrict__ from = pool + grec_i64(rv, 0);
  const bool is_int = lane < KI;
  const int fi = KI == 8 ? (lane & 7) : (is_int ? lane : lane - KI);   // row of the message (kept lanes) / pivot index (integrated lanes)
  const bool kept_live = lane >= KI && lane < KI + KK && fi < s;
  const bool row_live = is_int ? fi < ni : kept_live;
  const int up_lane = __shfl(cur.ub, fi);   // (inline map: lane l holds up[l & 15])
  int ua = u0 >= 0 ? u0 + fi : up_lane;
  int ubv[KK], cjv[KI], cbv[KK];
#pragma unroll
  for (int b = 0; b < KK; ++b) {
    const int rl = __builtin_amdgcn_readlane(cur.ub, b);
    ubv[b] = u0 >= 0 ? u0 + b : rl;
  }
  // position of this lane's variable in the sender, and of every column's (wave-uniform)
  const int q = is_int ? fi : ni + fi;                       // place in the order "integrated first, kept last"
  const int pq = __shfl(cur.pb, q < kGInlPerm ? q : 0);      // (inline map: lane l holds perm[l])
  int pi = k0 >= 0 ? (q < ni ? (q < k0 ? q : q + s) : k0 + (q - ni)) : pq;
#pragma unroll
  for (int j = 0; j < KI; ++j) {
    const int rl = __builtin_amdgcn_readlane(cur.pb, j);
    cjv[j] = k0 >= 0 ? (j < k0 ? j : j + s) : rl;
  }
#pragma unroll
  for (int b = 0; b < KK; ++b) {
    const int rl = __builtin_amdgcn_readlane(cur.pb, (ni + b) & 63);
    cbv[b] = k0 >= 0 ? k0 + b : rl;
  }
  // byte offsets of this lane's operands (DENSE: an unused column stands for column 0, see below); the sepset's and the
  // receiver's serve the loads at the top and the stores at the end
  unsigned int osep[KK], oto[KK], oX[KI], oY[KI], oZ[KK];
  const int rowbase = pi * mf;
#pragma unroll
  for (int b = 0; b < KK; ++b) {
    const int bb = (!DENSE || b < s) ? b : 0, ub = (!DENSE || b < s) ? ubv[b] : ubv[0];
    osep[b] = (unsigned int)(fi + bb * s) << 3;
    oto[b] = (unsigned int)(ua + ub * mt) << 3;
    const int cb = (!DENSE || b < s) ? cbv[b] : cjv[0];   // (DENSE, an unused kept column: any column of the sender)
    oZ[b] = (unsigned int)(is_int ? cb + rowbase : pi + cb * mf) << 3;   // J_SI' for a pivot row, J_S for a kept one
  }
#pragma unroll
  for (int j = 0; j < KI; ++j) {
    const int cj = (!DENSE || j < ni) ? cjv[j] : cjv[0];
    oX[j] = (unsigned int)(pi + cj * mf) << 3;      // J[this row, integrated column j]
    oY[j] = (unsigned int)(cj + rowbase) << 3;      // J[integrated row j, this column]: the upper triangle of J_I
  }
  unsigned int oseph = (unsigned int)(s * s + fi) << 3, otoh = (unsigned int)(mt * mt + ua) << 3,
               ofh = (unsigned int)(mf * mf + pi) << 3;
  if (pend != 0) {
    // (the decode is pinned in front of the synchronisation: nothing of it may sink behind the wait)
#pragma unroll
    for (int b = 0; b < KK; ++b) asm volatile("" : "+v"(osep[b]), "+v"(oto[b]), "+v"(oZ[b]));
#pragma unroll
    for (int j = 0; j < KI; ++j) asm volatile("" : "+v"(oX[j]), "+v"(oY[j]));
    asm volatile("" : "+v"(oseph), "+v"(otoh), "+v"(ofh));
    asm volatile("" : "+s"(sep), "+s"(to), "+s"(res), "+s"(from));
    if (pend == 1) __syncthreads();        // loop mode: the previous level of this workgroup's trees
    else __threadfence_block();            // the previous message of this task
  }
  int pz = 0;
  asm volatile("" : "+v"(pz));   // (vector loads of wave-uniform words: see generic_task)
  const int poisoned = S.poison[(int64_t)site * S.n_clusters + from_b + pz];
  // ---- receiver / sepset operands of the kept lanes (row a = fi of the message): requested first
  double psep[KK], pto[KK], pseph = 0.0, ptoh = 0.0, pre_sepg = 0.0, pre_tog = 0.0;
  if constexpr (DENSE) {
    // DENSE (the loop launches, whose frame is the smallest the message fits: few unused columns): the operands of a
    // message are requested in ONE basic block -- every column's load unconditional inside one masked region, an unused
    // column re-reading column 0 and zeroed afterwards by a wave-uniform select.  A lone wavefront on its SIMD (a narrow
    // pass) pays the full pipeline latency of every dependent instruction pair; with a branch per column each load was a
    // serial chain of its own (120 - 190 clocks per load: tools/stamp_generic.py); in one block the chains interleave.
#pragma unroll
    for (int b = 0; b < KK; ++b) {
      psep[b] = 0.0;
      pto[b] = 0.0;
    }
    if (kept_live) {   // (s >= 1 here)
#pragma unroll
      for (int b = 0; b < KK; ++b) {
        psep[b] = ld8o(sep, osep[b]);
        pto[b] = ld8o(to, oto[b]);
      }
      pseph = ld8o(sep, oseph);
      ptoh = ld8o(to, otoh);
    }
#pragma unroll
    for (int b = 0; b < KK; ++b) {
      psep[b] = b < s ? psep[b] : 0.0;
      pto[b] = b < s ? pto[b] : 0.0;
    }
  } else {
#pragma unroll
  for (int b = 0; b < KK; ++b) {
    psep[b] = 0.0;
    pto[b] = 0.0;
    if (b < s && kept_live) {
      psep[b] = ld8o(sep, osep[b]);
      pto[b] = ld8o(to, oto[b]);
    }
  }
  if (kept_live) {
    pseph = ld8o(sep, oseph);
    ptoh = ld8o(to, otoh);
  }
  }
  if (lane == 0) {
    pre_sepg = ld8(sep, s * s + s);
    pre_tog = ld8(to, mt * mt + mt);
  }
  const double thr_h = S.thr[s], thr_J = S.thr[PGBP_MAX_DIM + 1 + s];
  double gmsg = gmsg_io;
  bool fake = false;
  if (!en_reuse) {
    double X[KI], Y[KI], Z[KK], hv = 0.0;
    if constexpr (DENSE) {
#pragma unroll
      for (int j = 0; j < KI; ++j) {
        X[j] = 0.0;
        Y[j] = 0.0;
      }
#pragma unroll
      for (int b = 0; b < KK; ++b) Z[b] = 0.0;
      if (row_live) {
#pragma unroll
        for (int j = 0; j < KI; ++j) X[j] = ld8o(from, oX[j]);
        if (is_int) {
#pragma unroll
          for (int j = 0; j < KI; ++j) Y[j] = ld8o(from, oY[j]);
        }
#pragma unroll
        for (int b = 0; b < KK; ++b) Z[b] = ld8o(from, oZ[b]);
        hv = ld8o(from, ofh);
      }
#pragma unroll
      for (int j = 0; j < KI; ++j) {
        X[j] = j < ni ? X[j] : 0.0;
        Y[j] = j < ni ? Y[j] : 0.0;
      }
#pragma unroll
      for (int b = 0; b < KK; ++b) Z[b] = b < s ? Z[b] : 0.0;
    } else {
#pragma unroll
    for (int j = 0; j < KI; ++j) {
      X[j] = 0.0;
      Y[j] = 0.0;
      if (j < ni && row_live) {
        X[j] = ld8o(from, oX[j]);
        if (is_int) Y[j] = ld8o(from, oY[j]);
      }
    }
#pragma unroll
    for (int b = 0; b < KK; ++b) {
      Z[b] = 0.0;
      if (b < s && row_live) Z[b] = ld8o(from, oZ[b]);
    }
    if (row_live) hv = ld8o(from, ofh);
    }
    gmsg = ld8(from, mf * mf + mf + pz);
    // "fake" message: J_I, h_I, J_SI all ~ 0 (src/beliefupdates.jl:62-66), on the entries as stored
    bool nz = is_int && fabs(hv) > PGBP_EPS;
#pragma unroll
    for (int j = 0; j < KI; ++j) nz |= fabs(X[j]) > PGBP_EPS;
    fake = ni == 0 || !__any(nz);
    // Symmetric(J_I): its upper triangle only (:68)
#pragma unroll
    for (int j = 0; j < KI; ++j) F.row[j] = (is_int && j < fi) ? Y[j] : X[j];
#pragma unroll
    for (int b = 0; b < KK; ++b) F.row[KI + b] = Z[b];
    F.row[KI + KK] = hv;
  }
  PGBP_GST(2);
  if (__builtin_amdgcn_readfirstlane(poisoned)) {
    if (lane == 0) {
      S.poison[(int64_t)site * S.n_clusters + grec_dw(rv, 11)] = 1;
      for (int qn = next; qn >= 0; qn = recs[qn].next) S.poison[(int64_t)site * S.n_clusters + recs[qn].to_b] = 1;
    }
    return 1;
  }
  if (!en_reuse && !fake) {
    double mant = 1.0, quad = 0.0;
    int expo = 0, info = 0;
    if (PGBP_SMALL_DPP) {
      // (the other three rows of the wavefront hold zero frames: their lanes stop at the first pivot, nothing of theirs is used)
      Small4<KI, KK>::template pivot<0, decltype(F.row), DENSE>(F.row, ni, info, mant, expo, quad);
      info = __builtin_amdgcn_readfirstlane(info);
    } else {
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      if (k < ni && info == 0) {
        const double d = readlane_f64(F.row[k], k);
        const double hk = readlane_f64(F.row[KI + KK], k);
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
          const double f = F.row[k] * rd;
#pragma unroll
          for (int j = k + 1; j <= KI + KK; ++j) {
            const double pkj = readlane_f64(F.row[j], k);
            F.row[j] -= f * pkj;   // (rows <= k are dead from here on: no guard)
          }
        }
      }
    }
    }
    if (info != 0) {
      if (lane == 0) {
        S.status[(int64_t)site * S.n_msgs + en_msg] = info;
        S.poison[(int64_t)site * S.n_clusters + grec_dw(rv, 11)] = 1;
        for (int qn = next; qn >= 0; qn = recs[qn].next) S.poison[(int64_t)site * S.n_clusters + recs[qn].to_b] = 1;
        atomicMin(&S.fail[site], ((seq_base + (unsigned long long)(unsigned int)en_seq) << kInfoBits) |
                                     (unsigned long long)(unsigned int)info);
      }
      return 1;
    }
    const double logdet = log_by_table(S.logtab, mant) + (double)expo * 0.69314718055994530941723212145818;
    gmsg += 0.5 * ((double)ni * PGBP_LOG2PI - logdet + quad);  // :81
  }
  gmsg_io = gmsg;
  PGBP_GST(5);
  // Every operand requested at the top arrived during the elimination.  Said ONCE, here: loads and stores share one in-order
  // counter, so a wait the compiler places for one of these registers further down, behind the first stores (the thresholds
  // before the flag, lane 0's two g words), is a wait for every store issued before it -- a round trip to the L2 in the
  // middle of the store sequence and a second one for the flag behind it.
#if PGBP_SMALL_SETTLE
  double thr_hv = thr_h, thr_Jv = thr_J;
#pragma unroll
  for (int b = 0; b < KK; ++b) asm volatile("" : "+v"(psep[b]), "+v"(pto[b]));
  asm volatile("" : "+v"(pseph), "+v"(ptoh), "+v"(pre_sepg), "+v"(pre_tog), "+v"(gmsg), "+v"(thr_hv), "+v"(thr_Jv));
#else
  const double thr_hv = thr_h, thr_Jv = thr_J;
#endif
  // ---- divide! and mult!: kept lane 8 + a owns row a of the message
  double maxJ = 0.0, maxh = 0.0;
  if (kept_live) {
    if constexpr (DENSE) {
      // (one basic block as for the loads: an unused column stores column 0's values to column 0's places once more)
#pragma unroll
      for (int b = 0; b < KK; ++b) {
        const double msg = b < s ? F.row[KI + b] : F.row[KI];
        const double ps = b < s ? psep[b] : psep[0], pt = b < s ? pto[b] : pto[0];
        const double dJ = msg - ps;
        st8o(sep, osep[b], msg);
        st8o(res, osep[b], dJ);
        st8o(to, oto[b], pt + dJ);
        maxJ = (dJ != dJ) ? INFINITY : fmax(maxJ, fabs(dJ));
      }
    } else {
#pragma unroll
    for (int b = 0; b < KK; ++b) {
      if (b < s) {
        const double msg = F.row[KI + b];
        const double dJ = msg - psep[b];
        st8o(sep, osep[b], msg);
        st8o(res, osep[b], dJ);
        st8o(to, oto[b], pto[b] + dJ);
        maxJ = (dJ != dJ) ? INFINITY : fmax(maxJ, fabs(dJ));
      }
    }
    }
    const double msgh = F.row[KI + KK];
    const double dh = msgh - pseph;
    st8o(sep, oseph, msgh);
    st8o(res, oseph, dh);
    st8o(to, otoh, ptoh + dh);
    maxh = (dh != dh) ? INFINITY : fmax(maxh, fabs(dh));
  }
  if (lane == 0) {
    const double dg = gmsg - pre_sepg;
    st8(sep, s * s + s, gmsg);
    st8(to, mt * mt + mt, pre_tog + dg);
    S.status[(int64_t)site * S.n_msgs + en_msg] = 0;
  }
  if (S.update_resnorm) {
    const bool lane_ok = maxh <= thr_hv && maxJ <= thr_Jv;
    const bool ok = __all(lane_ok);
    if (lane == 0) S.flags[(int64_t)site * S.n_msgs + en_msg] = ok ? 1 : 0;
  }
  return 0;
}

// One task (its messages in order) by ONE wavefront, from the task's first record; perm / W: that wavefront's scratch in
// LDS.  WAVE: other wavefronts of the workgroup run other tasks beside it (the loop mode below), so every
// synchronisation in here is wave-local and nothing in here may be a workgroup barrier; a `return` ends the task (not
// the kernel).  Dependent loads of a message: record (fetched ahead) -> operands.
// SMALL_ONLY: every message of the launch fits the register-resident body (the planner's promise: Traversal::level_small,
// Chunk::small_only) -- the in-LDS body is not compiled in, which leaves the wide levels more wavefronts per SIMD.
// SPEC: the loop launches pick the smallest frame a message fits (small_message<WAVE, KI, KK>)
template <bool WAVE, bool SMALL_ONLY = false, bool SPEC = false>
__device__ __forceinline__ void generic_task(const DevState& S, const GRec* __restrict__ recs, GLoad cur, const int site,
                                             const int lane, unsigned long long seq_base, int32_t* perm, double* W,
                                             int pend = 0) {
  // pend: a synchronisation owed before the first access to a belief -- 1: the workgroup barrier between two levels of the
  // loop mode (every wavefront of the workgroup runs it exactly once per pass, here or in the kernel's loop), 2: the fence
  // between two messages of a task.  The register-resident body runs it behind its decode (small_message).
  double* __restrict__ pool = S.pool + (int64_t)site * S.pool_stride;
  double* __restrict__ rpool = S.rpool + (int64_t)site * S.rpool_stride;
  int mf = 0, ni = 0, ld = 1;
  double gmsg = 0.0;
  SmallFrame frame;
  bool small_prev = false;   // the previous message of the task went through small_message (its marginal is in `frame`)
  for (;;) {
#ifdef PGBP_GSTAMP
    unsigned int gst[kGStampN] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PGBP_GST(0);
    const unsigned int rv = cur.rv;
    const int next = grec_dw(rv, 15);
    GLoad nxt = cur;
    if (next >= 0) nxt = load_grec(recs, next, lane);   // beside everything below
    const int en_msg = grec_dw(rv, 8), en_seq = grec_dw(rv, 9), from_b = grec_dw(rv, 10);
    const int dims = grec_dw(rv, 16), fl = grec_dw(rv, 17);
    const int s = (dims >> 16) & 255, mt = (dims >> 8) & 255;
    const int k0 = (fl & 255) == 255 ? -1 : (fl & 255), u0 = ((fl >> 8) & 255) == 255 ? -1 : ((fl >> 8) & 255);
    const bool en_reuse = ((fl >> 16) & 255) != 0;
    const int inl = (fl >> 24) & 255;
    if (SMALL_ONLY || (((dims >> 24) & 255) <= kSmallI && s <= kSmallK && !(en_reuse && !small_prev))) {
      // the register-resident path (small_message); a reused marginal stays in the path that computed it
      small_prev = true;
      const int nim = (dims >> 24) & 255;
      int done;
#ifdef PGBP_GSTAMP
#define PGBP_SMALL(KI_, KK_) small_message<WAVE, KI_, KK_, SPEC>(S, recs, cur, site, lane, seq_base, pool, rpool, frame, gmsg, pend, gst)
#else
#define PGBP_SMALL(KI_, KK_) small_message<WAVE, KI_, KK_, SPEC>(S, recs, cur, site, lane, seq_base, pool, rpool, frame, gmsg, pend)
#endif
      // (a reused marginal has the dimensions of the message that computed it: the same instance, the same frame)
      if (SPEC && nim <= 4 && s <= 4) done = PGBP_SMALL(4, 4);
      else if (SPEC && nim <= 4) done = PGBP_SMALL(4, kSmallK);
      else if (SPEC && s <= 4) done = PGBP_SMALL(kSmallI, 4);
      else done = PGBP_SMALL(kSmallI, kSmallK);
#undef PGBP_SMALL
      if (done) return;
      mf = dims & 255;
      ni = (dims >> 24) & 255;
      PGBP_GST(6);
#ifdef PGBP_GSTAMP
      __builtin_amdgcn_s_waitcnt(0);
      PGBP_GST(7);
      if (lane == 0 && gridDim.x <= 512) {
        const unsigned int slot = atomicAdd(&g_gstamp_n, 1u);
        if (slot < kGStampSlots) {
          for (int i = 0; i < kGStampN; ++i) g_gstamp[slot][i] = gst[i];
          g_gstamp[slot][kGStampN] = blockIdx.x; g_gstamp[slot][kGStampN + 1] = threadIdx.x >> 6;
          g_gstamp[slot][kGStampN + 2] = (unsigned int)(mf | (ni << 8) | (s << 16) | ((WAVE ? 1 : 0) << 24) | (1u << 25));
          g_gstamp[slot][kGStampN + 3] = gridDim.x;
        }
      }
#endif
      if (next < 0) return;
      pend = 2;   // (the fence: behind the next message's decode)
      cur = nxt;
      continue;
    }
    if constexpr (SMALL_ONLY) __builtin_unreachable();
    small_prev = false;
    if (pend == 1) __syncthreads();
    else if (pend == 2) __threadfence_block();
    pend = 0;
    // the sender sits downstream of a failed message?  Requested with the operands (a vector load: in the loop mode the
    // mark may have been stored by another wavefront of this workgroup one level ago, which the scalar cache does not
    // see), looked at before anything of this message is recorded or stored
    int pz = 0;
    asm volatile("" : "+v"(pz));
    const int poisoned = S.poison[(int64_t)site * S.n_clusters + from_b + pz];
    double* __restrict__ sep = pool + grec_i64(rv, 4);
    double* __restrict__ to = pool + grec_i64(rv, 2);
    double* __restrict__ res = rpool + grec_i64(rv, 6);
    const bool upc = u0 >= 0;                         // update indices contiguous: no index map
    const bool upi = (inl & 2) != 0;                  // ... or in the record
    const int32_t* __restrict__ up = S.idx + grec_dw(rv, 13);
    // lane grid of the sepset block: a = row, b = b0, b0 + Rs, ...
    const int lgs = log2_ceil(s > 0 ? s : 1);
    const int a = lane & ((1 << lgs) - 1), b0 = lane >> lgs, Rs = kWave >> lgs;
    int ua = 0;
    if (upc) ua = u0 + a;
    else if (upi) ua = __shfl(cur.ub, a);
    else if (s > 0 && a < s) ua = up[a];
    // sepset and receiver operands of this lane, requested BEFORE the gather / elimination so that their latency
    // runs beside it (small sepsets only: at most 4 (a, b) pairs per lane; larger ones are read after the elimination)
    const bool pre = s > 0 && s <= 16;
    double pre_sep[4] = {0, 0, 0, 0}, pre_to[4] = {0, 0, 0, 0}, pre_seph = 0.0, pre_toh = 0.0;
    int ubq[4] = {0, 0, 0, 0};
    if (pre) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int b = b0 + q * Rs;
        int ub = 0;
        if (upc) ub = u0 + b;
        else if (upi) ub = __shfl(cur.ub, b & (kGInlUp - 1));
        else if (a < s && b < s) ub = up[b];
        ubq[q] = ub;
        if (a < s && b < s) {
          pre_sep[q] = sep[a + (int64_t)b * s];
          pre_to[q] = to[ua + (int64_t)ub * mt];
        }
      }
      if (a < s && b0 == 0) {
        pre_seph = sep[(int64_t)s * s + a];
        pre_toh = to[(int64_t)mt * mt + ua];
      }
    }
    double pre_sepg = 0.0, pre_tog = 0.0;             // the two g words (lane 0)
    if (lane == 0) {
      pre_sepg = sep[(int64_t)s * s + s];
      pre_tog = to[(int64_t)mt * mt + mt];
    }
    if (!en_reuse) {
      const double* __restrict__ from = pool + grec_i64(rv, 0);
      mf = dims & 255;
      ni = (dims >> 24) & 255;
      ld = (mf + 1) | 1;  // odd leading dimension: conflict-free column walks
      const int lgm = log2_ceil(mf > 0 ? mf : 1);
      const int ci = lane & ((1 << lgm) - 1), r0 = lane >> lgm, Rm = kWave >> lgm;
      task_sync<WAVE>();    // W / perm of the previous message no longer needed
      // integrated variables first, kept variables last.  Kept indices contiguous (every single-node sepset): the
      // permutation is arithmetic; else it came with the record, or (large senders) sits in the index pool
      if (k0 >= 0) {
        if (lane < mf) perm[lane] = (lane < ni) ? (lane < k0 ? lane : lane + s) : k0 + (lane - ni);
      } else if (inl & 1) {
        if (lane < mf) perm[lane] = cur.pb;
      } else {
        const int int_map = grec_dw(rv, 14), keep_map = grec_dw(rv, 12);
        for (int i = lane; i < mf; i += kWave) perm[i] = (i < ni) ? S.idx[int_map + i] : S.idx[keep_map + (i - ni)];
      }
      task_sync<WAVE>();
      // gather; h as the extra column
      if (ci < mf) {
        const int pi = perm[ci];
        for (int j = r0; j < mf; j += Rm) W[ci * ld + j] = from[pi + (int64_t)perm[j] * mf];
        if (r0 == 0) W[ci * ld + mf] = from[(int64_t)mf * mf + pi];
      }
      {
        // vector load on purpose: the sender may have been written through the vector cache a moment ago (a fused
        // chain, the previous step of the loop mode); a wave-uniform address would go through the scalar cache
        int z = 0;
        asm volatile("" : "+v"(z));
        gmsg = from[(int64_t)mf * mf + mf + z];
      }
      PGBP_GST(1);
      task_sync<WAVE>();
    }
    PGBP_GST(2);
    if (__builtin_amdgcn_readfirstlane(poisoned)) {
      // nothing downstream of a failed message runs: every receiver the rest of this task would have reached is marked
      if (lane == 0) {
        S.poison[(int64_t)site * S.n_clusters + grec_dw(rv, 11)] = 1;
        for (int q = next; q >= 0; q = recs[q].next) S.poison[(int64_t)site * S.n_clusters + recs[q].to_b] = 1;
      }
      return;
    }
    if (!en_reuse && ni > 0) {
      const int lgm = log2_ceil(mf > 0 ? mf : 1);
      const int ci = lane & ((1 << lgm) - 1), r0 = lane >> lgm, Rm = kWave >> lgm;
      // "fake" message: J_I, h_I, J_SI all ~ 0 (src/beliefupdates.jl:62-66)
      bool nz = false;
      if (ci < mf) {
        for (int j = r0; j < ni; j += Rm) nz |= fabs(W[ci * ld + j]) > PGBP_EPS;
        if (r0 == 0 && ci < ni) nz |= fabs(W[ci * ld + mf]) > PGBP_EPS;
      }
      const bool fake = !__any(nz);
      if (!fake) {
        // Symmetric(J_I): upper triangle only (:68); pivot rows get J_SI' for the kept columns (:77)
        if (ci < mf) {
          const int j = ci;
          for (int i = r0; i < ni; i += Rm) {
            if (j > i) {
              const double v = (j < ni) ? W[i * ld + j] : W[j * ld + i];
              W[i * ld + j] = v;
              if (j < ni) W[j * ld + i] = v;
            }
          }
        }
        task_sync<WAVE>();
        double logdet, quad;
        PGBP_GST(3);
        const int info = eliminate_leading<WAVE>(W, ld, mf, ni, lane, logdet, quad, S.logtab);
        PGBP_GST(4);
        if (info != 0) {
          if (lane == 0) {
            S.status[(int64_t)site * S.n_msgs + en_msg] = info;
            S.poison[(int64_t)site * S.n_clusters + grec_dw(rv, 11)] = 1;
            for (int q = next; q >= 0; q = recs[q].next) S.poison[(int64_t)site * S.n_clusters + recs[q].to_b] = 1;
            atomicMin(&S.fail[site], ((seq_base + (unsigned long long)(unsigned int)en_seq) << kInfoBits) |
                                         (unsigned long long)(unsigned int)info);
          }
          return;  // nothing of this message is applied; later messages of the task do not run
        }
        gmsg += 0.5 * ((double)ni * PGBP_LOG2PI - logdet + quad);  // :81
      }
    }
    // ---- divide! and mult!
    PGBP_GST(5);
    double maxJ = 0.0, maxh = 0.0;
    if (s > 0 && a < s) {
      if (pre) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int b = b0 + q * Rs;
          if (b < s) {
            const double msg = W[(ni + a) * ld + ni + b];
            const int64_t o = a + (int64_t)b * s;
            const double dJ = msg - pre_sep[q];
            sep[o] = msg;
            res[o] = dJ;
            to[ua + (int64_t)ubq[q] * mt] = pre_to[q] + dJ;
            maxJ = (dJ != dJ) ? INFINITY : fmax(maxJ, fabs(dJ));
          }
        }
      } else {
        for (int b = b0; b < s; b += Rs) {
          const double msg = W[(ni + a) * ld + ni + b];
          const int64_t o = a + (int64_t)b * s;
          const double dJ = msg - sep[o];
          sep[o] = msg;
          res[o] = dJ;
          to[ua + (int64_t)(upc ? u0 + b : up[b]) * mt] += dJ;
          maxJ = (dJ != dJ) ? INFINITY : fmax(maxJ, fabs(dJ));
        }
      }
      if (b0 == 0) {
        const double msg = W[(ni + a) * ld + mf];
        const int64_t o = (int64_t)s * s + a;
        const double dh = msg - (pre ? pre_seph : sep[o]);
        sep[o] = msg;
        res[o] = dh;
        to[(int64_t)mt * mt + ua] = (pre ? pre_toh : to[(int64_t)mt * mt + ua]) + dh;
        maxh = (dh != dh) ? INFINITY : fmax(maxh, fabs(dh));
      }
    }
    if (lane == 0) {
      const double dg = gmsg - pre_sepg;
      sep[(int64_t)s * s + s] = gmsg;
      to[(int64_t)mt * mt + mt] = pre_tog + dg;
      S.status[(int64_t)site * S.n_msgs + en_msg] = 0;
    }
    if (S.update_resnorm) {
      // iscalibrated_residnorm!: max|dh|/sqrt(s) <= atol && max|dJ|/s <= atol (src/beliefs.jl:994-1003), as a comparison
      // with the host's thresholds (DevState::thr); every lane tests its own maximum, one ballot instead of two reductions
      const bool lane_ok = maxh <= S.thr[s] && maxJ <= S.thr[PGBP_MAX_DIM + 1 + s];
      const bool ok = __all(lane_ok);
      if (lane == 0) S.flags[(int64_t)site * S.n_msgs + en_msg] = ok ? 1 : 0;
    }
    PGBP_GST(6);
#ifdef PGBP_GSTAMP
    __builtin_amdgcn_s_waitcnt(0);   // stores acknowledged
    PGBP_GST(7);
    if (lane == 0 && gridDim.x <= 512) {   // the narrow launches
      const unsigned int slot = atomicAdd(&g_gstamp_n, 1u);
      if (slot < kGStampSlots) {
        for (int i = 0; i < kGStampN; ++i) g_gstamp[slot][i] = gst[i];
        g_gstamp[slot][kGStampN] = blockIdx.x; g_gstamp[slot][kGStampN + 1] = threadIdx.x >> 6;
        g_gstamp[slot][kGStampN + 2] = (unsigned int)(mf | (ni << 8) | (s << 16) | ((WAVE ? 1 : 0) << 24));
        g_gstamp[slot][kGStampN + 3] = gridDim.x;
      }
    }
#endif
    if (next < 0) return;
    pend = 2;  // the next message of the task may read or read-modify-write what this one wrote
    cur = nxt;
  }
}

template <bool SMALL_ONLY>
__global__ __launch_bounds__(64) void bp_level_generic(DevState S, const GRec* __restrict__ recs, int rec0,
                                                       unsigned long long seq_base,
                                                       unsigned long long stop_below) {
  const int site = blockIdx.y;
  // A message of an EARLIER traversal failed: the reference has stopped (src/calibration.jl:82,129-132).
  // Failures inside the current traversal only stop what is downstream of them (poison), so that the
  // minimum fail key is the first failure of the reference's sequential order.
  if ((S.fail[site] >> kInfoBits) < stop_below) return;
  // (SMALL_ONLY launches are the narrow ones -- from kSmall4MinTasks tasks on a level runs on bp_level_small4 --: occupancy does
  // not matter, so they take the size-specialised, one-basic-block instances of the loop launches: SPEC)
  generic_task<false, SMALL_ONLY, SMALL_ONLY>(S, recs, load_grec(recs, rec0 + blockIdx.x, threadIdx.x), site, threadIdx.x, seq_base,
                      reinterpret_cast<int32_t*>(lds), lds + kPermDoubles);
}

// ---------------------------------------------------------------------------------------------------------
// FOUR small tasks per wavefront (the wide levels of a network's cluster graphs: cfg5).  small_message keeps one row of the
// 16 x 17 frame per lane, i.e. 16 of a wavefront's 64 lanes work and a wide level is bound by the instruction issue of its
// wavefronts.  Here every ROW OF 16 LANES (a DPP row) runs its own task: integrated variable k in lane k of the
// row, kept variable a in lane 8 + a.  What small_message holds wave-uniform (the record's words, pointers, dimensions, the
// pivot's status) is per row here, in vector registers; the record's words and inline maps are read by each lane from the
// record itself (one 128-byte line); the pivot row travels by v_mov_b64_dpp row_newbcast:k (lane k of EACH row to its 16
// lanes: one instruction where v_readlane needs two and serves one row); `any` / `all` over a row come from one ballot.
// A row whose task ends (last message, poisoned sender, failed pivot) leaves the loop; the wavefront ends with its longest
// task.  Arithmetic, its order and every store are those of small_message<false, 8, 8> (launch modes fuzz: bit-identical).
__device__ __forceinline__ bool row_any(bool p, int lane) {
  return ((__ballot(p) >> (lane & 48)) & 0xffffull) != 0;   // (inside a branch that whole rows take or skip)
}
__device__ __forceinline__ int byte_of(unsigned long long w, int k) { return (int)((w >> (8 * k)) & 255ull); }
__device__ __forceinline__ double log_by_table_lane(const double2* __restrict__ tab, double x) {   // log_by_table, index per lane
  int e;
  const double m = frexp(x, &e);
  const double2 t = tab[(__double2hiint(m) >> 13) & 127];
  const double r = fma(m, t.x, -1.0);
  double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
  p = fma(p, r, 0.2);
  p = fma(p, r, -0.25);
  p = fma(p, r, 1.0 / 3.0);
  p = fma(p, r, -0.5);
  return fma((double)e, 0.69314718055994530941723212145818, t.y) + fma(p * r, r, r);
}

// ROWS (postorder levels whose tasks have at most four messages: Traversal::rowmap): one MESSAGE per row instead of one
// task -- the k messages into one receiver sit in k consecutive rows of one wavefront (rowmap: record, position in the
// task, k), their marginals are computed side by side, and only mult! into the receiver goes in the task's order: position
// c of every task loads, adds and stores its entries of the receiver in turn c (a wave-wide fence in between), so the
// receiver's sums are those of the sequential task, bit for bit.  A message that fails (or whose sender is poisoned) ends
// its task as in the chain: the rows behind it in the task store nothing and report nothing.
template <bool ROWS>
__global__ __launch_bounds__(64) void bp_level_small4(DevState S, const GRec* __restrict__ recs, int rec0, int ntasks,
                                                      const int32_t* __restrict__ rowmap, unsigned long long seq_base,
                                                      unsigned long long stop_below) {
  constexpr int KI = kSmallI, KK = kSmallK;
  static_assert(KI == 8 && KK == 8, "a row of 16 lanes = 8 integrated + 8 kept variables");
  const int site = blockIdx.y;
  if ((S.fail[site] >> kInfoBits) < stop_below) return;
  const int lane = threadIdx.x, fi = lane & 7, wrow = lane >> 4;
  const bool is_int = (lane & 8) == 0, first = (lane & 15) == 0;
  const int task = blockIdx.x * 4 + wrow;   // ROWS: the row of the level
  double* __restrict__ pool = S.pool + (int64_t)site * S.pool_stride;
  double* __restrict__ rpool = S.rpool + (int64_t)site * S.rpool_stride;
  int32_t* __restrict__ poison = S.poison + (int64_t)site * S.n_clusters;
  double row[KI + KK + 1];
#pragma unroll
  for (int j = 0; j <= KI + KK; ++j) row[j] = 0.0;
  double gmsg = 0.0;
  int ri = rec0 + task, pos = 0, klen = 1;
  bool alive = task < ntasks;
  if constexpr (ROWS) {
    const int2 rm = reinterpret_cast<const int2*>(rowmap)[task];   // (ntasks = rows of the level, a multiple of four)
    ri = rm.x;
    alive = rm.x >= 0;
    pos = rm.y & 255;
    klen = (rm.y >> 8) & 255;
  }
  // ROWS: what the ordered turns at the end need of a row's message
  bool live = false;
  int r_mt = 0, r_s = 0, r_u0 = -1, r_ua = 0;
  unsigned long long r_uw = 0;
  double* r_to = pool;
  double r_psep[KK], r_pto[KK], r_pseph = 0.0, r_ptoh = 0.0, r_dg = 0.0, r_pre_tog = 0.0;
#pragma unroll
  for (int b = 0; b < KK; ++b) r_psep[b] = r_pto[b] = 0.0;
  bool r_kept_live = false;
  while (alive) {
    // ---- the record: 128 bytes, every lane of the row reads the words it needs (same line, same addresses over the row)
    const unsigned char* __restrict__ rb = reinterpret_cast<const unsigned char*>(recs + ri);
    const longlong2 o01 = *reinterpret_cast<const longlong2*>(rb);        // from_off, to_off
    const longlong2 o23 = *reinterpret_cast<const longlong2*>(rb + 16);   // sep_off, res_off
    const int4 w8 = *reinterpret_cast<const int4*>(rb + 32);              // msg, seq, from_b, to_b
    const int next = *reinterpret_cast<const int*>(rb + 60);
    const uint2 df = *reinterpret_cast<const uint2*>(rb + 64);            // dims, flags
    // perm[0 .. 15]: the field sits at byte 72 of the record -- 8-byte aligned, not 16: two 8-byte loads, not one 16-byte load
    // through a pointer that promises an alignment the address does not have
    ulonglong2 pw;
    pw.x = *reinterpret_cast<const unsigned long long*>(rb + offsetof(GRec, perm));
    pw.y = *reinterpret_cast<const unsigned long long*>(rb + offsetof(GRec, perm) + 8);
    const unsigned long long uw = *reinterpret_cast<const unsigned long long*>(rb + offsetof(GRec, up));   // up[0 .. 7]
    int touch = 0;   // the next record of the task: its line requested now, read at the top of the next turn
    if (!ROWS && next >= 0) touch = *reinterpret_cast<const int*>(recs + next);
    const int en_msg = w8.x, en_seq = w8.y, from_b = w8.z, to_b = w8.w;
    const int mf = df.x & 255, mt = (df.x >> 8) & 255, s = (df.x >> 16) & 255, ni = (df.x >> 24) & 255;
    const int k0 = (df.y & 255) == 255 ? -1 : (int)(df.y & 255), u0 = ((df.y >> 8) & 255) == 255 ? -1 : (int)((df.y >> 8) & 255);
    const bool en_reuse = ((df.y >> 16) & 255) != 0;
    const int poisoned = poison[from_b];
    double* __restrict__ sep = pool + o23.x;
    double* __restrict__ to = pool + o01.y;
    double* __restrict__ res = rpool + o23.y;
    const bool kept_live = !is_int && fi < s;
    const bool row_live = is_int ? fi < ni : kept_live;
    // ---- receiver / sepset operands of the kept lanes (row a = fi of the message); ROWS: the receiver's only in the
    // task's first row, the others read it in their turn
    const bool to_now = !ROWS || pos == 0;
    const int ua = u0 >= 0 ? u0 + fi : byte_of(uw, fi);
    double psep[KK], pto[KK], pseph = 0.0, ptoh = 0.0, pre_sepg = 0.0, pre_tog = 0.0;
#pragma unroll
    for (int b = 0; b < KK; ++b) {
      psep[b] = 0.0;
      pto[b] = 0.0;
      const int ub = u0 >= 0 ? u0 + b : byte_of(uw, b);
      if (b < s && kept_live) {
        psep[b] = sep[(unsigned)(fi + b * s)];
        if (to_now) pto[b] = to[(unsigned)(ua + ub * mt)];
      }
    }
    if (kept_live) {
      pseph = sep[(unsigned)(s * s + fi)];
      if (to_now) ptoh = to[(unsigned)(mt * mt + ua)];
    }
    if (first) {
      pre_sepg = sep[(unsigned)(s * s + s)];
      if (to_now) pre_tog = to[(unsigned)(mt * mt + mt)];
    }
    const double thr_h = S.thr[s], thr_J = S.thr[PGBP_MAX_DIM + 1 + s];
    bool fake = false;
    if (!en_reuse) {
      const double* __restrict__ from = pool + o01.x;
      const int q = is_int ? fi : ni + fi;                       // place in the order "integrated first, kept last"
      const int pq = q < 8 ? byte_of(pw.x, q) : byte_of(pw.y, q & 7);
      const int pi = k0 >= 0 ? (q < ni ? (q < k0 ? q : q + s) : k0 + (q - ni)) : pq;
      double X[KI], Y[KI], Z[KK], hv = 0.0;
#pragma unroll
      for (int j = 0; j < KI; ++j) {
        X[j] = 0.0;
        Y[j] = 0.0;
        const int cj = k0 >= 0 ? (j < k0 ? j : j + s) : byte_of(pw.x, j);
        if (j < ni && row_live) {
          X[j] = from[(unsigned)(pi + cj * mf)];
          if (is_int) Y[j] = from[(unsigned)(cj + pi * mf)];
        }
      }
#pragma unroll
      for (int b = 0; b < KK; ++b) {
        Z[b] = 0.0;
        const int t = ni + b;                                    // (<= 15: ni <= 8)
        const int cb = k0 >= 0 ? k0 + b : (t < 8 ? byte_of(pw.x, t) : byte_of(pw.y, t & 7));
        if (b < s && row_live) Z[b] = is_int ? from[(unsigned)(cb + pi * mf)] : from[(unsigned)(pi + cb * mf)];
      }
      if (row_live) hv = from[(unsigned)(mf * mf + pi)];
      gmsg = from[(unsigned)(mf * mf + mf)];
      bool nz = is_int && fabs(hv) > PGBP_EPS;
#pragma unroll
      for (int j = 0; j < KI; ++j) nz |= fabs(X[j]) > PGBP_EPS;
      fake = ni == 0 || !row_any(nz, lane);
#pragma unroll
      for (int j = 0; j < KI; ++j) row[j] = (is_int && j < fi) ? Y[j] : X[j];
#pragma unroll
      for (int b = 0; b < KK; ++b) row[KI + b] = Z[b];
      row[KI + KK] = hv;
    }
    int info = 0;
    if (!poisoned && !en_reuse && !fake) {
      double mant = 1.0, quad = 0.0;
      int expo = 0;
      Small4<KI, KK>::template pivot<0, decltype(row), (KI + KK <= 8)>(row, ni, info, mant, expo, quad);
      if (info == 0) {
        const double logdet = log_by_table_lane(S.logtab, mant) + (double)expo * 0.69314718055994530941723212145818;
        gmsg += 0.5 * ((double)ni * PGBP_LOG2PI - logdet + quad);
      }
    }
    const bool stops = poisoned || info != 0;   // the task ends at this message
    bool report = stops;
    if constexpr (ROWS) {
      // has a message in front of this one in its task stopped?  (rows wrow - pos .. wrow - 1 of this wavefront)
      const unsigned long long sm = __ballot(first && stops);
      const unsigned int four = (unsigned int)((sm & 1) | ((sm >> 15) & 2) | ((sm >> 30) & 4) | ((sm >> 45) & 8));
      const unsigned int before = (four >> (wrow - pos)) & ((1u << pos) - 1u);
      report = stops && before == 0;
      live = !stops && before == 0;
    }
    if (report && first) {
      if (info != 0) S.status[(int64_t)site * S.n_msgs + en_msg] = info;
      poison[to_b] = 1;
      if (!ROWS)
        for (int qn = next; qn >= 0; qn = recs[qn].next) poison[recs[qn].to_b] = 1;
      if (info != 0)
        atomicMin(&S.fail[site], ((seq_base + (unsigned long long)(unsigned int)en_seq) << kInfoBits) |
                                     (unsigned long long)(unsigned int)info);
    }
    if (ROWS ? !live : stops) break;
    // ---- divide! and mult!: kept lane 8 + a of the row owns row a of the message
    double maxJ = 0.0, maxh = 0.0;
    if (kept_live) {
#pragma unroll
      for (int b = 0; b < KK; ++b) {
        if (b < s) {
          const int ub = u0 >= 0 ? u0 + b : byte_of(uw, b);
          const double msg = row[KI + b];
          const double dJ = msg - psep[b];
          sep[(unsigned)(fi + b * s)] = msg;
          res[(unsigned)(fi + b * s)] = dJ;
          if (!ROWS) to[(unsigned)(ua + ub * mt)] = pto[b] + dJ;
          maxJ = (dJ != dJ) ? INFINITY : fmax(maxJ, fabs(dJ));
        }
      }
      const double msgh = row[KI + KK];
      const double dh = msgh - pseph;
      sep[(unsigned)(s * s + fi)] = msgh;
      res[(unsigned)(s * s + fi)] = dh;
      if (!ROWS) to[(unsigned)(mt * mt + ua)] = ptoh + dh;
      maxh = (dh != dh) ? INFINITY : fmax(maxh, fabs(dh));
    }
    if (first) {
      const double dg = gmsg - pre_sepg;
      sep[(unsigned)(s * s + s)] = gmsg;
      if (!ROWS) to[(unsigned)(mt * mt + mt)] = pre_tog + dg;
      r_dg = dg;
      S.status[(int64_t)site * S.n_msgs + en_msg] = 0;
    }
    if (S.update_resnorm) {
      const bool bad = !(maxh <= thr_h && maxJ <= thr_J);
      const bool any_bad = row_any(bad, lane);
      if (first) S.flags[(int64_t)site * S.n_msgs + en_msg] = any_bad ? 0 : 1;
    }
    if constexpr (ROWS) {
      r_mt = mt; r_s = s; r_u0 = u0; r_ua = ua; r_uw = uw; r_to = to; r_kept_live = kept_live;
#pragma unroll
      for (int b = 0; b < KK; ++b) { r_psep[b] = psep[b]; r_pto[b] = pto[b]; }
      r_pseph = pseph; r_ptoh = ptoh; r_pre_tog = pre_tog;
      break;
    }
    asm volatile("" ::"v"(touch));
    if (next < 0) break;
    __threadfence_block();   // the next message of the task may read or read-modify-write what this one wrote
    ri = next;
  }
  if constexpr (ROWS) {
    // ---- mult! in the order of each task: turn c = the messages at position c
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      if (!__any(live && klen > c)) break;
      if (live && pos == c) {
        if (r_kept_live) {
#pragma unroll
          for (int b = 0; b < KK; ++b) {
            if (b < r_s) {
              const int ub = r_u0 >= 0 ? r_u0 + b : byte_of(r_uw, b);
              const double dJ = row[KI + b] - r_psep[b];
              const double t0 = c == 0 ? r_pto[b] : r_to[(unsigned)(r_ua + ub * r_mt)];
              r_to[(unsigned)(r_ua + ub * r_mt)] = t0 + dJ;
            }
          }
          const double dh = row[KI + KK] - r_pseph;
          const double t0 = c == 0 ? r_ptoh : r_to[(unsigned)(r_mt * r_mt + r_ua)];
          r_to[(unsigned)(r_mt * r_mt + r_ua)] = t0 + dh;
        }
        if (first) {
          const double t0 = c == 0 ? r_pre_tog : r_to[(unsigned)(r_mt * r_mt + r_mt)];
          r_to[(unsigned)(r_mt * r_mt + r_mt)] = t0 + r_dg;
        }
      }
      __threadfence_block();   // turn c + 1 reads what turn c stored (same wavefront, same vector L1)
    }
  }
}

// LOOP MODE of the same task body: a chunk of fused narrow levels (pgbp_plan.cpp: build_chunks) of generic-class tasks.
// Workgroup b = one dependency-closed tree of tasks; it walks its groups [wg_off[b], wg_off[b + 1]) of up to
// kTailWaves first records of tasks (one wavefront per task, -1: none) with a workgroup barrier in between: the stores
// of a level are complete (vmcnt) and visible (same CU, same vector L1) before the next level's loads.  No workgroup of
// a launch depends on another.  per_wave: doubles of LDS scratch per wavefront (perm + the largest working matrix of
// the chunk).  The record of a wavefront's next task is fetched while it works on the current one.
template <bool SMALL_ONLY>
__global__ __launch_bounds__(kTailWaves * 64) void bp_chunk_generic(DevState S, const GRec* __restrict__ recs,
                                                                    const int32_t* __restrict__ grp_recs,
                                                                    const int32_t* __restrict__ wg_off, int per_wave,
                                                                    unsigned long long seq_base,
                                                                    unsigned long long stop_below) {
  const int site = blockIdx.y;
  if ((S.fail[site] >> kInfoBits) < stop_below) return;   // (uniform over the workgroup: no barrier is skipped by a few)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* scratch = lds + (size_t)wave * per_wave;
  const int g0 = wg_off[blockIdx.x], g1 = wg_off[blockIdx.x + 1];
  int ri = grp_recs[(int64_t)g0 * kTailWaves + wave];
  GLoad cur = {0u, 0, 0};
  if (ri >= 0) cur = load_grec(recs, ri, lane);
  for (int g = g0; g < g1; ++g) {
    int ri_next = -1;
    GLoad nxt = {0u, 0, 0};
    if (g + 1 < g1) {
      // (slot i of pass k runs on wavefront (i + 3 k) mod 8, kPassRotate: where a level has a task or two -- the top of every tree --,
      // consecutive passes fall to different wavefronts, and the one whose turn is next has its record decoded and waits
      // at the barrier while the current one still works: small_message, `pend`)
      ri_next = grp_recs[(int64_t)(g + 1) * kTailWaves + ((wave - kPassRotate * (g + 1 - g0)) & (kTailWaves - 1))];
      if (ri_next >= 0) nxt = load_grec(recs, ri_next, lane);
    }
    // the barrier between pass g - 1 and pass g: a wavefront with a task runs it behind the decode of the task's first
    // message (generic_task: pend), one without runs it here -- once per pass and wavefront either way
    if (ri >= 0)
      generic_task<true, SMALL_ONLY, true>(S, recs, cur, site, lane, seq_base, reinterpret_cast<int32_t*>(scratch), scratch + kPermDoubles,
                                           g > g0 ? 1 : 0);
    else if (g > g0)
      __syncthreads();
    ri = ri_next;
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------------
// LARGE beliefs (a sender, receiver or sepset of dimension 65 .. PGBP_MAX_DIM): one workgroup of 256 threads per task,
// the augmented sender [J | h] (up to 128 x 129 doubles = 132 KB of the CU's 160 KB of LDS) eliminated in place, flat index
// loops instead of the power-of-two lane grids of bp_level_generic.  The clique trees of real networks have a few such
// cliques (the 54-node clique of docs/src/man/clustergraphs.md:40-89); everything about the message itself is as in
// bp_level_generic: both early exits of marginalize, Symmetric(J_I) = upper triangle, potrf-style info, poison marks,
// first-failure key, residual-norm flag (src/beliefupdates.jl:55-83, 579-587, 483-488; src/beliefs.jl:994-1003).
// WS: senders of more than kLdsMaxDim variables (up to PGBP_MAX_DIM): the working matrix does not fit the CU's LDS and lives
// in a WORKSPACE in global memory, one slab per workgroup (engine: pgbp_engine::d_ws) -- a few hundred KB that stay in the
// XCD's L2 between the rank-1 updates; the barriers of the elimination order its accesses (same CU, same vector L1).
constexpr int kBigThreads = 1024;   // sixteen wavefronts per task (256 threads until round 4, last session: Mueller clique tree)
template <bool WS>
__global__ __launch_bounds__(kBigThreads) void bp_level_big(DevState S, const int32_t* __restrict__ task_off,
                                                            const Entry* __restrict__ entries, int task0,
                                                            unsigned long long seq_base, unsigned long long stop_below,
                                                            double* __restrict__ ws, int64_t ws_stride) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int site = blockIdx.y;
  if ((S.fail[site] >> kInfoBits) < stop_below) return;
  const int task = task0 + blockIdx.x;
  double* __restrict__ pool = S.pool + (int64_t)site * S.pool_stride;
  double* __restrict__ rpool = S.rpool + (int64_t)site * S.rpool_stride;
  int32_t* perm = reinterpret_cast<int32_t*>(lds);
  double* W = WS ? ws + ((int64_t)blockIdx.x + (int64_t)gridDim.x * blockIdx.y) * ws_stride : lds + kPermDoubles;
  __shared__ int s_info;
  __shared__ double s_red[kBigThreads];
  __shared__ int s_flag;
  const int e0 = task_off[task], e1 = task_off[task + 1];
  int mf = 0, ni = 0, ld = 1;
  double gmsg = 0.0;
  for (int e = e0; e < e1; ++e) {
    const Entry en = entries[e];
    const MsgDesc m = S.msgs[en.msg];
    if (S.poison[(int64_t)site * S.n_clusters + m.from_b]) {
      if (tid == 0)
        for (int e2 = e; e2 < e1; ++e2) S.poison[(int64_t)site * S.n_clusters + S.msgs[entries[e2].msg].to_b] = 1;
      return;
    }
    const int s = m.s, mt = m.mt;
    double* __restrict__ sep = pool + m.sep_off;
    double* __restrict__ to = pool + m.to_off;
    double* __restrict__ res = rpool + m.res_off;
    const int32_t* __restrict__ up = S.idx + m.up_map;
    if (!en.reuse) {
      const double* __restrict__ from = pool + m.from_off;
      mf = m.mf;
      ni = m.ni;
      ld = (mf + 1) | 1;
      __syncthreads();  // W / perm of the previous entry no longer needed
      for (int i = tid; i < mf; i += nthr)
        perm[i] = (i < ni) ? S.idx[m.int_map + i] : S.idx[m.keep_map + (i - ni)];  // integrated first, kept last
      __syncthreads();
      for (int idx = tid; idx < mf * mf; idx += nthr) {
        const int j = idx / mf, i = idx - j * mf;
        W[i * ld + j] = from[perm[i] + (int64_t)perm[j] * mf];
      }
      for (int i = tid; i < mf; i += nthr) W[i * ld + mf] = from[(int64_t)mf * mf + perm[i]];
      {
        int z = 0;
        asm volatile("" : "+v"(z));   // vector load: the sender may have been written by this workgroup (an earlier entry)
        gmsg = from[(int64_t)mf * mf + mf + z];
      }
      if (tid == 0) { s_info = 0; s_flag = 0; }
      __syncthreads();
      if (ni > 0) {
        // "fake" message: J_I, h_I, J_SI all ~ 0 (src/beliefupdates.jl:62-66): rows 0 .. mf-1, columns 0 .. ni-1, and h_I
        bool nz = false;
        for (int idx = tid; idx < mf * ni; idx += nthr) {
          const int j = idx / mf, i = idx - j * mf;
          nz |= fabs(W[i * ld + j]) > PGBP_EPS;
        }
        for (int i = tid; i < ni; i += nthr) nz |= fabs(W[i * ld + mf]) > PGBP_EPS;
        if (nz) s_flag = 1;
        __syncthreads();
        if (s_flag) {
          // Symmetric(J_I): upper triangle only (:68); pivot rows get J_SI' for the kept columns (:77)
          for (int idx = tid; idx < ni * mf; idx += nthr) {
            const int i = idx / mf, j = idx - i * mf;   // i < ni
            if (j > i && j >= ni) W[i * ld + j] = W[j * ld + i];
          }
          __syncthreads();
          for (int idx = tid; idx < ni * ni; idx += nthr) {
            const int i = idx / ni, j = idx - i * ni;
            if (j > i) W[j * ld + i] = W[i * ld + j];
          }
          __syncthreads();
          // right-looking elimination of the ni leading variables on [J | h]
          double mant = 1.0, quad = 0.0;
          int expo = 0;
          for (int k = 0; k < ni; ++k) {
            const double d = W[k * ld + k];
            if (!(d > 0.0)) {
              if (tid == 0) s_info = k + 1;
              break;
            }
            const double rd = 1.0 / d;
            const double hk = W[k * ld + mf];
            int ex;
            mant *= frexp(d, &ex);
            expo += ex;
            if ((k & 15) == 15) { mant = frexp(mant, &ex); expo += ex; }
            quad += hk * hk * rd;
            // rows k+1 .. mf-1 by wavefront, columns k+1 .. mf (h) by lane (every entry: the same operations as ever)
            if constexpr (!WS) {   // (in LDS)
              for (int i = k + 1 + (tid >> 6); i < mf; i += (nthr >> 6)) {
                const double lik = W[i * ld + k] * rd;
                for (int j = k + 1 + (tid & 63); j <= mf; j += kWave) W[i * ld + j] -= lik * W[k * ld + j];
              }
            } else {
              // the pivot row's entries of the lane's columns once per pivot, a row's entries requested together: in the
              // workspace every access is a round trip to the L2, and the compiler must assume W aliases itself
              constexpr int kChunks = (PGBP_MAX_DIM + 1 + kWave - 1) / kWave;
              const int j0 = k + 1 + (tid & 63);
              double rk[kChunks];
#pragma unroll
              for (int c = 0; c < kChunks; ++c) rk[c] = (j0 + c * kWave <= mf) ? W[k * ld + j0 + c * kWave] : 0.0;
              for (int i = k + 1 + (tid >> 6); i < mf; i += (nthr >> 6)) {
                const double lik = W[i * ld + k] * rd;
                double wv[kChunks];
#pragma unroll
                for (int c = 0; c < kChunks; ++c) wv[c] = (j0 + c * kWave <= mf) ? W[i * ld + j0 + c * kWave] : 0.0;
#pragma unroll
                for (int c = 0; c < kChunks; ++c)
                  if (j0 + c * kWave <= mf) W[i * ld + j0 + c * kWave] = wv[c] - lik * rk[c];
              }
            }
            __syncthreads();
          }
          __syncthreads();
          if (s_info != 0) {
            if (tid == 0) {
              S.status[(int64_t)site * S.n_msgs + en.msg] = s_info;
              for (int e2 = e; e2 < e1; ++e2) S.poison[(int64_t)site * S.n_clusters + S.msgs[entries[e2].msg].to_b] = 1;
              atomicMin(&S.fail[site], ((seq_base + (unsigned long long)en.seq) << kInfoBits) | (unsigned long long)s_info);
            }
            return;
          }
          const double logdet = log(mant) + (double)expo * 0.69314718055994530941723212145818;
          gmsg += 0.5 * ((double)ni * PGBP_LOG2PI - logdet + quad);  // :81
        }
      }
    }
    __syncthreads();
    // ---- divide! and mult!
    double maxJ = 0.0, maxh = 0.0;
    for (int idx = tid; idx < s * s; idx += nthr) {
      const int b = idx / s, a = idx - b * s;
      const double msg = W[(ni + a) * ld + ni + b];
      const int64_t o = a + (int64_t)b * s;
      const double dJ = msg - sep[o];
      sep[o] = msg;
      res[o] = dJ;
      to[up[a] + (int64_t)up[b] * mt] += dJ;
      maxJ = (dJ != dJ) ? INFINITY : fmax(maxJ, fabs(dJ));
    }
    for (int a = tid; a < s; a += nthr) {
      const double msg = W[(ni + a) * ld + mf];
      const int64_t o = (int64_t)s * s + a;
      const double dh = msg - sep[o];
      sep[o] = msg;
      res[o] = dh;
      to[(int64_t)mt * mt + up[a]] += dh;
      maxh = (dh != dh) ? INFINITY : fmax(maxh, fabs(dh));
    }
    if (tid == 0) {
      const double dg = gmsg - sep[(int64_t)s * s + s];
      sep[(int64_t)s * s + s] = gmsg;
      to[(int64_t)mt * mt + mt] += dg;
      S.status[(int64_t)site * S.n_msgs + en.msg] = 0;
    }
    if (S.update_resnorm) {
      // iscalibrated_residnorm! (src/beliefs.jl:994-1003): x -> fl(x / c) is monotone, every thread tests its own maximum
      const bool ok = maxh <= S.thr[s] && maxJ <= S.thr[PGBP_MAX_DIM + 1 + s];
      s_red[tid] = ok ? 1.0 : 0.0;
      __syncthreads();
      if (tid == 0) {
        bool all = true;
        for (int t = 0; t < nthr; ++t) all &= s_red[t] != 0.0;
        S.flags[(int64_t)site * S.n_msgs + en.msg] = all ? 1 : 0;
      }
    }
    __syncthreads();  // the next entry of the task may read or read-modify-write what this one wrote
  }
}

static void allow_large_lds(const void* kernel, size_t bytes) {
  // more than 64 KB of dynamic LDS is asked for explicitly (what the launch needs, not the device maximum: a kernel
  // with static LDS of its own would be refused the full 160 KB, and a refused attribute call is a sticky HIP error)
  // (only the call's own status is looked at: the thread's last-error word may hold an earlier launch's failure, which
  // the entry point still has to report)
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int64_t big_ws_doubles(int max_mf) { return max_mf > kLdsMaxDim ? (int64_t)max_mf * ((max_mf + 1) | 1) : 0; }

void launch_level_big(const DevState& S, const int32_t* d_task_off, const Entry* d_entries, int task0, int ntasks,
                      int n_sites, unsigned long long seq_base, unsigned long long stop_below, int max_mf, double* d_ws,
                      hipStream_t st) {
  if (ntasks <= 0) return;
  if (max_mf > kLdsMaxDim) {   // the working matrix in the workspace (ntasks * n_sites slabs of big_ws_doubles(max_mf))
    hipLaunchKernelGGL(bp_level_big<true>, dim3(ntasks, n_sites), dim3(kBigThreads), sizeof(double) * kPermDoubles, st, S,
                       d_task_off, d_entries, task0, seq_base, stop_below, d_ws, big_ws_doubles(max_mf));
    return;
  }
  const size_t bytes = generic_lds_bytes(max_mf);
  allow_large_lds(reinterpret_cast<const void*>(bp_level_big<false>), bytes);
  hipLaunchKernelGGL(bp_level_big<false>, dim3(ntasks, n_sites), dim3(kBigThreads), bytes, st, S, d_task_off, d_entries, task0,
                     seq_base, stop_below, (double*)nullptr, (int64_t)0);
}

// ---------------------------------------------------------------------------------------------------------
// Univariate / tiny beliefs (every dimension <= 2): ONE THREAD per (site, task), lanes = sites.
// The message descriptors are wave-uniform (scalar loads); each lane reads its own site's records.  This is the
// batch-of-independent-sites case (cfg4: thousands of univariate problems on one tree), where a wavefront per
// message would leave 63 of 64 lanes idle.  Same semantics as bp_level_generic (src/beliefupdates.jl:55-83,
// 483-488, 579-587, src/beliefs.jl:994-1003), closed forms for m_f <= 2.
// (element picks from 2- and 4-element register arrays by a wave-uniform index: a dynamically indexed local array would
// live in scratch memory)
__device__ __forceinline__ double pick4(const double (&v)[4], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3])); }
__device__ __forceinline__ double pick2(const double (&v)[2], int i) { return i == 0 ? v[0] : v[1]; }
__device__ __forceinline__ int pick2i(const int (&v)[2], int i) { return i == 0 ? v[0] : v[1]; }

template <bool SM>
__global__ __launch_bounds__(256) void bp_level_uni(DevState S, const int32_t* __restrict__ task_off,
                                                    const Entry* __restrict__ entries, int task0, int n_sites,
                                                    unsigned long long seq_base, unsigned long long stop_below) {
  const int site = blockIdx.y * blockDim.x + threadIdx.x;
  if (site >= n_sites) return;
  if ((S.fail[site] >> kInfoBits) < stop_below) return;
  const int task = task0 + blockIdx.x;
  double* __restrict__ pool = S.pool + (int64_t)site * S.pool_stride;
  double* __restrict__ rpool = S.rpool + (int64_t)site * S.rpool_stride;
  // element t of a belief / residual record: plain (this site's pool + padded offset) or site-minor (SM: lanes =
  // consecutive sites read consecutive doubles)
  const int64_t ns = sm_row(S.n_sites);   // (row stride of the site-minor layout)
  auto bel = [&](int b, int64_t plain_off, int t) -> double* {
    if constexpr (SM) return S.pool + (S.packed_off[b] + t) * ns + site;
    else return pool + plain_off + t;
  };
  // per-message / per-cluster words (flags, status, poison): [site][index] in the plain layout, [index][site] in the
  // site-minor one
  auto mword = [&](int32_t* base, int msg) -> int32_t* {
    if constexpr (SM) return base + (int64_t)msg * ns + site;
    else return base + (int64_t)site * S.n_msgs + msg;
  };
  auto cword = [&](int32_t* base, int c) -> int32_t* {
    if constexpr (SM) return base + (int64_t)c * ns + site;
    else return base + (int64_t)site * S.n_clusters + c;
  };
  auto rsd = [&](int msg, int64_t plain_off, int t) -> double* {
    if constexpr (SM) return S.rpool + (S.rpacked_off[msg] + t) * ns + site;
    else return rpool + plain_off + t;
  };
  const int e0 = task_off[task], e1 = task_off[task + 1];
  double mJs[2][2] = {{0, 0}, {0, 0}}, mh[2] = {0, 0}, gmsg = 0.0;  // message (s <= 2): mJs[a][b]
  for (int e = e0; e < e1; ++e) {
    const Entry en = entries[e];
    const MsgDesc m = S.msgs[en.msg];
    if (*cword(S.poison, m.from_b)) {
      *cword(S.poison, m.to_b) = 1;
      return;
    }
    const int mf = m.mf, s = m.s, mt = m.mt, ni = m.ni;
    if (!en.reuse) {
      double J[4] = {0, 0, 0, 0}, h[2] = {0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < mf * mf) J[t] = *bel(m.from_b, m.from_off, t);
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (t < mf) h[t] = *bel(m.from_b, m.from_off, mf * mf + t);
      gmsg = *bel(m.from_b, m.from_off, mf * mf + mf);
      int keep[2] = {0, 0}, integ[2] = {0, 0};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t < s) keep[t] = S.idx[m.keep_map + t];
        if (t < ni) integ[t] = S.idx[m.int_map + t];
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (b < s) mh[b] = pick2(h, keep[b]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
          if (a < s && b < s) mJs[a][b] = pick4(J, keep[a] + keep[b] * mf);
      }
      if (ni > 0) {
        // "fake" message: J_I, h_I, J_SI all ~ 0 (src/beliefupdates.jl:62-66)
        bool nz = false;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (b < ni) {
            nz |= fabs(pick2(h, integ[b])) > PGBP_EPS;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              if (a < ni) nz |= fabs(pick4(J, integ[a] + integ[b] * mf)) > PGBP_EPS;
              if (a < s) nz |= fabs(pick4(J, keep[a] + integ[b] * mf)) > PGBP_EPS;
            }
          }
        }
        if (nz) {
          int info = 0;
          // Cholesky of Symmetric(J_I) (upper triangle), ni <= 2
          const double d0 = pick4(J, integ[0] + integ[0] * mf);
          double u01 = 0.0, d1 = 1.0;
          if (!(d0 > 0.0)) info = 1;
          if (!info && ni == 2) {
            u01 = pick4(J, integ[0] + integ[1] * mf);       // upper entry (row integ[0] < integ[1])
            d1 = pick4(J, integ[1] + integ[1] * mf) - u01 * u01 / d0;
            if (!(d1 > 0.0)) info = 2;
          }
          if (info) {
            *mword(S.status, en.msg) = info;
            *cword(S.poison, m.to_b) = 1;
            atomicMin(&S.fail[site], ((seq_base + (unsigned long long)en.seq) << kInfoBits) | (unsigned long long)info);
            return;
          }
          // forward substitution with L = U': y = L^-1 [h_I | J_IS columns]
          const double y0 = pick2(h, integ[0]);
          const double y1 = ni == 2 ? pick2(h, integ[1]) - u01 / d0 * y0 : 0.0;
          double quad = y0 * y0 / d0 + (ni == 2 ? y1 * y1 / d1 : 0.0);
          const double logdet = log(d0) + (ni == 2 ? log(d1) : 0.0);
          // z_a = L^-1 J_SI[a, :]' (per kept variable a); message J -= z_a . z_b (D^-1 weighted), h -= z_a . y
          double z0[2] = {0, 0}, z1[2] = {0, 0};
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            if (a < s) {
              z0[a] = pick4(J, keep[a] + integ[0] * mf);
              z1[a] = ni == 2 ? pick4(J, keep[a] + integ[1] * mf) - u01 / d0 * z0[a] : 0.0;
            }
          }
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            if (b < s) {
              mh[b] -= z0[b] * y0 / d0 + (ni == 2 ? z1[b] * y1 / d1 : 0.0);
#pragma unroll
              for (int a = 0; a < 2; ++a)
                if (a < s) mJs[a][b] -= z0[a] * z0[b] / d0 + (ni == 2 ? z1[a] * z1[b] / d1 : 0.0);
            }
          }
          gmsg += 0.5 * ((double)ni * PGBP_LOG2PI - logdet + quad);
        }
      }
    }
    // ---- divide! and mult!
    auto sep = [&](int t) -> double& { return *bel(m.sep_b, m.sep_off, t); };
    const bool sz = S.sep_zero != 0;   // straight after a reset: the sepset is 1 (all zeros) and is not read (pgbp_engine.hip: fresh_sepsets_shortcut)
    auto to = [&](int t) -> double& { return *bel(m.to_b, m.to_off, t); };
    auto res = [&](int t) -> double& { return *rsd(en.msg, m.res_off, t); };
    int up[2] = {0, 0};
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < s) up[t] = S.idx[m.up_map + t];
    double maxJ = 0.0, maxh = 0.0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (b < s) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (a < s) {
            const int o = a + b * s;
            const double dJ = mJs[a][b] - (sz ? 0.0 : sep(o));
            sep(o) = mJs[a][b];
            res(o) = dJ;
            to(up[a] + up[b] * mt) += dJ;
            maxJ = (dJ != dJ) ? INFINITY : fmax(maxJ, fabs(dJ));
          }
        }
        const int o = s * s + b;
        const double dh = mh[b] - (sz ? 0.0 : sep(o));
        sep(o) = mh[b];
        res(o) = dh;
        to(mt * mt + up[b]) += dh;
        maxh = (dh != dh) ? INFINITY : fmax(maxh, fabs(dh));
      }
    }
    const int og = s * s + s;
    const double dg = gmsg - (sz ? 0.0 : sep(og));
    sep(og) = gmsg;
    to(mt * mt + mt) += dg;
    *mword(S.status, en.msg) = 0;
    if (S.update_resnorm) {
      const bool ok = maxh <= S.thr[s] && maxJ <= S.thr[PGBP_MAX_DIM + 1 + s];
      *mword(S.flags, en.msg) = ok ? 1 : 0;
      if (!ok && S.notcal) S.notcal[site] = 1;
    }
  }
}

// The same for engines whose sepsets hold at most ONE variable (univariate traits: clusters {child, parent} of at most
// two variables -- cfg4): every element a message needs is loaded by its ROLE (the kept diagonal entry, the pivot, the
// coupling ...), its index computed on the scalar unit, instead of loading the record and picking elements by runtime
// indices; one thread = one site, lanes = consecutive sites.  Expressions and their order are those of bp_level_uni.
// (the body: one task of one site; a `return` ends the task)
template <bool SM>
__device__ __forceinline__ void uni1_task(const DevState& S, const int32_t* __restrict__ task_off,
                                          const URec* __restrict__ urecs, const int task, const int site,
                                          unsigned long long seq_base) {
  double* __restrict__ pool = S.pool + (int64_t)site * S.pool_stride;
  double* __restrict__ rpool = S.rpool + (int64_t)site * S.rpool_stride;
  const int64_t ns = sm_row(S.n_sites);   // (row stride of the site-minor layout)
  auto bel = [&](int64_t packed, int64_t plain_off, int t) -> double* {
    if constexpr (SM) return S.pool + (packed + t) * ns + site;
    else return pool + plain_off + t;
  };
  auto mword = [&](int32_t* base, int msg) -> int32_t* {
    if constexpr (SM) return base + (int64_t)msg * ns + site;
    else return base + (int64_t)site * S.n_msgs + msg;
  };
  auto cword = [&](int32_t* base, int c) -> int32_t* {
    if constexpr (SM) return base + (int64_t)c * ns + site;
    else return base + (int64_t)site * S.n_clusters + c;
  };
  auto rsd = [&](int64_t packed, int64_t plain_off, int t) -> double* {
    if constexpr (SM) return S.rpool + (packed + t) * ns + site;
    else return rpool + plain_off + t;
  };
  const int e0 = task_off[task], e1 = task_off[task + 1];
  double mJ = 0.0, mh = 0.0, gmsg = 0.0;   // the message (s = 1)
  for (int e = e0; e < e1; ++e) {
    const URec m = urecs[e];   // (one record: entry, descriptor, index maps, offsets)
    // the sender's failure mark is REQUESTED here and looked at below, behind the requests of the message's operands: a check
    // up here put one more memory round trip (record -> mark -> operands) into every narrow level's chain of dependent loads
    const int poisoned = *cword(S.poison, m.from_b);
    const int mf = m.mf, s = m.s, mt = m.mt, ni = m.ni;
    auto from = [&](int t) -> double { return *bel(m.from_p, m.from_off, t); };
    // ... and so are the sepset's and the receiver's entries this message will read (divide! and mult! below): behind the
    // branches of the marginalisation they were a third round trip of the chain
    auto sep = [&](int t) -> double& { return *bel(m.sep_p, m.sep_off, t); };
    const bool sz = S.sep_zero != 0;   // straight after a reset: the sepset is 1 (all zeros) and is not read (pgbp_engine.hip: fresh_sepsets_shortcut)
    auto to = [&](int t) -> double& { return *bel(m.to_p, m.to_off, t); };
    const int og = s * s + s;
    double pre_sJ = 0.0, pre_sh = 0.0, pre_tJ = 0.0, pre_th = 0.0;
    // (STREAMING accesses, round 4: the sepset, the residual, the status and flag words are touched once per traversal; the
    // sender's and the receiver's entries are what the next level reads again)
    if (s == 1) {
      if (!sz) { pre_sJ = __builtin_nontemporal_load(&sep(0)); pre_sh = __builtin_nontemporal_load(&sep(1)); }
      pre_tJ = to(m.u + m.u * mt);
      pre_th = to(mt * mt + m.u);
    }
    const double pre_sg = sz ? 0.0 : __builtin_nontemporal_load(&sep(og)), pre_tg = to(mt * mt + mt);
    if (!m.reuse) {
      gmsg = from(mf * mf + mf);
      int info = 0;
      if (s == 1) {
        const int k = m.k;
        mJ = from(k + k * mf);
        mh = from(mf * mf + k);
        if (ni == 1) {
          const int i = m.i0;
          const double Jii = from(i + i * mf), Jki = from(k + i * mf), hi = from(mf * mf + i);
          // "fake" message: J_I, h_I, J_SI all ~ 0 (src/beliefupdates.jl:62-66)
          if (fabs(hi) > PGBP_EPS || fabs(Jii) > PGBP_EPS || fabs(Jki) > PGBP_EPS) {
            const double d0 = Jii;
            if (!(d0 > 0.0)) {
              info = 1;
            } else {
              const double y0 = hi;
              const double quad = y0 * y0 / d0 + 0.0;
              const double logdet = log(d0) + 0.0;
              const double z0 = Jki;
              mh -= z0 * y0 / d0 + 0.0;
              mJ -= z0 * z0 / d0 + 0.0;
              gmsg += 0.5 * ((double)ni * PGBP_LOG2PI - logdet + quad);
            }
          }
        }
      } else if (ni > 0) {
        // an empty sepset: every variable of the sender (one or two) is integrated, the message is its constant
        const int i0 = m.i0;
        const int i1 = ni == 2 ? m.i1 : 0;
        const double d0 = from(i0 + i0 * mf), y0 = from(mf * mf + i0);
        const double u01 = ni == 2 ? from(i0 + i1 * mf) : 0.0;          // upper entry (row i0 < i1)
        const double J10 = ni == 2 ? from(i1 + i0 * mf) : 0.0, J11 = ni == 2 ? from(i1 + i1 * mf) : 0.0;
        const double h1 = ni == 2 ? from(mf * mf + i1) : 0.0;
        bool nz = fabs(y0) > PGBP_EPS || fabs(d0) > PGBP_EPS;
        if (ni == 2) nz = nz || fabs(h1) > PGBP_EPS || fabs(u01) > PGBP_EPS || fabs(J10) > PGBP_EPS || fabs(J11) > PGBP_EPS;
        if (nz) {
          double d1 = 1.0;
          if (!(d0 > 0.0)) info = 1;
          if (!info && ni == 2) {
            d1 = J11 - u01 * u01 / d0;
            if (!(d1 > 0.0)) info = 2;
          }
          if (!info) {
            const double y1 = ni == 2 ? h1 - u01 / d0 * y0 : 0.0;
            const double quad = y0 * y0 / d0 + (ni == 2 ? y1 * y1 / d1 : 0.0);
            const double logdet = log(d0) + (ni == 2 ? log(d1) : 0.0);
            gmsg += 0.5 * ((double)ni * PGBP_LOG2PI - logdet + quad);
          }
        }
      }
      if (poisoned) {   // (nothing of this message has been recorded or stored yet)
        *cword(S.poison, m.to_b) = 1;
        return;
      }
      if (info) {
        *mword(S.status, m.msg) = info;
        *cword(S.poison, m.to_b) = 1;
        atomicMin(&S.fail[site], ((seq_base + (unsigned long long)m.seq) << kInfoBits) | (unsigned long long)info);
        return;
      }
    } else if (poisoned) {
      *cword(S.poison, m.to_b) = 1;
      return;
    }
    // ---- divide! and mult! (on the entries requested above: the same operations as read-modify-write in place)
    double maxJ = 0.0, maxh = 0.0;
    if (s == 1) {
      const int u = m.u;
      const double dJ = mJ - pre_sJ;
      __builtin_nontemporal_store(mJ, &sep(0));
      __builtin_nontemporal_store(dJ, rsd(m.res_p, m.res_off, 0));
      to(u + u * mt) = pre_tJ + dJ;
      maxJ = (dJ != dJ) ? INFINITY : fmax(maxJ, fabs(dJ));
      const double dh = mh - pre_sh;
      __builtin_nontemporal_store(mh, &sep(1));
      __builtin_nontemporal_store(dh, rsd(m.res_p, m.res_off, 1));
      to(mt * mt + u) = pre_th + dh;
      maxh = (dh != dh) ? INFINITY : fmax(maxh, fabs(dh));
    }
    const double dg = gmsg - pre_sg;
    __builtin_nontemporal_store(gmsg, &sep(og));
    to(mt * mt + mt) = pre_tg + dg;
    __builtin_nontemporal_store(0, mword(S.status, m.msg));
    if (S.update_resnorm) {
      const bool ok = maxh <= S.thr[s] && maxJ <= S.thr[PGBP_MAX_DIM + 1 + s];
      __builtin_nontemporal_store(ok ? 1 : 0, mword(S.flags, m.msg));
      if (!ok && S.notcal) S.notcal[site] = 1;   // (every writer stores the same 1: no atomic)
    }
  }
}

template <bool SM>
__global__ __launch_bounds__(256) void bp_level_uni1(DevState S, const int32_t* __restrict__ task_off,
                                                     const URec* __restrict__ urecs, int task0, int n_sites,
                                                     unsigned long long seq_base, unsigned long long stop_below, int ny) {
  // ny > 0: a one-dimensional grid, the ny site blocks of a task in CONSECUTIVE workgroups (they read and write adjacent
  // pieces of the same rows of the site-minor layout); ny == 0: grid (tasks, site blocks)
  const int yb = ny > 0 ? (int)(blockIdx.x % (unsigned)ny) : (int)blockIdx.y;
  const int tk = ny > 0 ? (int)(blockIdx.x / (unsigned)ny) : (int)blockIdx.x;
  const int site = yb * blockDim.x + threadIdx.x;
  if (site >= n_sites) return;
  if ((S.fail[site] >> kInfoBits) < stop_below) return;
  uni1_task<SM>(S, task_off, urecs, task0 + tk, site, seq_base);
}

// LOOP MODE of the same body: a chunk of fused narrow levels (pgbp_plan.cpp: build_chunks, plans of univariate site batches).
// Workgroup (b, y) = one dependency-closed tree of tasks for the 64 sites [64 y, 64 y + 64): wavefront w runs the task
// grp_tasks[g][w] (-1: none) of every group g of its walk, lanes = sites, with a workgroup barrier in between -- the
// stores of a level are complete (vmcnt) and visible (same CU, same vector L1) before the next level's loads of the same
// sites.  A level launch costs 6 - 15 us whatever it holds; a fused level its tasks.
template <bool SM>
__global__ __launch_bounds__(kTailWaves * 64) void bp_chunk_uni1(DevState S, const int32_t* __restrict__ task_off,
                                                                 const URec* __restrict__ urecs,
                                                                 const int32_t* __restrict__ grp_tasks,
                                                                 const int32_t* __restrict__ wg_off, int n_sites,
                                                                 unsigned long long seq_base, unsigned long long stop_below,
                                                                 int ny) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (one-dimensional grid, walk-major: the ny site blocks of a walk in consecutive workgroups, as in bp_level_uni1)
  const int walk = (int)(blockIdx.x / (unsigned)ny), yb = (int)(blockIdx.x % (unsigned)ny);
  const int site = yb * 64 + lane;
  // (a stopped or padding lane skips the work, never a barrier)
  const bool live = site < n_sites && !((S.fail[site < n_sites ? site : 0] >> kInfoBits) < stop_below);
  const int g0 = wg_off[walk], g1 = wg_off[walk + 1];
  for (int g = g0; g < g1; ++g) {
    const int t = grp_tasks[(int64_t)g * kTailWaves + wave];
    if (t >= 0 && live) uni1_task<SM>(S, task_off, urecs, t, site, seq_base);
    if (g + 1 < g1) __syncthreads();
  }
}

void launch_level_uni(const DevState& S, const int32_t* d_task_off, const Entry* d_entries, const URec* d_urecs, int task0,
                      int ntasks, int n_sites, unsigned long long seq_base, unsigned long long stop_below, int max_s,
                      hipStream_t st) {
  if (ntasks <= 0) return;
  const int bs = n_sites >= 256 ? 256 : 64;
  const dim3 grid(ntasks, (n_sites + bs - 1) / bs);
  if (max_s <= 1 && d_urecs) {   // every sepset of the engine holds at most one variable
    const int ny = (n_sites + bs - 1) / bs;
    // (round 4: task-major on a one-dimensional grid -- 5.9 -> 5.4 ms per sharded step at 1 000 problems, 31.0 -> 30.1 at 8 000:
    // consecutive workgroups touch adjacent pieces of the same rows; the two-dimensional grid only where the product overflows)
    const bool flat = (long long)ntasks * ny < (1ll << 31);
    const dim3 g1 = flat ? dim3((unsigned)(ntasks * ny)) : grid;
    if (S.sm) hipLaunchKernelGGL(bp_level_uni1<true>, g1, dim3(bs), 0, st, S, d_task_off, d_urecs, task0, n_sites, seq_base, stop_below, flat ? ny : 0);
    else hipLaunchKernelGGL(bp_level_uni1<false>, g1, dim3(bs), 0, st, S, d_task_off, d_urecs, task0, n_sites, seq_base, stop_below, flat ? ny : 0);
  } else {
    if (S.sm) hipLaunchKernelGGL(bp_level_uni<true>, grid, dim3(bs), 0, st, S, d_task_off, d_entries, task0, n_sites, seq_base, stop_below);
    else hipLaunchKernelGGL(bp_level_uni<false>, grid, dim3(bs), 0, st, S, d_task_off, d_entries, task0, n_sites, seq_base, stop_below);
  }
}

// a chunk of fused levels of a univariate site batch (every sepset <= 1 variable): n_wg trees of tasks x blocks of 64 sites
void launch_chunk_uni1(const DevState& S, const int32_t* d_task_off, const URec* d_urecs, const int32_t* d_grp_tasks,
                       const int32_t* d_wg_off, int n_wg, int n_sites, unsigned long long seq_base,
                       unsigned long long stop_below, hipStream_t st) {
  if (n_wg <= 0) return;
  const int ny = (n_sites + 63) / 64;
  const dim3 grid((unsigned)n_wg * (unsigned)ny), block(kTailWaves * 64);   // (n_wg <= a few thousand walks: no overflow)
  if (S.sm) hipLaunchKernelGGL(bp_chunk_uni1<true>, grid, block, 0, st, S, d_task_off, d_urecs, d_grp_tasks, d_wg_off, n_sites, seq_base, stop_below, ny);
  else hipLaunchKernelGGL(bp_chunk_uni1<false>, grid, block, 0, st, S, d_task_off, d_urecs, d_grp_tasks, d_wg_off, n_sites, seq_base, stop_below, ny);
}

size_t generic_lds_bytes(int max_mf) {
  const int mf = max_mf < 1 ? 1 : max_mf;
  const int ld = (mf + 1) | 1;
  return sizeof(double) * (size_t)(kPermDoubles + mf * ld);
}

void launch_level_generic(const DevState& S, const GRec* d_recs, int rec0, int ntasks, int n_sites,
                          unsigned long long seq_base, unsigned long long stop_below, int max_mf, bool small_only,
                          hipStream_t st, const int32_t* d_rowmap, int n_rows, int small4_min) {
  if (ntasks <= 0) return;
  // four tasks per wavefront from the width at which a level is bound by instruction issue, not by one message's latency
  // (small4_min: Tuning::small4_min of the engine's plan)
  if (small_only && small4_min >= 0 && ntasks >= small4_min && d_rowmap && n_rows > 0)
    hipLaunchKernelGGL(bp_level_small4<true>, dim3(n_rows / 4, n_sites), dim3(kWave), 0, st, S, d_recs, rec0, n_rows, d_rowmap,
                       seq_base, stop_below);
  else if (small_only && small4_min >= 0 && ntasks >= small4_min)
    hipLaunchKernelGGL(bp_level_small4<false>, dim3((ntasks + 3) / 4, n_sites), dim3(kWave), 0, st, S, d_recs, rec0, ntasks,
                       nullptr, seq_base, stop_below);
  else if (small_only)
    hipLaunchKernelGGL(bp_level_generic<true>, dim3(ntasks, n_sites), dim3(kWave), 0, st, S, d_recs, rec0, seq_base, stop_below);
  else
    hipLaunchKernelGGL(bp_level_generic<false>, dim3(ntasks, n_sites), dim3(kWave), generic_lds_bytes(max_mf), st, S, d_recs, rec0,
                       seq_base, stop_below);
}

void launch_chunk_generic(const DevState& S, const GRec* d_recs, const int32_t* d_grp_recs, const int32_t* d_wg_off, int n_wg,
                          int n_sites, unsigned long long seq_base, unsigned long long stop_below, int max_mf,
                          bool small_only, bool pair, hipStream_t st) {
  if (n_wg <= 0) return;
  const size_t per_wave = generic_lds_bytes(max_mf) / sizeof(double);
  if (small_only && pair)
    launch_chunk_pair(S, d_recs, d_grp_recs, d_wg_off, n_wg, n_sites, seq_base, stop_below, st);   // pgbp_pair.hip
  else if (small_only)
    hipLaunchKernelGGL(bp_chunk_generic<true>, dim3(n_wg, n_sites), dim3(kTailWaves * 64), 0, st, S, d_recs, d_grp_recs, d_wg_off,
                       0, seq_base, stop_below);
  else
    hipLaunchKernelGGL(bp_chunk_generic<false>, dim3(n_wg, n_sites), dim3(kTailWaves * 64), per_wave * sizeof(double) * kTailWaves,
                       st, S, d_recs, d_grp_recs, d_wg_off, (int)per_wave, seq_base, stop_below);
}

// integratebelief(h, J, g) (src/beliefupdates.jl:187-200): mu = J \ h, norm = g + (m log 2pi - logdet J + h'mu)/2
// WS: a belief of more than kLdsMaxDim variables: [J | h] in a workspace slab in global memory (one per site)
template <bool WS>
__global__ __launch_bounds__(WS ? 1024 : 64) void integrate_kernel(const double* __restrict__ pool_all, int64_t pool_stride,
                                                       int64_t rec_off, int m, int bs, int fp,
                                                       double* __restrict__ mu,
                                                       int mu_stride, double* __restrict__ norm,
                                                       int32_t* __restrict__ info_out, double* __restrict__ ws,
                                                       int64_t ws_stride) {
  const int lane = threadIdx.x, nthr = blockDim.x;   // (WS: sixteen wavefronts -- round 4, last session; else one)
  const int site = blockIdx.x;
  const double* __restrict__ rec = pool_all + (int64_t)site * pool_stride + rec_off;
  double* W = WS ? ws + (int64_t)site * ws_stride : lds + kPermDoubles;
  const int ld = (m + 1) | 1;
  const bool packed = bs && bs16::applies(m, fp);
  const double g = packed ? rec[bs16::g_off(m, fp)] : rec[(int64_t)m * m + m];
  if constexpr (!WS) {
    if (m <= 16 && mu == nullptr) {
      // the normalisation constant alone of a belief of at most 16 variables (the root of a log-likelihood evaluation): the
      // system in registers, one row per lane of the wavefront's first DPP row, the pivot row by row broadcast -- no LDS, no
      // barrier per pivot (Small4<16, 0>: the elimination of the register-resident small-message body with nothing kept).
      // Entry by entry the operations of eliminate_leading below, in its order (the mantissa product renormalised where it does).
      const int i = lane & 15;
      const bool live = lane < m;
      double row[17];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        row[j] = 0.0;
        if (live && j < m)
          row[j] = packed ? rec[bs16::J_off(m, i, j, fp)] : ((i <= j) ? rec[i + (int64_t)j * m] : rec[j + (int64_t)i * m]);
      }
      row[16] = live ? (packed ? rec[bs16::h_off(m, i, fp)] : rec[(int64_t)m * m + i]) : 0.0;
      bool nzr = live && row[16] != 0.0;
      if (live) {
        if (packed) {
#pragma unroll
          for (int j = 0; j < 16; ++j) nzr |= row[j] != 0.0;
        } else {   // (the generic path looks at every stored entry, both triangles)
          for (int j = 0; j < m; ++j) nzr |= rec[i + (int64_t)j * m] != 0.0;
        }
      }
      if (!__any(nzr)) {  // constant belief: norm = g (:189-191)
        if (lane == 0) {
          norm[site] = g;
          if (info_out) info_out[site] = 0;
        }
        return;
      }
      double mant = 1.0, quad = 0.0;
      int expo = 0, info = 0;
      Small4<16, 0>::template pivot<0, decltype(row), false>(row, m, info, mant, expo, quad);
      info = __builtin_amdgcn_readfirstlane(info);
      if (info != 0) {
        if (lane == 0) {
          norm[site] = NAN;
          if (info_out) info_out[site] = info;
        }
        return;
      }
      if (m == 16) { int ex; mant = frexp(mant, &ex); expo += ex; }   // (eliminate_leading: after every sixteenth pivot)
      const double logdet = log(mant) + (double)expo * 0.69314718055994530941723212145818;
      if (lane == 0) {
        norm[site] = g + 0.5 * ((double)m * PGBP_LOG2PI - logdet + quad);
        if (info_out) info_out[site] = 0;
      }
      return;
    }
  }
  bool nz = false;
  for (int idx = lane; idx < m * m; idx += nthr) {
    const int j = idx / m, i = idx - j * m;
    if (packed) {
      const double v = rec[bs16::J_off(m, i, j, fp)];  // symmetric by construction
      nz |= v != 0.0;
      W[i * ld + j] = v;
    } else {
      const double raw = rec[idx];
      nz |= raw != 0.0;
      // PDMat(Symmetric(J)): read the upper triangle
      W[i * ld + j] = (i <= j) ? raw : rec[j + (int64_t)i * m];
    }
  }
  for (int i = lane; i < m; i += nthr) {
    const double hv = packed ? rec[bs16::h_off(m, i, fp)] : rec[(int64_t)m * m + i];
    nz |= hv != 0.0;
    W[i * ld + m] = hv;
  }
  const bool any_nz = WS ? (__syncthreads_or(nz ? 1 : 0) != 0) : (__syncthreads(), __any(nz) != 0);
  if (!any_nz) {  // constant belief: mu = Inf, norm = g (:189-191)
    if (mu)
      for (int i = lane; i < m; i += nthr) mu[(int64_t)site * mu_stride + i] = INFINITY;
    if (lane == 0) {
      norm[site] = g;
      if (info_out) info_out[site] = 0;
    }
    return;
  }
  double logdet, quad;
  int info = 0;
  if constexpr (!WS) {
    info = eliminate_leading(W, ld, m, m, lane, logdet, quad);
  } else {
    // eliminate_leading by the whole workgroup on the workspace slab: rows by wavefront, columns by lane, the pivot row's
    // entries once per pivot and a row's entries requested together (every access is a round trip to the L2: bp_level_big)
    constexpr int kChunks = (PGBP_MAX_DIM + 1 + kWave - 1) / kWave;
    double mant = 1.0;
    int expo = 0;
    quad = 0.0;
    for (int k = 0; k < m; ++k) {
      const double d = W[k * ld + k];
      const double hk = W[k * ld + m];
      if (!(d > 0.0)) {
        info = k + 1;
        break;
      }
      double rd = __builtin_amdgcn_rcp(d);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      int ex;
      mant *= frexp(d, &ex);
      expo += ex;
      if ((k & 15) == 15) { mant = frexp(mant, &ex); expo += ex; }
      quad += hk * hk * rd;
      const int j0 = k + 1 + (lane & 63);
      double rk[kChunks];
#pragma unroll
      for (int c = 0; c < kChunks; ++c) rk[c] = (j0 + c * kWave <= m) ? W[k * ld + j0 + c * kWave] : 0.0;
      for (int i = k + 1 + (lane >> 6); i < m; i += (nthr >> 6)) {
        const double lik = W[i * ld + k] * rd;
        double wv[kChunks];
#pragma unroll
        for (int c = 0; c < kChunks; ++c) wv[c] = (j0 + c * kWave <= m) ? W[i * ld + j0 + c * kWave] : 0.0;
#pragma unroll
        for (int c = 0; c < kChunks; ++c)
          if (j0 + c * kWave <= m) W[i * ld + j0 + c * kWave] = wv[c] - lik * rk[c];
      }
      __syncthreads();
    }
    logdet = log(mant) + (double)expo * 0.69314718055994530941723212145818;
  }
  if (info != 0) {
    if (lane == 0) {
      norm[site] = NAN;
      if (info_out) info_out[site] = info;
    }
    return;
  }
  if (mu) {
    // back substitution on the upper-triangular system left by the elimination
    for (int k = m - 1; k >= 0; --k) {
      if (lane == 0) W[k * ld + m] = W[k * ld + m] / W[k * ld + k];
      __syncthreads();
      const double xk = W[k * ld + m];
      for (int i = lane; i < k; i += nthr) W[i * ld + m] -= W[i * ld + k] * xk;
      __syncthreads();
    }
    for (int i = lane; i < m; i += nthr) mu[(int64_t)site * mu_stride + i] = W[i * ld + m];
  }
  if (lane == 0) {
    norm[site] = g + 0.5 * ((double)m * PGBP_LOG2PI - logdet + quad);
    if (info_out) info_out[site] = 0;
  }
}

void launch_integrate(const double* pool, int64_t pool_stride, int64_t rec_off, int m, int bs16, int fast_p,
                      double* d_mu, int mu_stride, double* d_norm, int32_t* d_info, int n_sites, double* d_ws, hipStream_t st) {
  if (m > kLdsMaxDim) {   // (n_sites slabs of big_ws_doubles(m))
    hipLaunchKernelGGL(integrate_kernel<true>, dim3(n_sites), dim3(1024), 0, st, pool, pool_stride, rec_off, m, bs16, fast_p,
                       d_mu, mu_stride, d_norm, d_info, d_ws, big_ws_doubles(m));
    return;
  }
  allow_large_lds(reinterpret_cast<const void*>(integrate_kernel<false>), generic_lds_bytes(m));
  hipLaunchKernelGGL(integrate_kernel<false>, dim3(n_sites), dim3(kWave), generic_lds_bytes(m), st, pool, pool_stride,
                     rec_off, m, bs16, fast_p, d_mu, mu_stride, d_norm, d_info, (double*)nullptr, (int64_t)0);
}

// ---- BS16 <-> plain, in place, one workgroup per record (pgbp_bs16.hpp)
__global__ __launch_bounds__(256) void convert_layout_kernel(double* __restrict__ pool, int64_t stride,
                                                             const int64_t* __restrict__ off,
                                                             const int32_t* __restrict__ dim, int n_records,
                                                             int to_bs16, int is_residual, int fp) {
  __shared__ double buf[32 * 32 + 32 + 1];
  const int site = blockIdx.y;
  for (int r = blockIdx.x; r < n_records; r += gridDim.x) {
    const int m = dim[r];
    if (!bs16::applies(m, fp) || (is_residual && m != fp)) continue;  // uniform per workgroup
    double* __restrict__ rec = pool + (int64_t)site * stride + off[r];
    const int tail = is_residual ? m : m + 1;  // h (and g) after J
    const int plain_len = m * m + tail;
    const int hb = m == fp ? bs16::h1(fp) : bs16::h2(fp);  // where h starts in the packed record
    const int packed_len = hb + tail;
    const int src_len = to_bs16 ? plain_len : packed_len;
    __syncthreads();
    for (int t = threadIdx.x; t < src_len; t += blockDim.x) buf[t] = rec[t];
    __syncthreads();
    if (to_bs16) {
      for (int idx = threadIdx.x; idx < m * m; idx += blockDim.x) {
        const int c = idx / m, rr = idx - c * m;
        if (bs16::canonical(m, rr, c, fp)) rec[bs16::J_off(m, rr, c, fp)] = buf[idx];
      }
      for (int t = threadIdx.x; t < tail; t += blockDim.x) rec[hb + t] = buf[m * m + t];
    } else {
      for (int idx = threadIdx.x; idx < m * m; idx += blockDim.x) {
        const int c = idx / m, rr = idx - c * m;
        rec[idx] = buf[bs16::J_off(m, rr, c, fp)];
      }
      for (int t = threadIdx.x; t < tail; t += blockDim.x) rec[m * m + t] = buf[hb + t];
    }
  }
}

void launch_convert_layout(double* pool, int64_t stride, const int64_t* d_off, const int32_t* d_dim, int n_records,
                           int n_sites, int to_bs16, int is_residual, int fast_p, hipStream_t st) {
  if (n_records <= 0) return;
  const int gx = n_records < 65535 ? n_records : 65535;
  hipLaunchKernelGGL(convert_layout_kernel, dim3(gx, n_sites), dim3(256), 0, st, pool, stride, d_off, d_dim,
                     n_records, to_bs16, is_residual, fast_p);
}

__global__ __launch_bounds__(256) void check_symmetry_kernel(const double* __restrict__ pool, int64_t stride,
                                                             const int64_t* __restrict__ off,
                                                             const int32_t* __restrict__ dim, int n_records,
                                                             int32_t* __restrict__ flag, int fp) {
  __shared__ double red[2][4];
  const int site = blockIdx.y;
  for (int r = blockIdx.x; r < n_records; r += gridDim.x) {
    const int m = dim[r];
    if (!bs16::applies(m, fp)) continue;
    const double* __restrict__ rec = pool + (int64_t)site * stride + off[r];
    double asym = 0.0, mx = 0.0;
    for (int idx = threadIdx.x; idx < m * m; idx += blockDim.x) {
      const int c = idx / m, rr = idx - c * m;
      const double v = rec[idx];
      mx = fmax(mx, fabs(v));
      if (rr > c) {
        const double d = fabs(v - rec[c + rr * m]);
        asym = (d != d) ? INFINITY : fmax(asym, d);
      }
    }
    asym = wave_max(asym);
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = asym; red[1][threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double A = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
      const double M = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
      if (!(A <= 1e-10 * M)) atomicOr(flag, 1);
    }
  }
}

void launch_check_symmetry(const double* pool, int64_t stride, const int64_t* d_off, const int32_t* d_dim,
                           int n_records, int n_sites, int32_t* d_flag, int fast_p, hipStream_t st) {
  if (n_records <= 0) return;
  const int gx = n_records < 65535 ? n_records : 65535;
  hipLaunchKernelGGL(check_symmetry_kernel, dim3(gx, n_sites), dim3(256), 0, st, pool, stride, d_off, d_dim,
                     n_records, d_flag, fast_p);
}

// ---- free_energy / factored_energy (src/score.jl:105-182): one wavefront per belief ------------------------
// cluster i (factor (J_t, h_t, g_t), belief (J, h)):  average energy = (tr(J^-1 J_t) + mu'J_t mu)/2 - h_t'mu - g_t,
//   entropy = (m (log 2pi + 1) - log det J) / 2, mu = J^-1 h;  sepset: -entropy.
// Gauss-Jordan on the augmented system [J | J_t | h] in LDS (pivots = Cholesky pivots of Symmetric(J)).
// contrib[site][belief] = (energy term, entropy term); summed per site by free_energy_reduce_kernel in a
// fixed order (deterministic).  info[site] = smallest 1-based belief index whose J is not positive definite.
__device__ __forceinline__ double fe_elem(const double* __restrict__ rec, int m, bool packed, int fp, int i, int j) {
  return packed ? rec[bs16::J_off(m, i, j, fp)] : rec[(i <= j) ? i + (int64_t)j * m : j + (int64_t)i * m];
}

// Gauss-Jordan on the m x nc augmented system W = [A | B] (row stride ld) held in LDS (or in a workspace slab), by the
// whole workgroup (one wavefront: the LDS instances; sixteen: the workspace ones, round 4): on return the right block holds
// A^-1 B and logdet = log det A.  Pivots are the Cholesky pivots of A (no pivoting); returns false (uniformly) as soon as a
// pivot is not positive, i.e. A is not positive definite.  Every entry sees the same operations whatever the workgroup's size.
template <bool WS>
__device__ __forceinline__ bool gauss_jordan_spd(double* W, int m, int nc, int ld, int tid, double& logdet) {
  const int nthr = blockDim.x, wv = tid >> 6, ln = tid & 63, nw = nthr >> 6;
  double mant = 1.0;
  int expo = 0;
  for (int k = 0; k < m; ++k) {
    const double d = W[k * ld + k];
    if (!(d > 0.0)) return false;
    int ex;
    mant *= frexp(d, &ex);
    expo += ex;
    if ((k & 15) == 15) { mant = frexp(mant, &ex); expo += ex; }
    const double rd = 1.0 / d;
    __syncthreads();
    for (int j = k + 1 + tid; j < nc; j += nthr) W[k * ld + j] *= rd;  // normalise the pivot row
    __syncthreads();
    // eliminate column k from every other row (columns > k only: the rest is never read again)
    const int ncol = nc - (k + 1);
    if (nw <= 1) {   // one wavefront (small systems): the entries flat over the lanes
      if (ncol > 0) {
        for (int idx = tid; idx < (m - 1) * ncol; idx += kWave) {
          int i = idx / ncol;
          const int j = k + 1 + (idx - i * ncol);
          if (i >= k) ++i;
          W[i * ld + j] -= W[i * ld + k] * W[k * ld + j];
        }
      }
    } else if (!WS) {   // eight wavefronts in LDS: rows by wavefront, columns by lane, no division
      for (int i0 = wv; i0 < m - 1; i0 += nw) {
        const int i = i0 + (i0 >= k ? 1 : 0);
        const double wik = W[i * ld + k];
        for (int j = k + 1 + ln; j < nc; j += kWave) W[i * ld + j] -= wik * W[k * ld + j];
      }
    } else {            // sixteen wavefronts, the system in the workspace: every access is a round trip to the L2 and the
      // compiler must assume W aliases itself -- the pivot row's entries once per pivot, a row's entries requested together
      constexpr int kChunks = (2 * PGBP_MAX_DIM + 1 + kWave - 1) / kWave;
      const int j0 = k + 1 + ln;
      double rk[kChunks];
#pragma unroll
      for (int c = 0; c < kChunks; ++c) rk[c] = (j0 + c * kWave < nc) ? W[k * ld + j0 + c * kWave] : 0.0;
      for (int i0 = wv; i0 < m - 1; i0 += nw) {
        const int i = i0 + (i0 >= k ? 1 : 0);
        const double wik = W[i * ld + k];
        double wr[kChunks];
#pragma unroll
        for (int c = 0; c < kChunks; ++c) wr[c] = (j0 + c * kWave < nc) ? W[i * ld + j0 + c * kWave] : 0.0;
#pragma unroll
        for (int c = 0; c < kChunks; ++c)
          if (j0 + c * kWave < nc) W[i * ld + j0 + c * kWave] = wr[c] - wik * rk[c];
      }
    }
    __syncthreads();
  }
  logdet = log(mant) + (double)expo * 0.69314718055994530941723212145818;
  return true;
}

// sum of `v` over the workgroup (every thread gets it); one wavefront: shuffles only
__device__ __forceinline__ double workgroup_sum(double v, int tid) {
  __shared__ double part[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (blockDim.x <= kWave) return v;
  if ((tid & 63) == 0) part[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
  __syncthreads();
  return t;
}

// kb: columns of J_t per pass (m: one pass, the whole [J | J_t | h] at once; fewer when m x (2m + 1) doubles exceed the
// LDS a workgroup may have: the elimination of J is then repeated per block of columns, mu kept from the first pass)
// WS: the beliefs listed in big_idx (more than lds_max_m variables: their [J | J_t | h] does not fit a CU's LDS even a column
// block at a time), one workgroup each, the working matrix in a workspace slab in global memory (the barriers of the
// elimination order its accesses: same CU, same vector L1, as in bp_level_big<true>); the LDS instance skips those
template <bool WS>
__global__ __launch_bounds__(WS ? 1024 : 512) void free_energy_kernel(const double* __restrict__ pool, int64_t pool_stride,
                                                         const double* __restrict__ fpool, int64_t fpool_stride,
                                                         const int64_t* __restrict__ boff,
                                                         const int32_t* __restrict__ dim, int n_clusters,
                                                         int n_beliefs, int bs, int fp, int kb_max,
                                                         double2* __restrict__ contrib, int32_t* __restrict__ info,
                                                         const int32_t* __restrict__ big_idx, int lds_max_m,
                                                         double* __restrict__ ws, int64_t ws_stride) {
  const int lane = threadIdx.x, nthr = blockDim.x, b = WS ? big_idx[blockIdx.x] : (int)blockIdx.x, site = blockIdx.y;
  const int m = dim[b];
  if (!WS && m > lds_max_m) return;
  const bool is_cluster = b < n_clusters;
  const double* __restrict__ rec = pool + (int64_t)site * pool_stride + boff[b];
  const double* __restrict__ frec = fpool + (int64_t)site * fpool_stride + (is_cluster ? boff[b] : 0);
  double2* out = contrib + (int64_t)site * n_beliefs + b;
  if (m == 0) {
    // empty J: -g_t for a cluster (src/score.jl:170-171), entropy 0 for a sepset
    if (lane == 0) *out = make_double2(is_cluster ? -frec[0] : 0.0, 0.0);
    return;
  }
  const bool packed = bs && bs16::applies(m, fp);
  const int kb = is_cluster ? ((WS || m < kb_max) ? m : kb_max) : 0;   // columns of J_t per pass (workspace: all of them)
  const int ld = (m + kb + 1) | 1;
  double* W = WS ? ws + ((int64_t)blockIdx.x + (int64_t)gridDim.x * blockIdx.y) * ws_stride : lds;
  double* mu = W + (size_t)m * ld;                              // m doubles behind the working matrix
  double acc = 0.0, logdet = 0.0;
  for (int c0 = 0; c0 == 0 || c0 < (is_cluster ? m : 0); c0 += (kb > 0 ? kb : m)) {
    const int nb = is_cluster ? (m - c0 < kb ? m - c0 : kb) : 0;  // columns of J_t in this pass
    const int nc = m + nb + (c0 == 0 && is_cluster ? 1 : 0);      // + h in the first pass
    __syncthreads();
    for (int idx = lane; idx < m * m; idx += nthr) {
      const int j = idx / m, i = idx - j * m;
      W[i * ld + j] = fe_elem(rec, m, packed, fp, i, j);  // Symmetric(J): upper triangle
    }
    for (int idx = lane; idx < m * nb; idx += nthr) {
      const int jj = idx / m, i = idx - jj * m, j = c0 + jj;
      W[i * ld + m + jj] = packed ? frec[bs16::J_off(m, i, j, fp)] : frec[i + (int64_t)j * m];
    }
    if (c0 == 0 && is_cluster)
      for (int i = lane; i < m; i += nthr) W[i * ld + m + nb] = packed ? rec[bs16::h_off(m, i, fp)] : rec[(int64_t)m * m + i];
    __syncthreads();
    double ld_pass;
    if (!gauss_jordan_spd<WS>(W, m, nc, ld, lane, ld_pass)) {
      if (lane == 0) { atomicMin(&info[site], b + 1); *out = make_double2(NAN, NAN); }
      return;
    }
    if (c0 == 0) logdet = ld_pass;
    if (!is_cluster) break;
    // right block: J^-1 J_t[:, c0 .. c0 + nb) and, in the first pass, mu = J^-1 h behind it
    for (int jj = lane; jj < nb; jj += nthr) acc += 0.5 * W[(c0 + jj) * ld + m + jj];  // tr(J^-1 J_t) / 2
    if (c0 == 0)
      for (int i = lane; i < m; i += nthr) mu[i] = W[i * ld + m + nb];
  }
  __syncthreads();
  const double ent = 0.5 * ((double)m * (PGBP_LOG2PI + 1.0) - logdet);
  if (!is_cluster) {
    if (lane == 0) *out = make_double2(0.0, -ent);
    return;
  }
  for (int idx = lane; idx < m * m; idx += nthr) {
    const int j = idx / m, i = idx - j * m;
    const double jt = packed ? frec[bs16::J_off(m, i, j, fp)] : frec[i + (int64_t)j * m];
    acc += 0.5 * mu[i] * jt * mu[j];                                     // mu'J_t mu / 2
  }
  for (int i = lane; i < m; i += nthr)
    acc -= (packed ? frec[bs16::h_off(m, i, fp)] : frec[(int64_t)m * m + i]) * mu[i];  // - h_t'mu
  acc = workgroup_sum(acc, lane);
  if (lane == 0) *out = make_double2(acc - (packed ? frec[bs16::g_off(m, fp)] : frec[(int64_t)m * m + m]), ent);
}

__global__ __launch_bounds__(256) void free_energy_reduce_kernel(const double2* __restrict__ contrib, int n_beliefs,
                                                                 double* __restrict__ out3) {
  __shared__ double sa[256], se[256];
  const int site = blockIdx.x;
  double a = 0.0, e = 0.0;
  for (int b = threadIdx.x; b < n_beliefs; b += blockDim.x) {  // fixed assignment -> deterministic sums
    const double2 c = contrib[(int64_t)site * n_beliefs + b];
    a += c.x;
    e += c.y;
  }
  sa[threadIdx.x] = a;
  se[threadIdx.x] = e;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) { sa[threadIdx.x] += sa[threadIdx.x + s2]; se[threadIdx.x] += se[threadIdx.x + s2]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out3[3 * site + 0] = sa[0];
    out3[3 * site + 1] = se[0];
    out3[3 * site + 2] = sa[0] - se[0];
  }
}

constexpr int kLdsBigThreads = 512;   // the LDS instances where the launch's largest belief / sepset has more than 32 variables
constexpr int kWsThreads = 1024;   // the workspace instances of free_energy / residual_kldiv!: sixteen wavefronts per belief / message
int64_t free_energy_ws_doubles(int m) { return (int64_t)m * ((2 * m + 1) | 1) + m; }

void launch_free_energy(const double* pool, int64_t pool_stride, const double* fpool, int64_t fpool_stride,
                        const int64_t* d_boff, const int32_t* d_dim, int n_clusters, int n_beliefs, int max_dim, int bs16,
                        int fast_p, double* d_contrib, double* d_out3, int32_t* d_info, int n_sites, hipStream_t st,
                        const int32_t* d_big_idx, int n_big, double* d_ws) {
  const int mm = max_dim < 1 ? 1 : (max_dim > kFreeEnergyLdsMaxDim ? kFreeEnergyLdsMaxDim : max_dim);
  // columns of J_t per pass: all of them while m x (2m + 1) + m doubles fit in 150 KB of LDS (m <= 96), else a block
  const size_t cap = 150 * 1024 / sizeof(double);
  int kb = mm;
  while (kb > 1 && (size_t)mm * (size_t)((mm + kb + 1) | 1) + (size_t)mm > cap) --kb;
  const size_t ldsb = sizeof(double) * ((size_t)mm * (size_t)((mm + kb + 1) | 1) + (size_t)mm);
  allow_large_lds(reinterpret_cast<const void*>(free_energy_kernel<false>), ldsb);
  // (one wavefront per belief while the beliefs are small -- a network's 100 000 clusters of a dozen variables --, eight from
  // 33 variables on: the elimination of a 96-variable belief is 95 x 193 entries a pivot)
  hipLaunchKernelGGL(free_energy_kernel<false>, dim3(n_beliefs, n_sites), dim3(mm > 32 ? kLdsBigThreads : kWave), ldsb, st, pool, pool_stride, fpool,
                     fpool_stride, d_boff, d_dim, n_clusters, n_beliefs, bs16, fast_p, kb,
                     reinterpret_cast<double2*>(d_contrib), d_info, (const int32_t*)nullptr, kFreeEnergyLdsMaxDim,
                     (double*)nullptr, (int64_t)0);
  if (n_big > 0)   // beliefs above kFreeEnergyLdsMaxDim variables: n_big * n_sites slabs of free_energy_ws_doubles(max_dim)
    hipLaunchKernelGGL(free_energy_kernel<true>, dim3(n_big, n_sites), dim3(kWsThreads), 0, st, pool, pool_stride, fpool,
                       fpool_stride, d_boff, d_dim, n_clusters, n_beliefs, bs16, fast_p, max_dim,
                       reinterpret_cast<double2*>(d_contrib), d_info, d_big_idx, kFreeEnergyLdsMaxDim, d_ws,
                       free_energy_ws_doubles(max_dim));
  hipLaunchKernelGGL(free_energy_reduce_kernel, dim3(n_sites), dim3(256), 0, st,
                     reinterpret_cast<const double2*>(d_contrib), n_beliefs, d_out3);
}

static int grid_for(int64_t n, int n_sites) {
  int64_t g = (n + 255) / 256;
  const int64_t cap = n_sites >= 8 ? 256 : 2048;  // ~8 blocks per CU in total
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---- residual_kldiv! (src/beliefs.jl:1060-1075): one wavefront per message of a level ---------------------
// Runs right after the level's message kernels: the sepset holds the message just sent (J0, h0), the residual
// (dJ, dh) = after - before, so the belief before is (J1, h1) = (J0 - dJ, h0 - dh).
//   kl = ( -tr(J0^-1 dJ) + (mu1-mu0)' J1 (mu1-mu0) + logdet J0 - logdet J1 ) / 2
// If J0 or J1 is not positive definite nothing is written (the reference returns false and leaves kldiv and
// the flag alone).  Messages that did not run (failed / downstream of a failure) are skipped.
// WS: the entries listed in big_ent (sepsets of more than kKlLdsMaxS variables), the two systems in a workspace slab in
// global memory; the LDS instance skips those
template <bool WS>
__global__ __launch_bounds__(WS ? 1024 : 512) void residual_kldiv_kernel(DevState S, const Entry* __restrict__ entries, int e0,
                                                            double* __restrict__ kldiv, int32_t* __restrict__ klflags,
                                                            unsigned long long stop_below,
                                                            const int32_t* __restrict__ big_ent, double* __restrict__ ws,
                                                            int64_t ws_stride) {
  const int lane = threadIdx.x, nthr = blockDim.x, site = blockIdx.y;
  if ((S.fail[site] >> kInfoBits) < stop_below) return;
  const int msg = entries[WS ? big_ent[blockIdx.x] : e0 + (int)blockIdx.x].msg;
  const MsgDesc m = S.msgs[msg];
  const int s = m.s;
  if (s == 0) return;  // empty message: calibrated from birth
  if (!WS && s > kKlLdsMaxS) return;
  if (S.status[(int64_t)site * S.n_msgs + msg] != 0) return;
  if (S.poison[(int64_t)site * S.n_clusters + m.from_b] || S.poison[(int64_t)site * S.n_clusters + m.to_b]) return;
  const double* __restrict__ sep = S.pool + (int64_t)site * S.pool_stride + m.sep_off;
  const double* __restrict__ res = S.rpool + (int64_t)site * S.rpool_stride + m.res_off;
  const bool packed = S.bs16 && bs16::applies(s, S.fast_p);
  const int fp = S.fast_p;
  auto Jm = [&](int i, int j) {  // Symmetric(J0): upper triangle
    return packed ? sep[bs16::J_off(s, i, j, fp)] : sep[(i <= j) ? i + (int64_t)j * s : j + (int64_t)i * s];
  };
  auto dJ = [&](int i, int j) { return packed ? res[bs16::J_off(s, i, j, fp)] : res[i + (int64_t)j * s]; };
  auto J1u = [&](int i, int j) {  // Symmetric(J0 .- dJ)
    const int a = i <= j ? i : j, b = i <= j ? j : i;
    return packed ? sep[bs16::J_off(s, a, b, fp)] - res[bs16::J_off(s, a, b, fp)]
                  : sep[a + (int64_t)b * s] - res[a + (int64_t)b * s];
  };
  auto hm = [&](int i) { return packed ? sep[bs16::h_off(s, i, fp)] : sep[(int64_t)s * s + i]; };
  auto dh = [&](int i) { return packed ? res[bs16::h_off(s, i, fp)] : res[(int64_t)s * s + i]; };

  const int nc0 = 2 * s + 1, ld0 = nc0 | 1;
  double* W = WS ? ws + ((int64_t)blockIdx.x + (int64_t)gridDim.x * blockIdx.y) * ws_stride : lds;
  double* vec = W + (size_t)s * ld0;  // mu0, later mu1 - mu0
  for (int idx = lane; idx < s * s; idx += nthr) {
    const int j = idx / s, i = idx - j * s;
    W[i * ld0 + j] = Jm(i, j);
    W[i * ld0 + s + j] = dJ(i, j);
  }
  for (int i = lane; i < s; i += nthr) W[i * ld0 + 2 * s] = hm(i);
  __syncthreads();
  double logdet0, logdet1;
  if (!gauss_jordan_spd<WS>(W, s, nc0, ld0, lane, logdet0)) return;
  double tr = 0.0;
  for (int i = lane; i < s; i += nthr) {
    tr += W[i * ld0 + s + i];
    vec[i] = W[i * ld0 + 2 * s];
  }
  __syncthreads();
  const int nc1 = s + 1, ld1 = nc1 | 1;
  for (int idx = lane; idx < s * s; idx += nthr) {
    const int j = idx / s, i = idx - j * s;
    W[i * ld1 + j] = J1u(i, j);
  }
  for (int i = lane; i < s; i += nthr) W[i * ld1 + s] = hm(i) - dh(i);
  __syncthreads();
  if (!gauss_jordan_spd<WS>(W, s, nc1, ld1, lane, logdet1)) return;
  for (int i = lane; i < s; i += nthr) vec[i] = W[i * ld1 + s] - vec[i];
  __syncthreads();
  double quad = 0.0;
  for (int idx = lane; idx < s * s; idx += nthr) {
    const int j = idx / s, i = idx - j * s;
    quad += vec[i] * J1u(i, j) * vec[j];
  }
  const double acc = workgroup_sum(quad - tr, lane);
  if (lane == 0) {
    const double kl = 0.5 * (acc + logdet0 - logdet1);
    kldiv[(int64_t)site * S.n_msgs + msg] = kl;
    klflags[(int64_t)site * S.n_msgs + msg] = fabs(kl) <= S.atol ? 1 : 0;  // iscalibrated_kl! (src/beliefs.jl:1014-1016)
  }
}

int64_t kldiv_ws_doubles(int s) { return (int64_t)s * ((2 * s + 1) | 1) + s; }

void launch_residual_kldiv(const DevState& S, const Entry* d_entries, int e0, int n_entries, int max_s, double* d_kldiv,
                           int32_t* d_klflags, int n_sites, unsigned long long stop_below, hipStream_t st,
                           const int32_t* d_big_ent, int n_big, double* d_ws) {
  if (n_entries <= 0 || max_s <= 0) return;
  const int ms = max_s > kKlLdsMaxS ? kKlLdsMaxS : max_s;   // (the LDS a launch asks for: this level's largest sepset that fits)
  const size_t ldsb = sizeof(double) * (size_t)kldiv_ws_doubles(ms);
  allow_large_lds(reinterpret_cast<const void*>(residual_kldiv_kernel<false>), ldsb);
  hipLaunchKernelGGL(residual_kldiv_kernel<false>, dim3(n_entries, n_sites), dim3(ms > 32 ? kLdsBigThreads : kWave), ldsb, st, S, d_entries, e0, d_kldiv,
                     d_klflags, stop_below, (const int32_t*)nullptr, (double*)nullptr, (int64_t)0);
  if (n_big > 0)   // sepsets above kKlLdsMaxS variables: n_big * n_sites slabs of kldiv_ws_doubles(max_s)
    hipLaunchKernelGGL(residual_kldiv_kernel<true>, dim3(n_big, n_sites), dim3(kWsThreads), 0, st, S, d_entries, e0, d_kldiv,
                       d_klflags, stop_below, d_big_ent, d_ws, kldiv_ws_doubles(max_s));
}

// ---- regularizebeliefs_bycluster! (src/clustergraphbeliefs.jl:235-275) ------------------------------------
// Pass 1, one wavefront per (cluster, site): eps = max(eps(T), max|J|) BEFORE any edit (only the cluster itself
// edits its J), then for every incident sepset with a non-empty scope: J[i,i] += eps at the sepset's positions.
// Pass 2, one thread per (sepset, site): diag(J_sepset) += eps of its two clusters, lower cluster index first
// (the order of the reference's loop over labels).  The graphical model (product of cluster beliefs over
// product of sepset beliefs) is unchanged.  Plain layout only (the engine converts before launching).
__global__ __launch_bounds__(64) void regularize_cluster_kernel(double* __restrict__ pool, int64_t pool_stride,
                                                                const int64_t* __restrict__ boff,
                                                                const int32_t* __restrict__ dim,
                                                                const int32_t* __restrict__ nb_off,
                                                                const int32_t* __restrict__ nb_msg,
                                                                const MsgDesc* __restrict__ msgs,
                                                                const int32_t* __restrict__ idx,
                                                                double* __restrict__ eps_out, int n_clusters) {
  const int lane = threadIdx.x, c = blockIdx.x, site = blockIdx.y;
  const int m = dim[c];
  double* __restrict__ J = pool + (int64_t)site * pool_stride + boff[c];
  double mx = PGBP_EPS;
  for (int i = lane; i < m * m; i += kWave) mx = fmax(mx, fabs(J[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  if (lane == 0) eps_out[(int64_t)site * n_clusters + c] = mx;
  __syncthreads();
  for (int q = nb_off[c]; q < nb_off[c + 1]; ++q) {
    // nb_msg[q]: the message this cluster SENDS through the sepset: keep_map = scopeindex(sepset, cluster)
    const MsgDesc md = msgs[nb_msg[q]];
    for (int t = lane; t < md.s; t += kWave) {
      const int i = idx[md.keep_map + t];
      J[i + (int64_t)i * m] += mx;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void regularize_sepset_kernel(double* __restrict__ pool, int64_t pool_stride,
                                                                const int64_t* __restrict__ boff,
                                                                const int32_t* __restrict__ dim,
                                                                const int32_t* __restrict__ sepcl,
                                                                const double* __restrict__ eps, int n_clusters,
                                                                int n_sepsets) {
  const int site = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_sepsets; k += gridDim.x * blockDim.x) {
    const int s = dim[n_clusters + k];
    if (s == 0) continue;
    const int a = sepcl[2 * k], b = sepcl[2 * k + 1];
    const double e1 = eps[(int64_t)site * n_clusters + (a < b ? a : b)];
    const double e2 = eps[(int64_t)site * n_clusters + (a < b ? b : a)];
    double* __restrict__ J = pool + (int64_t)site * pool_stride + boff[n_clusters + k];
    for (int i = 0; i < s; ++i) {
      double v = J[i + (int64_t)i * s];
      v += e1;
      v += e2;
      J[i + (int64_t)i * s] = v;
    }
  }
}

void launch_regularize_bycluster(double* pool, int64_t pool_stride, const int64_t* d_boff, const int32_t* d_dim,
                                 const int32_t* d_nb_off, const int32_t* d_nb_msg, const MsgDesc* d_msgs,
                                 const int32_t* d_idx, const int32_t* d_sepcl, double* d_eps, int n_clusters,
                                 int n_sepsets, int n_sites, hipStream_t st) {
  if (n_clusters <= 0) return;
  hipLaunchKernelGGL(regularize_cluster_kernel, dim3(n_clusters, n_sites), dim3(kWave), 0, st, pool, pool_stride, d_boff,
                     d_dim, d_nb_off, d_nb_msg, d_msgs, d_idx, d_eps, n_clusters);
  if (n_sepsets <= 0) return;
  hipLaunchKernelGGL(regularize_sepset_kernel, dim3(grid_for(n_sepsets, n_sites), n_sites), dim3(256), 0, st, pool,
                     pool_stride, d_boff, d_dim, d_sepcl, d_eps, n_clusters, n_sepsets);
}

// ---- record gather/scatter between the ABI's packed layout and the padded device records
__global__ void records_kernel(const double* __restrict__ src, int64_t src_stride,
                               const int64_t* __restrict__ src_off, double* __restrict__ dst, int64_t dst_stride,
                               const int64_t* __restrict__ dst_off, const int64_t* __restrict__ len_off,
                               int n_records) {
  const int site = blockIdx.y;
  for (int r = blockIdx.x; r < n_records; r += gridDim.x) {
    const int64_t len = len_off[r + 1] - len_off[r];
    const double* __restrict__ s = src + (int64_t)site * src_stride + src_off[r];
    double* __restrict__ d = dst + (int64_t)site * dst_stride + dst_off[r];
    for (int64_t t = threadIdx.x; t < len; t += blockDim.x) d[t] = s[t];
  }
}

void launch_records(const double* src, int64_t src_stride, const int64_t* d_src_off, double* dst, int64_t dst_stride,
                    const int64_t* d_dst_off, const int64_t* d_len_off, int n_records, int n_sites, hipStream_t st) {
  if (n_records <= 0) return;
  const int gx = n_records < 65535 ? n_records : 65535;
  hipLaunchKernelGGL(records_kernel, dim3(gx, n_sites), dim3(256), 0, st, src, src_stride, d_src_off, dst,
                     dst_stride, d_dst_off, d_len_off, n_records);
}

__global__ void copy_strided_kernel(const double2* __restrict__ src, int64_t src_stride2, double2* __restrict__ dst,
                                    int64_t dst_stride2, int64_t n2) {
  const int site = blockIdx.y;
  const double2* __restrict__ s = src + (int64_t)site * src_stride2;
  double2* __restrict__ d = dst + (int64_t)site * dst_stride2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = s[i];
}

__global__ void zero_strided_kernel(double2* __restrict__ dst, int64_t dst_stride2, int64_t n2) {
  const int site = blockIdx.y;
  double2* __restrict__ d = dst + (int64_t)site * dst_stride2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = make_double2(0.0, 0.0);
}

// Record-aware copy (reset from factors in the BS16 layout): a packed record fills only the first ~55 % of its slot
// (577 of 1057 doubles for a 32-dim belief); one wavefront per record copies the part in use, in 16-byte pieces
// (slots start on 128-byte lines; the tail beyond the record is padding either way).
__global__ __launch_bounds__(256) void copy_records_kernel(const double* __restrict__ src, int64_t src_stride,
                                                           double* __restrict__ dst, int64_t dst_stride,
                                                           const int64_t* __restrict__ boff,
                                                           const int32_t* __restrict__ dim, int n_records, int bs, int fp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, site = blockIdx.y;
  for (int r = blockIdx.x * 4 + wave; r < n_records; r += gridDim.x * 4) {
    const int m = dim[r];
    const int len = (bs && bs16::applies(m, fp)) ? bs16::rec_len(m, fp) : m * m + m + 1;
    const double2* __restrict__ s2 = reinterpret_cast<const double2*>(src + (int64_t)site * src_stride + boff[r]);
    double2* __restrict__ d2 = reinterpret_cast<double2*>(dst + (int64_t)site * dst_stride + boff[r]);
    for (int t = lane; t < (len + 1) / 2; t += kWave) d2[t] = s2[t];
  }
}

void launch_copy_records(const double* src, int64_t src_stride, double* dst, int64_t dst_stride, const int64_t* d_boff,
                         const int32_t* d_dim, int n_records, int bs16, int fast_p, int n_sites, hipStream_t st) {
  if (n_records <= 0) return;
  const int gx = std::min((n_records + 3) / 4, 16384);
  hipLaunchKernelGGL(copy_records_kernel, dim3(gx, n_sites), dim3(256), 0, st, src, src_stride, dst, dst_stride, d_boff,
                     d_dim, n_records, bs16, fast_p);
}

// ---- the exchange buffer of a cut cluster graph (pgbp_pack_beliefs / pgbp_unpack_beliefs): the records of a list of
// beliefs of one site, back to back in the order of the list.  One workgroup per record (grid-stride), 8 bytes per thread
// and step: the records are a few hundred bytes to a few KB each and the list is a boundary, not the graph.
__global__ __launch_bounds__(256) void pack_records_kernel(double* __restrict__ pool_site, const int64_t* __restrict__ rec_off,
                                                           const int64_t* __restrict__ buf_off, int n,
                                                           double* __restrict__ buf, int to_buf) {
  for (int r = blockIdx.x; r < n; r += gridDim.x) {
    double* __restrict__ rec = pool_site + rec_off[r];
    double* __restrict__ b = buf + buf_off[r];
    const int64_t len = buf_off[r + 1] - buf_off[r];
    for (int64_t t = threadIdx.x; t < len; t += blockDim.x) {
      if (to_buf) b[t] = rec[t];
      else rec[t] = b[t];
    }
  }
}

void launch_pack_records(double* pool_site, const int64_t* d_rec_off, const int64_t* d_buf_off, int n, double* d_buf, int to_buf,
                         hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(pack_records_kernel, dim3(std::min(n, 4096)), dim3(256), 0, st, pool_site, d_rec_off, d_buf_off, n, d_buf,
                     to_buf);
}

// ---- site-minor layout (univariate batches) ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void site_minor_kernel(double* __restrict__ plain, int64_t plain_stride,
                                                         double* __restrict__ sm, const int64_t* __restrict__ off,
                                                         const int64_t* __restrict__ poff, int n_records, int n_sites,
                                                         int to_sm) {
  const int site = blockIdx.y * blockDim.x + threadIdx.x;
  if (site >= n_sites) return;
  for (int r = blockIdx.x; r < n_records; r += gridDim.x) {
    const int64_t p0 = poff[r], len = poff[r + 1] - p0;
    double* __restrict__ rec = plain + (int64_t)site * plain_stride + off[r];
    for (int64_t t = 0; t < len; ++t) {
      if (to_sm) sm[(p0 + t) * sm_row(n_sites) + site] = rec[t];
      else rec[t] = sm[(p0 + t) * sm_row(n_sites) + site];
    }
  }
}

void launch_site_minor(double* plain, int64_t plain_stride, double* sm, const int64_t* d_off, const int64_t* d_poff,
                       int n_records, int n_sites, int to_sm, hipStream_t st) {
  if (n_records <= 0) return;
  const int bs = n_sites >= 256 ? 256 : 64;
  const int gx = n_records < 16384 ? n_records : 16384;
  hipLaunchKernelGGL(site_minor_kernel, dim3(gx, (n_sites + bs - 1) / bs), dim3(bs), 0, st, plain, plain_stride, sm, d_off,
                     d_poff, n_records, n_sites, to_sm);
}

// integratebelief(h, J, g) (src/beliefupdates.jl:187-200) for m <= 2, site-minor layout, one thread per site
__global__ __launch_bounds__(256) void integrate_sm_kernel(const double* __restrict__ pool, int64_t p0, int m,
                                                           double* __restrict__ mu, int mu_stride,
                                                           double* __restrict__ norm, int32_t* __restrict__ info,
                                                           int n_sites) {
  const int site = blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= n_sites) return;
  const int64_t ns = sm_row(n_sites);
  auto E = [&](int t) { return pool[(p0 + t) * ns + site]; };
  double g = E(m * m + m);
  int bad = 0;
  double m0 = 0.0, m1 = 0.0, nrm = g;
  if (m == 1) {
    const double J = E(0), h = E(1);
    if (J == 0.0 && h == 0.0) { m0 = INFINITY; }               // iszero(h) && iszero(J): (:189-191)
    else if (!(J > 0.0)) bad = 1;
    else { m0 = h / J; nrm = g + 0.5 * (PGBP_LOG2PI - log(J) + h * m0); }
  } else if (m == 2) {
    const double a = E(0), b = E(2), d = E(3), h0 = E(4), h1 = E(5);  // Symmetric(J): upper triangle
    if (a == 0.0 && E(1) == 0.0 && b == 0.0 && d == 0.0 && h0 == 0.0 && h1 == 0.0) { m0 = m1 = INFINITY; }
    else if (!(a > 0.0)) bad = 1;
    else {
      const double s = d - b * b / a;
      if (!(s > 0.0)) bad = 2;
      else {
        m1 = (h1 - b / a * h0) / s;
        m0 = (h0 - b * m1) / a;
        nrm = g + 0.5 * (2.0 * PGBP_LOG2PI - (log(a) + log(s)) + h0 * m0 + h1 * m1);
      }
    }
  }
  if (mu) {
    if (m >= 1) mu[(int64_t)site * mu_stride] = m0;
    if (m >= 2) mu[(int64_t)site * mu_stride + 1] = m1;
  }
  norm[site] = bad ? NAN : nrm;
  if (info) info[site] = bad;
}

void launch_integrate_sm(const double* pool_sm, int64_t packed_off_b, int m, double* d_mu, int mu_stride, double* d_norm,
                         int32_t* d_info, int n_sites, hipStream_t st) {
  hipLaunchKernelGGL(integrate_sm_kernel, dim3((n_sites + 255) / 256), dim3(256), 0, st, pool_sm, packed_off_b, m, d_mu,
                     mu_stride, d_norm, d_info, n_sites);
}

// assignfactors! for a univariate BM on a tree (formulas: bm_tree_fill_kernel below with p = 1), site-minor layout
__global__ __launch_bounds__(256) void bm_tree_fill_uni_sm_kernel(double* __restrict__ pool, double* __restrict__ fpool,
                                                                  const int64_t* __restrict__ poff,
                                                                  const int32_t* __restrict__ dim,
                                                                  const int32_t* __restrict__ kind,
                                                                  const double* __restrict__ length,
                                                                  const int32_t* __restrict__ row,
                                                                  const double* __restrict__ data, int n_rows,
                                                                  const double* __restrict__ Rinv_all,
                                                                  const double* __restrict__ logdetR_all,
                                                                  const double* __restrict__ mu_all, int per_site,
                                                                  int n_clusters, int n_sites) {
  const int site = blockIdx.y * blockDim.x + threadIdx.x;
  if (site >= n_sites) return;
  const int64_t ns = sm_row(n_sites);
  const double rinv = Rinv_all[per_site ? site : 0], mu = mu_all[per_site ? site : 0];
  const double g_base = -0.5 * (PGBP_LOG2PI + logdetR_all[per_site ? site : 0]);
  for (int c = blockIdx.x; c < n_clusters; c += gridDim.x) {
    const int k = kind[c], m = dim[c];
    const int64_t p0 = poff[c];
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    int len = m * m + m + 1;
    if (k >= 0) {
      const double j = rinv / length[c];
      double g = g_base - 0.5 * log(length[c]);
      double x = 0.0;
      if (k >= 1) {
        x = (k >= 2) ? data[(int64_t)row[c] * ns + site] : mu;   // (tip data as [row][site]: a wavefront reads one line, not 64)
        if (k == 3) x -= mu;
        g -= 0.5 * j * x * x;
      }
      if (k == 0) { v[0] = j; v[1] = -j; v[2] = -j; v[3] = j; v[6] = g; }
      else if (k <= 2) { v[0] = j; v[1] = j * x; v[2] = g; }
      else { v[0] = g; }
    }
    for (int t = 0; t < len; ++t) {
      pool[(p0 + t) * ns + site] = v[t];
      if (fpool) fpool[(p0 + t) * ns + site] = v[t];
    }
  }
}

void launch_bm_tree_fill_uni_sm(double* pool_sm, double* fpool_sm, const int64_t* d_poff, const int32_t* d_dim,
                                const int32_t* d_kind, const double* d_length, const int32_t* d_row, const double* d_data,
                                int n_rows, const double* d_Rinv, const double* d_logdetR, const double* d_mu, int per_site,
                                int n_clusters, int n_sites, hipStream_t st) {
  if (n_clusters <= 0) return;
  const int bs = n_sites >= 256 ? 256 : 64;
  const int gx = n_clusters < 16384 ? n_clusters : 16384;
  hipLaunchKernelGGL(bm_tree_fill_uni_sm_kernel, dim3(gx, (n_sites + bs - 1) / bs), dim3(bs), 0, st, pool_sm, fpool_sm,
                     d_poff, d_dim, d_kind, d_length, d_row, d_data, n_rows, d_Rinv, d_logdetR, d_mu, per_site, n_clusters,
                     n_sites);
}

// all strides / offsets / counts are multiples of 2 doubles (records are padded to 16)
void launch_copy_strided(const double* src, int64_t src_stride, double* dst, int64_t dst_stride, int64_t n,
                         int n_sites, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(copy_strided_kernel, dim3(grid_for(n / 2, n_sites), n_sites), dim3(256), 0, st,
                     reinterpret_cast<const double2*>(src), src_stride / 2, reinterpret_cast<double2*>(dst),
                     dst_stride / 2, n / 2);
}

void launch_zero_strided(double* dst, int64_t dst_stride, int64_t n, int n_sites, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(zero_strided_kernel, dim3(grid_for(n / 2, n_sites), n_sites), dim3(256), 0, st,
                     reinterpret_cast<double2*>(dst), dst_stride / 2, n / 2);
}

// assignfactors! for MvFullBrownianMotion on a tree, complete data, fixed root (include/pgbp.h: pgbp_bm_tree).
//   factor_treeedge (src/evomodels/homogeneousbrownianmotion.jl:262-282): J = [j -j; -j j], j = R^-1 / t, h = 0,
//     g = g0 - p log(t) / 2,  g0 = -(p log 2pi + log det R) / 2
//   absorbevidence! (src/beliefupdates.jl:210-231) on the child's (tip data y) or the parent's (fixed root mu)
//     variables: g += h_a'y - y'J_aa y / 2, h_k -= J_ka y, J <- J_kk
__global__ __launch_bounds__(256) void bm_tree_fill_kernel(double* __restrict__ pool, int64_t pool_stride,
                                                           double* __restrict__ fpool, int64_t fpool_stride,
                                                           const int64_t* __restrict__ boff,
                                                           const int32_t* __restrict__ dim,
                                                           const int32_t* __restrict__ kind,
                                                           const double* __restrict__ length,
                                                           const int32_t* __restrict__ row,
                                                           const double* __restrict__ data, int n_rows, int p,
                                                           const double* __restrict__ Rinv_all,
                                                           const double* __restrict__ logdetR_all,
                                                           const double* __restrict__ mu_all, int per_site, int bs,
                                                           int fp, int n_clusters) {
  __shared__ double v[PGBP_MAX_DIM];   // the vector absorbed (y - or mu): R^-1 v / t
  __shared__ double jv[PGBP_MAX_DIM];
  const int site = blockIdx.y;
  const double* __restrict__ Rinv = Rinv_all + (per_site ? (int64_t)site * p * p : 0);
  const double* __restrict__ mu = mu_all + (per_site ? (int64_t)site * p : 0);
  const double logdetR = logdetR_all[per_site ? site : 0];
  for (int c = blockIdx.x; c < n_clusters; c += gridDim.x) {
    const int k = kind[c], m = dim[c];
    double* __restrict__ rec = pool + (int64_t)site * pool_stride + boff[c];
    double* __restrict__ frec = fpool + (int64_t)site * fpool_stride + boff[c];
    const bool packed = bs && bs16::applies(m, fp);
    const int len = packed ? bs16::rec_len(m, fp) : m * m + m + 1;
    __syncthreads();
    if (k < 0) {  // no factor: the constant function 1
      for (int t = threadIdx.x; t < len; t += blockDim.x) { rec[t] = 0.0; frec[t] = 0.0; }
      continue;
    }
    const double it = 1.0 / length[c];
    double g = -0.5 * ((double)p * PGBP_LOG2PI + logdetR) - 0.5 * (double)p * log(length[c]);
    // absorbed vector(s): kinds 1 (mu on the parent), 2 (y on the child), 3 (y - mu: both)
    if (k >= 1) {
      const double* __restrict__ y = (k >= 2) ? data + ((int64_t)site * n_rows + row[c]) * p : mu;
      for (int t = threadIdx.x; t < p; t += blockDim.x) v[t] = (k == 3) ? y[t] - mu[t] : y[t];
      __syncthreads();
      for (int t = threadIdx.x; t < p; t += blockDim.x) {
        double acc = 0.0;
        for (int u = 0; u < p; ++u) acc += Rinv[t + (int64_t)u * p] * v[u];
        jv[t] = acc * it;
      }
      __syncthreads();
      double q = 0.0;
      for (int u = 0; u < p; ++u) q += jv[u] * v[u];
      g -= 0.5 * q;
    }
    // J: kind 0: 2p x 2p [j -j; -j j]; kinds 1, 2: p x p block j; kind 3: nothing
    if (k <= 2) {
      for (int idx = threadIdx.x; idx < m * m; idx += blockDim.x) {
        const int cc = idx / m, rr = idx - cc * m;
        const double j = Rinv[(rr % p) + (int64_t)(cc % p) * p] * it;
        const double val = ((rr < p) == (cc < p)) ? j : -j;   // m == p: always +j
        if (packed) {
          if (bs16::canonical(m, rr, cc, fp)) { rec[bs16::J_off(m, rr, cc, fp)] = val; frec[bs16::J_off(m, rr, cc, fp)] = val; }
        } else {
          rec[idx] = val; frec[idx] = val;
        }
      }
      for (int t = threadIdx.x; t < m; t += blockDim.x) {
        // h_k -= J_ka v with J_ka = -j: h = +j v on the kept block (kinds 1, 2); 0 for kind 0
        const double hv = (k == 0) ? 0.0 : jv[t];
        const int o = packed ? bs16::h_off(m, t, fp) : m * m + t;
        rec[o] = hv; frec[o] = hv;
      }
    }
    if (threadIdx.x == 0) {
      const int o = packed ? bs16::g_off(m, fp) : m * m + m;
      rec[o] = g; frec[o] = g;
    }
  }
}

void launch_bm_tree_fill(double* pool, int64_t pool_stride, double* fpool, int64_t fpool_stride, const int64_t* d_boff,
                         const int32_t* d_dim, const int32_t* d_kind, const double* d_length, const int32_t* d_row,
                         const double* d_data, int n_rows, int p, const double* d_Rinv, const double* d_logdetR,
                         const double* d_mu, int per_site, int bs16, int fast_p, int n_clusters, int n_sites,
                         hipStream_t st) {
  if (n_clusters <= 0) return;
  const int gx = n_clusters < 65535 ? n_clusters : 65535;
  hipLaunchKernelGGL(bm_tree_fill_kernel, dim3(gx, n_sites), dim3(256), 0, st, pool, pool_stride, fpool, fpool_stride,
                     d_boff, d_dim, d_kind, d_length, d_row, d_data, n_rows, p, d_Rinv, d_logdetR, d_mu, per_site, bs16,
                     fast_p, n_clusters);
}

// init_messagecalibrationflags_reset! (src/beliefs.jl:973-979): empty messages stay calibrated
__global__ void reset_flags_kernel(const MsgDesc* __restrict__ msgs, int32_t* __restrict__ flags,
                                   int32_t* __restrict__ klflags, double* __restrict__ kldiv, int n_msgs,
                                   int reset_kl) {
  const int site = blockIdx.y;
  for (int d = blockIdx.x * blockDim.x + threadIdx.x; d < n_msgs; d += gridDim.x * blockDim.x) {
    // reset_kl: bit 0 = the KL divergences of non-empty messages back to -1, bit 1 = the KL flags (and the empty messages'
    // divergences) are written at all (the engine leaves them alone while nothing has touched them since the last reset)
    const bool empty = msgs[d].s == 0;
    flags[(int64_t)site * n_msgs + d] = empty ? 1 : 0;
    if (reset_kl & 2) {
      klflags[(int64_t)site * n_msgs + d] = empty ? 1 : 0;
      if (empty) kldiv[(int64_t)site * n_msgs + d] = 0.0;
    }
    if (!empty && (reset_kl & 1)) kldiv[(int64_t)site * n_msgs + d] = -1.0;
  }
}

// the same on the site-minor arrays ([message][site]): threads = sites
__global__ __launch_bounds__(256) void reset_flags_sm_kernel(const MsgDesc* __restrict__ msgs, int32_t* __restrict__ flags,
                                                             int32_t* __restrict__ klflags, double* __restrict__ kldiv,
                                                             int n_msgs, int n_sites, int reset_kl) {
  const int site = blockIdx.y * blockDim.x + threadIdx.x;
  if (site >= n_sites) return;
  for (int d = blockIdx.x; d < n_msgs; d += gridDim.x) {
    const bool empty = msgs[d].s == 0;
    const int64_t o = (int64_t)d * sm_row(n_sites) + site;
    __builtin_nontemporal_store(empty ? 1 : 0, &flags[o]);   // (2.6 GB at cfg4's size: streamed)
    if (reset_kl & 2) {
      klflags[o] = empty ? 1 : 0;
      if (empty) kldiv[o] = 0.0;
    }
    if (!empty && (reset_kl & 1)) kldiv[o] = -1.0;
  }
}

void launch_reset_flags(const MsgDesc* msgs, int32_t* flags, int32_t* klflags, double* kldiv, int n_msgs, int n_sites,
                        int reset_kl, hipStream_t st, int sm) {
  if (n_msgs <= 0) return;
  if (sm) {
    const int bs = n_sites >= 256 ? 256 : 64;
    hipLaunchKernelGGL(reset_flags_sm_kernel, dim3(n_msgs < 4096 ? n_msgs : 4096, (n_sites + bs - 1) / bs), dim3(bs), 0, st,
                       msgs, flags, klflags, kldiv, n_msgs, n_sites, reset_kl);
    return;
  }
  hipLaunchKernelGGL(reset_flags_kernel, dim3(grid_for(n_msgs, n_sites), n_sites), dim3(256), 0, st, msgs, flags,
                     klflags, kldiv, n_msgs, reset_kl);
}

// iscalibrated_residnorm(beliefs) = all flags (src/clustergraphbeliefs.jl:168-169).
// iscal[site] is preset to non-zero by a memset node; any block that sees a false flag clears it.
__global__ __launch_bounds__(256) void reduce_flags_kernel(const int32_t* __restrict__ flags, int n_msgs,
                                                           int32_t* __restrict__ iscal) {
  const int site = blockIdx.y;
  int bad = 0;
  for (int d = blockIdx.x * blockDim.x + threadIdx.x; d < n_msgs; d += gridDim.x * blockDim.x)
    bad |= (flags[(int64_t)site * n_msgs + d] == 0);
  if (__any(bad) && (threadIdx.x & 63) == 0) iscal[site] = 0;
}

// site-minor flags ([message][site]): threads = sites, each ANDs its column
__global__ __launch_bounds__(256) void reduce_flags_sm_kernel(const int32_t* __restrict__ flags, int n_msgs, int n_sites,
                                                              int32_t* __restrict__ iscal) {
  const int site = blockIdx.y * blockDim.x + threadIdx.x;
  if (site >= n_sites) return;
  int bad = 0;
  for (int d = blockIdx.x; d < n_msgs; d += gridDim.x) bad |= (flags[(int64_t)d * sm_row(n_sites) + site] == 0);
  if (bad) iscal[site] = 0;
}

// `auto` without a host round trip per schedule tree: once a site is calibrated, a key without failure information
// (info = 0) is min-ed into its fail word; every later traversal sees a key below its stop_below and returns at once
__global__ void halt_if_calibrated_kernel(const int32_t* __restrict__ iscal, unsigned long long* __restrict__ fail,
                                          unsigned long long key, int n_sites) {
  const int site = blockIdx.x * blockDim.x + threadIdx.x;
  if (site < n_sites && iscal[site] != 0) atomicMin(&fail[site], key);
}

void launch_halt_if_calibrated(const int32_t* d_iscal, unsigned long long* d_fail, unsigned long long key, int n_sites,
                               hipStream_t st) {
  hipLaunchKernelGGL(halt_if_calibrated_kernel, dim3((n_sites + 63) / 64), dim3(64), 0, st, d_iscal, d_fail, key, n_sites);
}

__global__ void iscal_from_notcal_kernel(const int32_t* __restrict__ notcal, int32_t* __restrict__ iscal, int n_sites) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n_sites) iscal[s] = notcal[s] == 0 ? 1 : 0;
}
void launch_iscal_from_notcal(const int32_t* d_notcal, int32_t* d_iscal, int n_sites, hipStream_t st) {
  hipLaunchKernelGGL(iscal_from_notcal_kernel, dim3((n_sites + 255) / 256), dim3(256), 0, st, d_notcal, d_iscal, n_sites);
}

void launch_reduce_flags(const int32_t* flags, int n_msgs, int n_sites, int32_t* d_iscal, hipStream_t st, int sm) {
  (void)hipMemsetAsync(d_iscal, 0x01, sizeof(int32_t) * (size_t)n_sites, st);
  if (n_msgs <= 0) return;
  if (sm) {
    const int bs = n_sites >= 256 ? 256 : 64;
    hipLaunchKernelGGL(reduce_flags_sm_kernel, dim3(n_msgs < 1024 ? n_msgs : 1024, (n_sites + bs - 1) / bs), dim3(bs), 0, st,
                       flags, n_msgs, n_sites, d_iscal);
    return;
  }
  hipLaunchKernelGGL(reduce_flags_kernel, dim3(grid_for(n_msgs, n_sites), n_sites), dim3(256), 0, st, flags,
                     n_msgs, d_iscal);
}

// [n_sites][n] <-> [n][n_sites] transposition of the per-message / per-cluster word arrays (site-minor layout switch);
// T = int32_t or double; to_sm != 0: src is [site][n]
template <class T>
__global__ __launch_bounds__(256) void transpose_words_kernel(const T* __restrict__ src, T* __restrict__ dst, int n,
                                                              int n_sites, int to_sm) {
  const int site = blockIdx.y * blockDim.x + threadIdx.x;
  if (site >= n_sites) return;
  for (int d = blockIdx.x; d < n; d += gridDim.x) {
    const int64_t a = (int64_t)site * n + d, b = (int64_t)d * sm_row(n_sites) + site;
    if (to_sm) dst[b] = src[a];
    else dst[a] = src[b];
  }
}

void launch_transpose_words_i32(const int32_t* src, int32_t* dst, int n, int n_sites, int to_sm, hipStream_t st) {
  if (n <= 0) return;
  const int bs = n_sites >= 256 ? 256 : 64;
  hipLaunchKernelGGL(transpose_words_kernel<int32_t>, dim3(n < 8192 ? n : 8192, (n_sites + bs - 1) / bs), dim3(bs), 0, st, src,
                     dst, n, n_sites, to_sm);
}
void launch_transpose_words_f64(const double* src, double* dst, int n, int n_sites, int to_sm, hipStream_t st) {
  if (n <= 0) return;
  const int bs = n_sites >= 256 ? 256 : 64;
  hipLaunchKernelGGL(transpose_words_kernel<double>, dim3(n < 8192 ? n : 8192, (n_sites + bs - 1) / bs), dim3(bs), 0, st, src,
                     dst, n, n_sites, to_sm);
}

}  // namespace pgbp

#ifdef PGBP_GSTAMP
extern "C" int pgbp_debug_gstamps(unsigned int* out, unsigned int cap, unsigned int* n) {
  if (hipDeviceSynchronize() != hipSuccess) return 4;
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(pgbp::g_gstamp_n), sizeof(unsigned int)) != hipSuccess) return 1;
  const unsigned int k = *n < cap ? *n : cap;
  if (k && hipMemcpyFromSymbol(out, HIP_SYMBOL(pgbp::g_gstamp), sizeof(unsigned int) * (size_t)k * 12) != hipSuccess) return 2;
  unsigned int z = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(pgbp::g_gstamp_n), &z, sizeof(z)) != hipSuccess) return 3;
  return 0;
}
#endif
