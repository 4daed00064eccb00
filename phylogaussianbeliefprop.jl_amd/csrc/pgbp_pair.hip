// The loop mode of the register-resident small-message body on two wavefronts per task (bp_chunk_pair): see below.
// Built with -mllvm -disable-machine-licm like the other loop kernels (Makefile): the body's constants (the coefficients of
// log(), lane geometry) hoisted out of the pass loop and kept live across it cost 20 spilled registers at the 128 a wavefront
// of a 16-wavefront workgroup may hold.
#include <hip/hip_runtime.h>

#include "pgbp_small_dev.hpp"

namespace pgbp {

#ifdef PGBP_GSTAMP  // experiment builds only (tools/stamp_pair.py): clock stamps of the phases of both halves of a message
constexpr int kPStampSlots = 1 << 16, kPStampN = 8;
__device__ unsigned int g_pstamp[kPStampSlots][kPStampN + 4];
__device__ unsigned int g_pstamp_n;
#define PGBP_PST(i) do { pst[i] = (unsigned int)__builtin_amdgcn_s_memtime(); } while (0)
#define PGBP_PST_FLUSH(role, a, b, c)                                                                                   \
  do {   /* (plain stores into a slot of this wavefront's own: no atomic, nothing to wait for) */                        \
    if (lane == 0 && gridDim.x <= 64) {                                                                                 \
      const unsigned int sl = ((blockIdx.x * 16u + (threadIdx.x >> 6)) << 6) | (e & 63u);                               \
      for (int i_ = 0; i_ < kPStampN; ++i_) g_pstamp[sl][i_] = pst[i_];                                                 \
      g_pstamp[sl][kPStampN] = blockIdx.x; g_pstamp[sl][kPStampN + 1] = (role);                                         \
      g_pstamp[sl][kPStampN + 2] = (unsigned int)((a) | ((b) << 8) | ((c) << 16)); g_pstamp[sl][kPStampN + 3] = e;      \
    }                                                                                                                   \
  } while (0)
#else
#define PGBP_PST(i) do { } while (0)
#define PGBP_PST_FLUSH(role, a, b, c) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------
// LOOP MODE of the register-resident body on TWO wavefronts per task (round 4; chunks whose messages all fit small_message).
// A wavefront alone on its SIMD issues one instruction every ~ 5 clocks whatever the instruction, so the ~ 1 400 instructions
// of a small message -- decode, addresses, requests, elimination, divide!, mult!, stores, flags -- ARE a narrow pass's
// 6 800 clocks (profiles/r03_e_cfg5_joingraph_phase_stamps_narrow.txt).  Here a task has a PROVIDER (wavefronts 0 .. 7: the
// sender's rows -> elimination -> the marginal into an LDS slot; it never stores to a belief) and a CONSUMER (wavefronts
// 8 .. 15: the sepset's and the receiver's entries at the top of the pass, then divide!, mult!, every store, status, flag):
// the consumer decodes, forms its addresses and requests its operands while the provider eliminates; the provider decodes
// the NEXT pass's record and forms its addresses while the consumer stores (both in front of the barrier between two passes:
// `pend`).  The hand-over is a sequence number per pair in LDS: pub = messages published so far, ack = messages whose stores
// have been issued; a message that fails or whose sender is poisoned is published with kPubAbort set, which ends the
// task on both sides.  Arithmetic, its order and every stored value are those of small_message (launch-mode fuzz: bit-identical).
struct PairSlot {
  double row[kSmallK][kSmallK + 2];   // marginal, row a = kept variable a: J entries b < KK, h at [kSmallK] (rows of 80 bytes)
  double g;
  unsigned int pub, ack, pad[2];
};
constexpr unsigned int kPubAbort = 0x80000000u;   // in `pub`: the message of this number did not happen (it ends the task)
__device__ __forceinline__ void wave_sync_lds() {   // LDS stores of this wavefront visible to its own lanes' loads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ unsigned int lds_acquire(const unsigned int* p) {
  return (unsigned int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
}

// the provider's half of message number cnt + 1 of this pair; returns 1 when the task ends here
template <int KI, int KK>
__device__ __forceinline__ int pair_provide(const DevState& S, const GRec* __restrict__ recs, const GLoad& cur, const int site,
                                            const int lane_in, unsigned long long seq_base, const double* __restrict__ pool,
                                            PairSlot* slot, unsigned int& cnt, double& gmsg_io, const int pend,
                                            const int prev_to_b, GLoad& ahead1, GLoad& ahead2, double* stage) {
  // (an opaque copy of the lane id per message: what derives from it -- masks, LDS addresses -- is then not invariant in the
  // pass loop and does not sit in a register, or in scratch, across it)
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
#ifdef PGBP_GSTAMP
  unsigned int pst[kPStampN] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PGBP_PST(0);
  const unsigned int rv = cur.rv;
  const int next = grec_dw(rv, 15);
  const int en_msg = grec_dw(rv, 8), en_seq = grec_dw(rv, 9), from_b = grec_dw(rv, 10);
  const int dims = grec_dw(rv, 16), fl = grec_dw(rv, 17);
  const int mf = dims & 255, s = (dims >> 16) & 255, ni = (dims >> 24) & 255;
  const int k0 = (fl & 255) == 255 ? -1 : (fl & 255);
  const bool en_reuse = ((fl >> 16) & 255) != 0;
  const double* __restrict__ from = pool + grec_i64(rv, 0);
  // FOUR COLUMN GROUPS: every row of 16 lanes (a DPP row) holds the frame's rows as small_message does -- integrated variable k in
  // lane k of the row, kept variable a in lane KI + a -- but only the columns of its GROUP g = lane / 16: all KI integrated
  // columns and h (every group keeps them: the multipliers and the pivot's terms are formed where they are used), and KK / 4 of
  // the kept columns.  A pivot then updates KI - k + KK / 4 entries per lane where the one-row frame updated KI - k + KK, its row
  // still travels by one DPP row broadcast per entry, and every entry goes through the operations it went through before.
  constexpr int KQ = KK / 4;
  const int grp = lane >> 4, rl16 = lane & 15;
  const bool is_int = rl16 < KI;
  const int fi = KI == 8 ? (rl16 & 7) : (is_int ? rl16 : rl16 - KI);
  const bool kept_live = rl16 >= KI && rl16 < KI + KK && fi < s;
  const bool row_live = is_int ? fi < ni : kept_live;
  int cjv[KI], cbv[KK];
  const int q = is_int ? fi : ni + fi;
  const int pq = __shfl(cur.pb, q < kGInlPerm ? q : 0);
  const int pi = k0 >= 0 ? (q < ni ? (q < k0 ? q : q + s) : k0 + (q - ni)) : pq;
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
  unsigned int oX[KI], oY[KI], oZ[KQ];
  const int rowbase = pi * mf;
#pragma unroll
  for (int qq = 0; qq < KQ; ++qq) {   // this group's kept columns b = grp KQ + qq
    int cb = cjv[0];
#pragma unroll
    for (int b = 0; b < KK; ++b)
      if (b == grp * KQ + qq && b < s) cb = cbv[b];
    oZ[qq] = (unsigned int)(is_int ? cb + rowbase : pi + cb * mf) << 3;
  }
#pragma unroll
  for (int j = 0; j < KI; ++j) {
    const int cj = j < ni ? cjv[j] : cjv[0];
    oX[j] = (unsigned int)(pi + cj * mf) << 3;
    oY[j] = (unsigned int)(cj + rowbase) << 3;
  }
  unsigned int ofh = (unsigned int)(mf * mf + pi) << 3;
  // the sender of this message is the receiver of the task's previous one (a unary cluster passed through inside a task):
  // its rows are requested once the consumer has issued that message's stores
  const bool chained = prev_to_b >= 0 && !en_reuse && from_b == prev_to_b;
  if (pend != 0) {
#pragma unroll
    for (int qq = 0; qq < KQ; ++qq) asm volatile("" : "+v"(oZ[qq]));
#pragma unroll
    for (int j = 0; j < KI; ++j) asm volatile("" : "+v"(oX[j]), "+v"(oY[j]));
    asm volatile("" : "+v"(ofh));
    asm volatile("" : "+s"(from));
    PGBP_PST(1);
    __syncthreads();   // the previous level of this workgroup's trees
  }
  PGBP_PST(2);
  settle(ahead1);   // (the records requested ahead, on every path through the message: see settle())
  settle(ahead2);
  bool acked = false;
  if (chained) {
    while (lds_acquire(&slot->ack) != cnt) __builtin_amdgcn_s_sleep(1);
    acked = true;
  }
  int pz = 0;
  asm volatile("" : "+v"(pz));
  const int poisoned = S.poison[(int64_t)site * S.n_clusters + from_b + pz];
  double gmsg = gmsg_io;
  bool fake = false;
  double row[KI + KQ + 1];   // this lane's entries: integrated columns, its group's kept columns, h
  if (!en_reuse) {
    // The sender's record (mf <= 16: at most 273 contiguous doubles) comes in with FIVE coalesced loads of the whole wavefront
    // and goes through the pair's LDS stage; every lane then picks its row's entries from there with the offsets formed above.
    // A vector-memory instruction costs this wavefront ~ 50 clocks whatever it moves (tools/stamp_pair.py: 25 loads of 16
    // lanes = 1 250 of a pass's 2 100 clocks to the frame); an LDS read ~ 10.  Same entries, same values.
    constexpr int kStageLoads = (16 * 16 + 16 + 1 + 63) / 64;
    const int len = mf * mf + mf + 1;
    double raw[kStageLoads];
#pragma unroll
    for (int t = 0; t < kStageLoads; ++t) {
      const int i = lane + 64 * t;
      raw[t] = 0.0;
      if (64 * t < len)   // (wave-uniform: a short record issues no empty load)
        raw[t] = i < len ? ld8(from, i) : 0.0;
    }
#pragma unroll
    for (int t = 0; t < kStageLoads; ++t) {
      const int i = lane + 64 * t;
      if (64 * t < len && i < len) stage[i] = raw[t];
    }
    wave_sync_lds();
    PGBP_PST(7);
    double X[KI], Y[KI], Z[KQ], hv = 0.0;
#pragma unroll
    for (int j = 0; j < KI; ++j) {
      X[j] = 0.0;
      Y[j] = 0.0;
    }
#pragma unroll
    for (int qq = 0; qq < KQ; ++qq) Z[qq] = 0.0;
    if (row_live) {
#pragma unroll
      for (int j = 0; j < KI; ++j) X[j] = ld8o(stage, oX[j]);
      if (is_int) {
#pragma unroll
        for (int j = 0; j < KI; ++j) Y[j] = ld8o(stage, oY[j]);
      }
#pragma unroll
      for (int qq = 0; qq < KQ; ++qq) Z[qq] = ld8o(stage, oZ[qq]);
      hv = ld8o(stage, ofh);
    }
#pragma unroll
    for (int j = 0; j < KI; ++j) {
      X[j] = j < ni ? X[j] : 0.0;
      Y[j] = j < ni ? Y[j] : 0.0;
    }
#pragma unroll
    for (int qq = 0; qq < KQ; ++qq) Z[qq] = grp * KQ + qq < s ? Z[qq] : 0.0;
    gmsg = stage[mf * mf + mf];
    wave_sync_lds();   // (the stage is free again: the next message of the task may overwrite it)
    bool nz = is_int && fabs(hv) > PGBP_EPS;
#pragma unroll
    for (int j = 0; j < KI; ++j) nz |= fabs(X[j]) > PGBP_EPS;
    fake = ni == 0 || !__any(nz);
#pragma unroll
    for (int j = 0; j < KI; ++j) row[j] = (is_int && j < fi) ? Y[j] : X[j];
#pragma unroll
    for (int qq = 0; qq < KQ; ++qq) row[KI + qq] = Z[qq];
    row[KI + KQ] = hv;
  }
  PGBP_PST(3);
  const unsigned int e = cnt + 1;
  cnt = e;
  int info = 0;
  const bool poison_stop = __builtin_amdgcn_readfirstlane(poisoned) != 0;
  double mant = 1.0, quad = 0.0;
  int expo = 0;
  if (!poison_stop && !en_reuse && !fake) {
    Small4<KI, KQ>::template pivot<0, decltype(row), true>(row, ni, info, mant, expo, quad);
    info = __builtin_amdgcn_readfirstlane(info);
  }
  if (poison_stop || info != 0) {
    if (lane == 0) {
      if (!poison_stop) S.status[(int64_t)site * S.n_msgs + en_msg] = info;
      S.poison[(int64_t)site * S.n_clusters + grec_dw(rv, 11)] = 1;
      for (int qn = next; qn >= 0; qn = recs[qn].next) S.poison[(int64_t)site * S.n_clusters + recs[qn].to_b] = 1;
      if (!poison_stop)
        atomicMin(&S.fail[site], ((seq_base + (unsigned long long)(unsigned int)en_seq) << kInfoBits) |
                                     (unsigned long long)(unsigned int)info);
      __hip_atomic_store(&slot->pub, e | kPubAbort, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return 1;
  }
  if (!en_reuse) {
    if (!fake) {
      const double logdet = log_by_table(S.logtab, mant) + (double)expo * 0.69314718055994530941723212145818;
      gmsg += 0.5 * ((double)ni * PGBP_LOG2PI - logdet + quad);  // :81
    }
    gmsg_io = gmsg;
    PGBP_PST(4);
    // the slot is free once the consumer is done with the previous message (its marginal may be reused until then); the first
    // message of a task needs no look: the consumer acknowledged the previous pass's last message in front of the barrier
    if (!acked && prev_to_b >= 0)
      while (lds_acquire(&slot->ack) != e - 1) __builtin_amdgcn_s_sleep(1);
    if (kept_live) {   // (every group its own columns of the marginal's row a = fi; h from group 0)
      if constexpr (KQ == 2)
        *reinterpret_cast<double2*>(&slot->row[fi][grp * 2]) = make_double2(row[KI], row[KI + 1]);
      else
        slot->row[fi][grp] = row[KI];
      if (grp == 0) slot->row[fi][kSmallK] = row[KI + KQ];
    }
    if (lane == 0) slot->g = gmsg;
  }
  PGBP_PST(5);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the whole wavefront's LDS stores, then lane 0's word)
  if (lane == 0) __hip_atomic_store(&slot->pub, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  PGBP_PST(6);
  PGBP_PST_FLUSH(0, mf, ni, s);
  return 0;
}

// the consumer's half: divide!, mult!, stores of message number cnt + 1.  Lane (a, b) = a + KK b owns ENTRY (a, b) of the
// message (the lane grid of the in-LDS body): the sepset's block, the receiver's block and the residual are one vector-memory
// instruction each for the whole s x s block, the three h vectors one each for lanes (a, 0) -- 6 loads and 10 stores a
// message where a row per lane took 20 and 30 at ~ 50 clocks apiece.  Every entry goes through the same operations as in
// small_message (dJ = msg - sepset, receiver + dJ, the two threshold tests ANDed over the lanes).
template <int KK>
__device__ __forceinline__ int pair_consume(const DevState& S, const GLoad& cur, const int site, const int lane_in,
                                            double* __restrict__ pool, double* __restrict__ rpool,
                                            PairSlot* slot, unsigned int& cnt, const int pend, GLoad& ahead1,
                                            GLoad& ahead2) {
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
#ifdef PGBP_GSTAMP
  unsigned int pst[kPStampN] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PGBP_PST(0);
  const unsigned int rv = cur.rv;
  const int en_msg = grec_dw(rv, 8);
  const int dims = grec_dw(rv, 16), fl = grec_dw(rv, 17);
  const int mt = (dims >> 8) & 255, s = (dims >> 16) & 255;
  const int u0 = ((fl >> 8) & 255) == 255 ? -1 : ((fl >> 8) & 255);
  double* __restrict__ sep = pool + grec_i64(rv, 4);
  double* __restrict__ to = pool + grec_i64(rv, 2);
  double* __restrict__ res = rpool + grec_i64(rv, 6);
  const int a = lane & (KK - 1), b = lane / KK;   // (KK = 4: lanes 0 .. 15; KK = 8: all 64)
  const bool live = lane < KK * KK && a < s && b < s;
  const bool hlive = lane < KK && a < s;          // lanes (a, 0): the h entries
  const int ua = u0 >= 0 ? u0 + a : __shfl(cur.ub, a);   // (inline map: lane l holds up[l & 15])
  const int ub = u0 >= 0 ? u0 + b : __shfl(cur.ub, b & (kGInlUp - 1));
  unsigned int osep = (unsigned int)(a + b * s) << 3, oto = (unsigned int)(ua + ub * mt) << 3;
  unsigned int oseph = (unsigned int)(s * s + a) << 3, otoh = (unsigned int)(mt * mt + ua) << 3;
  if (pend != 0) {
    asm volatile("" : "+v"(osep), "+v"(oto), "+v"(oseph), "+v"(otoh));
    asm volatile("" : "+s"(sep), "+s"(to), "+s"(res));
    PGBP_PST(1);
    __syncthreads();
  }
  PGBP_PST(2);
  settle(ahead1);   // (the records requested ahead: waited for in front of the stores, not behind them)
  settle(ahead2);
  double psep = 0.0, pto = 0.0, pseph = 0.0, ptoh = 0.0, pre_sepg = 0.0, pre_tog = 0.0;
  if (live) {
    psep = ld8o(sep, osep);
    pto = ld8o(to, oto);
  }
  if (hlive) {
    pseph = ld8o(sep, oseph);
    ptoh = ld8o(to, otoh);
  }
  if (lane == 0) {
    pre_sepg = ld8(sep, s * s + s);
    pre_tog = ld8(to, mt * mt + mt);
  }
  double thr_h = S.thr[s], thr_J = S.thr[PGBP_MAX_DIM + 1 + s];
  const unsigned int e = cnt + 1;
  cnt = e;
  PGBP_PST(3);
  // (no s_sleep: the provider runs on another SIMD, and every 64 clocks of the hand-over are on the pass; the abort mark rides
  // in the same word: one LDS round trip, not two, between the publication and the marginal)
  unsigned int pw;
  while (((pw = lds_acquire(&slot->pub)) & ~kPubAbort) < e) {}
  PGBP_PST(4);
  if (pw == (e | kPubAbort)) {
    if (lane == 0) __hip_atomic_store(&slot->ack, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    return 1;
  }
  double msgJ = live ? slot->row[a][b] : 0.0;
  double msgh = hlive ? slot->row[a][kSmallK] : 0.0;
  double gmsg = slot->g;
  // (every operand arrived while the provider eliminated: said once, in front of the stores -- small_message)
  asm volatile("" : "+v"(psep), "+v"(pto), "+v"(msgJ), "+v"(pseph), "+v"(ptoh), "+v"(pre_sepg), "+v"(pre_tog), "+v"(gmsg),
               "+v"(thr_h), "+v"(thr_J), "+v"(msgh));
  PGBP_PST(5);
  double maxJ = 0.0, maxh = 0.0;
  // (the RECEIVER's entries first: the next pass's provider loads them, and its load waits behind these stores on their way to
  // the L2 -- the sepset and the residual are nobody's operand before the next traversal)
  double dJ = 0.0, dh = 0.0, dg = 0.0;
  if (live) {
    dJ = msgJ - psep;
    st8o(to, oto, pto + dJ);
  }
  if (hlive) {
    dh = msgh - pseph;
    st8o(to, otoh, ptoh + dh);
  }
  if (lane == 0) {
    dg = gmsg - pre_sepg;
    st8(to, mt * mt + mt, pre_tog + dg);
  }
  if (live) {
    st8o(sep, osep, msgJ);
    st8o(res, osep, dJ);
    maxJ = (dJ != dJ) ? INFINITY : fmax(maxJ, fabs(dJ));
  }
  if (hlive) {
    st8o(sep, oseph, msgh);
    st8o(res, oseph, dh);
    maxh = (dh != dh) ? INFINITY : fmax(maxh, fabs(dh));
  }
  if (lane == 0) {
    st8(sep, s * s + s, gmsg);
    S.status[(int64_t)site * S.n_msgs + en_msg] = 0;
  }
  if (S.update_resnorm) {
    const bool lane_ok = maxh <= thr_h && maxJ <= thr_J;
    const bool ok = __all(lane_ok);
    if (lane == 0) S.flags[(int64_t)site * S.n_msgs + en_msg] = ok ? 1 : 0;
  }
  // the stores are issued: the provider may request this receiver as the task's next sender, and reuse the slot
  if (lane == 0) __hip_atomic_store(&slot->ack, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  PGBP_PST(6);
  PGBP_PST_FLUSH(1, mt, 0, s);
  return 0;
}

constexpr int kPairWaves = 2 * kTailWaves;
__global__ __launch_bounds__(kPairWaves * 64) void bp_chunk_pair(DevState S, const GRec* __restrict__ recs,
                                                                 const int32_t* __restrict__ grp_recs,
                                                                 const int32_t* __restrict__ wg_off,
                                                                 unsigned long long seq_base, unsigned long long stop_below) {
  __shared__ PairSlot slots[kTailWaves];
  __shared__ double stages[kTailWaves][16 * 16 + 16 + 8];   // a provider's copy of its sender's record (mf <= kSmallI + kSmallK)
  const int site = blockIdx.y;
  if ((S.fail[site] >> kInfoBits) < stop_below) return;   // (uniform over the workgroup: no barrier is skipped by a few)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wave & (kTailWaves - 1);
  const bool provider = wave < kTailWaves;
  PairSlot* slot = &slots[pair];
  if (threadIdx.x < kTailWaves) {
    slots[threadIdx.x].pub = 0;
    slots[threadIdx.x].ack = 0;
  }
  __syncthreads();
  double* __restrict__ pool = S.pool + (int64_t)site * S.pool_stride;
  double* __restrict__ rpool = S.rpool + (int64_t)site * S.rpool_stride;
  unsigned int cnt = 0;   // messages of this pair so far: both of its wavefronts count alike
  const int g0 = wg_off[blockIdx.x], g1 = wg_off[blockIdx.x + 1];
  // slot i of pass k runs on pair (i + 3 k) mod 8 (kPassRotate: a pass of up to three tasks shares no pair with the pass before,
  // so whoever works in pass k + 1 has decoded its record and formed its addresses while pass k ran; stride 1 / 3 / 4: cfg5 join
  // graph 0.958 / 0.931 / 0.934 ms per iteration, Bethe 1.174 / 1.145 / 1.148).  The task index of pass g + 2 is requested (a scalar load)
  // and the record of pass g + 1 (whose index came in a pass ago) is requested at the top of pass g: nothing of it is waited
  // for between a pass's last store and the barrier
  auto task_of = [&](int g) { return grp_recs[(int64_t)g * kTailWaves + ((pair - kPassRotate * (g - g0)) & (kTailWaves - 1))]; };
  int ri = task_of(g0);
  int ri_next = g0 + 1 < g1 ? task_of(g0 + 1) : -1;
  GLoad cur = {0u, 0, 0};
  if (ri >= 0) cur = load_grec(recs, ri, lane);
  for (int g = g0; g < g1; ++g) {
    GLoad nxt = {0u, 0, 0};
    const int ri_next2 = g + 2 < g1 ? task_of(g + 2) : -1;
    int pend = g > g0 ? 1 : 0;
    // (a wavefront without a task in this pass goes to the barrier first -- it may be the one the pass before ended on, its
    // last stores just issued -- and requests the next pass's record behind it)
    if (ri >= 0 && ri_next >= 0) nxt = load_grec(recs, ri_next, lane);
    if (ri >= 0) {
      // the task: its messages in order, the record of the next one requested beside the current one
      double gmsg = 0.0;
      int prev_to_b = -1;
      for (;;) {
        const unsigned int rv = cur.rv;
        const int next = grec_dw(rv, 15);
        GLoad nx = cur;
        if (next >= 0) nx = load_grec(recs, next, lane);
        const int dims = grec_dw(rv, 16);
        const int s = (dims >> 16) & 255, nim = (dims >> 24) & 255;
        int done;
        if (provider) {
          if (nim <= 4 && s <= 4) done = pair_provide<4, 4>(S, recs, cur, site, lane, seq_base, pool, slot, cnt, gmsg, pend, prev_to_b, nx, nxt, stages[pair]);
          else if (nim <= 4) done = pair_provide<4, kSmallK>(S, recs, cur, site, lane, seq_base, pool, slot, cnt, gmsg, pend, prev_to_b, nx, nxt, stages[pair]);
          else if (s <= 4) done = pair_provide<kSmallI, 4>(S, recs, cur, site, lane, seq_base, pool, slot, cnt, gmsg, pend, prev_to_b, nx, nxt, stages[pair]);
          else done = pair_provide<kSmallI, kSmallK>(S, recs, cur, site, lane, seq_base, pool, slot, cnt, gmsg, pend, prev_to_b, nx, nxt, stages[pair]);
        } else {
          if (s <= 4) done = pair_consume<4>(S, cur, site, lane, pool, rpool, slot, cnt, pend, nx, nxt);
          else done = pair_consume<kSmallK>(S, cur, site, lane, pool, rpool, slot, cnt, pend, nx, nxt);
        }
        pend = 0;
        if (done || next < 0) break;
        prev_to_b = grec_dw(rv, 11);
        cur = nx;
      }
    } else {
      // (no task in this pass: straight to the barrier -- the record requested a moment ago is waited for behind it)
      if (pend) __syncthreads();
      if (ri_next >= 0) nxt = load_grec(recs, ri_next, lane);
      settle(nxt);
    }
    ri = ri_next;
    ri_next = ri_next2;
    cur = nxt;
  }
}


void launch_chunk_pair(const DevState& S, const GRec* d_recs, const int32_t* d_grp_recs, const int32_t* d_wg_off, int n_wg,
                       int n_sites, unsigned long long seq_base, unsigned long long stop_below, hipStream_t st) {
  if (n_wg <= 0) return;
  hipLaunchKernelGGL(bp_chunk_pair, dim3(n_wg, n_sites), dim3(kPairWaves * 64), 0, st, S, d_recs, d_grp_recs, d_wg_off, seq_base,
                     stop_below);
}

}  // namespace pgbp

#ifdef PGBP_GSTAMP
extern "C" int pgbp_debug_pstamps(unsigned int* out, unsigned int cap, unsigned int* n) {
  // every slot (64 workgroups x 16 wavefronts x 64 messages); unused ones are zero; cleared afterwards
  if (hipDeviceSynchronize() != hipSuccess) return 4;
  *n = pgbp::kPStampSlots;
  const unsigned int k = *n < cap ? *n : cap;
  if (k && hipMemcpyFromSymbol(out, HIP_SYMBOL(pgbp::g_pstamp), sizeof(unsigned int) * (size_t)k * 12) != hipSuccess) return 2;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(pgbp::g_pstamp)) != hipSuccess) return 3;
  if (hipMemset(p, 0, sizeof(unsigned int) * (size_t)pgbp::kPStampSlots * 12) != hipSuccess) return 3;
  return 0;
}
#endif
