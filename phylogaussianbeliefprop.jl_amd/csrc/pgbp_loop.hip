// Loop launches of the register-resident message kernel in the packed layout (gfx950, wave64, even P): the single-workgroup
// TAIL of a traversal pair and the CHUNKS of fused narrow levels (pgbp_plan.cpp: Traversal::tail_levels, build_chunks).
//
// A workgroup walks its groups of 8 records, one group = one dependent step ("pass") of the schedule.  A narrow level lasts
// as long as ONE message, and a lone wavefront issues one instruction every four cycles or so whatever the instruction: the
// 3 000 instructions a wavefront of pgbp_fast.hip's loop mode spends on a message -- record, addresses, loads, elimination,
// divide!, mult!, stores, flags -- ARE the 12 000 cycles of its pass.  Here a record has TWO wavefronts:
//   * its PROVIDER (waves 0 .. 7) turns the sender into the marginal: operands, prologue, elimination, marginal into LDS.
//     It never stores to memory, so its memory counter only ever holds loads and the operands of pass g + 1 are requested
//     in the middle of pass g (EARLY LOADS), right behind its elimination, while the other half of the record is still
//     being worked on;
//   * its CONSUMER (waves 8 .. 15) loads the sepset and the receiver block at the top of the pass, waits for the marginal,
//     and does divide!, the accumulation over the task, mult!, every store, the flags, the failure bookkeeping -- and the
//     prologue's message where there is one (it forms the prologue's delta from the same operands: bit-identical);
//   * CHAINS: what pass g + 1 needs FROM pass g -- the block a consumer of pass g has just accumulated into cluster X, when
//     X sends in pass g + 1 -- does not go through memory: the consumer leaves (block, h, g, state) in its LDS chain slot,
//     the provider (and the consumer, for a prologue's X) of pass g + 1 patches it into what it loaded
//     (planner: FEntry::pad, link_chains);
//   * a pass whose records need anything else that pass g writes (a receiver block, a sepset: the post -> pre turn of the
//     tail) is LATE: every store of pass g complete first, loads at its top.
// Critical path of a pass: chain -> elimination -> marginal -> divide! / mult! -> chain.  Two workgroup barriers per pass,
// both LDS-only; the stores of a pass are issued behind the second one, beside the next elimination.
// Same records (FEntry / FPro), same arithmetic in the same order as pgbp_fast.hip: results are bit for bit the same.
#include <hip/hip_runtime.h>

#include "pgbp_fast_dev.hpp"

namespace pgbp {

extern __shared__ double loop_lds[];

#ifdef PGBP_STAMP  // experiment builds only (tools/stamp_loop.py): clock stamps of the phases of a pass
constexpr int kLStampSlots = 1 << 16, kLStampN = 12;
__device__ unsigned int g_lstamp[kLStampSlots][kLStampN + 4];
__device__ unsigned int g_lstamp_n;
#define PGBP_LT(i) do { stv[i] = (unsigned int)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PGBP_LT(i) do { } while (0)
#endif

namespace {

// LDS slot: 2 x 2 block per lane (256 doubles, lane-major), h (16), g, status, info
constexpr int kLSlotJ = 0, kLSlotH = 256, kLSlotG = 272, kLSlotStatus = 273, kLSlotInfo = 274, kLSlotDoubles = 288;
constexpr int kLoopWaves = 2 * kTailWaves;

// sender operands of one record exactly as loaded (nothing is computed from them before the pass that uses them: a select
// or a copy of a loaded register is a wait for that load, and these loads are in flight across half a pass)
struct SndOps {
  Blk ii, ss;       // 2P sender: its integrated and kept tiles; P-dim sender: ii = its tile
  double4 t;        // 2P sender: the block of T10 this lane needs
  double2 hi, hs;   // h of the integrated / kept variables (P-dim sender: hi = its h)
  double g;
  Blk xJ, s2;       // prologue: X's tile, the sepset (X, F)
  double2 xh, s2h;
  double xg, s2g;
  int poison, poison_x;   // of the sender and of X
};

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

}  // namespace

template <int P, bool PRO>
__global__ __launch_bounds__(kLoopWaves * 64) void bp_loop16(DevState S_arg, const FEntry* __restrict__ recs_arg,
                                                             const FPro* __restrict__ pros, int ngroups_arg, int split_arg,
                                                             unsigned long long seq_base_arg,
                                                             unsigned long long stop_a_arg,
                                                             unsigned long long stop_b_arg,
                                                             const int32_t* __restrict__ wg_off) {
  constexpr int W = kTailWaves, G = P / 2;
  DevState S = S_arg;
  const FEntry* __restrict__ recs = recs_arg;
  int ngroups = ngroups_arg, split = split_arg;
  unsigned long long seq_base = seq_base_arg, stop_a = stop_a_arg, stop_b = stop_b_arg;
  {
    unsigned long long th_bits = __double_as_longlong(S.thr_h_p), tj_bits = __double_as_longlong(S.thr_J_p);
    asm("; kernel arguments resident"
        : "+s"(S.pool_stride), "+s"(S.rpool_stride), "+s"(S.n_clusters), "+s"(S.n_msgs), "+s"(S.update_resnorm),
          "+s"(th_bits), "+s"(tj_bits), "+s"(ngroups), "+s"(split), "+s"(seq_base), "+s"(stop_a), "+s"(stop_b));
    S.thr_h_p = __longlong_as_double(th_bits);
    S.thr_J_p = __longlong_as_double(tj_bits);
  }
  const int wave16 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool consumer = wave16 >= W;
  const int wave = consumer ? wave16 - W : wave16;   // the record slot of this wavefront
  const int site = blockIdx.y;
  // the lane geometry is worked out anew in every pass (lane_geometry() below): as loop invariants, the two dozen offsets
  // and masks derived from it would each hold a register across the whole loop, and the loop has none to spare
  int lane = threadIdx.x & 63;
  bool act = lane < G * G;
  int a = act ? lane % G : 0, b = act ? lane / G : 0;
  bool up = act && a <= b;
  int kidx = (b * (b + 1) / 2 + a) * 4;
  auto lane_geometry = [&]() {
    asm volatile("; lane geometry" : "+v"(lane));
    act = lane < G * G;
    a = act ? lane % G : 0;
    b = act ? lane / G : 0;
    up = act && a <= b;
    kidx = (b * (b + 1) / 2 + a) * 4;
  };
  double* __restrict__ pool = S.pool + (int64_t)site * S.pool_stride;
  double* __restrict__ rpool = S.rpool + (int64_t)site * S.rpool_stride;
  double* const marg = loop_lds + wave * kLSlotDoubles;                        // provider -> consumers: the marginal
  double* const delta = loop_lds + (W + wave) * kLSlotDoubles;                 // consumer -> its task's first one: the sepset
  double* const chain = loop_lds + (2 * W + wave) * kLSlotDoubles;             // consumer -> the next pass
  double* const col = loop_lds + 3 * W * kLSlotDoubles + wave * kColDoubles;   // private strip of the elimination

  if (wg_off) ngroups = wg_off[blockIdx.x + 1];
  int g = wg_off ? wg_off[blockIdx.x] : 0;
  FEntry en = load_record(recs + ((int64_t)g * W + wave));
  FPro pr{};
  if constexpr (PRO) pr = load_pro(pros + ((int64_t)g * W + wave));
  unsigned long long failkey = S.fail[site];

  // the sender operands of a record, straight-line: every load is issued whatever the record is (an offset that does not
  // apply points at the start of the record or of the pool: valid memory, the value is ignored)
  auto issue_snd = [&](const FEntry& q, const FPro& qp, SndOps& s) {
    const bool heavy = q.mf == 2 * P, none = q.mf == 0, itrail = q.keep0 == 0;
    const double* __restrict__ from = pool + q.from_off;
    const int o_ii = heavy && itrail ? bs16::t11(P) : 0, o_ss = heavy && !itrail ? bs16::t11(P) : 0;
    const int o_t = heavy ? bs16::t10(P) : 0;
    const int o_hi = heavy ? bs16::h2(P) + (itrail ? P : 0) : (none ? 0 : bs16::h1(P));
    const int o_hs = heavy ? bs16::h2(P) + (itrail ? 0 : P) : 0;
    const int o_g = heavy ? bs16::g2(P) : (none ? 0 : bs16::g1(P));
    const int lk = none ? 0 : kidx, la = none ? 0 : 2 * a;                  // (a constant's record is one double long)
    const int lt = none ? 0 : (itrail ? (b + G * a) : (a + G * b)) * 4;
    {   // (every lane loads: a lane below the diagonal fetches some other block of the tile, zeroed where it is used)
      const double4 v = *reinterpret_cast<const double4*>(from + o_ii + lk);
      const double4 u = *reinterpret_cast<const double4*>(from + o_ss + lk);
      s.ii = Blk{v.x, v.y, v.z, v.w};
      s.ss = Blk{u.x, u.y, u.z, u.w};
    }
    s.t = *reinterpret_cast<const double4*>(from + o_t + lt);
    s.hi = *reinterpret_cast<const double2*>(from + o_hi + la);
    s.hs = *reinterpret_cast<const double2*>(from + o_hs + la);
    s.g = from[o_g];
    s.poison = S.poison[(int64_t)site * S.n_clusters + q.from_b];
    s.xJ = Blk{0, 0, 0, 0}; s.s2 = Blk{0, 0, 0, 0};
    s.xh = make_double2(0.0, 0.0); s.s2h = make_double2(0.0, 0.0);
    s.xg = 0.0; s.s2g = 0.0;
    s.poison_x = 0;
    if constexpr (PRO) {
      const double* __restrict__ xfrom = pool + qp.from_off;   // (no prologue: offsets 0, the start of the pool)
      const double* __restrict__ sep2 = pool + qp.sep_off;
      {
        const double4 v = *reinterpret_cast<const double4*>(xfrom + kidx);
        const double4 u = *reinterpret_cast<const double4*>(sep2 + kidx);
        s.xJ = Blk{v.x, v.y, v.z, v.w};
        s.s2 = Blk{u.x, u.y, u.z, u.w};
      }
      s.xh = *reinterpret_cast<const double2*>(xfrom + bs16::h1(P) + 2 * a);
      s.s2h = *reinterpret_cast<const double2*>(sep2 + bs16::h1(P) + 2 * a);
      s.xg = xfrom[bs16::g1(P)];
      s.s2g = sep2[bs16::g1(P)];
      s.poison_x = S.poison[(int64_t)site * S.n_clusters + qp.from_b];
    }
  };
  // the next record (and its prologue) out of the vector load of lanes 0 .. 3 (4, 5)
  auto decode_next = [&](const uint4& nxv, FEntry& nx, FPro& npr) {
    unsigned int q[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      q[4 * i + 0] = (unsigned int)__builtin_amdgcn_readlane((int)nxv.x, i);
      q[4 * i + 1] = (unsigned int)__builtin_amdgcn_readlane((int)nxv.y, i);
      q[4 * i + 2] = (unsigned int)__builtin_amdgcn_readlane((int)nxv.z, i);
      q[4 * i + 3] = (unsigned int)__builtin_amdgcn_readlane((int)nxv.w, i);
    }
    __builtin_memcpy(&nx, q, sizeof(FEntry));
    if constexpr (PRO) {
      unsigned int w8[8];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        w8[4 * i + 0] = (unsigned int)__builtin_amdgcn_readlane((int)nxv.x, 4 + i);
        w8[4 * i + 1] = (unsigned int)__builtin_amdgcn_readlane((int)nxv.y, 4 + i);
        w8[4 * i + 2] = (unsigned int)__builtin_amdgcn_readlane((int)nxv.z, 4 + i);
        w8[4 * i + 3] = (unsigned int)__builtin_amdgcn_readlane((int)nxv.w, 4 + i);
      }
      __builtin_memcpy(&npr, w8, sizeof(FPro));
    }
  };

  // Two loops, one per role: what one role carries from pass to pass (the early operands) is not live in the other.
  bool first_pass = true;
  if (!consumer) {
    SndOps cs{};
    issue_snd(en, pr, cs);   // the first pass of a walk loads at its top
    for (;;) {
#ifdef PGBP_STAMP
      unsigned int stv[kLStampN] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
      PGBP_LT(0);
      lane_geometry();
      const bool has_next = g + 1 < ngroups;
      // the next record (and its prologue) as a VECTOR load, lanes 0 .. 3 (4, 5): waited for where it is decoded (without
      // a next group it stays zero: an invalid record)
      uint4 nxv = make_uint4(0, 0, 0, 0);
      auto issue_next = [&]() {
        if (has_next) {
          if (PRO && (lane & 4))
            nxv = reinterpret_cast<const uint4*>(pros + ((int64_t)(g + 1) * W + wave))[lane & 1];
          else
            nxv = reinterpret_cast<const uint4*>(recs + ((int64_t)(g + 1) * W + wave))[lane & 3];
        }
      };
      // a failure in the postorder part of the tail must stop its preorder part: at the first preorder level the fail word
      // is read again, coherently
      if (g == split && split > 0) {
        failkey = __hip_atomic_load(&S.fail[site], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        failkey = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(failkey >> 32)) << 32) |
                  (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)failkey);
      }
      const unsigned long long stop_below = g >= split ? stop_b : stop_a;
      const bool live = en.valid && !((failkey >> kInfoBits) < stop_below);   // (not stopped)
      [[maybe_unused]] const bool has_block = en.s > 0;
      [[maybe_unused]] const bool own = (en.mode & kFOwn) != 0;
      [[maybe_unused]] const bool accum = (en.mode & kFAccum) != 0;
      [[maybe_unused]] const bool recv_blk = has_block || (accum && !(en.mode & kFNoBlock));
      const bool provider = en.src_wave == wave;   // the record eliminates (the others reuse a sibling's marginal)
      [[maybe_unused]] const int first_wave = en.grp_base;
      [[maybe_unused]] const bool pro = PRO && (en.mode & kFPro) != 0;
      // (the first pass of a walk has loaded everything from memory, whatever its records say: a tail launch may start at
      // its preorder half)
      const int chain_kind = first_pass ? 0 : en.pad[0], chain_src = en.pad[1];
      FEntry nx{};
      FPro npr{};

      // ================================================================================ PROVIDER: sender -> marginal
      // state 0: nothing to do / stopped; 1: marginal available; 2: failed (not PD); 3: sender poisoned
      Blk mJ{0, 0, 0, 0};
      double mh0 = 0.0, mh1 = 0.0, gmsg = 0.0;
      int info = 0, state = (live && provider) ? 1 : 0;
      int poison_v = 0;
      Frag f;
      Blk d2{0, 0, 0, 0};   // the prologue's delta: X minus the sepset (X, F)
      double d2h0 = 0.0, d2h1 = 0.0, d2g = 0.0;
      if (state == 1) {
        poison_v = cs.poison | (pro ? cs.poison_x : 0);
        // ---- CHAIN: what this record's sender (or its prologue's X) received in the pass before, from the owner's slot
        double4 cv = make_double4(0.0, 0.0, 0.0, 0.0);
        double2 cu = make_double2(0.0, 0.0);
        double cg = 0.0;
        if (chain_kind != 0) {
          const double* src = loop_lds + (2 * W + chain_src) * kLSlotDoubles;
          cv = *reinterpret_cast<const double4*>(src + kLSlotJ + 4 * lane);
          cu = *reinterpret_cast<const double2*>(src + kLSlotH + 2 * a);
          cg = src[kLSlotG];
          if ((int)src[kLSlotStatus] != 1) poison_v = 1;   // the cluster it reads failed or was skipped in the pass before
        }
        // the prologue's delta first: its operands are dead before the sender's are unpacked
        if constexpr (PRO) {
          const bool xc = chain_kind == 2;   // X comes through the chain
          const bool sz = S.sep_zero != 0;
          const Blk xJ = xc ? Blk{cv.x, cv.y, cv.z, cv.w} : cs.xJ;
          const Blk s2 = sz ? Blk{0, 0, 0, 0} : cs.s2;
          if (up) d2 = Blk{xJ.x - s2.x, xJ.y - s2.y, xJ.z - s2.z, xJ.w - s2.w};
          d2h0 = (xc ? cu.x : cs.xh.x) - (sz ? 0.0 : cs.s2h.x);
          d2h1 = (xc ? cu.y : cs.xh.y) - (sz ? 0.0 : cs.s2h.y);
          d2g = (xc ? cg : cs.xg) - (sz ? 0.0 : cs.s2g);
        }
        const bool heavy = en.mf == 2 * P, itrail = en.keep0 == 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) f.w[i][j] = 0.0;
        if (up) {
          f.w[0][0] = cs.ii.x; f.w[1][0] = cs.ii.y; f.w[0][1] = cs.ii.z; f.w[1][1] = cs.ii.w;
        }
        f.h[0] = cs.hi.x; f.h[1] = cs.hi.y; f.h[2] = 0.0; f.h[3] = 0.0;
        if (heavy) {
          if (up) {
            f.w[2][2] = cs.ss.x; f.w[3][2] = cs.ss.y; f.w[2][3] = cs.ss.z; f.w[3][3] = cs.ss.w;
          }
          f.w[2][0] = cs.t.x; f.w[3][0] = itrail ? cs.t.z : cs.t.y; f.w[2][1] = itrail ? cs.t.y : cs.t.z; f.w[3][1] = cs.t.w;
          f.h[2] = cs.hs.x; f.h[3] = cs.hs.y;
        }
        gmsg = cs.g;
        if (chain_kind == 1 || chain_kind == 3) {   // the integrated block of a 2P sender; a P-dim sender's whole belief
          if (up) {
            f.w[0][0] = cv.x; f.w[1][0] = cv.y; f.w[0][1] = cv.z; f.w[1][1] = cv.w;
          }
          f.h[0] = cu.x; f.h[1] = cu.y;
          gmsg = cg;
        }
      }
      PGBP_LT(1);
      if (state == 1) {
        if (en.mf == 0) {
          // a constant factor
        } else if (en.mf == P && has_block) {
          // nothing to integrate: the message is the sender's belief (src/beliefupdates.jl:56)
          mJ = Blk{f.w[0][0], f.w[1][0], f.w[0][1], f.w[1][1]};
          mh0 = f.h[0]; mh1 = f.h[1];
        } else {
          if (pro && !__builtin_amdgcn_readfirstlane(poison_v)) {
            // PROLOGUE: X -> F lands on F's integrated block (mult!, src/beliefupdates.jl:483-488); what it stores is
            // the consumer's business, which forms the same delta from the same operands
            f.w[0][0] += d2.x; f.w[1][0] += d2.y; f.w[0][1] += d2.z; f.w[1][1] += d2.w;
            f.h[0] += d2h0; f.h[1] += d2h1;
            gmsg += d2g;
          }
          // Symmetric(J_I) (src/beliefupdates.jl:68): the lanes a > b take all four entries from lane (b, a)
          const int tl = a * G + b;
          const double t00 = __shfl(f.w[0][0], tl), t01 = __shfl(f.w[1][0], tl);
          const double t10 = __shfl(f.w[0][1], tl), t11 = __shfl(f.w[1][1], tl);
          if (2 * a + 0 > 2 * b + 0) f.w[0][0] = t00;
          if (2 * a + 0 > 2 * b + 1) f.w[0][1] = t01;
          if (2 * a + 1 > 2 * b + 0) f.w[1][0] = t10;
          if (2 * a + 1 > 2 * b + 1) f.w[1][1] = t11;
          // "fake" message: J_I, J_SI, h_I all ~ 0 (:62-66)
          bool nz = fabs(f.h[0]) > PGBP_EPS || fabs(f.h[1]) > PGBP_EPS;
#pragma unroll
          for (int i = 0; i < 4; ++i) nz |= fabs(f.w[i][0]) > PGBP_EPS || fabs(f.w[i][1]) > PGBP_EPS;
          if (__any(nz)) {
            double mant = 1.0, quad = 0.0;
            int expo = 0;
            PGBP_LT(2);
            info = eliminate2<P, 0>(f, a, b, act, col, mant, expo, quad);
            PGBP_LT(3);
            if (info == 0) {
              const double logdet = log_by_table(S.logtab, mant) + (double)expo * PGBP_LN2;
              gmsg += 0.5 * ((double)P * PGBP_LOG2PI - logdet + quad);  // :81
            }
          }
          mJ = Blk{f.w[2][2], f.w[3][2], f.w[2][3], f.w[3][3]};
          mh0 = f.h[2]; mh1 = f.h[3];
        }
        if (__builtin_amdgcn_readfirstlane(poison_v)) state = 3;
        else if (info != 0) state = 2;
      }
      // (the next record is requested HERE: behind the last use of the early operands -- the wait for those is a wait for
      // everything in flight -- and behind the elimination, which has no register to spare for it)
      issue_next();
      // ---- the marginal, for this record's consumer and for those of the records that reuse it
      if (en.valid && provider) {
        if (state == 1) {
          *reinterpret_cast<double4*>(marg + kLSlotJ + 4 * lane) = make_double4(mJ.x, mJ.y, mJ.z, mJ.w);
          if (act && b == 0) *reinterpret_cast<double2*>(marg + kLSlotH + 2 * a) = make_double2(mh0, mh1);
        }
        if (lane == 0) {
          marg[kLSlotG] = gmsg;
          marg[kLSlotStatus] = (double)state;
          marg[kLSlotInfo] = (double)info;
        }
      }
      PGBP_LT(4);
      lds_barrier();   // B1: the marginals are out; every store of the pass before is complete
      PGBP_LT(5);
      // ---- the next record and its sender operands, EARLY (a late group's are fetched again behind the full barrier)
      decode_next(nxv, nx, npr);
      issue_snd(nx, npr, cs);
      PGBP_LT(6);
      lds_barrier();   // B2: the chain slots of this pass are written
      PGBP_LT(7);
#ifdef PGBP_STAMP
      if ((threadIdx.x & 63) == 0) {
        const unsigned int sl = atomicAdd(&g_lstamp_n, 1u);
        if (sl < kLStampSlots) {
          for (int i = 0; i < kLStampN; ++i) g_lstamp[sl][i] = stv[i];
          g_lstamp[sl][kLStampN] = blockIdx.x; g_lstamp[sl][kLStampN + 1] = wave16; g_lstamp[sl][kLStampN + 2] = g;
          g_lstamp[sl][kLStampN + 3] = (unsigned int)(gridDim.x * 4 + (first_pass ? 1 : 0) + (en.pad[2] ? 2 : 0));
        }
      }
#endif
      ++g;
      if (g >= ngroups) break;
      if (nx.pad[2] != 0) {   // (the same in every record of a group: all sixteen wavefronts take the same branch)
        // the next pass reads from memory what this one wrote (a receiver block, a sepset): every store complete first
        __syncthreads();
        issue_snd(nx, npr, cs);
      }
      en = nx;
      pr = npr;
      first_pass = false;
    }
  } else {
    for (;;) {
#ifdef PGBP_STAMP
      unsigned int stv[kLStampN] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
      PGBP_LT(0);
      lane_geometry();
      const bool has_next = g + 1 < ngroups;
      // the next record (and its prologue) as a VECTOR load, lanes 0 .. 3 (4, 5): waited for where it is decoded (without
      // a next group it stays zero: an invalid record)
      uint4 nxv = make_uint4(0, 0, 0, 0);
      auto issue_next = [&]() {
        if (has_next) {
          if (PRO && (lane & 4))
            nxv = reinterpret_cast<const uint4*>(pros + ((int64_t)(g + 1) * W + wave))[lane & 1];
          else
            nxv = reinterpret_cast<const uint4*>(recs + ((int64_t)(g + 1) * W + wave))[lane & 3];
        }
      };
      // a failure in the postorder part of the tail must stop its preorder part: at the first preorder level the fail word
      // is read again, coherently
      if (g == split && split > 0) {
        failkey = __hip_atomic_load(&S.fail[site], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        failkey = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(failkey >> 32)) << 32) |
                  (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)failkey);
      }
      const unsigned long long stop_below = g >= split ? stop_b : stop_a;
      const bool live = en.valid && !((failkey >> kInfoBits) < stop_below);   // (not stopped)
      [[maybe_unused]] const bool has_block = en.s > 0;
      [[maybe_unused]] const bool own = (en.mode & kFOwn) != 0;
      [[maybe_unused]] const bool accum = (en.mode & kFAccum) != 0;
      [[maybe_unused]] const bool recv_blk = has_block || (accum && !(en.mode & kFNoBlock));
      const bool provider = en.src_wave == wave;   // the record eliminates (the others reuse a sibling's marginal)
      [[maybe_unused]] const int first_wave = en.grp_base;
      [[maybe_unused]] const bool pro = PRO && (en.mode & kFPro) != 0;
      // (the first pass of a walk has loaded everything from memory, whatever its records say: a tail launch may start at
      // its preorder half)
      const int chain_kind = first_pass ? 0 : en.pad[0], chain_src = en.pad[1];
      FEntry nx{};
      FPro npr{};

      // ================================================================================ CONSUMER: divide!, mult!, stores
      double* __restrict__ sep = pool + en.sep_off;
      double* __restrict__ to = pool + en.to_off;
      double* __restrict__ res = rpool + en.res_off;
      const int mt = en.mt, up0 = en.up0;
      const bool tpk = (mt == P || mt == 2 * P);
      const int sepG = has_block ? bs16::g1(P) : 0;
      const int64_t tJ0 = tpk ? ((mt == 2 * P && up0 == P) ? bs16::t11(P) : 0) : (up0 + (int64_t)mt * up0);
      const int64_t tH0 = (tpk ? (mt == P ? bs16::h1(P) : bs16::h2(P)) : (int64_t)mt * mt) + up0;
      const int64_t tG0 = tpk ? (mt == P ? bs16::g1(P) : bs16::g2(P)) : (int64_t)mt * mt + mt;
      const bool sepz = S.sep_zero != 0;
      issue_next();
      const int f_ii = en.keep0 == 0 ? bs16::t11(P) : 0, f_hi = bs16::h2(P) + (en.keep0 == 0 ? P : 0);
      // ---- its loads, at the top of the pass: the sepset, the receiver block, the prologue's operands
      Blk sJ{0, 0, 0, 0}, tJ{0, 0, 0, 0};
      double2 sh = make_double2(0.0, 0.0);
      double sg = 0.0, th0 = 0.0, th1 = 0.0, tg = 0.0;
      // the prologue: X (what the sepset (X, F) becomes), the residual, F's integrated block after mult!
      Blk p_x{0, 0, 0, 0}, p_d{0, 0, 0, 0}, p_ii{0, 0, 0, 0};
      double p_xh0 = 0.0, p_xh1 = 0.0, p_dh0 = 0.0, p_dh1 = 0.0, p_xg = 0.0, p_h0 = 0.0, p_h1 = 0.0, p_g = 0.0;
      bool p_ok = false;
      if (live) {
        if (!sepz) {
          if (has_block) {
            sJ = load_blk<true, false, !PRO>(sep, P, a, b, up, kidx);
            if (b == 0) sh = load_pair<false>(sep + bs16::h1(P), a, P);
          }
          sg = sep[sepG];
        }
        if (own) {
          if (recv_blk) {
            tJ = load_blk<true>(to + tJ0, mt, a, b, up, kidx);
            if (b == 0) {
              const double2 t2 = load_pair<false>(to + tH0, a, P);
              th0 = t2.x; th1 = t2.y;
            }
          }
          tg = to[tG0];
        }
        if (pro && provider) {
          const double* __restrict__ xfrom = pool + pr.from_off;
          const double* __restrict__ sep2 = pool + pr.sep_off;
          const double* __restrict__ fw = pool + en.from_off;
          int poison_p = S.poison[(int64_t)site * S.n_clusters + en.from_b] | S.poison[(int64_t)site * S.n_clusters + pr.from_b];
          Blk s2{0, 0, 0, 0};
          double2 xh, s2h = make_double2(0.0, 0.0);
          double s2g = 0.0;
          if (chain_kind == 2) {
            const double* src = loop_lds + (2 * W + chain_src) * kLSlotDoubles;
            const double4 v = *reinterpret_cast<const double4*>(src + kLSlotJ + 4 * lane);
            p_x = up ? Blk{v.x, v.y, v.z, v.w} : Blk{0, 0, 0, 0};
            xh = *reinterpret_cast<const double2*>(src + kLSlotH + 2 * a);
            p_xg = src[kLSlotG];
            if ((int)src[kLSlotStatus] != 1) poison_p = 1;
          } else {
            p_x = load_blk<true>(xfrom, P, a, b, up, kidx);
            xh = load_pair<false>(xfrom + bs16::h1(P), a, P);
            p_xg = xfrom[bs16::g1(P)];
          }
          if (!sepz) {
            s2 = load_blk<true>(sep2, P, a, b, up, kidx);
            s2h = load_pair<false>(sep2 + bs16::h1(P), a, P);
            s2g = sep2[bs16::g1(P)];
          }
          const Blk fi = load_blk<true>(fw + f_ii, P, a, b, up, kidx);
          const double2 fh = load_pair<false>(fw + f_hi, a, P);
          const double fg = fw[bs16::g2(P)];
          // the message X -> F as bp_fast16 forms it: divide! by the sepset (X, F), mult! onto F's record
          p_ok = !__builtin_amdgcn_readfirstlane(poison_p);
          p_xh0 = xh.x; p_xh1 = xh.y;
          p_d = Blk{p_x.x - s2.x, p_x.y - s2.y, p_x.z - s2.z, p_x.w - s2.w};
          p_dh0 = xh.x - s2h.x; p_dh1 = xh.y - s2h.y;
          const double d2g = p_xg - s2g;
          p_ii = Blk{fi.x + p_d.x, fi.y + p_d.y, fi.z + p_d.z, fi.w + p_d.w};
          p_h0 = fh.x + p_dh0; p_h1 = fh.y + p_dh1;
          p_g = fg + d2g;
        }
      }
      // the later consumers of a task hand their sepset to its first one, which does all of the task's mult!
      if (accum && wave > first_wave && live) {
        *reinterpret_cast<double4*>(delta + kLSlotJ + 4 * lane) = make_double4(sJ.x, sJ.y, sJ.z, sJ.w);
        if (act && b == 0) *reinterpret_cast<double2*>(delta + kLSlotH + 2 * a) = sh;
        if (lane == 0) delta[kLSlotG] = sg;
      }
      // everything this wavefront stored in the pass before is complete (the barrier below orders it before the early
      // loads of the providers), and so are its own loads of this pass
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PGBP_LT(4);
      lds_barrier();   // B1: the marginals are out
      PGBP_LT(5);
      Blk mJ{0, 0, 0, 0};
      double mh0 = 0.0, mh1 = 0.0, gmsg = 0.0;
      int state = 0, info = 0;
      if (live) {
        const double* src = loop_lds + en.src_wave * kLSlotDoubles;
        state = (int)src[kLSlotStatus];
        info = (int)src[kLSlotInfo];
        if (state == 1) {
          const double4 v = *reinterpret_cast<const double4*>(src + kLSlotJ + 4 * lane);
          mJ = Blk{v.x, v.y, v.z, v.w};
          if (b == 0) {
            const double2 u = *reinterpret_cast<const double2*>(src + kLSlotH + 2 * a);
            mh0 = u.x; mh1 = u.y;
          }
          gmsg = src[kLSlotG];
        } else if (!provider && state != 0) {
          state = 3;   // the marginal it reuses failed or was skipped: as good as a poisoned sender
        }
      }
      // ---- divide! (src/beliefupdates.jl:579-587)
      Blk dJ{0, 0, 0, 0};
      double dh0 = 0.0, dh1 = 0.0, dg = 0.0, maxJ = 0.0, maxh = 0.0;
      if (state == 1) {
        if (has_block) {
          dJ = Blk{mJ.x - sJ.x, mJ.y - sJ.y, mJ.z - sJ.z, mJ.w - sJ.w};
          if (up) {
            maxJ = fmax(fmax(fabs(dJ.x), fabs(dJ.y)), fmax(fabs(dJ.z), fabs(dJ.w)));
            if (dJ.x != dJ.x || dJ.y != dJ.y || dJ.z != dJ.z || dJ.w != dJ.w) maxJ = INFINITY;
          }
          if (act && b == 0) {
            dh0 = mh0 - sh.x; dh1 = mh1 - sh.y;
            maxh = (dh0 != dh0 || dh1 != dh1) ? INFINITY : fmax(fabs(dh0), fabs(dh1));
          }
        }
        dg = gmsg - sg;
      }
      // ---- mult! (src/beliefupdates.jl:483-488): the first consumer of a task adds up the task's messages in the
      // reference's order -- the others' it forms itself, marginal minus sepset, as their own consumers do for their stores
      bool whole = state == 1;   // the receiver block this wavefront owns holds every message of its task
      if (state == 1) {
        tJ = Blk{tJ.x + dJ.x, tJ.y + dJ.y, tJ.z + dJ.z, tJ.w + dJ.w};
        th0 += dh0; th1 += dh1;
        tg += dg;
        if (accum && wave == first_wave) {
          for (int w = 1; w < en.grp_len; ++w) {
            const double* mw = loop_lds + (first_wave + w) * kLSlotDoubles;        // (an accumulating record eliminates)
            const double* sw = loop_lds + (W + first_wave + w) * kLSlotDoubles;
            if ((int)mw[kLSlotStatus] != 1) {  // the reference stops at the first failing message
              whole = false;
              break;
            }
            const double4 m = *reinterpret_cast<const double4*>(mw + kLSlotJ + 4 * lane);
            const double4 v = *reinterpret_cast<const double4*>(sw + kLSlotJ + 4 * lane);
            tJ = Blk{tJ.x + (m.x - v.x), tJ.y + (m.y - v.y), tJ.z + (m.z - v.z), tJ.w + (m.w - v.w)};
            if (b == 0) {
              const double2 mu = *reinterpret_cast<const double2*>(mw + kLSlotH + 2 * a);
              const double2 u = *reinterpret_cast<const double2*>(sw + kLSlotH + 2 * a);
              th0 += mu.x - u.x; th1 += mu.y - u.y;
            }
            tg += mw[kLSlotG] - sw[kLSlotG];
          }
        }
      }
      PGBP_LT(6);
      // ---- CHAIN: what this wavefront owns of its receiver goes into its chain slot for the next pass
      if (own && en.valid) {
        *reinterpret_cast<double4*>(chain + kLSlotJ + 4 * lane) = make_double4(tJ.x, tJ.y, tJ.z, tJ.w);
        if (act && b == 0) *reinterpret_cast<double2*>(chain + kLSlotH + 2 * a) = make_double2(th0, th1);
        if (lane == 0) {
          chain[kLSlotG] = tg;
          chain[kLSlotStatus] = whole ? 1.0 : 3.0;
        }
      }
      PGBP_LT(7);
      lds_barrier();   // B2: the chain slots of this pass are written
      PGBP_LT(8);
      // ---- the stores of the pass, off the critical path (the providers are at the next elimination): the prologue's,
      // divide!'s, mult!'s
      if (p_ok) {
        double* __restrict__ sep2 = pool + pr.sep_off;
        double* __restrict__ res2 = rpool + pr.res_off;
        double* __restrict__ fw = pool + en.from_off;
        constexpr int xH = bs16::h1(P), xG = bs16::g1(P);
        store_blk<true>(sep2, P, a, b, up, act, kidx, p_x);
        store_blk<true, false, true>(res2, P, a, b, up, act, kidx, p_d);
        double maxJ2 = 0.0, maxh2 = 0.0;
        if (up) {
          maxJ2 = fmax(fmax(fabs(p_d.x), fabs(p_d.y)), fmax(fabs(p_d.z), fabs(p_d.w)));
          if (p_d.x != p_d.x || p_d.y != p_d.y || p_d.z != p_d.z || p_d.w != p_d.w) maxJ2 = INFINITY;
        }
        if (act && b == 0) {
          store_pair<false>(sep2 + xH, a, p_xh0, p_xh1, P);
          store_pair<false, true>(res2 + xH, a, p_dh0, p_dh1, P);
          maxh2 = (p_dh0 != p_dh0 || p_dh1 != p_dh1) ? INFINITY : fmax(fabs(p_dh0), fabs(p_dh1));
        }
        if (lane == 0) {
          sep2[xG] = p_xg;
          S.status[(int64_t)site * S.n_msgs + pr.msg] = 0;
        }
        if (S.update_resnorm) {
          const bool all_ok2 = __all(maxh2 <= S.thr_h_p && maxJ2 <= S.thr_J_p);
          if (lane == 0) S.flags[(int64_t)site * S.n_msgs + pr.msg] = all_ok2 ? 1 : 0;
        }
        store_blk<true>(fw + f_ii, P, a, b, up, act, kidx, p_ii);
        if (act && b == 0) store_pair<false>(fw + f_hi, a, p_h0, p_h1, P);
        if (lane == 0) fw[bs16::g2(P)] = p_g;
      }
      if (state == 1) {
        if (has_block) {
          // (streaming stores, as in bp_fast16: nothing reads the residual, and the sepset not before the other traversal)
          store_blk<true, false, !PRO>(sep, P, a, b, up, act, kidx, mJ);
          store_blk<true, false, true>(res, P, a, b, up, act, kidx, dJ);
          if (act && b == 0) {
            store_pair<false, !PRO>(sep + bs16::h1(P), a, mh0, mh1, P);
            store_pair<false, true>(res + bs16::h1(P), a, dh0, dh1, P);
          }
        }
        if (lane == 0) {
          sep[sepG] = gmsg;
          S.status[(int64_t)site * S.n_msgs + en.msg] = 0;
        }
        if (S.update_resnorm) {
          // iscalibrated_residnorm! (src/beliefs.jl:994-1003); an empty message is calibrated
          const bool lane_ok = maxh <= S.thr_h_p && maxJ <= S.thr_J_p;
          const bool all_ok = __all(lane_ok);
          if (lane == 0) S.flags[(int64_t)site * S.n_msgs + en.msg] = (!has_block || all_ok) ? 1 : 0;
        }
        if (own) {
          if (recv_blk) {
            store_blk<true>(to + tJ0, mt, a, b, up, act, kidx, tJ);
            if (act && b == 0) store_pair<false>(to + tH0, a, th0, th1, P);
          }
          if (lane == 0) to[tG0] = tg;
        }
      } else if (state >= 2 && lane == 0) {
        // not positive definite, or downstream of a failure: nothing of this message is applied
        S.poison[(int64_t)site * S.n_clusters + en.to_b] = 1;
        if (state == 2) {
          S.status[(int64_t)site * S.n_msgs + en.msg] = info;
          atomicMin(&S.fail[site], ((seq_base + (unsigned long long)en.seq) << kInfoBits) | (unsigned long long)info);
        }
      }
      decode_next(nxv, nx, npr);
      PGBP_LT(9);
#ifdef PGBP_STAMP
      if ((threadIdx.x & 63) == 0) {
        const unsigned int sl = atomicAdd(&g_lstamp_n, 1u);
        if (sl < kLStampSlots) {
          for (int i = 0; i < kLStampN; ++i) g_lstamp[sl][i] = stv[i];
          g_lstamp[sl][kLStampN] = blockIdx.x; g_lstamp[sl][kLStampN + 1] = wave16; g_lstamp[sl][kLStampN + 2] = g;
          g_lstamp[sl][kLStampN + 3] = (unsigned int)(gridDim.x * 4 + (first_pass ? 1 : 0) + (en.pad[2] ? 2 : 0));
        }
      }
#endif
      ++g;
      if (g >= ngroups) break;
      if (nx.pad[2] != 0) {   // (the same in every record of a group: all sixteen wavefronts take the same branch)
        // the next pass reads from memory what this one wrote (a receiver block, a sepset): every store complete first
        __syncthreads();
      }
      en = nx;
      pr = npr;
      first_pass = false;
    }
  }
}

namespace {

template <int P>
void launch_loop_p(const DevState& S, const FEntry* d_recs, const FPro* d_pros, int ngroups, int split, int n_sites,
                   unsigned long long seq_base, unsigned long long stop_a, unsigned long long stop_b, hipStream_t st,
                   const int32_t* d_wg_off, int n_wg) {
  const dim3 grid(d_wg_off ? n_wg : 1, n_sites), block(kLoopWaves * 64);
  const size_t lds = sizeof(double) * (size_t)kTailWaves * (size_t)(3 * kLSlotDoubles + kColDoubles);
  if (d_pros)
    hipLaunchKernelGGL((bp_loop16<P, true>), grid, block, lds, st, S, d_recs, d_pros, ngroups, split, seq_base, stop_a, stop_b,
                       d_wg_off);
  else
    hipLaunchKernelGGL((bp_loop16<P, false>), grid, block, lds, st, S, d_recs, d_pros, ngroups, split, seq_base, stop_a, stop_b,
                       d_wg_off);
}

}  // namespace

// the loop launches (tail, chunks) in the packed layout, even P; arguments as for launch_fast16's mode kFastTail
void launch_loop16(const DevState& S, const FEntry* d_recs, const FPro* d_pros, int ngroups, int split, int n_sites,
                   unsigned long long seq_base, unsigned long long stop_a, unsigned long long stop_b, hipStream_t st,
                   const int32_t* d_wg_off, int n_wg) {
  if (ngroups <= 0) return;
#define PGBP_LOOP(PP) launch_loop_p<PP>(S, d_recs, d_pros, ngroups, split, n_sites, seq_base, stop_a, stop_b, st, d_wg_off, n_wg); break
  switch (S.fast_p) {
    case 16: PGBP_LOOP(16);
#ifndef PGBP_ONLY_P16
    case 14: PGBP_LOOP(14);
    case 12: PGBP_LOOP(12);
    case 10: PGBP_LOOP(10);
    case 8: PGBP_LOOP(8);
    case 6: PGBP_LOOP(6);
    case 4: PGBP_LOOP(4);
    case 2: PGBP_LOOP(2);
#endif
    default: break;
  }
#undef PGBP_LOOP
}

}  // namespace pgbp

#ifdef PGBP_STAMP
extern "C" int pgbp_debug_lstamps(unsigned int* out, unsigned int cap, unsigned int* n) {
  if (hipDeviceSynchronize() != hipSuccess) return 4;
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(pgbp::g_lstamp_n), sizeof(unsigned int)) != hipSuccess) return 1;
  const unsigned int k = *n < cap ? *n : cap;
  if (k && hipMemcpyFromSymbol(out, HIP_SYMBOL(pgbp::g_lstamp), sizeof(unsigned int) * (size_t)k * 16) != hipSuccess) return 2;
  unsigned int z = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(pgbp::g_lstamp_n), &z, sizeof(z)) != hipSuccess) return 3;
  return 0;
}
#endif
