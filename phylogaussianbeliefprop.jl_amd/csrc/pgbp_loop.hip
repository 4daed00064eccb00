// Loop launches of the register-resident message kernel in the packed layout (gfx950, wave64, even P): the single-workgroup
// TAIL of a traversal pair and the CHUNKS of fused narrow levels (pgbp_plan.cpp: Traversal::tail_levels, build_chunks).
//
// A workgroup of 8 wavefronts walks its groups of 8 records, one group = one dependent step ("pass") of the schedule; a
// narrow level lasts as long as ONE message, so what counts here is the chain of dependent latencies of a pass, not
// throughput.  pgbp_fast.hip's loop mode (still used for the plain layouts) pays per pass: a workgroup barrier that waits
// for every store of the pass, then the operand loads of the next one (two memory round trips) in front of its
// elimination.  Here a pass starts with its operands already in registers:
//   * EARLY LOADS: the operands of pass g + 1 are requested in the middle of pass g -- behind its elimination, where the
//     registers of the elimination are free again -- and arrive while pass g divides, multiplies and stores;
//   * CHAINS: what pass g + 1 needs FROM pass g -- the block a wavefront of pass g has just accumulated into cluster X, when
//     X sends in pass g + 1 -- does not go through memory: the owner leaves (block, h, g, state) in its LDS chain slot and
//     the sender of pass g + 1 patches them into the operands it loaded early (planner: FEntry::chain, link_chains);
//   * a pass whose records need anything else that pass g writes (a receiver block, a sepset: the post -> pre turn of the
//     tail) is LATE: full barrier, loads at its top, as before;
//   * the barriers inside and between passes order LDS only; each wavefront waits for its own stores of pass g - 1 (long
//     done) before the first barrier of pass g, so that whatever pass g + 1 loads early, written in pass g - 1 or before,
//     is complete.
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

// hand-over / chain slot of one wave: 2 x 2 block per lane (256 doubles, lane-major), h (16), g, status
constexpr int kLSlotJ = 0, kLSlotH = 256, kLSlotG = 272, kLSlotStatus = 273, kLSlotDoubles = 288;

// sender operands of one record exactly as loaded (no arithmetic on them before the pass that uses them: a select or a
// copy of a loaded register is a wait for that load, and these loads are in flight across half a pass)
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
// ... its sepset and receiver block
struct RcvOps {
  Blk sJ, tJ;
  double2 sh;
  double sg, th0, th1, tg;
};

}  // namespace

template <int P, bool PRO>
__global__ __launch_bounds__(kTailWaves * 64) void bp_loop16(DevState S_arg, const FEntry* __restrict__ recs_arg,
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
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int site = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const bool act = lane < G * G;
  const int a = act ? lane % G : 0, b = act ? lane / G : 0;
  const bool up = act && a <= b;
  const int kidx = (b * (b + 1) / 2 + a) * 4;
  double* __restrict__ pool = S.pool + (int64_t)site * S.pool_stride;
  double* __restrict__ rpool = S.rpool + (int64_t)site * S.rpool_stride;
  double* const slot = loop_lds + wave * kLSlotDoubles;                        // hand-over inside a pass
  double* const chain = loop_lds + (W + wave) * kLSlotDoubles;                 // what this wave leaves for the next pass
  double* const col = loop_lds + 2 * W * kLSlotDoubles + wave * kColDoubles;   // private strip of the elimination

  if (wg_off) ngroups = wg_off[blockIdx.x + 1];
  int g = wg_off ? wg_off[blockIdx.x] : 0;
  FEntry en = load_record(recs + ((int64_t)g * W + wave));
  FPro pr{};
  if constexpr (PRO) pr = load_pro(pros + ((int64_t)g * W + wave));
  unsigned long long failkey = S.fail[site];

  // the sender operands of a record -- what its elimination waits for -- requested EARLY, in the middle of the pass
  // before (chained parts included: what they fetch is overwritten from the chain slot).  Straight-line code: every load is
  // issued whatever the record is (an offset that does not apply points at the start of the record or of the pool: valid
  // memory, the value is ignored), nothing is computed from a loaded value here.
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
  // its sepset and its receiver block: requested at the top of the pass, used behind the elimination.  Straight-line like
  // issue_snd: every load is issued (what does not apply fetches the start of the record), masks at the point of use.
  auto issue_rcv = [&](const FEntry& q, RcvOps& v) {
    const bool has_block = q.s > 0;
    const bool rblk = has_block || ((q.mode & kFAccum) && !(q.mode & kFNoBlock));
    const double* __restrict__ sep = pool + q.sep_off;
    const double* __restrict__ to = pool + q.to_off;
    const int mt = q.mt, up0 = q.up0;
    const bool tpk = (mt == P || mt == 2 * P);
    const int64_t tJ0 = tpk ? ((mt == 2 * P && up0 == P) ? bs16::t11(P) : 0) : (up0 + (int64_t)mt * up0);
    const int64_t tH0 = (tpk ? (mt == P ? bs16::h1(P) : bs16::h2(P)) : (int64_t)mt * mt) + up0;
    const int64_t tG0 = tpk ? (mt == P ? bs16::g1(P) : bs16::g2(P)) : (int64_t)mt * mt + mt;
    const double4 sj = *reinterpret_cast<const double4*>(sep + (has_block ? kidx : 0));
    v.sJ = Blk{sj.x, sj.y, sj.z, sj.w};
    v.sh = *reinterpret_cast<const double2*>(sep + (has_block ? bs16::h1(P) + 2 * a : 0));
    v.sg = sep[has_block ? bs16::g1(P) : 0];
    const double4 tj = *reinterpret_cast<const double4*>(to + (rblk ? tJ0 + kidx : 0));
    v.tJ = Blk{tj.x, tj.y, tj.z, tj.w};
    const double2 t2 = *reinterpret_cast<const double2*>(to + (rblk ? tH0 + 2 * a : 0));
    v.th0 = t2.x; v.th1 = t2.y;
    v.tg = to[(q.mode & kFOwn) ? tG0 : 0];
  };

  // every register of a set of sender operands passes through an empty asm: the compiler waits for each load HERE -- where
  // they are the youngest memory operations in flight -- and knows of no pending load when the next pass uses them
  auto resident = [&](SndOps& s) {
    asm volatile("; early operands resident"
                 : "+v"(s.ii.x), "+v"(s.ii.y), "+v"(s.ii.z), "+v"(s.ii.w), "+v"(s.ss.x), "+v"(s.ss.y), "+v"(s.ss.z),
                   "+v"(s.ss.w), "+v"(s.t.x), "+v"(s.t.y), "+v"(s.t.z), "+v"(s.t.w), "+v"(s.hi.x), "+v"(s.hi.y),
                   "+v"(s.hs.x), "+v"(s.hs.y), "+v"(s.g), "+v"(s.poison));
    if constexpr (PRO)
      asm volatile("; early prologue operands resident"
                   : "+v"(s.xJ.x), "+v"(s.xJ.y), "+v"(s.xJ.z), "+v"(s.xJ.w), "+v"(s.s2.x), "+v"(s.s2.y), "+v"(s.s2.z),
                     "+v"(s.s2.w), "+v"(s.xh.x), "+v"(s.xh.y), "+v"(s.s2h.x), "+v"(s.s2h.y), "+v"(s.xg), "+v"(s.s2g),
                     "+v"(s.poison_x));
  };
  SndOps cs{};
  issue_snd(en, pr, cs);
  resident(cs);   // the first pass of a walk loads at its top
  bool first_pass = true;

  for (;;) {
#ifdef PGBP_STAMP
    unsigned int stv[kLStampN] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PGBP_LT(0);
    const bool has_next = g + 1 < ngroups;
    // the next record (and its prologue) as a VECTOR load, lanes 0 .. 3 (4, 5): waited for where it is used, behind the
    // elimination
    uint4 nxv = make_uint4(0, 0, 0, 0);
    if (has_next) {
      if (PRO && (lane & 4))
        nxv = reinterpret_cast<const uint4*>(pros + ((int64_t)(g + 1) * W + wave))[lane & 1];
      else
        nxv = reinterpret_cast<const uint4*>(recs + ((int64_t)(g + 1) * W + wave))[lane & 3];
    }
    // a failure in the postorder part of the tail must stop its preorder part: at the first preorder level the fail word
    // is read again, coherently
    if (g == split && split > 0) {
      failkey = __hip_atomic_load(&S.fail[site], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      failkey = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(failkey >> 32)) << 32) |
                (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)failkey);
    }
    const unsigned long long stop_below = g >= split ? stop_b : stop_a;

    // state 0: nothing to do / stopped; 1: message available; 2: failed (not PD); 3: sender poisoned
    int state = (en.valid && !((failkey >> kInfoBits) < stop_below)) ? 1 : 0;
    const bool has_block = en.s > 0;
    const bool own = (en.mode & kFOwn) != 0;
    const bool accum = (en.mode & kFAccum) != 0;
    const bool recv_blk = has_block || (accum && !(en.mode & kFNoBlock));
    const bool provider = en.src_wave == wave;
    const int first_wave = en.grp_base;
    const bool pro = PRO && (en.mode & kFPro) != 0;
    // (the first pass of a walk has loaded everything from memory, whatever its records say: a tail launch may start at
    // its preorder half)
    const int chain_kind = first_pass ? 0 : en.pad[0], chain_src = en.pad[1];

    double* __restrict__ sep = pool + en.sep_off;
    double* __restrict__ to = pool + en.to_off;
    double* __restrict__ res = rpool + en.res_off;
    const int mt = en.mt, up0 = en.up0;
    const bool tpk = (mt == P || mt == 2 * P);
    const int sepG = has_block ? bs16::g1(P) : 0;
    const int64_t tJ0 = tpk ? ((mt == 2 * P && up0 == P) ? bs16::t11(P) : 0) : (up0 + (int64_t)mt * up0);
    const int64_t tH0 = (tpk ? (mt == P ? bs16::h1(P) : bs16::h2(P)) : (int64_t)mt * mt) + up0;
    const int64_t tG0 = tpk ? (mt == P ? bs16::g1(P) : bs16::g2(P)) : (int64_t)mt * mt + mt;

    Blk mJ{0, 0, 0, 0};
    double mh[2] = {0, 0}, gmsg = 0.0;
    int info = 0;
    RcvOps cr;
    issue_rcv(en, cr);
    const bool sepz = S.sep_zero != 0;
    // what the prologue leaves to be stored at the end of the pass
    Blk p_x{0, 0, 0, 0}, p_d{0, 0, 0, 0}, p_ii{0, 0, 0, 0};
    double p_xh0 = 0.0, p_xh1 = 0.0, p_dh0 = 0.0, p_dh1 = 0.0, p_xg = 0.0, p_h0 = 0.0, p_h1 = 0.0, p_g = 0.0;
    bool p_go = false, p_ok = false;
    if (state == 1) {
      int poison_v = cs.poison | (pro ? cs.poison_x : 0);
      // ---- the operands as the elimination wants them (nothing was computed from them when they were loaded)
      const bool heavy = en.mf == 2 * P, itrail = en.keep0 == 0;
      Frag f;
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
      Blk xJ = up ? cs.xJ : Blk{0, 0, 0, 0}, s2 = up ? cs.s2 : Blk{0, 0, 0, 0};
      double2 xh = cs.xh, s2h = cs.s2h;
      double xg = cs.xg, s2g = cs.s2g;
      if (S.sep_zero) {
        s2 = Blk{0, 0, 0, 0};
        s2h = make_double2(0.0, 0.0);
        s2g = 0.0;
      }
      // ---- CHAIN: what this record's sender (or its prologue's X) received in the previous pass, from the owner's slot
      if (chain_kind != 0 && provider) {
        const double* src = loop_lds + (W + chain_src) * kLSlotDoubles;
        const double4 v = *reinterpret_cast<const double4*>(src + kLSlotJ + 4 * lane);
        const double2 u = *reinterpret_cast<const double2*>(src + kLSlotH + 2 * a);
        const double cg = src[kLSlotG];
        if ((int)src[kLSlotStatus] != 1) poison_v = 1;   // the cluster it reads failed or was skipped in the previous pass
        if (chain_kind == 2) {      // the prologue's X
          xJ = Blk{v.x, v.y, v.z, v.w};
          xh = u;
          xg = cg;
        } else {                    // 1: the integrated block of a 2P sender; 3: a P-dim sender's whole belief
          f.w[0][0] = v.x; f.w[1][0] = v.y; f.w[0][1] = v.z; f.w[1][1] = v.w;
          f.h[0] = u.x; f.h[1] = u.y;
          gmsg = cg;
        }
      }
      PGBP_LT(1);
      if (provider) {
        if (en.mf == 0) {
          // a constant factor
        } else if (en.mf == P && has_block) {
          // nothing to integrate: the message is the sender's belief (src/beliefupdates.jl:56)
          mJ = Blk{f.w[0][0], f.w[1][0], f.w[0][1], f.w[1][1]};
          mh[0] = f.h[0]; mh[1] = f.h[1];
        } else {
          if (pro) {
            // ---- PROLOGUE (bp_fast16): the message X -> F: divide!, then mult! onto F's integrated block; its stores wait
            // for the end of the pass like all the others
            p_go = true;
            if (!__builtin_amdgcn_readfirstlane(poison_v)) {
              p_ok = true;
              p_x = xJ; p_xh0 = xh.x; p_xh1 = xh.y; p_xg = xg;
              p_d = Blk{xJ.x - s2.x, xJ.y - s2.y, xJ.z - s2.z, xJ.w - s2.w};
              p_dh0 = xh.x - s2h.x; p_dh1 = xh.y - s2h.y;
              const double d2g = xg - s2g;
              f.w[0][0] += p_d.x; f.w[1][0] += p_d.y; f.w[0][1] += p_d.z; f.w[1][1] += p_d.w;
              f.h[0] += p_dh0; f.h[1] += p_dh1;
              gmsg += d2g;
              p_ii = Blk{f.w[0][0], f.w[1][0], f.w[0][1], f.w[1][1]};
              p_h0 = f.h[0]; p_h1 = f.h[1]; p_g = gmsg;
            }
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
          mh[0] = f.h[2]; mh[1] = f.h[3];
        }
      }
      if (__builtin_amdgcn_readfirstlane(poison_v)) state = 3;
      else if (info != 0) state = 2;
    }
    // this wave's stores of the previous pass are complete (they were issued an elimination ago): behind the barrier
    // below every wave may load, early, whatever the previous pass or an earlier one wrote
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PGBP_LT(4);
    // ---- hand the marginal over to the waves that reuse it
    if (provider && en.valid && !accum && en.grp_len > 1) {
      if (state == 1) {
        *reinterpret_cast<double4*>(slot + kLSlotJ + 4 * lane) = make_double4(mJ.x, mJ.y, mJ.z, mJ.w);
        if (act && b == 0) *reinterpret_cast<double2*>(slot + kLSlotH + 2 * a) = make_double2(mh[0], mh[1]);
      }
      if (lane == 0) {
        slot[kLSlotG] = gmsg;
        slot[kLSlotStatus] = (double)state;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    PGBP_LT(5);
    // ---- the next record, and its sender operands EARLY (unless it is a late one: then at the bottom, behind the full
    // barrier)
    FEntry nx{};
    FPro npr{};
    SndOps ns;
    bool next_late = false;
    {   // (unconditionally: without a next group nxv is zero, an invalid record whose loads fetch the start of the pool)
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
      next_late = nx.pad[2] != 0;   // (the same in every record of a group: all eight waves take the same branch)
      issue_snd(nx, npr, ns);       // (a late group's are fetched again behind the full barrier)
    }
    PGBP_LT(6);
    if (!provider && state == 1) {
      const double* src = loop_lds + en.src_wave * kLSlotDoubles;
      const int pst = (int)src[kLSlotStatus];
      if (pst == 1) {
        const double4 v = *reinterpret_cast<const double4*>(src + kLSlotJ + 4 * lane);
        mJ = Blk{v.x, v.y, v.z, v.w};
        if (b == 0) {
          const double2 u = *reinterpret_cast<const double2*>(src + kLSlotH + 2 * a);
          mh[0] = u.x; mh[1] = u.y;
        }
        gmsg = src[kLSlotG];
      } else {
        state = 3;
      }
    }
    // ---- divide! (src/beliefupdates.jl:579-587): the arithmetic now, the stores at the end of the pass
    Blk dJ{0, 0, 0, 0};
    double dh0 = 0.0, dh1 = 0.0, dg = 0.0;
    double maxJ = 0.0, maxh = 0.0;
    if (state == 1) {
      if (has_block) {
        const Blk sJ = (up && !sepz) ? cr.sJ : Blk{0, 0, 0, 0};
        dJ = Blk{mJ.x - sJ.x, mJ.y - sJ.y, mJ.z - sJ.z, mJ.w - sJ.w};
        if (up) {
          maxJ = fmax(fmax(fabs(dJ.x), fabs(dJ.y)), fmax(fabs(dJ.z), fabs(dJ.w)));
          if (dJ.x != dJ.x || dJ.y != dJ.y || dJ.z != dJ.z || dJ.w != dJ.w) maxJ = INFINITY;
        }
        if (act && b == 0) {
          dh0 = mh[0] - (sepz ? 0.0 : cr.sh.x); dh1 = mh[1] - (sepz ? 0.0 : cr.sh.y);
          maxh = (dh0 != dh0 || dh1 != dh1) ? INFINITY : fmax(fabs(dh0), fabs(dh1));
        }
      }
      dg = gmsg - (sepz ? 0.0 : cr.sg);
    }
    // ---- mult! (src/beliefupdates.jl:483-488)
    if (accum && wave > first_wave) {
      if (state == 1) {
        *reinterpret_cast<double4*>(slot + kLSlotJ + 4 * lane) = make_double4(dJ.x, dJ.y, dJ.z, dJ.w);
        if (act && b == 0) *reinterpret_cast<double2*>(slot + kLSlotH + 2 * a) = make_double2(dh0, dh1);
      }
      if (lane == 0) {
        slot[kLSlotG] = dg;
        slot[kLSlotStatus] = (double)state;
      }
    }
    PGBP_LT(7);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    PGBP_LT(8);
    Blk tJ = (up && own && recv_blk) ? cr.tJ : Blk{0, 0, 0, 0};
    double th0 = (own && recv_blk && b == 0) ? cr.th0 : 0.0, th1 = (own && recv_blk && b == 0) ? cr.th1 : 0.0;
    double tg = own ? cr.tg : 0.0;
    bool whole = state == 1;   // the receiver block this wave owns holds every message of its task
    if (state == 1) {
      tJ = Blk{tJ.x + dJ.x, tJ.y + dJ.y, tJ.z + dJ.z, tJ.w + dJ.w};
      th0 += dh0; th1 += dh1;
      tg += dg;
      if (accum && wave == first_wave) {
        for (int w = 1; w < en.grp_len; ++w) {
          const double* src = loop_lds + (first_wave + w) * kLSlotDoubles;
          if ((int)src[kLSlotStatus] != 1) {  // the reference stops at the first failing message
            whole = false;
            break;
          }
          const double4 v = *reinterpret_cast<const double4*>(src + kLSlotJ + 4 * lane);
          tJ = Blk{tJ.x + v.x, tJ.y + v.y, tJ.z + v.z, tJ.w + v.w};
          if (b == 0) {
            const double2 u = *reinterpret_cast<const double2*>(src + kLSlotH + 2 * a);
            th0 += u.x; th1 += u.y;
          }
          tg += src[kLSlotG];
        }
      }
    }
    // ---- CHAIN: what this wave owns of its receiver goes into its chain slot for the senders of the next pass
    if (own && en.valid) {
      *reinterpret_cast<double4*>(chain + kLSlotJ + 4 * lane) = make_double4(tJ.x, tJ.y, tJ.z, tJ.w);
      if (act && b == 0) *reinterpret_cast<double2*>(chain + kLSlotH + 2 * a) = make_double2(th0, th1);
      if (lane == 0) {
        chain[kLSlotG] = tg;
        chain[kLSlotStatus] = whole ? 1.0 : 3.0;
      }
    }
    // ---- the early loads have arrived: they become the next pass's operands NOW, before this pass's stores join the
    // queue of outstanding memory operations behind them (the counter is in order: with the stores in front, the next pass
    // would wait for their acknowledgement to be sure of its operands)
    cs = ns;
    resident(cs);
    PGBP_LT(9);
    // ---- the stores of the pass: the prologue's, divide!'s, mult!'s
    if (p_go && p_ok) {
      double* __restrict__ sep2 = pool + pr.sep_off;
      double* __restrict__ res2 = rpool + pr.res_off;
      constexpr int xH = bs16::h1(P), xG = bs16::g1(P);
      store_blk<true>(sep2, P, a, b, up, act, kidx, p_x);
      store_blk<true>(res2, P, a, b, up, act, kidx, p_d);
      double maxJ2 = 0.0, maxh2 = 0.0;
      if (up) {
        maxJ2 = fmax(fmax(fabs(p_d.x), fabs(p_d.y)), fmax(fabs(p_d.z), fabs(p_d.w)));
        if (p_d.x != p_d.x || p_d.y != p_d.y || p_d.z != p_d.z || p_d.w != p_d.w) maxJ2 = INFINITY;
      }
      if (act && b == 0) {
        store_pair<false>(sep2 + xH, a, p_xh0, p_xh1, P);
        store_pair<false>(res2 + xH, a, p_dh0, p_dh1, P);
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
      double* __restrict__ fw = pool + en.from_off;
      store_blk<true>(fw + (en.keep0 == 0 ? bs16::t11(P) : 0), P, a, b, up, act, kidx, p_ii);
      if (act && b == 0) store_pair<false>(fw + bs16::h2(P) + (en.keep0 == 0 ? P : 0), a, p_h0, p_h1, P);
      if (lane == 0) fw[bs16::g2(P)] = p_g;
    }
    if (state == 1) {
      if (has_block) {
        store_blk<true>(sep, P, a, b, up, act, kidx, mJ);
        store_blk<true>(res, P, a, b, up, act, kidx, dJ);
        if (act && b == 0) {
          store_pair<false>(sep + bs16::h1(P), a, mh[0], mh[1], P);
          store_pair<false>(res + bs16::h1(P), a, dh0, dh1, P);
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
#ifdef PGBP_STAMP
    if ((threadIdx.x & 63) == 0) {
      const unsigned int sl = atomicAdd(&g_lstamp_n, 1u);
      if (sl < kLStampSlots) {
        for (int i = 0; i < kLStampN; ++i) g_lstamp[sl][i] = stv[i];
        g_lstamp[sl][kLStampN] = blockIdx.x; g_lstamp[sl][kLStampN + 1] = wave; g_lstamp[sl][kLStampN + 2] = g;
        g_lstamp[sl][kLStampN + 3] = (unsigned int)(gridDim.x * 4 + (first_pass ? 1 : 0) + (en.pad[2] ? 2 : 0));
      }
    }
#endif
    ++g;
    if (g >= ngroups) break;
    if (next_late) {
      // the next pass reads from memory what this one wrote (a receiver block, a sepset): every store complete first
      __syncthreads();
      issue_snd(nx, npr, cs);
      resident(cs);
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    en = nx;
    pr = npr;
    first_pass = false;
  }
}

namespace {

template <int P>
void launch_loop_p(const DevState& S, const FEntry* d_recs, const FPro* d_pros, int ngroups, int split, int n_sites,
                   unsigned long long seq_base, unsigned long long stop_a, unsigned long long stop_b, hipStream_t st,
                   const int32_t* d_wg_off, int n_wg) {
  const dim3 grid(d_wg_off ? n_wg : 1, n_sites), block(kTailWaves * 64);
  const size_t lds = sizeof(double) * (size_t)kTailWaves * (size_t)(2 * kLSlotDoubles + kColDoubles);
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
