// Register-resident message kernel for gfx950 (wave64), sepsets of dimension P (2 .. 16; described for P = 16).
//
// One wavefront = one message; the waves of a workgroup = the messages of a few tasks (ordered lists of messages
// sharing a receiver or a sender); every HBM access a 16- or 32-byte-per-lane coalesced vector access issued up front.
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
// marginalize (src/beliefupdates.jl:55-83) = 8 rounds of 2 x 2-blocked symmetric elimination
//     W <- W - X D^-1 X',   X = columns 2R, 2R+1 of W, D = their 2 x 2 pivot block
// which reads exactly what the reference reads: upper(J_I) (the integrated block is symmetrised from its
// upper triangle at load, like PDMat(Symmetric(J_I)) at :68), J_SI (Jki at :59) and J_S.  J_IS is never
// loaded.  The pivot columns go through a wave-private LDS strip.
// log det J_I is accumulated as mantissa product + exponent sum (one log per message).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "pgbp_bs16.hpp"
#include "pgbp_kernels.hpp"

#include "pgbp_fast_dev.hpp"

namespace pgbp {

// LDS hand-over slot of one wave: 2 x 2 block per lane (256 doubles, lane-major), h (16), g, status
constexpr int kSlotJ = 0, kSlotH = 256, kSlotG = 272, kSlotStatus = 273, kSlotDoubles = 288;

extern __shared__ double fast_lds[];

// ---------------------------------------------------------------------------------------------------------------------
// The message kernel.  A launch walks GROUPS of W records (FEntry, pgbp_internal.hpp): wave w of the workgroup runs
// record w of the group; the records of one task (messages sharing a receiver in a postorder, a sender in a preorder)
// are consecutive inside their group:
//   * accumulate tasks (postorder, several children into one receiver block): every wave computes its message and
//     divides; the task's later waves hand their delta to its first wave through LDS, which adds them in the
//     reference's order and stores the receiver block once;
//   * reuse (preorder, one sender, several children): the providing wave computes the marginal once and hands it to the
//     others through LDS; every wave divides by its own sepset and updates its own receiver.
// Every wave of the workgroup executes the same two workgroup barriers per group whatever its record holds (an invalid
// record = a wave with nothing to do: it skips the work, not the barriers).
//
// Two launch modes of the same body:
//   kLevel  one group per workgroup, W = 4: one launch per level of the schedule (pgbp_plan.cpp);
//   kTail   the narrow levels at the root end of the schedule tree: ONE workgroup of 8 waves walks them, one group per
//           level, a workgroup barrier between levels instead of a kernel boundary (what a level hands to the next one
//           never leaves this CU's L2 slice); a postorder's tail and the following preorder's head go out as one launch.
// BS: beliefs / residuals are in the BS16 symmetric block-packed layout (pgbp_bs16.hpp).
// ODD: the sepsets really have PR = P - 1 variables (an odd trait count): same lane grid, one phantom variable per
// block, unit precision where it is integrated so that its pivot is 1 (log det and quadratic term unchanged).
constexpr int kLevel = 0, kTail = 2;   // (1 was the streaming launch of rounds 1 - 2, built and measured slower: DESIGN.md section 4.2)

#ifdef PGBP_STAMP  // experiment builds only (tools/stamp_passes.py): clock stamps of the phases of a pass
constexpr int kStampSlots = 1 << 16, kStampN = 12;
__device__ unsigned int g_stamp[kStampSlots][kStampN + 4];
__device__ unsigned int g_stamp_n;
#define PGBP_ST(i) do { stv[i] = (unsigned int)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PGBP_ST(i) do { } while (0)
#endif
// LDS exchange between the waves of a workgroup: release / acquire fences restricted to the LDS address space around the
// barrier, i.e. s_waitcnt lgkmcnt(0) + s_barrier.  Unlike __syncthreads() this does NOT wait for outstanding global
// loads / stores / LDS-DMA (vmcnt): the prefetch of the next sender stays in flight across it.  (The barrier has to be
// the builtin, which the compiler knows as a convergent operation: an inline-asm s_barrier may be duplicated into the
// two arms of a lane-divergent branch, and a wave that runs both arms then arrives twice.)
__device__ __forceinline__ void wg_barrier_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// level boundary inside the tail launch: this wave's global stores are complete (vmcnt) before the other waves of the
// workgroup -- same CU, same vector L1 -- load them after the barrier: the workgroup-scope release / acquire of
// __syncthreads()
__device__ __forceinline__ void wg_barrier_global() { __syncthreads(); }

// PRO: records may carry a PROLOGUE (kFPro; FPro, pgbp_internal.hpp): a message X -> F that integrates nothing (X's whole
// belief, e.g. a variable cluster of a Bethe graph) and lands exactly on the block of F that this record's own message
// F -> Y integrates out.  The wavefront first sends it -- divide! by the sepset (X, F), store sepset and residual, add the
// delta to F's block in registers and in memory (mult!) -- and goes straight on with F's elimination: the level in which
// X -> F would have run alone (a launch or a loop pass, with its memory round trips) no longer exists.  Same arithmetic,
// same order as the two messages one after the other (src/beliefupdates.jl:650-665 twice).
template <int P, bool BS, bool ODD, int MODE, bool PRO>
// (the level launches live on four waves per SIMD, 128 registers: left to itself the scheduler trades occupancy for them)
__attribute__((amdgpu_waves_per_eu(MODE == kTail ? 2 : 4, 4)))
__global__ __launch_bounds__(MODE == kTail ? kTailWaves * 64 : kFastMaxWaves * 64) void bp_fast16(
    DevState S_arg, const FEntry* __restrict__ recs_arg, const FPro* __restrict__ pros, int ngroups_arg, int split_arg,
    unsigned long long seq_base_arg, unsigned long long stop_a_arg, unsigned long long stop_b_arg,
    const int32_t* __restrict__ wg_off) {
  constexpr int W = MODE == kTail ? kTailWaves : kFastMaxWaves;
  static_assert(!(ODD && BS), "odd dimensions run in the plain layout");
  // The kernarg segment spans three cache lines and the compiler fetches arguments one group at a time, each a
  // dependent round trip on the critical path of a narrow level.  Pinning the plain scalars of all three lines in SGPRs
  // here makes ONE batch of scalar loads touch every line; the pointer arguments are left alone (an asm operand would
  // cost them their global-address-space provenance) and hit the scalar cache.
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
  constexpr int PR = P - (ODD ? 1 : 0);  // the real sepset dimension
  constexpr int G = P / 2;               // lane grid G x G (all 64 lanes for P = 16)
  // STREAMING accesses (round 4, last session): what a calibrate touches once -- the residual and the sepset it writes, the old
  // sepset and a leaf's belief it reads, a postorder's 2P-dim sender -- goes past the caches with the non-temporal hint, so
  // that what IS read again (the receiver blocks: the next level's senders) stays in them: cfg3 0.838 -> 0.796 ms per
  // calibrate, 1 940 -> 2 110 log-likelihood evaluations/s on one box, hint by hint.  Not in the
  // prologue instances (Bethe graphs: a variable cluster is read by every factor around it; cfg2 lost 2 % with the hints).
  constexpr bool kStream = !PRO;
  double* __restrict__ pool = S.pool + (int64_t)site * S.pool_stride;
  double* __restrict__ rpool = S.rpool + (int64_t)site * S.rpool_stride;
  double* const slot = fast_lds + wave * kSlotDoubles;
  double* const col = fast_lds + W * kSlotDoubles + wave * kColDoubles;  // private strip of this wave

  // kTail with wg_off: workgroup b walks the groups [wg_off[b], wg_off[b + 1]) -- its own dependency-closed piece of a
  // run of fused levels (pgbp_plan.cpp: build_chunks); without: the one workgroup walks all of them
  if constexpr (MODE == kTail) {
    if (wg_off) ngroups = wg_off[blockIdx.x + 1];
  }
  int g = MODE == kTail ? (wg_off ? wg_off[blockIdx.x] : 0) : blockIdx.x;
  constexpr int gstride = 1;
  FEntry en = load_record(recs + ((int64_t)g * W + wave));
  FPro pr{};
  if constexpr (PRO) pr = load_pro(pros + ((int64_t)g * W + wave));
  unsigned long long failkey = S.fail[site];

  for (;;) {
#ifdef PGBP_STAMP
    unsigned int stv[kStampN] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PGBP_ST(0);
    // lane geometry, re-derived per pass from an opaque copy of the lane id so that nothing of it is carried in
    // registers across the loop (the loop-free kLevel mode compiles to what it was)
    int lane = threadIdx.x & 63;
    if constexpr (MODE != kLevel) asm volatile("" : "+v"(lane));
    const bool act = lane < G * G;  // lanes beyond the grid shadow lane (0, 0) and never store
    const int a = act ? lane % G : 0, b = act ? lane / G : 0;
    const bool up = act && a <= b;               // this lane's block is stored in the packed layout
    const int kidx = (b * (b + 1) / 2 + a) * 4;  // its offset inside a packed symmetric tile

    const bool has_next = MODE != kLevel && g + gstride < ngroups;
    // kTail: the whole next record is requested here as a VECTOR load (lanes 0 .. 3 fetch 16 bytes each): the compiler
    // tracks it like any other load and waits for it where it is used, at the end of the pass -- a scalar load pinned in
    // SGPRs would be waited for on the spot, a dependent round trip (about 900 clocks) in every level of the tail
    uint4 nxv = make_uint4(0, 0, 0, 0);
    if (MODE == kTail && has_next) {
      if (PRO && (lane & 4))   // (PRO: lanes 4, 5 fetch the 32 bytes of the next record's prologue)
        nxv = reinterpret_cast<const uint4*>(pros + ((int64_t)(g + gstride) * W + wave))[lane & 1];
      else
        nxv = reinterpret_cast<const uint4*>(recs + ((int64_t)(g + gstride) * W + wave))[lane & 3];
    }
    if constexpr (MODE == kTail) {
      // a failure in the postorder part of this launch must stop its preorder part: at the first preorder level the fail
      // word is read again, coherently (every other level keeps the value read at the start: a dependent memory round
      // trip less on the critical path of a level)
      if (g == split && split > 0) {
        failkey = __hip_atomic_load(&S.fail[site], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the builtin returns int: without the unsigned casts a low word with bit 31 set would sign-extend over the high one)
        failkey = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(failkey >> 32)) << 32) |
                  (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)failkey);
      }
    }
    const unsigned long long stop_below = (MODE == kTail && g >= split) ? stop_b : stop_a;

    // state 0: nothing to do / stopped; 1: message available; 2: failed (not PD); 3: sender poisoned
    int state = (en.valid && !((failkey >> kInfoBits) < stop_below)) ? 1 : 0;
    const bool has_block = en.s > 0;                // the message has a J/h part
    const bool own = (en.mode & kFOwn) != 0;        // this wave loads/stores the receiver block
    const bool accum = (en.mode & kFAccum) != 0;
    // the receiver has a J / h block to load and store (an accumulate task of constants only touches its g)
    const bool recv_blk = has_block || (accum && !(en.mode & kFNoBlock));
    const bool provider = en.src_wave == wave;      // computes the marginal itself
    const int first_wave = en.grp_base;             // first wave of this record's task

    double* __restrict__ sep = pool + en.sep_off;
    double* __restrict__ to = pool + en.to_off;
    double* __restrict__ res = rpool + en.res_off;
    const int mt = en.mt, up0 = en.up0;
    // offsets inside the sepset / receiver / residual records
    const bool tpk = BS && (mt == P || mt == 2 * P);                                // receiver record is packed
    const int sepH = BS ? bs16::h1(P) : PR * PR, sepG = has_block ? (BS ? bs16::g1(P) : PR * PR + PR) : 0;
    const int64_t tJ0 = tpk ? ((mt == 2 * P && up0 == P) ? bs16::t11(P) : 0) : (up0 + (int64_t)mt * up0);
    const int64_t tH0 = (tpk ? (mt == P ? bs16::h1(P) : bs16::h2(P)) : (int64_t)mt * mt) + up0;
    const int64_t tG0 = tpk ? (mt == P ? bs16::g1(P) : bs16::g2(P)) : (int64_t)mt * mt + mt;

    Blk mJ{0, 0, 0, 0}, tJ{0, 0, 0, 0}, sJ{0, 0, 0, 0};
    double mh[2] = {0, 0}, gmsg = 0.0, th[2] = {0, 0}, tg = 0.0;
    double2 sh = make_double2(0.0, 0.0);
    double sg = 0.0;
    int info = 0;
    PGBP_ST(1);
    if (state == 1) {
      // (kTail: the mark may have been set by an earlier level of this very launch -- by a wave of this workgroup, i.e.
      // through this CU's own vector L1, behind the level's barrier: a plain load sees it)
      // In the loop modes the load goes through a per-lane (opaque zero) offset: as a wave-uniform value the compiler
      // fetches it and waits for it on the spot -- a dependent memory round trip in front of the operand loads of every
      // level of the tail; like this it is waited for where it is used, behind the elimination.
      int zero_v = 0;
      if constexpr (MODE != kLevel) asm volatile("" : "+v"(zero_v));
      int poison_v = S.poison[(int64_t)site * S.n_clusters + en.from_b + zero_v];
      const bool pro = PRO && (en.mode & kFPro) != 0;
      if (pro) poison_v |= S.poison[(int64_t)site * S.n_clusters + pr.from_b + zero_v];   // X poisoned: so is F
      // ---- every load of this message is issued before any arithmetic; the sender (the largest operand and
      // the one the elimination waits for) goes first, the sepset and the receiver block right behind it
      auto load_sep_to = [&]() {
        if (!S.sep_zero) {
          if (has_block) {
            sJ = load_blk<BS, ODD, kStream>(sep, PR, a, b, up, kidx, PR);
            if (b == 0) sh = load_pair<ODD>(sep + sepH, a, PR);
          }
          sg = sep[sepG];
        }
        if (own) {
          if (recv_blk) {
            tJ = load_blk<BS, ODD>(to + tJ0, mt, a, b, up, kidx, PR);
            if (b == 0) {
              const double2 t2 = load_pair<ODD>(to + tH0, a, PR);
              th[0] = t2.x; th[1] = t2.y;
            }
          }
          tg = to[tG0];
        }
      };
      if (!provider) load_sep_to();
      if (provider) {
        const double* __restrict__ gfrom = pool + en.from_off;
        const double* __restrict__ from = gfrom;
        if (en.mf == 0) {
          gmsg = gfrom[0];  // a constant factor
          load_sep_to();
        } else if (en.mf == PR && has_block) {
          // nothing to integrate: the message is the sender's belief (src/beliefupdates.jl:56)
          mJ = load_blk<BS, ODD, kStream>(from, PR, a, b, up, kidx, PR);
          const double2 ch = load_pair<ODD>(from + (BS ? bs16::h1(P) : PR * PR), a, PR);
          mh[0] = ch.x; mh[1] = ch.y;
          gmsg = from[BS ? bs16::g1(P) : PR * PR + PR];
          load_sep_to();
        } else {
          Frag f;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) f.w[i][j] = 0.0;
          if (en.mf == PR) {
            // everything is integrated (dimension-0 sepset): the 16 x 16 precision is the integrated block
            const Blk v = load_blk<BS, ODD>(from, PR, a, b, up, kidx, PR);
            const double2 vh = load_pair<ODD>(from + (BS ? bs16::h1(P) : PR * PR), a, PR);
            f.w[0][0] = v.x; f.w[1][0] = v.y; f.w[0][1] = v.z; f.w[1][1] = v.w;
            if (ODD && a == G - 1 && b == G - 1) f.w[1][1] = 1.0;  // the phantom variable: decoupled, unit precision
            f.h[0] = vh.x; f.h[1] = vh.y; f.h[2] = 0.0; f.h[3] = 0.0;
            gmsg = from[BS ? bs16::g1(P) : PR * PR + PR];
          } else if constexpr (ODD) {
            // 2 PR-dim sender, odd PR: logical index l in [0, 2P) (integrated block first) -> physical index, -1: phantom
            const int rot = (en.keep0 == 0) ? PR : 0;
            auto phys = [&](int l) -> int {
              const int Kb = l >= P ? 1 : 0, i = l - Kb * P;
              if (i >= PR) return -1;
              const int q = Kb * PR + i + rot;
              return q >= 2 * PR ? q - 2 * PR : q;
            };
            int pr[4], pc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              pr[i] = phys((i & 1) + 2 * a + (i >> 1) * P);
              pc[i] = phys((i & 1) + 2 * b + (i >> 1) * P);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (!(i < 2 && j >= 2) && pr[i] >= 0 && pc[j] >= 0) f.w[i][j] = from[pr[i] + (int64_t)pc[j] * (2 * PR)];
            if (a == G - 1 && b == G - 1) f.w[1][1] = 1.0;  // the phantom integrated variable: decoupled, unit precision
#pragma unroll
            for (int i = 0; i < 4; ++i) f.h[i] = pr[i] >= 0 ? from[4 * PR * PR + pr[i]] : 0.0;
            gmsg = from[4 * PR * PR + 2 * PR];
          } else if constexpr (BS) {
            // 32-dim sender, packed: tiles T00 | T10 | T11.  integrated block = tile 0 (postorder, keep0 = 16)
            // or tile 1 (preorder, keep0 = 0)
            const bool itrail = en.keep0 == 0;
            // (postorder: the sender is read by this message alone and not again before the preorder comes back to it as a
            // receiver -- a streaming load; preorder: the siblings' messages read the same sender -- a plain one)
            Blk ii{0, 0, 0, 0}, ss{0, 0, 0, 0};
            double4 t;
            if (itrail || !kStream) {
              ii = load_blk<true>(from + (itrail ? bs16::t11(P) : 0), P, a, b, up, kidx);
              ss = load_blk<true>(from + (itrail ? 0 : bs16::t11(P)), P, a, b, up, kidx);
              // J_SI block (rows of S = my a, cols of I = my b): block (a, b) of T10, or block (b, a) transposed
              t = *reinterpret_cast<const double4*>(from + bs16::t10(P) + (itrail ? (b + G * a) : (a + G * b)) * 4);
            } else {
              if (up) {
                const pgbp_d4v vi = __builtin_nontemporal_load(reinterpret_cast<const pgbp_d4v*>(from + kidx));
                const pgbp_d4v vs = __builtin_nontemporal_load(reinterpret_cast<const pgbp_d4v*>(from + bs16::t11(P) + kidx));
                ii = Blk{vi.x, vi.y, vi.z, vi.w};
                ss = Blk{vs.x, vs.y, vs.z, vs.w};
              }
              // ... block (a, b) of T10
              const pgbp_d4v tv = __builtin_nontemporal_load(reinterpret_cast<const pgbp_d4v*>(from + bs16::t10(P) + (a + G * b) * 4));
              t = make_double4(tv.x, tv.y, tv.z, tv.w);
            }
            f.w[0][0] = ii.x; f.w[1][0] = ii.y; f.w[0][1] = ii.z; f.w[1][1] = ii.w;
            f.w[2][2] = ss.x; f.w[3][2] = ss.y; f.w[2][3] = ss.z; f.w[3][3] = ss.w;
            f.w[2][0] = t.x; f.w[3][0] = itrail ? t.z : t.y; f.w[2][1] = itrail ? t.y : t.z; f.w[3][1] = t.w;
            const double2 hi = *reinterpret_cast<const double2*>(from + bs16::h2(P) + (itrail ? P : 0) + 2 * a);
            const double2 hs = *reinterpret_cast<const double2*>(from + bs16::h2(P) + (itrail ? 0 : P) + 2 * a);
            f.h[0] = hi.x; f.h[1] = hi.y; f.h[2] = hs.x; f.h[3] = hs.y;
            gmsg = from[bs16::g2(P)];
          } else {
            // logical index = original index rotated so that the integrated block comes first
            const int rot = (en.keep0 == 0) ? P : 0;
            const int r0 = (2 * a + rot) % (2 * P), r1 = (2 * a + P + rot) % (2 * P);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int cl = 2 * b + (j & 1) + (j >> 1) * P;  // logical column C_b(j)
              const int64_t co = (int64_t)((cl + rot) % (2 * P)) * (2 * P);
              if (j < 2) {
                const double2 v = *reinterpret_cast<const double2*>(from + r0 + co);
                f.w[0][j] = v.x; f.w[1][j] = v.y;
              }
              const double2 u = *reinterpret_cast<const double2*>(from + r1 + co);
              f.w[2][j] = u.x; f.w[3][j] = u.y;
            }
            const double2 v = *reinterpret_cast<const double2*>(from + 4 * P * P + r0);
            const double2 u = *reinterpret_cast<const double2*>(from + 4 * P * P + r1);
            f.h[0] = v.x; f.h[1] = v.y; f.h[2] = u.x; f.h[3] = u.y;
            gmsg = from[4 * P * P + 2 * P];
          }
          load_sep_to();
          if (pro) {
            // ---- PROLOGUE: the message X -> F (nothing to integrate: X's belief), then mult! onto F's integrated block
            const double* __restrict__ xfrom = pool + pr.from_off;
            double* __restrict__ sep2 = pool + pr.sep_off;
            double* __restrict__ res2 = rpool + pr.res_off;
            constexpr int xH = BS ? bs16::h1(P) : PR * PR, xG = BS ? bs16::g1(P) : PR * PR + PR;
            const Blk xJ = load_blk<BS, ODD>(xfrom, PR, a, b, up, kidx, PR);
            const double2 xh = load_pair<ODD>(xfrom + xH, a, PR);
            const double xg = xfrom[xG];
            Blk s2{0, 0, 0, 0};
            double2 s2h = make_double2(0.0, 0.0);
            double s2g = 0.0;
            if (!S.sep_zero) {
              s2 = load_blk<BS, ODD>(sep2, PR, a, b, up, kidx, PR);
              s2h = load_pair<ODD>(sep2 + xH, a, PR);
              s2g = sep2[xG];
            }
            // nothing of the prologue is applied from a poisoned X (or F): the whole record is then skipped below
            if (!__builtin_amdgcn_readfirstlane(poison_v)) {
              // divide! (src/beliefupdates.jl:579-587)
              const Blk d2{xJ.x - s2.x, xJ.y - s2.y, xJ.z - s2.z, xJ.w - s2.w};
              const double d2h0 = xh.x - s2h.x, d2h1 = xh.y - s2h.y, d2g = xg - s2g;
              store_blk<BS, ODD>(sep2, PR, a, b, up, act, kidx, xJ, PR);
              store_blk<BS, ODD, true>(res2, PR, a, b, up, act, kidx, d2, PR);
              double maxJ2 = 0.0, maxh2 = 0.0;
              if (BS ? up : act) {
                maxJ2 = fmax(fmax(fabs(d2.x), fabs(d2.y)), fmax(fabs(d2.z), fabs(d2.w)));
                if (d2.x != d2.x || d2.y != d2.y || d2.z != d2.z || d2.w != d2.w) maxJ2 = INFINITY;
              }
              if (act && b == 0) {
                store_pair<ODD>(sep2 + xH, a, xh.x, xh.y, PR);
                store_pair<ODD, true>(res2 + xH, a, d2h0, d2h1, PR);
                maxh2 = (d2h0 != d2h0 || d2h1 != d2h1) ? INFINITY : fmax(fabs(d2h0), fabs(d2h1));
              }
              if (lane == 0) {
                sep2[xG] = xg;
                S.status[(int64_t)site * S.n_msgs + pr.msg] = 0;
              }
              if (S.update_resnorm) {   // iscalibrated_residnorm! (src/beliefs.jl:994-1003)
                const bool all_ok2 = __all(maxh2 <= S.thr_h_p && maxJ2 <= S.thr_J_p);
                if (lane == 0) S.flags[(int64_t)site * S.n_msgs + pr.msg] = all_ok2 ? 1 : 0;
              }
              // mult! (src/beliefupdates.jl:483-488): F's integrated block, in the registers of the elimination ...
              f.w[0][0] += d2.x; f.w[1][0] += d2.y; f.w[0][1] += d2.z; f.w[1][1] += d2.w;
              f.h[0] += d2h0; f.h[1] += d2h1;
              gmsg += d2g;
              // ... and in F's record: F's belief holds the message it received, as after the two-level schedule
              double* __restrict__ fw = pool + en.from_off;
              const int i0 = (en.keep0 == 0) ? PR : 0;   // first integrated variable of F (plain layouts)
              double* __restrict__ iJ = BS ? fw + (en.keep0 == 0 ? bs16::t11(P) : 0) : fw + i0 + (int64_t)(2 * PR) * i0;
              double* __restrict__ iH = BS ? fw + bs16::h2(P) + (en.keep0 == 0 ? P : 0) : fw + 4 * PR * PR + i0;
              double* __restrict__ iG = BS ? fw + bs16::g2(P) : fw + 4 * PR * PR + 2 * PR;
              // (ODD: the phantom corner holds the unit pivot in registers only: store_blk skips indices >= PR)
              store_blk<BS, ODD>(iJ, 2 * PR, a, b, up, act, kidx, Blk{f.w[0][0], f.w[1][0], f.w[0][1], f.w[1][1]}, PR);
              if (act && b == 0) store_pair<ODD>(iH, a, f.h[0], f.h[1], PR);
              if (lane == 0) iG[0] = gmsg;
            }
          }
          // Symmetric(J_I): entries below the diagonal take the value of their transpose (:68)
          // (in BS16 the lanes a > b hold nothing yet: all four of their entries come from lane (b, a))
          {
            const int tl = a * G + b;  // lane holding the transposed 2 x 2 block
            const double t00 = __shfl(f.w[0][0], tl), t01 = __shfl(f.w[1][0], tl);
            const double t10 = __shfl(f.w[0][1], tl), t11 = __shfl(f.w[1][1], tl);
            // the "fake"-message test below must see the RAW lower triangle in the plain layout
            bool nzraw = false;
            if constexpr (!BS) {
#pragma unroll
              for (int i = 0; i < 2; ++i) nzraw |= fabs(f.w[i][0]) > PGBP_EPS || fabs(f.w[i][1]) > PGBP_EPS;
            }
            if (2 * a + 0 > 2 * b + 0) f.w[0][0] = t00;
            if (2 * a + 0 > 2 * b + 1) f.w[0][1] = t01;
            if (2 * a + 1 > 2 * b + 0) f.w[1][0] = t10;
            if (2 * a + 1 > 2 * b + 1) f.w[1][1] = t11;
            // "fake" message: J_I, J_SI, h_I all ~ 0 (src/beliefupdates.jl:62-66)
            bool nz = nzraw || fabs(f.h[0]) > PGBP_EPS || fabs(f.h[1]) > PGBP_EPS;
#pragma unroll
            for (int i = 0; i < 4; ++i) nz |= fabs(f.w[i][0]) > PGBP_EPS || fabs(f.w[i][1]) > PGBP_EPS;
            if (__any(nz)) {
              double mant = 1.0, quad = 0.0;
              int expo = 0;
              PGBP_ST(2);
              info = eliminate2<P, 0>(f, a, b, act, col, mant, expo, quad);
              PGBP_ST(3);
              if (info == 0) {
                const double logdet = log_by_table(S.logtab, mant) + (double)expo * PGBP_LN2;
                gmsg += 0.5 * ((double)PR * PGBP_LOG2PI - logdet + quad);  // :81
              }
            }
          }
          mJ = Blk{f.w[2][2], f.w[3][2], f.w[2][3], f.w[3][3]};
          mh[0] = f.h[2]; mh[1] = f.h[3];
        }
      }
      if (__builtin_amdgcn_readfirstlane(poison_v)) state = 3;
      else if (info != 0) state = 2;
    }
    // ---- hand the marginal over to the waves that reuse it
    if (provider && en.valid && !accum && en.grp_len > 1) {
      if (state == 1) {
        *reinterpret_cast<double4*>(slot + kSlotJ + 4 * lane) = make_double4(mJ.x, mJ.y, mJ.z, mJ.w);
        if (act && b == 0) *reinterpret_cast<double2*>(slot + kSlotH + 2 * a) = make_double2(mh[0], mh[1]);
      }
      if (lane == 0) {
        slot[kSlotG] = gmsg;
        slot[kSlotStatus] = (double)state;
      }
    }
    PGBP_ST(4);
    wg_barrier_lds();
    PGBP_ST(5);
    if (!provider && state == 1) {
      const double* src = fast_lds + en.src_wave * kSlotDoubles;
      const int pst = (int)src[kSlotStatus];
      if (pst == 1) {
        const double4 v = *reinterpret_cast<const double4*>(src + kSlotJ + 4 * lane);
        mJ = Blk{v.x, v.y, v.z, v.w};
        if (b == 0) {
          const double2 u = *reinterpret_cast<const double2*>(src + kSlotH + 2 * a);
          mh[0] = u.x; mh[1] = u.y;
        }
        gmsg = src[kSlotG];
      } else {
        state = 3;  // the marginal it depends on failed or was skipped: as good as a poisoned sender
      }
    }
    // ---- divide! (src/beliefupdates.jl:579-587): every wave for its own sepset
    Blk dJ{0, 0, 0, 0};
    double dh0 = 0.0, dh1 = 0.0, dg = 0.0;
    if (state == 1) {
      double maxJ = 0.0, maxh = 0.0;
      if (has_block) {
        dJ = Blk{mJ.x - sJ.x, mJ.y - sJ.y, mJ.z - sJ.z, mJ.w - sJ.w};
        store_blk<BS, ODD, kStream>(sep, PR, a, b, up, act, kidx, mJ, PR);
        store_blk<BS, ODD, true>(res, PR, a, b, up, act, kidx, dJ, PR);
        if (BS ? up : act) {
          maxJ = fmax(fmax(fabs(dJ.x), fabs(dJ.y)), fmax(fabs(dJ.z), fabs(dJ.w)));
          if (dJ.x != dJ.x || dJ.y != dJ.y || dJ.z != dJ.z || dJ.w != dJ.w) maxJ = INFINITY;
        }
        if (act && b == 0) {
          dh0 = mh[0] - sh.x; dh1 = mh[1] - sh.y;
          store_pair<ODD, kStream>(sep + sepH, a, mh[0], mh[1], PR);
          store_pair<ODD, true>(res + (BS ? bs16::h1(P) : PR * PR), a, dh0, dh1, PR);
          maxh = (dh0 != dh0 || dh1 != dh1) ? INFINITY : fmax(fabs(dh0), fabs(dh1));
        }
      }
      dg = gmsg - sg;
      if (lane == 0) {
        sep[sepG] = gmsg;
        S.status[(int64_t)site * S.n_msgs + en.msg] = 0;
      }
      if (S.update_resnorm) {
        // iscalibrated_residnorm! (src/beliefs.jl:994-1003); an empty message is calibrated
        // x -> fl(x / c) is monotone, so "max over the wave, divide, compare" equals "every lane divides and
        // compares its own maximum": one ballot instead of two 6-step wave reductions on the critical path
        const bool lane_ok = maxh <= S.thr_h_p && maxJ <= S.thr_J_p;   // (DevState::thr: no division here)
        const bool all_ok = __all(lane_ok);
        if (lane == 0) S.flags[(int64_t)site * S.n_msgs + en.msg] = (!has_block || all_ok) ? 1 : 0;
      }
    } else if (state >= 2 && lane == 0) {
      // not positive definite, or downstream of a failure: nothing of this message is applied
      S.poison[(int64_t)site * S.n_clusters + en.to_b] = 1;
      if (state == 2) {
        S.status[(int64_t)site * S.n_msgs + en.msg] = info;
        atomicMin(&S.fail[site], ((seq_base + (unsigned long long)en.seq) << kInfoBits) | (unsigned long long)info);
      }
    }
    // ---- mult! (src/beliefupdates.jl:483-488)
    if (accum && wave > first_wave) {
      // the task's later waves publish their delta; its first wave adds them in the reference's order
      if (state == 1) {
        *reinterpret_cast<double4*>(slot + kSlotJ + 4 * lane) = make_double4(dJ.x, dJ.y, dJ.z, dJ.w);
        if (act && b == 0) *reinterpret_cast<double2*>(slot + kSlotH + 2 * a) = make_double2(dh0, dh1);
      }
      if (lane == 0) {
        slot[kSlotG] = dg;
        slot[kSlotStatus] = (double)state;
      }
    }
    PGBP_ST(7);
    wg_barrier_lds();
    PGBP_ST(8);
    FEntry nx{};
    if (MODE == kTail && has_next) {
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
        __builtin_memcpy(&pr, w8, sizeof(FPro));
      }
    }
    PGBP_ST(9);
    if (state == 1) {
      tJ = Blk{tJ.x + dJ.x, tJ.y + dJ.y, tJ.z + dJ.z, tJ.w + dJ.w};
      th[0] += dh0; th[1] += dh1;
      tg += dg;
      if (accum && wave == first_wave) {
        for (int w = 1; w < en.grp_len; ++w) {
          const double* src = fast_lds + (first_wave + w) * kSlotDoubles;
          if ((int)src[kSlotStatus] != 1) break;  // the reference stops at the first failing message
          const double4 v = *reinterpret_cast<const double4*>(src + kSlotJ + 4 * lane);
          tJ = Blk{tJ.x + v.x, tJ.y + v.y, tJ.z + v.z, tJ.w + v.w};
          if (b == 0) {
            const double2 u = *reinterpret_cast<const double2*>(src + kSlotH + 2 * a);
            th[0] += u.x; th[1] += u.y;
          }
          tg += src[kSlotG];
        }
      }
      if (own) {
        if (recv_blk) {
          store_blk<BS, ODD>(to + tJ0, mt, a, b, up, act, kidx, tJ, PR);
          if (act && b == 0) store_pair<ODD>(to + tH0, a, th[0], th[1], PR);
        }
        if (lane == 0) to[tG0] = tg;
      }
    }
    PGBP_ST(10);
#ifdef PGBP_STAMP
    if ((blockIdx.x & 63) == 0 && (threadIdx.x & 63) == 0) {  // a sample: every 64th workgroup
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stv[11] = (unsigned int)__builtin_amdgcn_s_memtime();
      const unsigned int slot = atomicAdd(&g_stamp_n, 1u);
      if (slot < kStampSlots) {
        for (int i = 0; i < kStampN; ++i) g_stamp[slot][i] = stv[i];
        g_stamp[slot][kStampN] = blockIdx.x; g_stamp[slot][kStampN + 1] = wave; g_stamp[slot][kStampN + 2] = g;
        g_stamp[slot][kStampN + 3] = (unsigned int)(MODE * 1000000 + gridDim.x);
      }
    }
#endif
    if constexpr (MODE == kLevel) break;
    g += gstride;
    if (g >= ngroups) break;
    // the hand-over slots are reused by the next pass: every wave has finished reading them (kTail: and every global
    // store of this level is complete before the next level's loads)
    wg_barrier_global();
    en = nx;
  }
}

namespace {

size_t fast_lds_bytes(int mode) {
  const int W = mode == kTail ? kTailWaves : kFastMaxWaves;
  return sizeof(double) * (size_t)W * (size_t)(kSlotDoubles + kColDoubles);
}

template <int P, bool ODD>
void launch_fast_p(const DevState& S, const FEntry* d_recs, const FPro* d_pros, int mode, int ngroups, int split, int n_sites,
                   unsigned long long seq_base, unsigned long long stop_a, unsigned long long stop_b, hipStream_t st,
                   const int32_t* d_wg_off, int n_wg) {
  const bool tail = mode == kTail;
  const dim3 grid(tail ? (d_wg_off ? n_wg : 1) : ngroups, n_sites), block((tail ? kTailWaves : kFastMaxWaves) * 64);
  const size_t lds = fast_lds_bytes(mode);
  const int sp = tail ? split : ngroups;
  const int32_t* wg = tail ? d_wg_off : nullptr;
#define PGBP_GO(BSV, ODV, MD, PROV) \
  hipLaunchKernelGGL((bp_fast16<P, BSV, ODV, MD, PROV>), grid, block, lds, st, S, d_recs, d_pros, ngroups, sp, seq_base, stop_a, stop_b, wg)
#define PGBP_GO2(BSV, ODV)                                  \
  do {                                                      \
    if (tail) { if (d_pros) PGBP_GO(BSV, ODV, kTail, true); else PGBP_GO(BSV, ODV, kTail, false); } \
    else { if (d_pros) PGBP_GO(BSV, ODV, kLevel, true); else PGBP_GO(BSV, ODV, kLevel, false); }    \
  } while (0)
  if constexpr (!ODD) {
    if (S.bs16) {
      PGBP_GO2(true, false);
      return;
    }
  }
  PGBP_GO2(false, ODD);
#undef PGBP_GO2
#undef PGBP_GO
}

}  // namespace

// ---- assignfactors! for MvFullBrownianMotion on a tree (pgbp_bm_tree of include/pgbp.h), lane-blocked ---------------
// One wavefront per cluster, same lane geometry as the message kernel: lane (a, b) holds the 2 x 2 block
// R^-1[2a..2a+1][2b..2b+1] for the whole launch and writes it, scaled by 1/t and signed, into the tiles of the record
// (32-byte stores, whole lines), h = R^-1 v / t by a butterfly over b, g from v'R^-1 v / t by a butterfly over a.
// Formulas: pgbp_kernels.hip, bm_tree_fill_kernel (the general-dimension version of the same fill).
// clusters per wavefront of bm_tree_fill_fast (1 / 2 / 3 / 4 / 8 / 16 / 32 / 64: 1 970 / 1 990 / 2 000 / 1 990 / 1 968 / 1 953 / 1 949 /
// 1 893 log-likelihood evaluations per second on cfg3: the records want many wavefronts more than they want their descriptions
// in one load)
constexpr int kFillChunk = 2;
template <int P, bool BS, bool ODD>
__global__ __launch_bounds__(256) void bm_tree_fill_fast(double* __restrict__ pool, int64_t pool_stride,
                                                         double* __restrict__ fpool, int64_t fpool_stride,
                                                         const int64_t* __restrict__ boff,
                                                         const int32_t* __restrict__ dim,
                                                         const int32_t* __restrict__ kind,
                                                         const double2* __restrict__ ithl,
                                                         const int32_t* __restrict__ row,
                                                         const double* __restrict__ data, int n_rows,
                                                         const double* __restrict__ Rinv_all,
                                                         const double* __restrict__ logdetR_all,
                                                         const double* __restrict__ mu_all, int per_site, int n_clusters,
                                                         int chunk) {
  static_assert(!(ODD && BS), "odd dimensions are filled in the plain layout");
  constexpr int PR = P - (ODD ? 1 : 0);  // the real trait count (ODD: the lane grid of P with a phantom index PR)
  constexpr int G = P / 2;
  [[maybe_unused]] __shared__ double obuf[BS ? 4 : 1][BS ? (bs16::g2(P) + 1 + 15) / 16 * 16 : 1];   // a wavefront's record (the 2P one: 577 doubles for P = 16)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, site = blockIdx.y;
  const bool act = lane < G * G;
  const int a = act ? lane % G : 0, b = act ? lane / G : 0;
  const bool up = act && a <= b;
  const int kidx = (b * (b + 1) / 2 + a) * 4;
  const double* __restrict__ Rinv = Rinv_all + (per_site ? (int64_t)site * PR * PR : 0);
  const double* __restrict__ mu = mu_all + (per_site ? (int64_t)site * PR : 0);
  const double g_base = -0.5 * ((double)PR * PGBP_LOG2PI + logdetR_all[per_site ? site : 0]);
  const Blk Rb = load_blk<false, ODD>(Rinv, PR, a, b, up, kidx, PR);
  auto vec = [&](const double* __restrict__ v, int i) { return (!ODD || i < PR) ? v[i] : 0.0; };
  // A wavefront fills `chunk` consecutive clusters.  Their descriptions -- kind, dimension, record offset, (1 / t, (p / 2) log t),
  // data row -- come in with ONE coalesced load each (lane l: cluster c0 + l) and are handed out by v_readlane, and the absorbed
  // vector of cluster i + 1 is requested before cluster i is formed.
  const int c0 = (blockIdx.x * 4 + wave) * chunk;   // (chunk <= 64: a lane per cluster description)
  if (c0 >= n_clusters) return;
  const int nloc = n_clusters - c0 < chunk ? n_clusters - c0 : chunk;
  const int cl = c0 + (lane < nloc ? lane : 0);
  const int kind_v = kind[cl], dim_v = dim[cl], row_v = row[cl];
  const long long boff_v = boff[cl];
  const double2 il_v = ithl[cl];
  auto absorbed = [&](int kk, int rw, double (&v)[4]) {   // v = {vb0, vb1, va0, va1} of a cluster of kind kk >= 1
    const double* __restrict__ y = (kk >= 2) ? data + ((int64_t)site * n_rows + rw) * PR : mu;
    v[0] = vec(y, 2 * b); v[1] = vec(y, 2 * b + 1); v[2] = vec(y, 2 * a); v[3] = vec(y, 2 * a + 1);
    if (kk == 3) { v[0] -= vec(mu, 2 * b); v[1] -= vec(mu, 2 * b + 1); v[2] -= vec(mu, 2 * a); v[3] -= vec(mu, 2 * a + 1); }
  };
  double vnext[4] = {0.0, 0.0, 0.0, 0.0};
  {
    const int k_first = __builtin_amdgcn_readlane(kind_v, 0);
    if (k_first >= 1) absorbed(k_first, __builtin_amdgcn_readlane(row_v, 0), vnext);
  }
  for (int i = 0; i < nloc; ++i) {
    const int k = __builtin_amdgcn_readlane(kind_v, i), m = __builtin_amdgcn_readlane(dim_v, i);
    const long long bo = (long long)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(boff_v >> 32), i) << 32) |
                                     (unsigned int)__builtin_amdgcn_readlane((int)boff_v, i));
    double* __restrict__ rec = pool + (int64_t)site * pool_stride + bo;
    double* __restrict__ frec = fpool ? fpool + (int64_t)site * fpool_stride + bo : nullptr;
    const double vcur[4] = {vnext[0], vnext[1], vnext[2], vnext[3]};
    if (i + 1 < nloc) {
      const int k_next = __builtin_amdgcn_readlane(kind_v, i + 1);
      if (k_next >= 1) absorbed(k_next, __builtin_amdgcn_readlane(row_v, i + 1), vnext);
    }
    if (k < 0) {  // no factor: the constant function 1
      const int len = (BS && bs16::applies(m, P)) ? bs16::rec_len(m, P) : m * m + m + 1;
      for (int t = lane; t < len; t += kWave) {
        rec[t] = 0.0;
        if (frec) frec[t] = 0.0;
      }
      continue;
    }
    // (1 / t, (p / 2) log t) of the cluster's branch: formed once, when the tree is set up (bm_ithl_kernel) -- a division and a
    // log() less in every wavefront of every evaluation
    auto lane_f64 = [&](double x) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
      const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)bits, i);
      const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(bits >> 32), i);
      return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    const double it = lane_f64(il_v.x);
    double g = g_base - lane_f64(il_v.y);
    double jv0 = 0.0, jv1 = 0.0;
    if (k >= 1) {
      // absorbed vector v: mu on the parent (1), the tip's data on the child (2), their difference (3)
      const double vb0 = vcur[0], vb1 = vcur[1], va0 = vcur[2], va1 = vcur[3];
      const double t0 = act ? fma(Rb.x, vb0, Rb.z * vb1) : 0.0, t1 = act ? fma(Rb.y, vb0, Rb.w * vb1) : 0.0;
      // rows 2a, 2a+1 of R^-1 v: the sum over the G lanes (a, 0 .. G-1), in a fixed order (any G, not only powers of 2)
      double p0 = 0.0, p1 = 0.0;
#pragma unroll
      for (int bb = 0; bb < G; ++bb) { p0 += __shfl(t0, bb * G + a); p1 += __shfl(t1, bb * G + a); }
      jv0 = p0 * it;
      jv1 = p1 * it;
      const double qa = fma(jv0, va0, jv1 * va1);  // this lane's rows; the same on every lane of a given a
      double q = 0.0;
#pragma unroll
      for (int aa = 0; aa < G; ++aa) q += __shfl(qa, aa);
      g -= 0.5 * q;
    }
    const Blk Jp{Rb.x * it, Rb.y * it, Rb.z * it, Rb.w * it}, Jm{-Jp.x, -Jp.y, -Jp.z, -Jp.w};
    if constexpr (BS) {
      // The packed record is put together in LDS and goes out as ONE contiguous stream of 16 bytes per lane: every store
      // instruction then covers whole 128-byte lines.  (A lane's 2 x 2 block is 32 bytes = two 16-byte stores at a stride of
      // 32 bytes between lanes: each instruction left half of every line it touched to the other, the L2 took twice the write
      // requests, and the kernel ran at 3.3 TB/s where a plain fill of the same 292 MB reaches 6.1.)
      double* r = obuf[wave];
      int rl = 1;
      if (k == 0) {  // 2P x 2P: [j -j; -j j], h = 0
        store_blk<true>(r, P, a, b, up, act, kidx, Jp);
        store_blk<true>(r + bs16::t11(P), P, a, b, up, act, kidx, Jp);
        if (act) *reinterpret_cast<double4*>(r + bs16::t10(P) + (a + G * b) * 4) = make_double4(Jm.x, Jm.y, Jm.z, Jm.w);
        if (act && b == 0) {
          *reinterpret_cast<double2*>(r + bs16::h2(P) + 2 * a) = make_double2(0.0, 0.0);
          *reinterpret_cast<double2*>(r + bs16::h2(P) + P + 2 * a) = make_double2(0.0, 0.0);
        }
        if (lane == 0) r[bs16::g2(P)] = g;
        rl = bs16::g2(P) + 1;
      } else if (k <= 2) {  // P x P block j on the kept variables, h = +j v
        store_blk<true>(r, P, a, b, up, act, kidx, Jp);
        if (act && b == 0) *reinterpret_cast<double2*>(r + bs16::h1(P) + 2 * a) = make_double2(jv0, jv1);
        if (lane == 0) r[bs16::g1(P)] = g;
        rl = bs16::g1(P) + 1;
      } else {  // everything absorbed: a constant
        if (lane == 0) r[0] = g;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int which = 0; which < (frec ? 2 : 1); ++which) {
        double* __restrict__ dst = which ? frec : rec;
        for (int t = lane; t < (rl >> 1); t += kWave) {
          const double2 q = reinterpret_cast<const double2*>(r)[t];
          __builtin_nontemporal_store(pgbp_d2v{q.x, q.y}, reinterpret_cast<pgbp_d2v*>(dst) + t);   // (292 MB at cfg3's size: streamed)
        }
        if ((rl & 1) && lane == 0) dst[rl - 1] = r[rl - 1];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();   // (the next cluster's record overwrites the buffer)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      continue;
    }
    for (int which = 0; which < (frec ? 2 : 1); ++which) {
      double* __restrict__ r = which ? frec : rec;
      if (k == 0) {  // 2P x 2P: [j -j; -j j], h = 0
        if constexpr (BS) {
          store_blk<true>(r, P, a, b, up, act, kidx, Jp);
          store_blk<true>(r + bs16::t11(P), P, a, b, up, act, kidx, Jp);
          if (act) *reinterpret_cast<double4*>(r + bs16::t10(P) + (a + G * b) * 4) = make_double4(Jm.x, Jm.y, Jm.z, Jm.w);
          if (act && b == 0) {
            *reinterpret_cast<double2*>(r + bs16::h2(P) + 2 * a) = make_double2(0.0, 0.0);
            *reinterpret_cast<double2*>(r + bs16::h2(P) + P + 2 * a) = make_double2(0.0, 0.0);
          }
          if (lane == 0) r[bs16::g2(P)] = g;
        } else {
          store_blk<false, ODD>(r, 2 * PR, a, b, up, act, kidx, Jp, PR);
          store_blk<false, ODD>(r + PR, 2 * PR, a, b, up, act, kidx, Jm, PR);
          store_blk<false, ODD>(r + (int64_t)2 * PR * PR, 2 * PR, a, b, up, act, kidx, Jm, PR);
          store_blk<false, ODD>(r + (int64_t)2 * PR * PR + PR, 2 * PR, a, b, up, act, kidx, Jp, PR);
          if (act && b == 0) {
            store_pair<ODD>(r + 4 * PR * PR, a, 0.0, 0.0, PR);
            store_pair<ODD>(r + 4 * PR * PR + PR, a, 0.0, 0.0, PR);
          }
          if (lane == 0) r[4 * PR * PR + 2 * PR] = g;
        }
      } else if (k <= 2) {  // P x P block j on the kept variables, h = +j v
        store_blk<BS, ODD>(r, PR, a, b, up, act, kidx, Jp, PR);
        if (act && b == 0) store_pair<ODD>(r + (BS ? bs16::h1(P) : PR * PR), a, jv0, jv1, PR);
        if (lane == 0) r[BS ? bs16::g1(P) : PR * PR + PR] = g;
      } else {  // everything absorbed: a constant
        if (lane == 0) r[0] = g;
      }
    }
  }
}

template <int P, bool ODD>
static void launch_fill_p(double* pool, int64_t pool_stride, double* fpool, int64_t fpool_stride, const int64_t* d_boff,
                          const int32_t* d_dim, const int32_t* d_kind, const double2* d_length, const int32_t* d_row,
                          const double* d_data, int n_rows, const double* d_Rinv, const double* d_logdetR,
                          const double* d_mu, int per_site, int bs16, int n_clusters, int n_sites, hipStream_t st) {
  const int chunk = kFillChunk;
  const int nchunks = (n_clusters + chunk - 1) / chunk;
  const int gx = (nchunks + 3) / 4;   // four wavefronts per workgroup, kFillChunk clusters per wavefront
  if constexpr (!ODD) {
    if (bs16) {
      hipLaunchKernelGGL((bm_tree_fill_fast<P, true, false>), dim3(gx, n_sites), dim3(256), 0, st, pool, pool_stride, fpool,
                         fpool_stride, d_boff, d_dim, d_kind, d_length, d_row, d_data, n_rows, d_Rinv, d_logdetR, d_mu,
                         per_site, n_clusters, chunk);
      return;
    }
  }
  hipLaunchKernelGGL((bm_tree_fill_fast<P, false, ODD>), dim3(gx, n_sites), dim3(256), 0, st, pool, pool_stride, fpool,
                     fpool_stride, d_boff, d_dim, d_kind, d_length, d_row, d_data, n_rows, d_Rinv, d_logdetR, d_mu,
                     per_site, n_clusters, chunk);
}

// (1 / t, (p / 2) log t) per cluster: what every evaluation of the fill needs of a branch length, formed once
__global__ void bm_ithl_kernel(const double* __restrict__ length, const int32_t* __restrict__ kind, int p,
                               double2* __restrict__ out, int n) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double2 v = make_double2(0.0, 0.0);
  if (kind[c] >= 0) {
    const double tlen = length[c];
    v.x = 1.0 / tlen;
    v.y = 0.5 * (double)p * log(tlen);
  }
  out[c] = v;
}
void launch_bm_ithl(const double* d_length, const int32_t* d_kind, int p, double2* d_out, int n, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(bm_ithl_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_length, d_kind, p, d_out, n);
}

bool launch_bm_tree_fill_fast(double* pool, int64_t pool_stride, double* fpool, int64_t fpool_stride,
                              const int64_t* d_boff, const int32_t* d_dim, const int32_t* d_kind, const double2* d_length,
                              const int32_t* d_row, const double* d_data, int n_rows, int p, const double* d_Rinv,
                              const double* d_logdetR, const double* d_mu, int per_site, int bs16, int n_clusters,
                              int n_sites, hipStream_t st) {
  if (n_clusters <= 0) return true;
#define PGBP_FILL(PP, OD)                                                                                              \
  launch_fill_p<PP, OD>(pool, pool_stride, fpool, fpool_stride, d_boff, d_dim, d_kind, d_length, d_row, d_data, n_rows, \
                        d_Rinv, d_logdetR, d_mu, per_site, bs16, n_clusters, n_sites, st);                             \
  return true
  switch (p) {  // the real trait count; odd ones on the next even lane grid
    case 16: PGBP_FILL(16, false);
    case 15: PGBP_FILL(16, true);
    case 14: PGBP_FILL(14, false);
    case 13: PGBP_FILL(14, true);
    case 12: PGBP_FILL(12, false);
    case 11: PGBP_FILL(12, true);
    case 10: PGBP_FILL(10, false);
    case 9: PGBP_FILL(10, true);
    case 8: PGBP_FILL(8, false);
    case 7: PGBP_FILL(8, true);
    case 6: PGBP_FILL(6, false);
    case 5: PGBP_FILL(6, true);
    case 4: PGBP_FILL(4, false);
    case 3: PGBP_FILL(4, true);
    case 2: PGBP_FILL(2, false);
    default: return false;
  }
#undef PGBP_FILL
}

// The kernel is instantiated for every even sepset dimension P <= 16 ((P/2)^2 lanes: all 64 for P = 16, 16 for P = 8,
// 1 for P = 2), and each instance once more for the odd dimension P - 1 (plain layout, a phantom variable per block);
// the small-P instances trade lane utilisation for the same per-level latency (one wave per message).
// mode: kLevel (0) one group of kFastMaxWaves records per workgroup; kTail (2) one workgroup walks `ngroups` groups of
// kTailWaves records, groups >= split stop below stop_b instead of stop_a (a postorder's tail followed by a preorder's
// head); with d_wg_off (n_wg + 1 group offsets) n_wg workgroups each walk their own range of groups (a chunk of fused
// levels).  d_pros: the prologues of the records (same indexing), or null if none of them has one.
void launch_fast16(const DevState& S, const FEntry* d_recs, const FPro* d_pros, int mode, int ngroups, int split, int n_sites,
                   unsigned long long seq_base, unsigned long long stop_a, unsigned long long stop_b, hipStream_t st,
                   const int32_t* d_wg_off, int n_wg) {
  if (ngroups <= 0) return;
#ifdef PGBP_ONLY_P16  // experiment builds: one instance, seconds to compile
  if (S.fast_p == 16) launch_fast_p<16, false>(S, d_recs, d_pros, mode, ngroups, split, n_sites, seq_base, stop_a, stop_b, st, d_wg_off, n_wg);
#else
#define PGBP_FAST(PP, OD) launch_fast_p<PP, OD>(S, d_recs, d_pros, mode, ngroups, split, n_sites, seq_base, stop_a, stop_b, st, d_wg_off, n_wg); break
  switch (S.fast_p) {  // the real sepset dimension; odd ones run on the next even instance with a phantom variable
    case 16: PGBP_FAST(16, false);
    case 15: PGBP_FAST(16, true);
    case 14: PGBP_FAST(14, false);
    case 13: PGBP_FAST(14, true);
    case 12: PGBP_FAST(12, false);
    case 11: PGBP_FAST(12, true);
    case 10: PGBP_FAST(10, false);
    case 9: PGBP_FAST(10, true);
    case 8: PGBP_FAST(8, false);
    case 7: PGBP_FAST(8, true);
    case 6: PGBP_FAST(6, false);
    case 5: PGBP_FAST(6, true);
    case 4: PGBP_FAST(4, false);
    case 3: PGBP_FAST(4, true);
    case 2: PGBP_FAST(2, false);
    default: break;  // the planner never marks a task fast for another P
  }
#undef PGBP_FAST
#endif
}

}  // namespace pgbp

#ifdef PGBP_STAMP
extern "C" int pgbp_debug_stamps(unsigned int* out, unsigned int cap, unsigned int* n) {
  if (hipDeviceSynchronize() != hipSuccess) return 4;
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(pgbp::g_stamp_n), sizeof(unsigned int)) != hipSuccess) return 1;
  const unsigned int k = *n < cap ? *n : cap;
  if (k && hipMemcpyFromSymbol(out, HIP_SYMBOL(pgbp::g_stamp), sizeof(unsigned int) * (size_t)k * 16) != hipSuccess) return 2;
  unsigned int z = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(pgbp::g_stamp_n), &z, sizeof(z)) != hipSuccess) return 3;
  return 0;
}
#endif
