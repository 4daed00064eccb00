// Internal structures shared by the host planner (pgbp_plan.cpp), the engine
// (pgbp_engine.hip) and the kernels (pgbp_kernels.hip).  Not part of the ABI.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pgbp.h"

namespace pgbp {

constexpr int kRecAlign = 16;  // records padded to 16 doubles = 128 B (one L2 line)
// site-minor layout: element t of belief b of site s at pool[(packed_off[b] + t) * sm_row(n_sites) + s] -- the row of one
// element padded to 32 entries, so that every row of doubles (and of 32-bit words) starts on a 128-byte line whatever the number
// of sites: with 1 000 sites (a rank's share of cfg4 at 8 GPUs) the unpadded rows of 8 000 bytes start mid-line and every
// wavefront's 512-byte access touches five lines instead of four (measured: 6.19 us per problem and sharded step at 1 000
// problems against 5.70 at 1 008 and 5.69 at 1 024: profiles/r04_cfg4_row_alignment.txt)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int64_t sm_row(int64_t n_sites) { return (n_sites + 31) / 32 * 32; }
constexpr int kWave = 64;

// One directed message (sepset k, direction dir); static for the life of the engine.
// Offsets are in doubles inside ONE SITE's belief pool / residual pool.
struct MsgDesc {
  int64_t from_off, to_off, sep_off, res_off;
  int32_t mf, mt, s, ni;           // dims: sender, receiver, sepset; ni = mf - s integrated
  int32_t keep_map, up_map, int_map;  // offsets into the int32 index pool
  int32_t keep0;                   // first keep index if the keep indices are contiguous, else -1
  int32_t up0;                     // first update index if contiguous, else -1
  int32_t from_b, to_b, sep_b;     // belief indices (diagnostics / failure text)
};
static_assert(sizeof(MsgDesc) == 72 || sizeof(MsgDesc) == 80, "MsgDesc layout");

// One message inside a task of a level.
struct Entry {
  int32_t msg;    // directed message id
  int32_t edge;   // position in the tree's edge list
  int32_t reuse;  // 1: same sender and same keep indices as the previous entry of the task
  int32_t seq;    // position in the reference's sequential order of one (post, pre) pair
  int32_t tflags; // fast kernels: bit 0 = load the receiver's block before this entry, bit 1 = store it after
  int32_t pro;    // 1: this entry is the PROLOGUE of the next one (a message X -> F that integrates nothing, sent by the
                  // wavefront that then sends F's own message: build_traversals, bp_fast16)
  int32_t pad[2];
};
constexpr int kTLoad = 1, kTStore = 2;

// One message of a task of the thread-per-site kernel for sepsets of at most one variable (bp_level_uni1 / bp_chunk_uni1),
// everything it needs resolved: entry + descriptor + the three index maps' single indices + the record offsets in both
// layouts (plain: per-site record offsets; site-minor: packed offsets).  A level's launch is five dependent reads long
// otherwise -- task, entries, descriptor, index maps, operands -- and a narrow level IS that chain.  Parallel to
// Traversal::entries (same index).
struct URec {
  int32_t msg, seq, reuse, from_b, to_b;
  int32_t mf, s, mt, ni;
  int32_t k, i0, i1, u;   // keep index (s = 1), integrated indices, update index in the receiver (s = 1)
  int32_t pad[3];
  int64_t from_off, sep_off, to_off, res_off;   // plain layout
  int64_t from_p, sep_p, to_p, res_p;           // site-minor layout
};
static_assert(sizeof(URec) == 128, "URec layout");

// Self-contained record of one message for the wave-per-task kernels (bp_level_generic, bp_chunk_generic in
// pgbp_kernels.hip): descriptor, position in its task and -- for senders of up to kGInlPerm variables and sepsets of
// up to kGInlUp -- the two index maps, in ONE 128-byte line.  A wavefront fetches it with one dword load per lane
// (fields by v_readlane) and two byte loads, so the chain of dependent loads of a message is  record -> operands
// instead of  task -> entry -> descriptor -> index maps -> operands  (about a microsecond per hop on a narrow level).
// Records of one traversal: per level the FIRST record of each of its generic-class tasks, in task order
// (Traversal::level_gbase), then the later records of those tasks, chained through `next`.
constexpr int kGInlPerm = 40, kGInlUp = 16;
constexpr int kSmallI = 8, kSmallK = 8;   // the register-resident small-message body: at most 8 integrated and 8 kept variables
struct GRec {
  int64_t from_off, to_off, sep_off, res_off;  // dwords 0 .. 7: doubles, inside one site's pools
  int32_t msg, seq, from_b, to_b;              // 8 .. 11
  int32_t keep_map, up_map, int_map;           // 12 .. 14: offsets into the index pool (the maps when not inline)
  int32_t next;                                // 15: record of the task's next message, -1: this is the last
  uint8_t mf, mt, s, ni;                       // 16
  uint8_t keep0, up0;                          // 17: first kept / updated index when contiguous, 255: not
  uint8_t reuse;                               //     same sender and kept indices as the previous message of the task
  uint8_t inl;                                 //     bit 0: perm[] holds the sender's order, bit 1: up[] the receiver's positions
  uint8_t perm[kGInlPerm];                     // 18 .. 27: integrated variables first, kept last
  uint8_t up[kGInlUp];                         // 28 .. 31
};
static_assert(sizeof(GRec) == 128 && offsetof(GRec, perm) == 72 && offsetof(GRec, up) == 112, "GRec is one 128-byte record");

// Self-contained message record of the register-resident kernel (pgbp_fast.hip): everything a wavefront
// needs is in ONE 64-byte line, so the dependent-load chain is  record -> data.  The fast-class tasks of a level
// (at most kFastMaxWaves messages each) are packed into GROUPS of W records: one group = one workgroup pass, one
// wavefront per record; a task's records are consecutive inside its group (grp_base .. grp_base + grp_len - 1).
// W = kFastMaxWaves for the level and streaming launches, kTailWaves for the single-workgroup tail launch; records
// beyond a group's last task have valid = 0 (their waves only join the barriers).
struct FEntry {
  int64_t from_off, to_off, sep_off, res_off;  // doubles, inside one site's pools
  int32_t msg, seq, from_b, to_b;
  uint8_t valid;
  uint8_t mf, mt, s;     // dims of sender, receiver, sepset
  uint8_t keep0;         // first kept index in the sender (0 or P)
  uint8_t up0;           // first index of the receiver's block
  uint8_t src_wave;      // wave of the workgroup that computes this record's marginal (== own index unless reused)
  uint8_t mode;          // bit 0: this wave loads+stores the receiver block itself; bit 1: accumulate task; bit 2: kFNoBlock; bit 3: kFPro
                         // (its first wave owns the receiver block, the other waves hand their delta over through LDS)
  uint8_t grp_base;      // wave (record index inside the group) of the first message of this record's task
  uint8_t grp_len;       // number of messages of the task
  // loop launches in the packed layout (pgbp_loop.hip; set by link_chains): pad[0] = CHAIN kind -- what this record's
  // sender (1: the integrated block of a 2P sender, 3: a P-dim sender's whole belief) or its prologue's X (2) received
  // in the previous pass of the walk is taken from the LDS chain slot of wave pad[1] instead of memory; pad[2] != 0: the
  // record's group is LATE (it reads from memory something else the previous pass wrote: full barrier, loads at its top)
  uint8_t pad[6];
};
static_assert(sizeof(FEntry) == 64, "FEntry must be one 64-byte record");
// walks: ranges [first group, one past the last) of groups of kTailWaves records that ONE workgroup walks pass after pass
void link_chains(std::vector<FEntry>& recs, const std::vector<struct FPro>& pros, const std::vector<std::pair<int64_t, int64_t>>& walks,
                 int P);
// Prologue of a record (kFPro): the message X -> F, F = the record's sender, that integrates nothing and lands on the block
// of F the record's own message integrates out (a variable cluster's message into a factor cluster of a Bethe graph).
// What is not here comes from the record: F's record (from_off), the block (keep0), F's belief index (from_b).
struct FPro {
  int64_t from_off, sep_off, res_off;  // X's record, the sepset (X, F), the residual of X -> F; doubles, inside one site's pools
  int32_t msg, from_b;                 // directed message id of X -> F, belief index of X
};
static_assert(sizeof(FPro) == 32, "FPro is half a 64-byte line");
constexpr int kFOwn = 1, kFAccum = 2;
constexpr int kFPro = 8;           // the record has a PROLOGUE (FPro, same index in the parallel array): see bp_fast16
constexpr int kFNoBlock = 4;       // accumulate task whose messages are all constants (dimension-0 sepsets): only the receiver's g
constexpr int kFastMaxWaves = 4;   // messages per fast-class task at most = records per group of a level launch
constexpr int kPassRotate = 3;     // loop launches of the wave-per-task class: slot i of pass k runs on wavefront / pair (i + 3 k) mod kTailWaves
constexpr int kTailWaves = 8;      // records per step of the tail launch (one workgroup of 8 wavefronts)
constexpr int kGenericMaxDim = 64;    // largest sender the wave-per-task generic kernel's lane grids handle
// a message of the large-belief kernel (bp_level_big: descriptors with 32-bit dimensions): its sender is beyond the wave-per-task
// kernel's lane grids, or its receiver beyond the byte fields of a GRec
inline bool big_msg(const MsgDesc& m) { return m.mf > kGenericMaxDim || m.mt > 254; }
constexpr int kChunkMaxTasks = 2400;  // a level joins a chunk of fused levels if it is all fast-class and has at most this many RECORDS (messages):
                                      // with the trees of a chunk's forest packed into at most kChunkBins workgroups (round 4) a pass of the loop
                                      // kernel holds up to 8 records per CU and the chip 2 048 per pass -- 7.6 us per fused level at that width
                                      // against 11.8 - 14.5 us for the level's own launch (cfg3: 384 tasks / 768 / 1 200 / 1 700 / 2 400 / 3 400 / 5 000
                                      // records: 0.840 / 0.823 / 0.831 / 0.836 / 0.811 - 0.832 / 0.831 / 0.807 ms; cfg2 0.331 / 0.321 / 0.317 / 0.317 /
                                      // 0.313 / 0.313 / 0.313: flat from 2 400)
constexpr int kChunkGenericMaxTasks = 4096;  // ... of generic-class tasks (round 4 with bp_chunk_pair, trees packed into kChunkBins workgroups, cfg5 join graph / Bethe: 1 536 / 3 072 / 4 096 / 6 144 / 8 192 tasks: 0.988 / 0.964 / 0.958 / 0.961 / 0.980 and 1.214 / 1.174 / 1.175 / 1.200 / 1.205 ms per iteration; one wavefront per task: 768 / 1 536 / 3 072: 1.142 / 1.103 / 1.102, unpacked 1.168; round 2, unpacked: 384 / 768 / 1 536 / 3 072: 1.559 / 1.541 / 1.552 / 1.690)
constexpr int kChunkUniMaxThreads = 65536;  // ... of a batch of univariate sites (thread-per-site kernels): tasks x sites of a level that joins a chunk
constexpr int kChunkGenericMaxMf = 24;  // generic-class chunks: 8 wavefronts x (perm + mf x (mf + 1)) doubles of LDS per workgroup
constexpr int kChunkDepth = 4;        // levels per chunk
constexpr int kChunkBins = 256;       // workgroups of a chunk launch at most (one per CU): above that the trees of its forest share workgroups
constexpr int kChunkGenericDepth = 6; // ... of generic-class tasks (cfg5 join graph, depth 3 / 4 / 6 / 8: 1.329 / 1.333 / 1.312 / 1.318 ms per iteration)
constexpr int kSmall4MinTasksDefault = 1024;  // level launches of small generic-class tasks: four tasks per wavefront (bp_level_small4) from this width; PGBP_TUNING small4_min=n overrides, -1: never
constexpr size_t kMixedLevelFastMin = 2048;  // fewer fast-class tasks than this in a level that also has generic ones: all generic
constexpr size_t kMixedLevelFastMinNarrow = (size_t)1 << 30;  // ... where the sepsets have at most 4 variables: never split.  The register-resident
                                                   // kernel's instance works on 4 lanes of 64 there; with bp_level_small4 (four tasks per wavefront) the
                                                   // whole level in one launch is the faster (cfg5 on the same box, split from 8 192 / never: join
                                                   // graph 1.382 / 1.366, Bethe 1.579 / 1.534 ms per iteration; before small4: 2 048 / 8 192 / never =
                                                   // 1.817 / 1.780 / 1.787)

// Launch tuning of ONE plan / engine, read ONCE when the plan is built (plan_build) from the single environment variable
// PGBP_TUNING -- comma-separated `key` or `key=value` tokens -- and carried in the plan: what the differential fuzz of the
// launch modes switches (tests/test_gpu_parity.py), plus the few numeric thresholds worth re-sweeping on other trees.
// Everything else that used to be an environment switch of a finished experiment is gone (DESIGN.md section 4 keeps
// the measurements).  Unknown tokens are an error of pgbp_create / pgbp_plan_create.
struct Tuning {
  bool tail = true;            // no_tail: no single-workgroup tail and no chunks: one launch per level
  bool chunks = true;          // no_chunks: no chunks of fused levels
  bool prologues = true;       // no_prologue: Bethe graphs of trees on the two-level schedule (no prologue fusion)
  bool chain_fusion = false;   // chain_fusion: unary clusters passed through inside one task of the wave-per-task kernel (opt-in)
  bool pair2 = true;           // pair=0: chunks of small messages on bp_chunk_generic (one wavefront per task) instead of bp_chunk_pair
  bool loop2 = true;           // loop=0: tail and chunks of the packed layout on bp_fast16's own loop mode (one wavefront per record)
  bool packed_layouts = true;  // plain_layout: keep the ABI's record layout on the device (no BS16, no site-minor)
  long long mixed_fast_min = -1;   // mixed_fast_min=n: a level with both task classes splits from n fast-class tasks on (-1: default)
  int small4_min = kSmall4MinTasksDefault;   // small4_min=n: four tasks per wavefront from n tasks on (-1: never)
  int chunk_bins = -1;         // chunk_bins=n: workgroups of a chunk launch at most (0: one per tree of its forest; -1: kChunkBins)
  int chunk_max_recs = -1, chunk_max_tasks = -1;   // chunk_max_recs / chunk_max_tasks=n: widest fused level (-1: defaults)
  long long chunk_uni_max = -1;   // chunk_uni_max=n: site batches: a level joins a chunk while tasks x sites <= n
  int chunk_depth = -1, chunk_depth_generic = -1;   // chunk_depth / chunk_depth_generic=n: levels per chunk (-1: kChunkDepth / kChunkGenericDepth)
};
// parses PGBP_TUNING; false + message on an unknown token
bool read_tuning(Tuning& t, std::string& err);

struct Traversal {
  std::vector<int32_t> level_off;  // [n_levels+1] -> tasks; inside a level the fast-class tasks come first
  std::vector<int32_t> level_nfast;  // [n_levels] how many of the level's tasks run on the register-resident kernel
  std::vector<int32_t> level_nbig;   // [n_levels] how many of its LAST tasks hold a sender of dimension > 64 (bp_level_big)
  int32_t max_mf_big = 0;            // largest sender dimension among those
  std::vector<FEntry> fentries;      // packed groups of the fast tasks (kFastMaxWaves records each), level after level
  // prologues of the records (kFPro), same indexing as fentries / tentries / centries; has_pro: some record has one
  std::vector<FPro> fpros, tpros, cpros;
  bool has_pro = false;
  std::vector<int64_t> level_fbase;  // [n_levels] first record of the level in fentries
  std::vector<int32_t> level_ngroups;  // [n_levels] groups of the level
  std::vector<int32_t> level_nrecs;    // [n_levels] messages (valid records) of the level's fast tasks
  // TAIL: the run of narrow levels at the root end of the schedule tree (the last tail_levels levels of a postorder, the
  // first of a preorder), every one all fast-class with at most kTailWaves messages: ONE single-workgroup launch walks
  // them with a workgroup barrier per level instead of a kernel boundary.  tentries = tail_levels groups of kTailWaves.
  int32_t tail_levels = 0;
  std::vector<FEntry> tentries;
  // CHUNKS: runs of consecutive narrow all-fast levels below the tail, fused into one launch each.  Inside a chunk the
  // tasks form a forest (a task hangs below the task that carries its receiver's own message -- postorder -- or that
  // delivers its sender's message -- preorder); every tree of it is ONE workgroup of kTailWaves wavefronts that walks
  // its levels one after the other with a workgroup barrier in between, and no workgroup depends on another one of the
  // same launch.  centries = groups of kTailWaves records; chunk c owns the groups [group0, group0 + n_groups) and its
  // workgroup b the groups [wg_off[wg0 + b], wg_off[wg0 + b + 1]) (offsets relative to group0).
  struct Chunk {
    int32_t level0 = 0, level1 = 0;   // levels [level0, level1) of this traversal
    int32_t n_wg = 0, wg0 = 0;        // workgroups; first entry of this chunk in chunk_wg_off (n_wg + 1 entries)
    int64_t group0 = 0;
    int32_t n_groups = 0;
    int32_t generic = 0;              // 1: its groups are kTailWaves task ids each (cgroups), run by bp_chunk_generic
    int32_t max_mf = 0;               // largest sender among its tasks (LDS scratch per wavefront of bp_chunk_generic)
    int32_t small_only = 0;           // generic chunk: every message fits the small-message body (no LDS scratch at all)
  };
  std::vector<Chunk> chunks;
  std::vector<int32_t> chunk_wg_off;
  std::vector<FEntry> centries;
  std::vector<int32_t> cgroups;      // generic chunks: kTailWaves task ids per group (-1: none); Chunk::group0 indexes groups
  std::vector<GRec> grecs;           // records of the generic-class tasks (see GRec)
  std::vector<int32_t> level_gbase;  // [n_levels] record of the level's first generic-class task (its tasks follow in order)
  std::vector<int32_t> task_grec;    // [n_tasks] first record of the task (-1: a fast-class task)
  std::vector<uint8_t> level_small;  // [n_levels] every message of the level's generic-class tasks fits the register-resident small-message body
  // POSTORDER levels whose generic-class tasks are all small and have at most four messages: the level's messages one per
  // ROW of 16 lanes (bp_level_small4<true>) -- pairs (record, position in its task | messages of the task << 8), -1: an
  // empty row; the rows of a task are consecutive and never straddle a wavefront (four rows)
  std::vector<int32_t> rowmap;
  std::vector<int64_t> level_rowbase;  // [n_levels] first row of the level in rowmap (in rows)
  std::vector<int32_t> level_nrows;    // [n_levels] rows of the level (a multiple of four), 0: no row form
  std::vector<int32_t> task_off;   // [n_tasks+1]  -> entries
  std::vector<Entry> entries;
  int32_t max_mf = 0;
};

struct Tree {
  std::vector<int32_t> pa, ch, sep;  // per edge: parent cluster, child cluster, sepset (0-based among sepsets)
  Traversal post, pre;
  // the tail launch's records: the postorder's tail groups followed by the preorder's (one walk: link_chains), and their
  // prologues (same indexing; all zero where no record has one)
  std::vector<FEntry> tail;
  std::vector<FPro> tail_pros;
};

struct Plan {
  int32_t n_clusters = 0, n_sepsets = 0, n_sites = 1, device = 0;
  std::vector<int32_t> dims;
  std::vector<int32_t> sepset_clusters;
  std::vector<int64_t> scope_off;
  std::vector<int32_t> scope_idx;
  // layout
  std::vector<int64_t> boff;        // [n_beliefs+1] padded record offsets (doubles)
  std::vector<int64_t> packed_off;  // [n_beliefs+1] unpadded (ABI "packed")
  std::vector<int64_t> roff;        // [n_msgs+1] padded residual records
  std::vector<int64_t> rpacked_off; // [n_msgs+1]
  std::vector<MsgDesc> msgs;        // [2*n_sepsets]
  std::vector<int32_t> idxpool;
  std::vector<Tree> trees;
  int32_t max_dim = 0;
  Tuning tune;         // PGBP_TUNING as read when the plan was built
  int32_t fast_p = 0;  // sepset dimension the register-resident kernel is instantiated for (0: none)
  bool all_fast = false;  // every task of every scheduled traversal runs on the register-resident kernel
  std::string err;

  int32_t n_beliefs() const { return n_clusters + n_sepsets; }
  int32_t n_msgs() const { return 2 * n_sepsets; }
  int64_t pool_stride() const { return boff.back(); }
  int64_t cluster_stride() const { return boff[n_clusters]; }
  int64_t rpool_stride() const { return roff.back(); }
};

int plan_build(Plan& p, const pgbp_desc* d);
GRec make_grec(const Plan& p, const Entry& en, int32_t next);  // the record of one message (next: see GRec)
int plan_set_schedule(Plan& p, int32_t n_trees, const int32_t* tree_off, const int32_t* pa_j,
                      const int32_t* ch_j);
double plan_bytes_per_calibrate(const Plan& p, int64_t* n_messages);

}  // namespace pgbp

struct pgbp_plan {
  pgbp::Plan p;
};
