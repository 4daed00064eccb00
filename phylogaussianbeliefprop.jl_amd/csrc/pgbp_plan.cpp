// Host-only planner: belief/residual record layout, the static table of directed
// messages with their precomputed scope index maps, and the level-synchronous
// schedule of each spanning tree.  No GPU needed (covered by the CPU test-suite).
//
// Reference behaviour restated here:
//   * scopeindex(sepset, cluster) is computed ONCE per (sepset, side) by the caller
//     instead of twice per message (src/beliefs.jl:389-405, src/beliefupdates.jl:659,663);
//   * integrate_index = setdiff(1:m, keep_index), ascending (src/beliefupdates.jl:52);
//   * postorder = edges in reverse, child -> parent, residual key (pa, ch);
//     preorder = edges in order, parent -> child, key (ch, pa) (src/calibration.jl:121-151).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <unordered_map>

#include "pgbp_internal.hpp"

namespace pgbp {

static int64_t pad_to(int64_t n, int64_t a) { return (n + a - 1) / a * a; }

bool read_tuning(Tuning& t, std::string& err) {
  t = Tuning{};
  const char* v = getenv("PGBP_TUNING");   // the library's one tuning variable (the other getenv: PGBP_RCCL_LIB, pgbp_dist.hip)
  if (!v) return true;
  std::string all(v);
  size_t at = 0;
  while (at <= all.size()) {
    size_t end = all.find(',', at);
    if (end == std::string::npos) end = all.size();
    std::string tok = all.substr(at, end - at);
    at = end + 1;
    while (!tok.empty() && tok.front() == ' ') tok.erase(tok.begin());
    while (!tok.empty() && tok.back() == ' ') tok.pop_back();
    if (tok.empty()) continue;
    const size_t eq = tok.find('=');
    const std::string key = tok.substr(0, eq), val = eq == std::string::npos ? "" : tok.substr(eq + 1);
    const long long num = val.empty() ? 0 : atoll(val.c_str());
    if (key == "no_tail") t.tail = false;
    else if (key == "no_chunks") t.chunks = false;
    else if (key == "no_prologue") t.prologues = false;
    else if (key == "chain_fusion") t.chain_fusion = true;
    else if (key == "pair") t.pair2 = num != 0;
    else if (key == "loop") t.loop2 = num != 0;
    else if (key == "plain_layout") t.packed_layouts = false;
    else if (key == "mixed_fast_min") t.mixed_fast_min = num;
    else if (key == "small4_min") t.small4_min = (int)num;
    else if (key == "chunk_bins") t.chunk_bins = (int)num;
    else if (key == "chunk_max_recs") t.chunk_max_recs = (int)num;
    else if (key == "chunk_max_tasks") t.chunk_max_tasks = (int)num;
    else if (key == "chunk_uni_max") t.chunk_uni_max = num;
    else if (key == "chunk_depth") t.chunk_depth = (int)num;
    else if (key == "chunk_depth_generic") t.chunk_depth_generic = (int)num;
    else {
      err = "PGBP_TUNING: unknown token '" + tok + "'";
      return false;
    }
  }
  return true;
}

int plan_build(Plan& p, const pgbp_desc* d) {
  if (!d || d->n_clusters <= 0 || d->n_sepsets < 0 || !d->dims || d->n_sites < 1) {
    p.err = "invalid description (null pointers, no clusters, or n_sites < 1)";
    return PGBP_ERR_INVALID;
  }
  if (d->n_sepsets > 0 && (!d->sepset_clusters || !d->scope_off || !d->scope_idx)) {
    p.err = "invalid description: sepset arrays missing";
    return PGBP_ERR_INVALID;
  }
  if (!read_tuning(p.tune, p.err)) return PGBP_ERR_INVALID;
  p.n_clusters = d->n_clusters;
  p.n_sepsets = d->n_sepsets;
  p.n_sites = d->n_sites;
  p.device = d->device;
  const int nb = p.n_beliefs();
  p.dims.assign(d->dims, d->dims + nb);
  p.max_dim = 0;
  for (int b = 0; b < nb; ++b) {
    if (p.dims[b] < 0) {
      p.err = "negative belief dimension";
      return PGBP_ERR_INVALID;
    }
    p.max_dim = std::max(p.max_dim, p.dims[b]);
  }
  if (p.max_dim > PGBP_MAX_DIM) {
    p.err = "belief dimension " + std::to_string(p.max_dim) + " exceeds PGBP_MAX_DIM=" +
            std::to_string(PGBP_MAX_DIM);
    return PGBP_ERR_TOO_LARGE;
  }
  if (p.n_sites > 65535 && p.max_dim > 2) {
    // the wavefront-per-message kernels put the site on grid.y (only the thread-per-site kernel of univariate batches,
    // every dimension <= 2, has no such bound)
    p.err = "n_sites = " + std::to_string(p.n_sites) + " exceeds 65535 (split the batch into several engines)";
    return PGBP_ERR_TOO_LARGE;
  }
  p.sepset_clusters.assign(d->sepset_clusters, d->sepset_clusters + 2 * (size_t)p.n_sepsets);
  p.scope_off.assign(d->scope_off, d->scope_off + 2 * (size_t)p.n_sepsets + 1);
  p.scope_idx.assign(d->scope_idx, d->scope_idx + (p.n_sepsets ? p.scope_off.back() : 0));

  // record layout: [J m*m | h m | g] padded to 128 B; clusters first, then sepsets
  p.boff.assign(nb + 1, 0);
  p.packed_off.assign(nb + 1, 0);
  for (int b = 0; b < nb; ++b) {
    int64_t m = p.dims[b];
    int64_t len = m * m + m + 1;
    p.packed_off[b + 1] = p.packed_off[b] + len;
    p.boff[b + 1] = p.boff[b] + pad_to(len, kRecAlign);
  }
  const int nm = p.n_msgs();
  p.roff.assign(nm + 1, 0);
  p.rpacked_off.assign(nm + 1, 0);
  p.msgs.assign(nm, MsgDesc{});
  p.idxpool.clear();
  for (int k = 0; k < p.n_sepsets; ++k) {
    const int a = p.sepset_clusters[2 * k], b = p.sepset_clusters[2 * k + 1];
    const int sb = p.n_clusters + k;
    const int s = p.dims[sb];
    if (a < 0 || a >= p.n_clusters || b < 0 || b >= p.n_clusters || a == b) {
      p.err = "sepset " + std::to_string(k) + ": bad incident clusters";
      return PGBP_ERR_INVALID;
    }
    int32_t mapoff[2], intoff[2], first[2];
    for (int side = 0; side < 2; ++side) {
      const int c = side == 0 ? a : b;
      const int m = p.dims[c];
      const int64_t o0 = p.scope_off[2 * k + side], o1 = p.scope_off[2 * k + side + 1];
      if (o1 - o0 != s) {
        p.err = "sepset " + std::to_string(k) + ": scope index length != sepset dimension";
        return PGBP_ERR_INVALID;
      }
      mapoff[side] = (int32_t)p.idxpool.size();
      int prev = -1;
      bool contig = true;
      for (int64_t t = o0; t < o1; ++t) {
        int v = p.scope_idx[t];
        // src/beliefs.jl:398-401: labels in order, subset, in the cluster's scope
        if (v <= prev || v >= m) {
          p.err = "sepset " + std::to_string(k) + ": scope indices must be strictly increasing and inside the cluster";
          return PGBP_ERR_INVALID;
        }
        if (prev >= 0 && v != prev + 1) contig = false;
        prev = v;
        p.idxpool.push_back(v);
      }
      first[side] = (s > 0 && contig) ? p.scope_idx[o0] : (s == 0 ? 0 : -1);
      // integrate indices = complement, ascending
      intoff[side] = (int32_t)p.idxpool.size();
      int64_t t = o0;
      for (int v = 0; v < m; ++v) {
        if (t < o1 && p.scope_idx[t] == v) {
          ++t;
        } else {
          p.idxpool.push_back(v);
        }
      }
    }
    for (int dir = 0; dir < 2; ++dir) {
      // dir 0: received by a, sent by b
      const int to = dir == 0 ? a : b, from = dir == 0 ? b : a;
      const int sfrom = dir == 0 ? 1 : 0, sto = dir == 0 ? 0 : 1;
      MsgDesc& m = p.msgs[2 * k + dir];
      m.from_off = p.boff[from];
      m.to_off = p.boff[to];
      m.sep_off = p.boff[sb];
      m.mf = p.dims[from];
      m.mt = p.dims[to];
      m.s = s;
      m.ni = m.mf - s;
      m.keep_map = mapoff[sfrom];
      m.int_map = intoff[sfrom];
      m.up_map = mapoff[sto];
      m.keep0 = first[sfrom];
      m.up0 = first[sto];
      m.from_b = from;
      m.to_b = to;
      m.sep_b = sb;
      const int64_t rlen = (int64_t)s * s + s;
      const int id = 2 * k + dir;
      p.rpacked_off[id + 1] = p.rpacked_off[id] + rlen;
      p.roff[id + 1] = p.roff[id] + pad_to(rlen, kRecAlign);
      m.res_off = p.roff[id];
    }
  }
  p.trees.clear();
  // the register-resident kernel has an instance for every sepset dimension from 2 to 16 (pgbp_fast.hip: 2 x 2 blocks
  // per lane; odd dimensions on the next even instance with a phantom variable): pick the one most sepsets have
  p.fast_p = 0;
  {
    int cnt[17] = {0};
    for (int k = 0; k < p.n_sepsets; ++k) {
      const int s = p.dims[p.n_clusters + k];
      if (s >= 2 && s <= 16) ++cnt[s];
    }
    int best = 0;
    for (int q = 16; q >= 2; --q)
      if (cnt[q] > best) { best = cnt[q]; p.fast_p = q; }
  }
  return PGBP_OK;
}

// Shape classes of the register-resident kernel (pgbp_fast.hip), P = sepset dimension it is built for:
//   sender  : dim P, nothing integrated | dim 2P, the other P-block integrated
//             | dim P, everything integrated into a dimension-0 sepset | dim 0 (a constant) into a dimension-0 sepset
//   receiver: dim P or 2P with the sepset on one contiguous P-block; anything (only g is touched) for s = 0.
// This is every message of a BM clique tree / Bethe graph of a tree without missing data
// (m = k*p, s = p, plus the {root} sepset of a fixed-root model).
static bool fast_msg(const MsgDesc& m, int P) {
  if (P <= 0) return false;
  if (m.s == 0) return (m.mf == P || m.mf == 0) && m.mt <= 255;
  if (m.s != P) return false;
  const bool snd = (m.mf == P && m.ni == 0 && m.keep0 == 0) ||
                   (m.mf == 2 * P && m.ni == P && (m.keep0 == 0 || m.keep0 == P));
  const bool rcv = (m.mt == P && m.up0 == 0) || (m.mt == 2 * P && (m.up0 == 0 || m.up0 == P));
  return snd && rcv;
}

// wavefronts a task takes on the register-resident kernel: its messages less the prologues, which ride with the next one
static int task_waves(const Traversal& tr, int t) {
  int n = 0;
  for (int e = tr.task_off[t]; e < tr.task_off[t + 1]; ++e) n += tr.entries[e].pro ? 0 : 1;
  return n;
}

// A PROLOGUE pair (bp_fast16, FPro): `light` = X -> F sends X's whole belief (nothing integrated) and lands exactly on
// the block of F that `heavy` = F -> Y integrates out.
static bool prologue_pair(const MsgDesc& light, const MsgDesc& heavy, int P) {
  return P > 0 && fast_msg(light, P) && fast_msg(heavy, P) && light.to_b == heavy.from_b && light.mf == P && light.s == P &&
         light.ni == 0 && light.mt == 2 * P && heavy.mf == 2 * P && heavy.ni == P && heavy.s == P &&
         light.up0 == (heavy.keep0 == 0 ? P : 0);
}

// Can the whole task run as ONE workgroup of the register-resident kernel, one wave per message?
static bool fast_task(const Plan& p, const Traversal& tr, int t, bool postorder, int* block_up0, int* block_mt) {
  const int e0 = tr.task_off[t], e1 = tr.task_off[t + 1];
  int up0 = -1, mt = -1, first_main = -1, mains = 0;
  for (int e = e0; e < e1; ++e) {
    const MsgDesc& m = p.msgs[tr.entries[e].msg];
    if (!fast_msg(m, p.fast_p)) return false;
    if (tr.entries[e].pro) {  // the prologue of the next entry
      if (e + 1 >= e1 || tr.entries[e + 1].pro || !prologue_pair(m, p.msgs[tr.entries[e + 1].msg], p.fast_p)) return false;
      continue;
    }
    if (first_main < 0) first_main = e;
    if (++mains > kFastMaxWaves) return false;
    // a fused chain (build_traversals) passes through several receivers / senders: generic kernel
    const MsgDesc& m0 = p.msgs[tr.entries[first_main].msg];
    if (postorder ? m.to_b != m0.to_b : m.from_b != m0.from_b) return false;
    if (postorder && m.s > 0) {  // all deltas must land on the same receiver block
      if (up0 >= 0 && (m.up0 != up0 || m.mt != mt)) return false;
      up0 = m.up0;
      mt = m.mt;
    }
  }
  if (mains == 0) return false;
  // (several constant messages into one receiver -- the two root-child factors of a Bethe graph under a fixed root --
  // accumulate into its g alone: block_mt = -1 marks "no block", kFNoBlock in the records)
  *block_up0 = up0 < 0 ? 0 : up0;
  *block_mt = mt;
  return true;
}

// The W-record group of fast-class tasks `members` (sum of their wavefronts <= W): one record per message, a task's
// records consecutive, invalid records behind the last task; a prologue entry goes into the record of the entry behind
// it (kFPro + the parallel array out_pro).
static void append_group(const Plan& p, const Traversal& tr, const std::vector<int>& members,
                         const std::vector<std::pair<int, int>>& blk_of_task, int t_first, bool postorder, int W,
                         std::vector<FEntry>& out, std::vector<FPro>& out_pro, bool* any_pro) {
  int fill = 0;
  for (int t : members) {
    const int e0 = tr.task_off[t], e1 = tr.task_off[t + 1], len = task_waves(tr, t);
    const bool accum = postorder && len > 1;
    const std::pair<int, int>& blk = blk_of_task[t - t_first];
    int provider = fill, w = 0;
    const Entry* pending = nullptr;
    for (int e = e0; e < e1; ++e) {
      const Entry& en = tr.entries[e];
      if (en.pro) {
        pending = &en;
        continue;
      }
      const MsgDesc& m = p.msgs[en.msg];
      FEntry f{};
      FPro fp{};
      f.from_off = m.from_off; f.to_off = m.to_off; f.sep_off = m.sep_off; f.res_off = m.res_off;
      f.msg = en.msg; f.seq = en.seq; f.from_b = m.from_b; f.to_b = m.to_b;
      f.valid = 1;
      f.mf = (uint8_t)m.mf; f.mt = (uint8_t)m.mt; f.s = (uint8_t)m.s;
      f.keep0 = (uint8_t)(m.keep0 < 0 ? 0 : m.keep0);
      f.up0 = (uint8_t)(m.s > 0 ? m.up0 : 0);
      if (!en.reuse) provider = fill + w;
      f.src_wave = (uint8_t)provider;
      f.grp_base = (uint8_t)fill;
      f.grp_len = (uint8_t)len;
      if (accum) {
        // the task's first wave owns the shared receiver block (even if its own message is g-only)
        f.mode = (uint8_t)(kFAccum | (w == 0 ? kFOwn : 0) | (blk.second < 0 ? kFNoBlock : 0));
        if (w == 0 && m.s == 0) {
          f.up0 = (uint8_t)blk.first;
          f.mt = (uint8_t)(blk.second >= 0 ? blk.second : m.mt);
        }
      } else {
        f.mode = kFOwn;
      }
      if (pending) {
        const MsgDesc& x = p.msgs[pending->msg];
        f.mode |= kFPro;
        fp.from_off = x.from_off; fp.sep_off = x.sep_off; fp.res_off = x.res_off;
        fp.msg = pending->msg; fp.from_b = x.from_b;
        pending = nullptr;
        if (any_pro) *any_pro = true;
      }
      out.push_back(f);
      out_pro.push_back(fp);
      ++w;
    }
    fill += len;
  }
  for (; fill < W; ++fill) {
    out.push_back(FEntry{});
    out_pro.push_back(FPro{});
  }
}

// CHAINS and LATE groups of the loop launches (pgbp_loop.hip).  A workgroup walks its groups pass after pass; the operands
// of pass g + 1 are loaded while pass g still runs, so nothing pass g + 1 reads from memory may be written by pass g:
//   * the block a wavefront of pass g accumulates into a cluster that SENDS in pass g + 1 -- the integrated block of a 2P
//     sender (kind 1), a P-dim sender's whole belief (3), the X of a prologue (2) -- travels through that wavefront's LDS
//     chain slot;
//   * any other overlap (a receiver block or a receiver's g, a sepset, the kept block of a sender) makes group g + 1 LATE:
//     the workgroup waits for every store of pass g and loads at the top of pass g + 1;
//   * the first group of a walk is always late.
void link_chains(std::vector<FEntry>& recs, const std::vector<FPro>& pros, const std::vector<std::pair<int64_t, int64_t>>& walks,
                 int P) {
  constexpr int W = kTailWaves;
  struct Wr { int wave; bool block; int mt, up0; };
  for (const auto& wk : walks) {
    for (int64_t g = wk.first; g < wk.second; ++g) {
      FEntry* B = &recs[(size_t)g * W];
      for (int w = 0; w < W; ++w) B[w].pad[0] = B[w].pad[1] = B[w].pad[2] = 0;
      bool late = g == wk.first;
      if (!late) {
        const FEntry* A = &recs[(size_t)(g - 1) * W];
        std::unordered_map<int, Wr> wrote;          // cluster -> the (one) record of pass g - 1 that stores into it
        std::unordered_map<int, int> wrote_n;
        std::set<int64_t> seps;                     // sepsets pass g - 1 stores
        for (int w = 0; w < W; ++w) {
          const FEntry& r = A[w];
          if (!r.valid) continue;
          seps.insert(r.sep_off);
          if (r.mode & kFPro) {
            seps.insert(pros[(size_t)(g - 1) * W + w].sep_off);
            wrote[r.from_b] = Wr{w, false, 0, 0};   // the prologue's F: not chainable
            wrote_n[r.from_b] += 2;
          }
          if (r.mode & kFOwn) {
            const bool blk = r.s > 0 || ((r.mode & kFAccum) && !(r.mode & kFNoBlock));
            wrote[r.to_b] = Wr{w, blk, r.mt, r.up0};
            wrote_n[r.to_b] += 1;
          }
        }
        for (int w = 0; w < W && !late; ++w) {
          FEntry& r = B[w];
          if (!r.valid) continue;
          const bool has_pro = (r.mode & kFPro) != 0;
          const FPro* q = has_pro ? &pros[(size_t)g * W + w] : nullptr;
          if (seps.count(r.sep_off) || (q && seps.count(q->sep_off))) late = true;
          if ((r.mode & kFOwn) && wrote.count(r.to_b)) late = true;
          if (r.src_wave == w) {   // it reads its sender
            auto it = wrote.find(r.from_b);
            if (it != wrote.end()) {
              const Wr& x = it->second;
              const bool one = wrote_n[r.from_b] == 1 && x.block;
              if (one && r.mf == 2 * P && x.mt == 2 * P && x.up0 == (r.keep0 == 0 ? P : 0) && !has_pro) {
                r.pad[0] = 1; r.pad[1] = (uint8_t)x.wave;
              } else if (one && r.mf == P && x.mt == P) {
                r.pad[0] = 3; r.pad[1] = (uint8_t)x.wave;
              } else {
                late = true;
              }
            }
            if (q) {
              auto ix = wrote.find(q->from_b);
              if (ix != wrote.end()) {
                const Wr& x = ix->second;
                if (wrote_n[q->from_b] == 1 && x.block && x.mt == P && r.pad[0] == 0) {
                  r.pad[0] = 2; r.pad[1] = (uint8_t)x.wave;
                } else {
                  late = true;
                }
              }
            }
          }
          // (a record that reuses a sibling's marginal reads nothing of the sender: what the previous pass did to it,
          // a failure included, reaches it through the providing record's state)
        }
      }
      if (late)
        for (int w = 0; w < W; ++w) {
          B[w].pad[0] = B[w].pad[1] = 0;
          B[w].pad[2] = 1;
        }
    }
  }
}

static void build_chunks(const Plan& p, Traversal& tr, bool postorder);
static void build_grecs(const Plan& p, Traversal& tr, bool postorder);

// many tiny problems: every belief dimension <= 2 and at least 8 sites run on the thread-per-site kernels (lanes = sites),
// whatever the class of their tasks (pgbp_engine.hip: enqueue_levels) ...
// ... and their loop mode (bp_chunk_uni1) takes sepsets of at most one variable: such plans get chunks of TASKS over all
// of their narrow levels and no single-workgroup tail
static bool plan_uni(const Plan& p) {
  if (!(p.max_dim <= 2 && p.n_sites >= 8)) return false;
  for (int k = 0; k < p.n_sepsets; ++k)
    if (p.dims[p.n_clusters + k] > 1) return false;
  return true;
}

// Reorder the tasks of every level so that fast-class tasks come first, pack them into groups of kFastMaxWaves records
// (first fit, largest task first: no wave of a workgroup idles beside a shorter task), set the receiver load/store
// flags of the generic tasks, and cut the tail (Traversal::tail_levels).
static void finalize_traversal(const Plan& p, Traversal& tr, bool postorder) {
  const int nlev = (int)tr.level_off.size() - 1;
  std::vector<int32_t> new_task_off{0}, new_level_off{0};
  std::vector<Entry> new_entries;
  new_entries.reserve(tr.entries.size());
  tr.level_nfast.assign(nlev, 0);
  tr.level_nbig.assign(nlev, 0);
  tr.max_mf_big = 0;
  tr.level_fbase.assign(nlev, 0);
  tr.level_ngroups.assign(nlev, 0);
  tr.level_nrecs.assign(nlev, 0);
  tr.fentries.clear();
  tr.tentries.clear();
  tr.fpros.clear();
  tr.tpros.clear();
  tr.has_pro = false;
  tr.tail_levels = 0;
  tr.max_mf = 0;
  std::vector<std::vector<int>> level_fast(nlev);               // fast tasks of each level (old task ids)
  std::vector<std::vector<std::pair<int, int>>> level_blk(nlev);  // (up0, mt) of the receiver block, per task of the level
  for (int L = 0; L < nlev; ++L) {
    std::vector<int> fast, slow;
    const int t_first = tr.level_off[L];
    std::vector<std::pair<int, int>>& blk = level_blk[L];
    blk.assign(tr.level_off[L + 1] - t_first, {0, 0});
    for (int t = t_first; t < tr.level_off[L + 1]; ++t) {
      int up0 = 0, mt = 0;
      if (fast_task(p, tr, t, postorder, &up0, &mt)) {
        fast.push_back(t);
        blk[t - t_first] = {up0, mt};
      } else {
        slow.push_back(t);
      }
    }
    // A level that needs the generic kernel anyway and has too few fast-class tasks to fill the chip runs entirely on
    // the generic kernel: one launch instead of two (a narrow level costs its slowest task plus a kernel boundary per
    // launch: loopy cluster graphs of networks, where hybrid families sit beside tree-edge clusters in every level).
    // (measured: polytomy trees with 8 and 16 traits prefer 2 048 to 8 192 by 2 - 5 %; cfg5, 4 traits -- where the fast
    // kernel's instance works on 4 lanes of 64 -- prefers 8 192 by 2 %)
    const size_t mixed_min = p.tune.mixed_fast_min >= 0 ? (size_t)p.tune.mixed_fast_min : (p.fast_p <= 4 ? kMixedLevelFastMinNarrow : kMixedLevelFastMin);
    if (!slow.empty() && fast.size() < mixed_min) {
      slow.insert(slow.end(), fast.begin(), fast.end());
      std::sort(slow.begin(), slow.end());
      fast.clear();
    }
    // tasks with a sender of more than kGenericMaxDim variables go last: they run on bp_level_big (and so do those whose
    // receiver has more variables than a message record's byte fields hold: big_msg)
    {
      auto is_big = [&](int t) {
        for (int e = tr.task_off[t]; e < tr.task_off[t + 1]; ++e)
          if (big_msg(p.msgs[tr.entries[e].msg])) return true;
        return false;
      };
      std::stable_partition(slow.begin(), slow.end(), [&](int t) { return !is_big(t); });
      for (int t : slow)
        if (is_big(t)) ++tr.level_nbig[L];
    }
    tr.level_nfast[L] = (int32_t)fast.size();
    tr.level_fbase[L] = (int64_t)tr.fentries.size();
    // first-fit-decreasing into groups of kFastMaxWaves wave slots; tasks of one size keep their order
    {
      std::vector<std::vector<int>> by_len(kFastMaxWaves + 1);
      for (int t : fast) by_len[task_waves(tr, t)].push_back(t);
      std::vector<size_t> next(kFastMaxWaves + 1, 0);
      size_t left = fast.size();
      std::vector<int> members;
      while (left > 0) {
        members.clear();
        int room = kFastMaxWaves;
        for (int len = kFastMaxWaves; len >= 1;) {
          if (len <= room && next[len] < by_len[len].size()) {
            members.push_back(by_len[len][next[len]++]);
            room -= len;
            --left;
          } else {
            --len;
          }
        }
        append_group(p, tr, members, blk, t_first, postorder, kFastMaxWaves, tr.fentries, tr.fpros, &tr.has_pro);
        ++tr.level_ngroups[L];
      }
      for (int t : fast) tr.level_nrecs[L] += task_waves(tr, t);
    }
    level_fast[L] = fast;
    for (const auto* grp : {&fast, &slow})
      for (int t : *grp) {
        const int e0 = tr.task_off[t], e1 = tr.task_off[t + 1];
        bool same_block = postorder;
        for (int e = e0 + 1; e < e1 && same_block; ++e) {
          const MsgDesc& x = p.msgs[tr.entries[e0].msg];
          const MsgDesc& y = p.msgs[tr.entries[e].msg];
          same_block = x.to_b == y.to_b && x.up0 == y.up0 && x.s == y.s && x.up0 >= 0;
        }
        bool task_is_big = false;
        for (int e = e0; e < e1; ++e) task_is_big |= big_msg(p.msgs[tr.entries[e].msg]);
        // A generic-class PREORDER task (one sender, a message to each of its children: different receivers, different
        // sepsets, the sender itself untouched) becomes one task per message: a wavefront works through its messages
        // one after the other at several microseconds each, and on the narrow levels that latency is the level's time.
        // (Not in the univariate site-minor engines, whose tasks are spread over sites, not over messages.)
        bool split = grp == &slow && !postorder && !task_is_big && p.max_dim > 2;
        for (int e = e0 + 1; e < e1; ++e)   // (a chain-fused task passes through several senders: its messages depend on each other)
          split = split && p.msgs[tr.entries[e].msg].from_b == p.msgs[tr.entries[e0].msg].from_b;
        for (int e = e0; e < e1; ++e) {
          Entry en = tr.entries[e];
          en.tflags = same_block ? ((e == e0 ? kTLoad : 0) | (e == e1 - 1 ? kTStore : 0)) : (kTLoad | kTStore);
          if (split) en.reuse = 0;
          new_entries.push_back(en);
          if (grp == &slow) {
            const int mfe = p.msgs[en.msg].mf;
            if (task_is_big) tr.max_mf_big = std::max(tr.max_mf_big, mfe);
            else tr.max_mf = std::max(tr.max_mf, mfe);
          }
          if (split && e + 1 < e1) new_task_off.push_back((int32_t)new_entries.size());
        }
        new_task_off.push_back((int32_t)new_entries.size());
      }
    new_level_off.push_back((int32_t)new_task_off.size() - 1);
  }
  // the tail: levels at the root end, all fast-class, at most kTailWaves messages each
  auto tail_ok = [&](int L) {
    const int nt = tr.level_off[L + 1] - tr.level_off[L];
    return nt > 0 && tr.level_nfast[L] == nt && tr.level_nrecs[L] <= kTailWaves;
  };
  if (plan_uni(p)) {
    // (the thread-per-site kernels have no single-workgroup tail: their narrow levels are all fused as chunks)
  } else if (postorder) {
    while (tr.tail_levels < nlev && tail_ok(nlev - 1 - tr.tail_levels)) ++tr.tail_levels;
  } else {
    while (tr.tail_levels < nlev && tail_ok(tr.tail_levels)) ++tr.tail_levels;
  }
  for (int q = 0; q < tr.tail_levels; ++q) {
    const int L = postorder ? nlev - tr.tail_levels + q : q;
    append_group(p, tr, level_fast[L], level_blk[L], tr.level_off[L], postorder, kTailWaves, tr.tentries, tr.tpros, &tr.has_pro);
  }
  tr.task_off.swap(new_task_off);
  tr.entries.swap(new_entries);
  tr.level_off.swap(new_level_off);
  build_grecs(p, tr, postorder);
  build_chunks(p, tr, postorder);
}

GRec make_grec(const Plan& p, const Entry& en, int32_t next) {
  const MsgDesc& m = p.msgs[en.msg];
  GRec r;
  std::memset(&r, 0, sizeof(r));
  r.from_off = m.from_off; r.to_off = m.to_off; r.sep_off = m.sep_off; r.res_off = m.res_off;
  r.msg = en.msg; r.seq = en.seq; r.from_b = m.from_b; r.to_b = m.to_b;
  r.keep_map = m.keep_map; r.up_map = m.up_map; r.int_map = m.int_map;
  r.next = next;
  r.mf = (uint8_t)m.mf; r.mt = (uint8_t)m.mt; r.s = (uint8_t)m.s; r.ni = (uint8_t)m.ni;
  r.keep0 = m.keep0 < 0 ? 255 : (uint8_t)m.keep0;
  r.up0 = m.up0 < 0 ? 255 : (uint8_t)m.up0;
  r.reuse = (uint8_t)(en.reuse != 0);
  if (m.keep0 < 0 && m.mf <= kGInlPerm) {
    for (int i = 0; i < m.ni; ++i) r.perm[i] = (uint8_t)p.idxpool[m.int_map + i];
    for (int i = 0; i < m.s; ++i) r.perm[m.ni + i] = (uint8_t)p.idxpool[m.keep_map + i];
    r.inl |= 1;
  }
  if (m.up0 < 0 && m.s <= kGInlUp) {
    for (int i = 0; i < m.s; ++i) r.up[i] = (uint8_t)p.idxpool[m.up_map + i];
    r.inl |= 2;
  }
  return r;
}

// Records of the generic-class tasks (GRec), on the final task / entry arrays of a traversal.
static void build_grecs(const Plan& p, Traversal& tr, bool postorder) {
  const int nlev = (int)tr.level_off.size() - 1;
  const int ntasks = (int)tr.task_off.size() - 1;
  tr.grecs.clear();
  tr.level_gbase.assign(nlev, 0);
  tr.task_grec.assign(ntasks, -1);
  tr.level_small.assign(nlev, 1);
  for (int L = 0; L < nlev; ++L) {
    const int t0 = tr.level_off[L] + tr.level_nfast[L], t1 = tr.level_off[L + 1];
    tr.level_gbase[L] = (int32_t)tr.grecs.size();
    int32_t later = (int32_t)tr.grecs.size() + (t1 - t0);   // where the tasks' later records go
    for (int t = t0; t < t1; ++t) {
      tr.task_grec[t] = (int32_t)tr.grecs.size();
      const int n = tr.task_off[t + 1] - tr.task_off[t];
      tr.grecs.push_back(make_grec(p, tr.entries[tr.task_off[t]], n > 1 ? later : -1));
      later += n - 1;
    }
    for (int t = t0; t < t1; ++t)
      for (int e = tr.task_off[t] + 1; e < tr.task_off[t + 1]; ++e)
        tr.grecs.push_back(make_grec(p, tr.entries[e], e + 1 < tr.task_off[t + 1] ? (int32_t)tr.grecs.size() + 1 : -1));
    for (int t = t0; t < t1; ++t)
      for (int e = tr.task_off[t]; e < tr.task_off[t + 1]; ++e) {
        const MsgDesc& m = p.msgs[tr.entries[e].msg];
        if (m.ni > kSmallI || m.s > kSmallK) tr.level_small[L] = 0;
      }
  }
  // the row form of the small postorder levels (one message per row of 16 lanes, mult! in task order)
  tr.rowmap.clear();
  tr.level_rowbase.assign(nlev, 0);
  tr.level_nrows.assign(nlev, 0);
  if (!postorder) return;
  for (int L = 0; L < nlev; ++L) {
    const int t0 = tr.level_off[L] + tr.level_nfast[L], t1 = tr.level_off[L + 1];
    if (t1 <= t0 || !tr.level_small[L] || tr.level_nbig[L] != 0) continue;
    bool ok = true;
    std::vector<std::vector<int>> by_len(5);
    for (int t = t0; t < t1 && ok; ++t) {
      const int n = tr.task_off[t + 1] - tr.task_off[t];
      const int32_t to_b = p.msgs[tr.entries[tr.task_off[t]].msg].to_b;
      ok = n >= 1 && n <= 4;
      for (int e = tr.task_off[t]; e < tr.task_off[t + 1] && ok; ++e)   // one receiver, no reused marginal, no prologue
        ok = p.msgs[tr.entries[e].msg].to_b == to_b && tr.entries[e].reuse == 0 && tr.entries[e].pro == 0;
      if (ok) by_len[n].push_back(t);
    }
    if (!ok) continue;
    tr.level_rowbase[L] = (int64_t)(tr.rowmap.size() / 2);
    // first fit, longest task first, into wavefronts of four rows; tasks of one length keep their order
    std::vector<size_t> next(5, 0);
    size_t left = (size_t)(t1 - t0);
    int32_t rows = 0;
    while (left > 0) {
      int room = 4;
      for (int len = 4; len >= 1;) {
        if (len <= room && next[len] < by_len[len].size()) {
          const int t = by_len[len][next[len]++];
          int32_t rec = tr.task_grec[t];
          for (int c = 0; c < len; ++c) {
            tr.rowmap.push_back(rec);
            tr.rowmap.push_back(c | (len << 8));
            rec = tr.grecs[rec].next;
          }
          room -= len;
          --left;
        } else {
          --len;
        }
      }
      for (; room > 0; --room) {
        tr.rowmap.push_back(-1);
        tr.rowmap.push_back(0);
      }
      rows += 4;
    }
    tr.level_nrows[L] = rows;
  }
}

// CHUNKS of fused levels (Traversal::chunks).  Called on the final task / entry arrays of a traversal.
static void build_chunks(const Plan& p, Traversal& tr, bool postorder) {
  tr.chunks.clear();
  tr.chunk_wg_off.clear();
  tr.centries.clear();
  tr.cpros.clear();
  tr.cgroups.clear();
  // (chain fusion makes tasks that pass through several receivers: the forest below assumes one receiver / sender per task)
  const bool off = !p.tune.chunks || !p.tune.tail || p.tune.chain_fusion;
  const int depth_fast = p.tune.chunk_depth >= 2 ? p.tune.chunk_depth : kChunkDepth;
  const int depth_generic = p.tune.chunk_depth_generic >= 2 ? p.tune.chunk_depth_generic : kChunkGenericDepth;
  const int max_tasks = p.tune.chunk_max_recs > 0 ? p.tune.chunk_max_recs : kChunkMaxTasks;
  const int max_tasks_generic = p.tune.chunk_max_tasks > 0 ? p.tune.chunk_max_tasks : kChunkGenericMaxTasks;
  const int nlev = (int)tr.level_off.size() - 1;
  if (off || nlev <= 0) return;
  const int ntasks = (int)tr.task_off.size() - 1;
  std::vector<int32_t> task_level(ntasks, 0);
  for (int L = 0; L < nlev; ++L)
    for (int t = tr.level_off[L]; t < tr.level_off[L + 1]; ++t) task_level[t] = L;
  // the task in which a cluster sends (postorder) / receives (preorder) in this traversal
  std::vector<int32_t> task_of(p.n_clusters, -1);
  for (int t = 0; t < ntasks; ++t)
    for (int e = tr.task_off[t]; e < tr.task_off[t + 1]; ++e) {
      const MsgDesc& m = p.msgs[tr.entries[e].msg];
      task_of[postorder ? m.from_b : m.to_b] = t;
    }
  // levels below the tail that may be fused
  const int lo = postorder ? 0 : tr.tail_levels, hi = postorder ? nlev - tr.tail_levels : nlev;
  // a level may be fused if it is narrow and holds no large-belief task; generic-class tasks need their working matrix
  // eight times over in one workgroup's LDS
  std::vector<int32_t> level_mf(nlev, 0);  // largest sender of the level
  for (int t = 0; t < ntasks; ++t)
    for (int e = tr.task_off[t]; e < tr.task_off[t + 1]; ++e)
      level_mf[task_level[t]] = std::max(level_mf[task_level[t]], p.msgs[tr.entries[e].msg].mf);
  const bool uni = plan_uni(p);   // chunks of TASKS (one wavefront = one task of 64 sites), whatever the tasks' class
  auto all_fast = [&](int L) { return !uni && tr.level_nfast[L] == tr.level_off[L + 1] - tr.level_off[L]; };
  auto eligible = [&](int L) {
    const int nt = tr.level_off[L + 1] - tr.level_off[L];
    // (a level of a site batch joins a chunk while its launch could not fill the chip: tasks x sites threads)
    // (measured, three repetitions each: 65 536 threads best for 2 000 and 8 000 sites; 262 144 another 2 % for 1 000 -- the
    // fewer the sites, the less a level's own launch has to do)
    const long long uni_max = p.tune.chunk_uni_max >= 0 ? p.tune.chunk_uni_max : (long long)kChunkUniMaxThreads * (p.n_sites <= 1024 ? 4 : 1);
    if (uni) return nt > 0 && (long long)nt * p.n_sites <= uni_max;
    // (register-resident levels are measured in RECORDS -- wavefront pairs of a pass --, wave-per-task ones in tasks)
    return nt > 0 && (all_fast(L) ? tr.level_nrecs[L] <= max_tasks : nt <= max_tasks_generic) && tr.level_nbig[L] == 0 &&
           (all_fast(L) || (tr.level_nfast[L] == 0 && level_mf[L] <= kChunkGenericMaxMf));   // (a generic chunk walks message
           // records, which the fast-class tasks of a mixed level do not have)
  };
  auto width = [&](int L) { return tr.level_off[L + 1] - tr.level_off[L]; };
  // levels per chunk, counted from the root-most level (width w0): `depth` of them, and beyond that (up to 4 x depth)
  // only while the levels stay no wider than 2 x w0 -- a chain of narrow levels is as long as it is whatever the launch
  // count, but a workgroup that owns a widening subtree serialises its lower levels (measured: blind depth 16 on the
  // root end of cfg3 costs 0.25 ms)
  auto extend = [&](int depth, int w0, int n, int w) { return n < depth || (n < 4 * depth && w <= 2 * w0); };
  std::vector<std::pair<int, int>> spans;  // chunks as level ranges
  for (int L = lo; L < hi;) {
    if (!eligible(L)) { ++L; continue; }
    int R = L;
    while (R < hi && eligible(R) && all_fast(R) == all_fast(L)) ++R;   // one kernel class per run (and per chunk)
    // the run [L, R): cut into chunks from the root end, so that the odd short chunk is the wide one
    const int depth = (all_fast(L) || uni) ? depth_fast : depth_generic;   // (wave-per-task chunks only: the thread-per-site ones keep 4)
    if (postorder) {
      for (int b = R; b > L;) {
        int a = b - 1;
        while (a > L && extend(depth, width(b - 1), b - a, width(a - 1))) --a;
        spans.push_back({a, b});
        b = a;
      }
    } else {
      for (int a = L; a < R;) {
        int b = a + 1;
        while (b < R && extend(depth, width(a), b - a, width(b))) ++b;
        spans.push_back({a, b});
        a = b;
      }
    }
    L = R;
  }
  std::sort(spans.begin(), spans.end());
  std::vector<int32_t> wg_of(ntasks, -1);
  for (const auto& sp : spans) {
    const int L0 = sp.first, L1 = sp.second;
    if (L1 - L0 < 2) continue;  // a single level: the level launch does as well
    // workgroup of every task: the tree of the chunk's forest it belongs to (roots first)
    int n_wg = 0;
    auto parent_task = [&](int t) -> int {
      // (a task's LAST entry is never a prologue: its receiver is the task's; its FIRST entry's sender is the cluster the
      // task waits for -- the prologue's X where there is one)
      const MsgDesc& m = p.msgs[tr.entries[postorder ? tr.task_off[t + 1] - 1 : tr.task_off[t]].msg];
      const int c = postorder ? m.to_b : m.from_b;  // the cluster whose own message (post) / receipt (pre) is the parent
      const int q = task_of[c];
      if (q < 0) return -1;
      const int Lq = task_level[q];
      return (Lq >= L0 && Lq < L1 && Lq != task_level[t]) ? q : -1;
    };
    if (postorder) {
      // a receiver that sends only after this chunk may still receive at several of its levels (children of different
      // heights): those tasks write the same belief and must stay in one workgroup, in level order
      std::unordered_map<int, int> wg_of_receiver;
      for (int L = L1 - 1; L >= L0; --L)
        for (int t = tr.level_off[L]; t < tr.level_off[L + 1]; ++t) {
          const int q = parent_task(t);
          if (q >= 0) {
            wg_of[t] = wg_of[q];
          } else {
            const int R = p.msgs[tr.entries[tr.task_off[t + 1] - 1].msg].to_b;
            auto it = wg_of_receiver.find(R);
            if (it == wg_of_receiver.end()) it = wg_of_receiver.emplace(R, n_wg++).first;
            wg_of[t] = it->second;
          }
        }
    } else {
      for (int L = L0; L < L1; ++L)
        for (int t = tr.level_off[L]; t < tr.level_off[L + 1]; ++t) {
          const int q = parent_task(t);
          wg_of[t] = q < 0 ? n_wg++ : wg_of[q];
        }
    }
    // PACKING: a launch wider than the chip is several generations of workgroups whose passes are mostly empty (a tree of
    // a narrow chunk holds one to three tasks per level, a pass has room for kTailWaves records).  Above `bins` trees the
    // trees share workgroups: largest first, each into the workgroup whose walk it lengthens least (passes = sum over the
    // levels of ceil(slots / kTailWaves)), the emptier one on a tie.  A tree stays whole, so the launch is as dependency-
    // closed as before; the chains of a walk (link_chains) are found on the merged walk like on any other.
    {
      int bins = p.tune.chunk_bins >= 0 ? p.tune.chunk_bins : kChunkBins;
      if (uni) bins = std::max(1, bins * 4 / std::max(1, (p.n_sites + 63) / 64));   // (workgroup = (walk, block of 64 sites), four per CU)
      if (bins > 0 && n_wg > bins) {
        const int nl = L1 - L0;
        const bool by_task = uni || !all_fast(L0);
        std::vector<int32_t> need((size_t)n_wg * nl, 0), total(n_wg, 0);
        for (int L = L0; L < L1; ++L)
          for (int t = tr.level_off[L]; t < tr.level_off[L + 1]; ++t) {
            const int w = by_task ? 1 : task_waves(tr, t);
            need[(size_t)wg_of[t] * nl + (L - L0)] += w;
            total[wg_of[t]] += w;
          }
        std::vector<int> by_size(n_wg);
        for (int w = 0; w < n_wg; ++w) by_size[w] = w;
        std::stable_sort(by_size.begin(), by_size.end(), [&](int x, int y) { return total[x] > total[y]; });
        std::vector<int32_t> load((size_t)bins * nl, 0), load_total(bins, 0), bin_of(n_wg, 0);
        for (int w : by_size) {
          int best = 0;
          long best_passes = -1;
          for (int b = 0; b < bins; ++b) {
            long passes = 0;
            for (int l = 0; l < nl; ++l) passes += (load[(size_t)b * nl + l] + need[(size_t)w * nl + l] + kTailWaves - 1) / kTailWaves;
            if (best_passes < 0 || passes < best_passes || (passes == best_passes && load_total[b] < load_total[best])) {
              best = b;
              best_passes = passes;
            }
          }
          bin_of[w] = best;
          for (int l = 0; l < nl; ++l) load[(size_t)best * nl + l] += need[(size_t)w * nl + l];
          load_total[best] += total[w];
        }
        for (int L = L0; L < L1; ++L)
          for (int t = tr.level_off[L]; t < tr.level_off[L + 1]; ++t) wg_of[t] = bin_of[wg_of[t]];
        n_wg = bins;   // (n_wg > bins trees, every one of them non-empty, emptiest workgroup first on a tie: none stays empty)
      }
    }
    // per workgroup, per level: its tasks, in level order; groups of kTailWaves records (a task never straddles two)
    std::vector<std::vector<int>> tasks_of_wg(n_wg);
    for (int L = L0; L < L1; ++L)
      for (int t = tr.level_off[L]; t < tr.level_off[L + 1]; ++t) tasks_of_wg[wg_of[t]].push_back(t);  // ascending level
    std::vector<int> order(n_wg);
    for (int w = 0; w < n_wg; ++w) order[w] = w;
    std::stable_sort(order.begin(), order.end(),
                     [&](int x, int y) { return tasks_of_wg[x].size() > tasks_of_wg[y].size(); });  // long ones first
    Traversal::Chunk ch;
    ch.level0 = L0; ch.level1 = L1; ch.n_wg = n_wg; ch.wg0 = (int32_t)tr.chunk_wg_off.size();
    ch.generic = all_fast(L0) ? 0 : 1;
    ch.max_mf = 0;
    for (int L = L0; L < L1; ++L) ch.max_mf = std::max(ch.max_mf, level_mf[L]);
    ch.small_only = ch.generic;
    for (int L = L0; L < L1; ++L) ch.small_only = ch.small_only && tr.level_small[L];
    ch.group0 = ch.generic ? (int64_t)(tr.cgroups.size() / kTailWaves) : (int64_t)(tr.centries.size() / kTailWaves);
    int32_t ngroups = 0;
    for (int w : order) {
      tr.chunk_wg_off.push_back(ngroups);
      const std::vector<int>& ts = tasks_of_wg[w];
      size_t i = 0;
      while (ch.generic && i < ts.size()) {
        // generic-class chunk: one wavefront per TASK, up to kTailWaves tasks of one level per group
        const int L = task_level[ts[i]];
        int fill = 0;
        for (; i < ts.size() && task_level[ts[i]] == L && fill < kTailWaves; ++i, ++fill) tr.cgroups.push_back(ts[i]);
        for (; fill < kTailWaves; ++fill) tr.cgroups.push_back(-1);
        ++ngroups;
      }
      while (i < ts.size()) {
        const int L = task_level[ts[i]];
        std::vector<int> members;
        int fill = 0;
        while (i < ts.size() && task_level[ts[i]] == L) {
          const int len = task_waves(tr, ts[i]);
          if (fill + len > kTailWaves) break;
          members.push_back(ts[i]);
          fill += len;
          ++i;
        }
        // receiver blocks of the members (fast_task is cheap: recomputed instead of stored per task)
        int tmin = members[0], tmax = members[0];
        for (int t : members) {
          tmin = std::min(tmin, t);
          tmax = std::max(tmax, t);
        }
        std::vector<std::pair<int, int>> blk_by_task(tmax - tmin + 1, {0, 0});
        for (int t : members) {
          int up0 = 0, mt = 0;
          (void)fast_task(p, tr, t, postorder, &up0, &mt);
          blk_by_task[t - tmin] = {up0, mt};
        }
        append_group(p, tr, members, blk_by_task, tmin, postorder, kTailWaves, tr.centries, tr.cpros, &tr.has_pro);
        ++ngroups;
      }
    }
    tr.chunk_wg_off.push_back(ngroups);
    ch.n_groups = ngroups;
    tr.chunks.push_back(ch);
  }
  // chains / late groups of the register-resident chunks: one walk per workgroup
  {
    std::vector<std::pair<int64_t, int64_t>> walks;
    for (const Traversal::Chunk& ch : tr.chunks) {
      if (ch.generic) continue;
      for (int b = 0; b < ch.n_wg; ++b)
        walks.push_back({ch.group0 + tr.chunk_wg_off[ch.wg0 + b], ch.group0 + tr.chunk_wg_off[ch.wg0 + b + 1]});
    }
    if (tr.cpros.size() < tr.centries.size()) tr.cpros.resize(tr.centries.size());
    link_chains(tr.centries, tr.cpros, walks, p.fast_p);
  }
}

// Level-synchronous schedules of one spanning tree.
// fuse = 1: CHAIN FUSION (opt-in).  A cluster with exactly one child in the schedule tree receives one message and can send
// at once: the wave that delivered the message goes on with the cluster's own message(s) instead of leaving them to the
// next level (= the next kernel launch).  Postorder: the single message into a unary cluster X is prepended to the
// entry X -> parent(X) of the task of parent(X); preorder: the task of a cluster S whose parent has no other child is
// appended to the parent's task.  Message order into every receiver, sequence numbers and arithmetic are unchanged;
// only the grouping into waves and levels is: Bethe graphs (factor clusters are unary) lose half of their levels.
// Fused tasks run on the generic kernel (entries of a task are executed in order by one wavefront).
// fuse = 2: PROLOGUE FUSION (default where it applies).  Only the pairs the register-resident kernel runs as one record
// (prologue_pair): F has one child C and a parent Y; its one incoming message of a traversal (postorder: C -> F,
// preorder: Y -> F) integrates nothing and lands on the block its own message (F -> Y / F -> C) integrates out: the
// incoming message becomes the PROLOGUE entry (Entry::pro) in front of F's own.  Every factor cluster of a tree's Bethe
// graph is such an F: half of the levels go, and the tasks stay on the register-resident kernel.
static bool tree_all_fast(const Tree& t);
static void build_traversals(const Plan& p, Tree& t, int fuse) {
  const int n = (int)t.pa.size();
  // message id of edge i in each direction: sepset k = (a, b); dir 0 is received by a
  auto msg_to = [&](int i, int receiver) {
    const int k = t.sep[i];
    return 2 * k + (p.sepset_clusters[2 * k] == receiver ? 0 : 1);
  };
  // child edges and parent edge of every cluster of the tree
  std::unordered_map<int, int> nchild, only_child_edge, parent_edge;
  for (int i = 0; i < n; ++i) {
    if (++nchild[t.pa[i]] == 1) only_child_edge[t.pa[i]] = i;
    parent_edge[t.ch[i]] = i;
  }
  auto unary = [&](int cluster) {
    auto it = nchild.find(cluster);
    return fuse == 1 && it != nchild.end() && it->second == 1;
  };
  // fuse = 2: is F's one incoming message of the postorder / preorder the prologue of its own?
  auto pro_cluster = [&](int F, bool post) {
    if (fuse != 2) return false;
    auto ic = nchild.find(F);
    auto ip = parent_edge.find(F);
    if (ic == nchild.end() || ic->second != 1 || ip == parent_edge.end()) return false;
    const int ec = only_child_edge[F], ep = ip->second;
    const int light = post ? msg_to(ec, F) : msg_to(ep, F);
    const int heavy = post ? msg_to(ep, t.pa[ep]) : msg_to(ec, t.ch[ec]);
    return prologue_pair(p.msgs[light], p.msgs[heavy], p.fast_p);
  };
  // ---- postorder: level = height of the child in the schedule tree (chains collapsed)
  std::unordered_map<int, int> hnode;  // cluster -> level at which it can send (its height; leaves are absent: 0)
  int post_levels = 0;
  {
    std::vector<int> lvl(n);
    std::vector<uint8_t> chained_edge(n, 0);  // the message of edge i rides in front of the edge above its receiver
    for (int i = 0; i < n; ++i)
      chained_edge[i] = (unary(t.pa[i]) && parent_edge.count(t.pa[i])) || pro_cluster(t.pa[i], true);
    int nlev = 0;
    for (int i = n - 1; i >= 0; --i) {
      int h = 0;
      auto it = hnode.find(t.ch[i]);
      if (it != hnode.end()) h = it->second;
      lvl[i] = h;
      int& hp = hnode[t.pa[i]];
      // a unary parent that has an edge above it sends in the same level, right after this message
      hp = std::max(hp, chained_edge[i] ? h : h + 1);
      nlev = std::max(nlev, h + 1);
    }
    // tasks: group by (level, target parent); entries in reference order (decreasing i)
    std::vector<std::vector<int>> bylevel(nlev);
    for (int i = n - 1; i >= 0; --i)
      if (!chained_edge[i]) bylevel[lvl[i]].push_back(i);  // chained edges ride in front of the edge above their receiver
    Traversal& tr = t.post;
    tr = Traversal{};
    tr.level_off.push_back(0);
    tr.task_off.push_back(0);
    std::vector<int> chain;
    for (int L = 0; L < nlev; ++L) {
      std::unordered_map<int, int> task_of_target;
      std::vector<std::vector<int>> tasks;
      for (int i : bylevel[L]) {
        auto it = task_of_target.find(t.pa[i]);
        if (it == task_of_target.end()) {
          task_of_target[t.pa[i]] = (int)tasks.size();
          tasks.push_back({i});
        } else {
          tasks[it->second].push_back(i);
        }
      }
      for (auto& tk : tasks) {
        for (int top : tk) {
          // the chain below `top`: single child edges of unary / prologue clusters, deepest first
          chain.assign(1, top);
          while (unary(t.ch[chain.back()]) || pro_cluster(t.ch[chain.back()], true))
            chain.push_back(only_child_edge[t.ch[chain.back()]]);
          for (int q = (int)chain.size() - 1; q >= 0; --q) {
            const int i = chain[q];
            Entry e{};
            e.msg = msg_to(i, t.pa[i]);
            e.edge = i;
            e.reuse = 0;
            e.seq = n - 1 - i;
            e.pro = (fuse == 2 && q > 0) ? 1 : 0;
            tr.entries.push_back(e);
            tr.max_mf = std::max(tr.max_mf, p.msgs[e.msg].mf);
          }
        }
        tr.task_off.push_back((int)tr.entries.size());
      }
      if (!tasks.empty() || !fuse) tr.level_off.push_back((int)tr.task_off.size() - 1);
    }
    post_levels = nlev;
    finalize_traversal(p, tr, true);
  }
  // ---- preorder; tasks group by sender.  A sender may send as soon as it has received from its own parent (level =
  // its depth: as soon as possible) and no later than its height allows (level = tree height - its height: as late as
  // possible); both take as many levels as the tree is deep.  Default: AS LATE AS POSSIBLE, the mirror image of the
  // postorder -- the levels near the root are then as narrow as the postorder's last ones (they join the single-workgroup
  // tail launch), and the bulk of the messages sits in a few very wide levels at the leaf end, instead of a bell of
  // mid-sized levels that each pay a full launch latency.  Chain fusion keeps the depth order (a fused chain starts where its
  // first message may).
  {
    const bool alap = fuse != 1;
    auto height = [&](int cluster) {
      auto it = hnode.find(cluster);
      return it == hnode.end() ? 0 : it->second;
    };
    std::unordered_map<int, int> dnode;   // cluster -> level at which it sends
    std::unordered_map<int, int> head;    // cluster -> the cluster whose task carries its messages
    std::vector<int> lvl(n);
    std::vector<uint8_t> pro_edge(n, 0);  // fuse = 2: the message of edge i is the prologue of its receiver's own task
    int nlev = 0;
    if (fuse == 2) {
      // Levels from the dependencies themselves (the postorder's heights do not carry over: there a prologue shortens the
      // path through F, here it moves X -> F from X's task into F's).  The task of a sender S waits for the task that
      // delivered into S -- its parent's -- or, where parent(S) -> S is S's prologue, for the one that delivered into
      // parent(S): its grandparent's.  depth = the longest chain of such waits above, ph = below; as late as possible:
      // level = (largest depth) - ph.
      for (int i = 0; i < n; ++i) pro_edge[i] = pro_cluster(t.ch[i], false);
      std::unordered_map<int, int> depth, ph;
      auto dep_of = [&](int S) -> int {   // the cluster whose task S's task waits for (-1: none)
        auto ip = parent_edge.find(S);
        if (ip == parent_edge.end()) return -1;
        const int X = t.pa[ip->second];
        if (!pro_edge[ip->second]) return X;
        auto ix = parent_edge.find(X);
        return ix == parent_edge.end() ? -1 : t.pa[ix->second];
      };
      int maxdepth = 0;
      for (int i = 0; i < n; ++i) {   // senders in preorder: a sender's ancestors come before it
        const int S = t.pa[i];
        if (depth.count(S)) continue;
        const int d = dep_of(S);
        depth[S] = d < 0 ? 0 : depth[d] + 1;
        maxdepth = std::max(maxdepth, depth[S]);
      }
      // (a cluster all of whose child edges are prologue edges has no task of its own: its depth is only a relay)
      for (int i = n - 1; i >= 0; --i) {
        const int S = t.pa[i];
        const int d = dep_of(S);
        if (!ph.count(S)) ph[S] = 0;
        if (d >= 0) ph[d] = std::max(ph.count(d) ? ph[d] : 0, ph[S] + 1);
      }
      for (int i = 0; i < n; ++i) {
        const int S = t.pa[i];
        lvl[i] = alap ? maxdepth - ph[S] : depth[S];
      }
      // (levels of prologue edges are not used: they follow their receiver's task)
      nlev = maxdepth + 1;
    } else {
      for (int i = 0; i < n; ++i) {
        int dpt = 0;
        auto it = dnode.find(t.pa[i]);
        if (it != dnode.end()) dpt = it->second;
        if (alap) dpt = post_levels - height(t.pa[i]);
        lvl[i] = dpt;
        const bool chained = unary(t.pa[i]);   // the child's task follows this message in the same wave
        dnode[t.ch[i]] = chained ? dpt : dpt + 1;
        auto ih = head.find(t.pa[i]);
        const int hp = ih == head.end() ? t.pa[i] : ih->second;
        head[t.ch[i]] = chained ? hp : t.ch[i];
        nlev = std::max(nlev, dpt + 1);
      }
    }
    std::vector<std::vector<int>> bylevel(nlev);
    for (int i = 0; i < n; ++i)
      if (!pro_edge[i]) bylevel[lvl[i]].push_back(i);
    Traversal& tr = t.pre;
    tr = Traversal{};
    tr.level_off.push_back(0);
    tr.task_off.push_back(0);
    for (int L = 0; L < nlev; ++L) {
      std::unordered_map<int, int> task_of_sender;
      std::vector<std::vector<int>> tasks;
      for (int i : bylevel[L]) {   // increasing i: a chained child's edges come after the edge into it
        auto ih = head.find(t.pa[i]);
        const int key = ih == head.end() ? t.pa[i] : ih->second;
        auto it = task_of_sender.find(key);
        if (it == task_of_sender.end()) {
          task_of_sender[key] = (int)tasks.size();
          tasks.push_back({i});
        } else {
          tasks[it->second].push_back(i);
        }
      }
      for (auto& tk : tasks) {
        int prev_msg = -1;
        if (fuse == 2) {
          // the prologue of this sender's task: the message its parent sends into it
          auto ip = parent_edge.find(t.pa[tk[0]]);
          if (ip != parent_edge.end() && pro_edge[ip->second]) {
            const int i = ip->second;
            Entry e{};
            e.msg = msg_to(i, t.ch[i]);
            e.edge = i;
            e.seq = n + i;
            e.reuse = 0;
            e.pro = 1;
            tr.entries.push_back(e);
            tr.max_mf = std::max(tr.max_mf, p.msgs[e.msg].mf);
          }
        }
        for (int i : tk) {
          Entry e{};
          e.msg = msg_to(i, t.ch[i]);
          e.edge = i;
          e.seq = n + i;
          e.reuse = 0;
          if (prev_msg >= 0) {
            const MsgDesc& a = p.msgs[prev_msg];
            const MsgDesc& b = p.msgs[e.msg];
            if (a.from_b == b.from_b && a.s == b.s && b.ni > 0 &&
                std::equal(p.idxpool.begin() + a.keep_map, p.idxpool.begin() + a.keep_map + a.s,
                           p.idxpool.begin() + b.keep_map))
              e.reuse = 1;
          }
          prev_msg = e.msg;
          tr.entries.push_back(e);
          tr.max_mf = std::max(tr.max_mf, p.msgs[e.msg].mf);
        }
        tr.task_off.push_back((int)tr.entries.size());
      }
      if (!tasks.empty() || !fuse) tr.level_off.push_back((int)tr.task_off.size() - 1);
    }
    finalize_traversal(p, tr, false);
  }
}

static bool tree_all_fast(const Tree& t) {
  for (const Traversal* tr : {&t.post, &t.pre})
    for (size_t L = 0; L + 1 < tr->level_off.size(); ++L)
      if (tr->level_nfast[L] != tr->level_off[L + 1] - tr->level_off[L]) return false;
  return true;
}

int plan_set_schedule(Plan& p, int32_t n_trees, const int32_t* tree_off, const int32_t* pa_j,
                      const int32_t* ch_j) {
  if (n_trees < 0 || (n_trees > 0 && (!tree_off || !pa_j || !ch_j))) {
    p.err = "invalid schedule arguments";
    return PGBP_ERR_INVALID;
  }
  std::map<std::pair<int, int>, int> sepmap;
  for (int k = 0; k < p.n_sepsets; ++k) {
    int a = p.sepset_clusters[2 * k], b = p.sepset_clusters[2 * k + 1];
    sepmap[{std::min(a, b), std::max(a, b)}] = k;  // sdict: Set of the 2 labels -> sepset (clustergraphbeliefs.jl:73-76)
  }
  std::vector<Tree> trees(n_trees);
  for (int t = 0; t < n_trees; ++t) {
    const int n = tree_off[t + 1] - tree_off[t];
    if (n < 0) {
      p.err = "tree_off must be non-decreasing";
      return PGBP_ERR_INVALID;
    }
    Tree& T = trees[t];
    T.pa.assign(pa_j + tree_off[t], pa_j + tree_off[t + 1]);
    T.ch.assign(ch_j + tree_off[t], ch_j + tree_off[t + 1]);
    T.sep.resize(n);
    std::unordered_map<int, int> seen;  // cluster -> 1 if already placed in the tree
    for (int i = 0; i < n; ++i) {
      const int a = T.pa[i], c = T.ch[i];
      if (a < 0 || a >= p.n_clusters || c < 0 || c >= p.n_clusters) {
        p.err = "schedule tree " + std::to_string(t) + ": cluster index out of range";
        return PGBP_ERR_INVALID;
      }
      if (i == 0) seen[a] = 1;
      if (!seen.count(a) || seen.count(c)) {
        p.err = "schedule tree " + std::to_string(t) + ", edge " + std::to_string(i) +
                ": not a preorder edge list of a tree (parent unseen or child seen twice)";
        return PGBP_ERR_NOT_TREE;
      }
      seen[c] = 1;
      auto it = sepmap.find({std::min(a, c), std::max(a, c)});
      if (it == sepmap.end()) {
        p.err = "schedule tree " + std::to_string(t) + ", edge " + std::to_string(i) +
                ": no sepset between clusters " + std::to_string(a) + " and " + std::to_string(c);
        return PGBP_ERR_INVALID;
      }
      T.sep[i] = it->second;
    }
    build_traversals(p, T, 0);
    // Prologue fusion (fuse = 2) where some cluster qualifies (every factor cluster of a tree's Bethe graph) AND the
    // register-resident kernel runs the whole tree: on a mixed schedule (cfg5's Bethe graph: hybrid families beside
    // tree edges in every level) the levels go to the wave-per-task kernel whole, where a prologue is one more message
    // for the same wavefront -- half the levels at twice the time each (measured: 1.86 against 1.78 ms per iteration).
    // The univariate site batches run on the thread-per-site kernel and keep the plain levels.  (PGBP_TUNING no_prologue: A/B.)
    if (tree_all_fast(T)) {
      const bool no_pro = !p.tune.prologues || p.tune.chain_fusion;
      const bool uni = p.max_dim <= 2 && p.n_sites >= 8;
      if (!no_pro && !uni && p.fast_p > 0) {
        std::unordered_map<int, int> nch, par, only;
        for (int i = 0; i < n; ++i) {
          if (++nch[T.pa[i]] == 1) only[T.pa[i]] = i;
          par[T.ch[i]] = i;
        }
        auto msg_to = [&](int i, int receiver) {
          const int k = T.sep[i];
          return 2 * k + (p.sepset_clusters[2 * k] == receiver ? 0 : 1);
        };
        bool any = false;
        for (const auto& kv : nch) {
          if (kv.second != 1 || !par.count(kv.first)) continue;
          const int F = kv.first, ec = only[F], ep = par[F];
          if (prologue_pair(p.msgs[msg_to(ec, F)], p.msgs[msg_to(ep, T.pa[ep])], p.fast_p) ||
              prologue_pair(p.msgs[msg_to(ep, F)], p.msgs[msg_to(ec, T.ch[ec])], p.fast_p)) {
            any = true;
            break;
          }
        }
        if (any) {
          build_traversals(p, T, 2);
          if (!tree_all_fast(T)) build_traversals(p, T, 0);
        }
      }
    }
    // Chain fusion is OPT-IN (PGBP_TUNING chain_fusion): measured on the cfg5 network (Bethe graph, 20 000 tips) it trades
    // 398 launches for 152 but a fused level lasts as long as its longest chain (about 5 us per message inside a wave
    // against about 10 us per launch): 4.8 ms per iteration against 3.9 ms.  It pays on path-like
    // schedule trees (nodesubtree_clusterlist schedules).  Schedules that the register-resident kernel runs whole and
    // the thread-per-site kernel of univariate batches always keep the plain levels.
    const bool fuse_on = p.tune.chain_fusion;
    const bool uni_batch = p.max_dim <= 2 && p.n_sites >= 8;
    if (fuse_on && !uni_batch && !tree_all_fast(T)) build_traversals(p, T, 1);
    // the tail launch walks the postorder's last levels and the preorder's first ones as ONE sequence of passes
    T.tail = T.post.tentries;
    T.tail.insert(T.tail.end(), T.pre.tentries.begin(), T.pre.tentries.end());
    T.tail_pros = T.post.tpros;
    T.tail_pros.insert(T.tail_pros.end(), T.pre.tpros.begin(), T.pre.tpros.end());
    T.tail_pros.resize(T.tail.size());
    link_chains(T.tail, T.tail_pros, {{0, (int64_t)(T.tail.size() / kTailWaves)}}, p.fast_p);
  }
  p.trees.swap(trees);
  p.all_fast = !p.trees.empty();
  for (const Tree& t : p.trees)
    for (const Traversal* tr : {&t.post, &t.pre})
      for (size_t L = 0; L + 1 < tr->level_off.size(); ++L)
        if (tr->level_nfast[L] != tr->level_off[L + 1] - tr->level_off[L]) p.all_fast = false;
  return PGBP_OK;
}

double plan_bytes_per_calibrate(const Plan& p, int64_t* n_messages) {
  // 8*[(mf^2+mf+1) + 4*(s^2+s+1) + (s^2+s)] per message
  double bytes = 0;
  int64_t nm = 0;
  for (const Tree& t : p.trees) {
    for (const Traversal* tr : {&t.post, &t.pre}) {
      for (const Entry& e : tr->entries) {
        const MsgDesc& m = p.msgs[e.msg];
        const double mf = m.mf, s = m.s;
        bytes += 8.0 * ((mf * mf + mf + 1) + 4.0 * (s * s + s + 1) + (s * s + s));
        ++nm;
      }
    }
  }
  if (n_messages) *n_messages = nm * p.n_sites;
  return bytes * p.n_sites;
}

}  // namespace pgbp

// ------------------------------------------------------------------ C ABI (host-only part)
using pgbp::Plan;

extern "C" {

int pgbp_plan_create(const pgbp_desc* desc, pgbp_plan** out) {
  if (!out) return PGBP_ERR_INVALID;
  pgbp_plan* pl = new pgbp_plan();
  int rc = pgbp::plan_build(pl->p, desc);
  *out = pl;  // returned even on error so that the message can be read
  return rc;
}

void pgbp_plan_destroy(pgbp_plan* p) { delete p; }

int pgbp_plan_set_schedule(pgbp_plan* p, int32_t n_trees, const int32_t* tree_off, const int32_t* pa_j,
                           const int32_t* ch_j) {
  if (!p) return PGBP_ERR_INVALID;
  return pgbp::plan_set_schedule(p->p, n_trees, tree_off, pa_j, ch_j);
}

int64_t pgbp_plan_packed_size(const pgbp_plan* p) { return p ? p->p.packed_off.back() : -1; }
int64_t pgbp_plan_residual_size(const pgbp_plan* p) { return p ? p->p.rpacked_off.back() : -1; }
int32_t pgbp_plan_n_messages(const pgbp_plan* p) { return p ? p->p.n_msgs() : -1; }

static const pgbp::Traversal* get_trav(const pgbp_plan* p, int32_t tree, int32_t dir) {
  if (!p || tree < 0 || tree >= (int)p->p.trees.size() || dir < 0 || dir > 1) return nullptr;
  return dir == 0 ? &p->p.trees[tree].post : &p->p.trees[tree].pre;
}

int pgbp_plan_traversal_sizes(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* n_levels,
                              int32_t* n_tasks, int32_t* n_entries) {
  const pgbp::Traversal* tr = get_trav(p, tree, dir);
  if (!tr) return PGBP_ERR_INVALID;
  if (n_levels) *n_levels = (int32_t)tr->level_off.size() - 1;
  if (n_tasks) *n_tasks = (int32_t)tr->task_off.size() - 1;
  if (n_entries) *n_entries = (int32_t)tr->entries.size();
  return PGBP_OK;
}

int pgbp_plan_level_nfast(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* level_nfast) {
  const pgbp::Traversal* tr = get_trav(p, tree, dir);
  if (!tr || !level_nfast) return PGBP_ERR_INVALID;
  std::copy(tr->level_nfast.begin(), tr->level_nfast.end(), level_nfast);
  return PGBP_OK;
}

int pgbp_plan_traversal(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* level_off, int32_t* task_off,
                        int32_t* entry_msg, int32_t* entry_edge, int32_t* entry_reuse) {
  const pgbp::Traversal* tr = get_trav(p, tree, dir);
  if (!tr) return PGBP_ERR_INVALID;
  if (level_off) std::copy(tr->level_off.begin(), tr->level_off.end(), level_off);
  if (task_off) std::copy(tr->task_off.begin(), tr->task_off.end(), task_off);
  for (size_t i = 0; i < tr->entries.size(); ++i) {
    if (entry_msg) entry_msg[i] = tr->entries[i].msg;
    if (entry_edge) entry_edge[i] = tr->entries[i].edge;
    if (entry_reuse) entry_reuse[i] = tr->entries[i].pro ? 2 : tr->entries[i].reuse;
  }
  return PGBP_OK;
}

int pgbp_plan_groups(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* level_ngroups, int32_t* tail_levels,
                     int32_t* records, int32_t* tail_records) {
  const pgbp::Traversal* tr = get_trav(p, tree, dir);
  if (!tr) return PGBP_ERR_INVALID;
  if (level_ngroups) std::copy(tr->level_ngroups.begin(), tr->level_ngroups.end(), level_ngroups);
  if (tail_levels) *tail_levels = tr->tail_levels;
  auto dump = [](const std::vector<pgbp::FEntry>& v, int32_t* out) {
    for (size_t i = 0; i < v.size(); ++i) {
      const pgbp::FEntry& f = v[i];
      int32_t* r = out + 6 * i;
      r[0] = f.valid; r[1] = f.msg; r[2] = f.grp_base; r[3] = f.grp_len; r[4] = f.src_wave; r[5] = f.mode;
    }
  };
  if (records) dump(tr->fentries, records);
  if (tail_records) dump(tr->tentries, tail_records);
  return PGBP_OK;
}

int pgbp_plan_chunks(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* n_chunks, int32_t* info, int32_t* wg_off,
                     int32_t* records) {
  const pgbp::Traversal* tr = get_trav(p, tree, dir);
  if (!tr || !n_chunks) return PGBP_ERR_INVALID;
  *n_chunks = (int32_t)tr->chunks.size();
  for (size_t c = 0; c < tr->chunks.size(); ++c) {
    const auto& ch = tr->chunks[c];
    if (info) {
      int32_t* r = info + 4 * c;
      r[0] = ch.level0; r[1] = ch.level1; r[2] = ch.n_wg; r[3] = ch.generic ? -ch.n_groups : ch.n_groups;
    }
  }
  if (wg_off) std::copy(tr->chunk_wg_off.begin(), tr->chunk_wg_off.end(), wg_off);
  if (records) {
    // chunk after chunk: a register-resident chunk's groups as 8 records of 6 words; a generic one's as 8 "records"
    // {1 or 0, task id, 0, 0, 0, 0}
    size_t o = 0;
    for (const auto& ch : tr->chunks)
      for (int g = 0; g < ch.n_groups; ++g)
        for (int w = 0; w < pgbp::kTailWaves; ++w, ++o) {
          int32_t* r = records + 6 * o;
          if (ch.generic) {
            const int32_t t = tr->cgroups[(size_t)(ch.group0 + g) * pgbp::kTailWaves + w];
            r[0] = t >= 0; r[1] = t; r[2] = r[3] = r[4] = r[5] = 0;
          } else {
            const pgbp::FEntry& f = tr->centries[(size_t)(ch.group0 + g) * pgbp::kTailWaves + w];
            r[0] = f.valid; r[1] = f.msg; r[2] = f.grp_base; r[3] = f.grp_len; r[4] = f.src_wave; r[5] = f.mode;
          }
        }
  }
  return PGBP_OK;
}

int pgbp_plan_prologues(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* level_pro, int32_t* tail_pro,
                        int32_t* chunk_pro) {
  const pgbp::Traversal* tr = get_trav(p, tree, dir);
  if (!tr) return PGBP_ERR_INVALID;
  auto dump = [](const std::vector<pgbp::FEntry>& v, const std::vector<pgbp::FPro>& q, size_t i0, size_t i1, int32_t* out) {
    for (size_t i = i0; i < i1; ++i) out[i - i0] = (v[i].mode & pgbp::kFPro) ? q[i].msg : -1;
  };
  if (level_pro) dump(tr->fentries, tr->fpros, 0, tr->fentries.size(), level_pro);
  if (tail_pro) dump(tr->tentries, tr->tpros, 0, tr->tentries.size(), tail_pro);
  if (chunk_pro) {
    size_t o = 0;
    for (const auto& ch : tr->chunks) {
      const size_t nrec = (size_t)ch.n_groups * pgbp::kTailWaves;
      if (ch.generic)
        std::fill(chunk_pro + o, chunk_pro + o + nrec, -1);
      else
        dump(tr->centries, tr->cpros, (size_t)ch.group0 * pgbp::kTailWaves, (size_t)ch.group0 * pgbp::kTailWaves + nrec, chunk_pro + o);
      o += nrec;
    }
  }
  return PGBP_OK;
}

int pgbp_plan_chains(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* tail_chain, int32_t* chunk_chain) {
  const pgbp::Traversal* tr = get_trav(p, tree, dir);
  if (!tr) return PGBP_ERR_INVALID;
  auto word = [](const pgbp::FEntry& f) { return (int32_t)f.pad[0] | ((int32_t)f.pad[1] << 8) | ((int32_t)f.pad[2] << 16); };
  if (tail_chain) {
    const pgbp::Tree& T = p->p.trees[tree];
    const size_t first = dir == 0 ? 0 : T.post.tentries.size();
    for (size_t i = 0; i < tr->tentries.size(); ++i) tail_chain[i] = word(T.tail[first + i]);
  }
  if (chunk_chain) {
    size_t o = 0;
    for (const auto& ch : tr->chunks) {
      const size_t nrec = (size_t)ch.n_groups * pgbp::kTailWaves;
      for (size_t i = 0; i < nrec; ++i)
        chunk_chain[o + i] = ch.generic ? 0 : word(tr->centries[(size_t)ch.group0 * pgbp::kTailWaves + i]);
      o += nrec;
    }
  }
  return PGBP_OK;
}

int pgbp_plan_records(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* n_records, int32_t* level_first,
                      int32_t* task_first, uint8_t* records) {
  const pgbp::Traversal* tr = get_trav(p, tree, dir);
  if (!tr || !n_records) return PGBP_ERR_INVALID;
  *n_records = (int32_t)tr->grecs.size();
  if (level_first) std::copy(tr->level_gbase.begin(), tr->level_gbase.end(), level_first);
  if (task_first) std::copy(tr->task_grec.begin(), tr->task_grec.end(), task_first);
  if (records && !tr->grecs.empty()) std::memcpy(records, tr->grecs.data(), tr->grecs.size() * sizeof(pgbp::GRec));
  return PGBP_OK;
}

int pgbp_plan_rows(const pgbp_plan* p, int32_t tree, int32_t dir, int64_t* n_rows, int64_t* level_row0, int32_t* level_nrows,
                   int32_t* rowmap) {
  const pgbp::Traversal* tr = get_trav(p, tree, dir);
  if (!tr || !n_rows) return PGBP_ERR_INVALID;
  *n_rows = (int64_t)(tr->rowmap.size() / 2);
  if (level_row0) std::copy(tr->level_rowbase.begin(), tr->level_rowbase.end(), level_row0);
  if (level_nrows) std::copy(tr->level_nrows.begin(), tr->level_nrows.end(), level_nrows);
  if (rowmap) std::copy(tr->rowmap.begin(), tr->rowmap.end(), rowmap);
  return PGBP_OK;
}

const char* pgbp_plan_last_error(const pgbp_plan* p) { return p ? p->p.err.c_str() : "null plan"; }

}  // extern "C"
