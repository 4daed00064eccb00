// Kernel launch wrappers (defined in pgbp_kernels.hip), called by the engine.
#pragma once
#include <hip/hip_runtime.h>

#include "pgbp_internal.hpp"

namespace pgbp {

constexpr unsigned long long kNoFail = ~0ull;
constexpr int kInfoBits = 20;  // fail key = (global sequence number << kInfoBits) | info

// Everything a message kernel needs, passed by value.
struct DevState {
  double* pool;
  int64_t pool_stride;  // doubles per site
  double* rpool;
  int64_t rpool_stride;
  const MsgDesc* msgs;
  const int32_t* idx;
  int32_t* flags;              // [n_sites][n_msgs] iscalibrated_resid
  int32_t* status;             // [n_sites][n_msgs] 0 ok, >0 PosDefException.info of the last attempt
  unsigned long long* fail;    // [n_sites] min over failing messages of the fail key
  int32_t* poison;             // [n_sites][n_clusters] 1: the cluster sits downstream of a failed message
  int32_t n_clusters;
  int32_t n_msgs;
  int32_t update_resnorm;
  double atol;
  // iscalibrated_residnorm! without a division on the device: thr[s] = the largest x with fl(x / fl(sqrt(s))) <= atol,
  // thr[PGBP_MAX_DIM + 1 + s] = the largest x with fl(x / s) <= atol (x -> fl(x / c) is monotone, so max|dh| <= thr[s] is
  // the reference's test bit for bit; s = 0: +inf); thr_h_p / thr_J_p: the two entries of the fast kernel's sepset dimension
  const double* thr;
  double thr_h_p, thr_J_p;
  // log of the pivots' mantissa product without the library routine (about a hundred dependent double-double
  // operations on the critical path of every message): logtab[i] = {1 / c_i rounded, -log(that)} for the 128 intervals
  // [0.5 + i / 256, 0.5 + (i + 1) / 256) of a mantissa with centre c_i (the host fills it in extended precision)
  const double2* logtab;
  int32_t bs16;    // beliefs of dimension P / 2P and residuals of P-dim sepsets are in the packed layout
  int32_t fast_p;  // P: sepset dimension of the register-resident kernel (2 .. 16, the real dimension; 0: none)
  // site-minor layout (univariate batches, every dimension <= 2): element t of belief b of site s lives at
  // pool[(packed_off[b] + t) * n_sites + s] (residuals: rpacked_off[msg]); pool / rpool then point at those buffers
  int32_t sm;
  int32_t n_sites;
  const int64_t* packed_off;
  const int64_t* rpacked_off;
  // thread-per-site kernels only, may be null: notcal[site] is set by a message whose residual-norm flag comes out false.  When
  // a postorder + preorder pair over a spanning tree that holds EVERY sepset has run (each of the 2 n_sepsets flags rewritten
  // once), iscalibrated_residnorm(beliefs) (src/clustergraphbeliefs.jl:168-169) of a site is "no message set the mark": the
  // engine then skips the reduction over the flag array (80 M words per calibrate of a 1 000-site batch on a 20 000-tip tree)
  int32_t* notcal;
  // every sepset this launch touches is known to hold the constant 1 (J = h = g = 0: straight after a reset, in a
  // postorder that overwrites all of them): the register-resident kernel then does not read them
  int32_t sep_zero;
};

size_t generic_lds_bytes(int max_mf);

#if defined(__HIPCC__)
// log(x) for a positive, normal x (the mantissa product of up to 16 pivots, a single pivot): x = m 2^e with m in
// [0.5, 1); r = m / c' - 1 with c' = the tabulated centre's reciprocal, |r| <= 2^-8; log x = e ln 2 + log c' + log1p(r),
// the series of log1p to r^7 (the next term is below 2^-66).  Absolute error about one unit in the last place of the
// result's leading term (1e-16); the table word is fetched through the scalar cache (the index is wave-uniform wherever
// this is used: every lane holds the same product).
__device__ __forceinline__ double log_by_table(const double2* __restrict__ tab, double x) {
  int e;
  const double m = frexp(x, &e);
  const int i = __builtin_amdgcn_readfirstlane((__double2hiint(m) >> 13) & 127);
  // (through the CONSTANT address space: with a wave-uniform index the compiler then issues a scalar load -- a hundred
  // clocks out of the scalar cache on the critical path of a message instead of a vector load's trip to L2)
  typedef double pgbp_cdouble2 __attribute__((ext_vector_type(2)));
  const __attribute__((address_space(4))) pgbp_cdouble2* ctab =
      reinterpret_cast<const __attribute__((address_space(4))) pgbp_cdouble2*>(reinterpret_cast<unsigned long long>(tab));
  const pgbp_cdouble2 tv = ctab[i];
  const double2 t = make_double2(tv.x, tv.y);
  const double r = fma(m, t.x, -1.0);
  double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
  p = fma(p, r, 0.2);
  p = fma(p, r, -0.25);
  p = fma(p, r, 1.0 / 3.0);
  p = fma(p, r, -0.5);
  return fma((double)e, 0.69314718055994530941723212145818, t.y) + fma(p * r, r, r);
}
#endif

// wave-per-task kernel: block b of the launch runs the task whose first record is d_recs[rec0 + b] (GRec, pgbp_internal.hpp)
// small_only: every message of these tasks fits the register-resident body (Traversal::level_small): the instance without the
// in-LDS body (fewer registers: more wavefronts per SIMD on the wide levels, half the code)
// d_rowmap / n_rows: the level's messages one per row of 16 lanes (Traversal::rowmap: pairs (record, position | k << 8), a
// multiple of four rows, tasks never straddle a wavefront), nullptr / 0: none -- wide small-only postorder levels then run
// one message per row with mult! in task order (bp_level_small4<true>)
void launch_level_generic(const DevState& S, const GRec* d_recs, int rec0, int ntasks, int n_sites,
                          unsigned long long seq_base, unsigned long long stop_below, int max_mf, bool small_only,
                          hipStream_t st, const int32_t* d_rowmap = nullptr, int n_rows = 0,
                          int small4_min = kSmall4MinTasksDefault);

// loop mode of the generic task body: n_wg workgroups of kTailWaves wavefronts, workgroup b walks the groups
// [d_wg_off[b], d_wg_off[b + 1]) of kTailWaves first records of tasks (-1: none) with a workgroup barrier in between
// (a chunk of fused levels)
void launch_chunk_pair(const DevState& S, const GRec* d_recs, const int32_t* d_grp_recs, const int32_t* d_wg_off, int n_wg,
                       int n_sites, unsigned long long seq_base, unsigned long long stop_below, hipStream_t st);
void launch_chunk_generic(const DevState& S, const GRec* d_recs, const int32_t* d_grp_recs, const int32_t* d_wg_off, int n_wg,
                          int n_sites, unsigned long long seq_base, unsigned long long stop_below, int max_mf,
                          bool small_only, bool pair, hipStream_t st);

// tasks with a belief of dimension 65 .. PGBP_MAX_DIM: one workgroup of 256 threads per task, the sender in up to 132 KB of LDS
// (a sender of more than kLdsMaxDim variables: the working matrix in d_ws, ntasks * n_sites slabs of big_ws_doubles(max_mf))
void launch_level_big(const DevState& S, const int32_t* d_task_off, const Entry* d_entries, int task0, int ntasks,
                      int n_sites, unsigned long long seq_base, unsigned long long stop_below, int max_mf, double* d_ws,
                      hipStream_t st);
constexpr int kLdsMaxDim = 128;   // largest working matrix [J | h] that fits a CU's LDS (128 x 129 doubles = 132 KB of 160)
int64_t big_ws_doubles(int max_mf);   // doubles of one workspace slab for a working matrix of dimension max_mf (0: fits LDS)

// thread-per-(site, task) kernel for graphs whose beliefs all have dimension <= 2 (univariate batches)
void launch_level_uni(const DevState& S, const int32_t* d_task_off, const Entry* d_entries, const URec* d_urecs, int task0,
                      int ntasks, int n_sites, unsigned long long seq_base, unsigned long long stop_below, int max_s,
                      hipStream_t st);
// loop mode of the thread-per-site body (sepsets of at most one variable): n_wg trees of tasks (groups of kTailWaves task
// ids, -1: none; workgroup b walks [d_wg_off[b], d_wg_off[b + 1])) x blocks of 64 sites
void launch_chunk_uni1(const DevState& S, const int32_t* d_task_off, const URec* d_urecs, const int32_t* d_grp_tasks,
                       const int32_t* d_wg_off, int n_wg, int n_sites, unsigned long long seq_base,
                       unsigned long long stop_below, hipStream_t st);
// (max_s: the engine's largest sepset dimension; <= 1 selects the instance that loads a message's elements by role)

// register-resident message kernel (pgbp_fast.hip): `ngroups` groups of records (FEntry) of a traversal.
// mode 0 (level): one group of kFastMaxWaves records per workgroup; 2 (tail): ONE workgroup walks the groups (kTailWaves
// records each) with a workgroup barrier between them; groups >= split stop below stop_b instead of stop_a; with d_wg_off
// (n_wg + 1 offsets into the groups): n_wg workgroups, workgroup b walks the groups [d_wg_off[b], d_wg_off[b + 1]) -- a
// chunk of fused levels.  d_pros: the records' prologues (FPro, same indexing) or null when none has one.
constexpr int kFastLevel = 0, kFastTail = 2;
void launch_fast16(const DevState& S, const FEntry* d_recs, const FPro* d_pros, int mode, int ngroups, int split, int n_sites,
                   unsigned long long seq_base, unsigned long long stop_a, unsigned long long stop_b, hipStream_t st,
                   const int32_t* d_wg_off = nullptr, int n_wg = 0);

// the loop launches in the packed layout, even P (pgbp_loop.hip): arguments as for launch_fast16's mode kFastTail
void launch_loop16(const DevState& S, const FEntry* d_recs, const FPro* d_pros, int ngroups, int split, int n_sites,
                   unsigned long long seq_base, unsigned long long stop_a, unsigned long long stop_b, hipStream_t st,
                   const int32_t* d_wg_off = nullptr, int n_wg = 0);

void launch_integrate(const double* pool, int64_t pool_stride, int64_t rec_off, int m, int bs16, int fast_p,
                      double* d_mu, int mu_stride, double* d_norm, int32_t* d_info, int n_sites, double* d_ws, hipStream_t st);

// In-place layout conversion of the records listed by (d_off, d_dim): to_bs16 != 0: plain -> BS16, else back.
// is_residual: records are [dJ | dh] (no g).  One workgroup per record.
void launch_convert_layout(double* pool, int64_t stride, const int64_t* d_off, const int32_t* d_dim, int n_records,
                           int n_sites, int to_bs16, int is_residual, int fast_p, hipStream_t st);
// d_flag[0] |= 1 if some record's J is not symmetric to 1e-10 * max|J| (plain layout)
void launch_check_symmetry(const double* pool, int64_t stride, const int64_t* d_off, const int32_t* d_dim,
                           int n_records, int n_sites, int32_t* d_flag, int fast_p, hipStream_t st);

// dst[site*dst_stride + dst_off[r] + t] = src[site*src_stride + src_off[r] + t], t < src_off'[r+1]-..: record copy
void launch_records(const double* src, int64_t src_stride, const int64_t* d_src_off, double* dst, int64_t dst_stride,
                    const int64_t* d_dst_off, const int64_t* d_len_off, int n_records, int n_sites, hipStream_t st);

void launch_copy_strided(const double* src, int64_t src_stride, double* dst, int64_t dst_stride, int64_t n,
                         int n_sites, hipStream_t st);
void launch_zero_strided(double* dst, int64_t dst_stride, int64_t n, int n_sites, hipStream_t st);

// site-minor layout (DevState::sm): transposition of the records listed by (d_off: padded plain offsets, d_poff: packed
// offsets) between plain[site * stride + off[r] + t] and sm[(poff[r] + t) * n_sites + site]; to_sm != 0: plain -> sm
void launch_site_minor(double* plain, int64_t plain_stride, double* sm, const int64_t* d_off, const int64_t* d_poff,
                       int n_records, int n_sites, int to_sm, hipStream_t st);
// integratebelief! of one belief of dimension <= 2 in the site-minor layout, one thread per site
void launch_integrate_sm(const double* pool_sm, int64_t packed_off_b, int m, double* d_mu, int mu_stride, double* d_norm,
                         int32_t* d_info, int n_sites, hipStream_t st);
// assignfactors! for a univariate BM on a tree (pgbp_bm_tree, p = 1) straight into the site-minor layout;
// fpool_sm may be null (beliefs only)
// (d_data: [row][site], rows padded to sm_row(n_sites): the transposed copy the engine keeps for this kernel)
void launch_bm_tree_fill_uni_sm(double* pool_sm, double* fpool_sm, const int64_t* d_poff, const int32_t* d_dim,
                                const int32_t* d_kind, const double* d_length, const int32_t* d_row, const double* d_data,
                                int n_rows, const double* d_Rinv, const double* d_logdetR, const double* d_mu, int per_site,
                                int n_clusters, int n_sites, hipStream_t st);
// record-aware copy: only the part of each record slot that the current layout uses (pgbp_kernels.hip)
// records of listed beliefs of one site <-> one contiguous buffer (the exchange buffer of a cut cluster graph)
void launch_pack_records(double* pool_site, const int64_t* d_rec_off, const int64_t* d_buf_off, int n, double* d_buf, int to_buf,
                         hipStream_t st);
void launch_copy_records(const double* src, int64_t src_stride, double* dst, int64_t dst_stride, const int64_t* d_boff,
                         const int32_t* d_dim, int n_records, int bs16, int fast_p, int n_sites, hipStream_t st);

// assignfactors! for a homogeneous BM on a tree (pgbp_bm_tree of include/pgbp.h): writes the cluster record into
// `pool` and `fpool` (current layout), one workgroup per (cluster, site)
void launch_bm_tree_fill(double* pool, int64_t pool_stride, double* fpool, int64_t fpool_stride, const int64_t* d_boff,
                         const int32_t* d_dim, const int32_t* d_kind, const double* d_length, const int32_t* d_row,
                         const double* d_data, int n_rows, int p, const double* d_Rinv, const double* d_logdetR,
                         const double* d_mu, int per_site, int bs16, int fast_p, int n_clusters, int n_sites,
                         hipStream_t st);

// assignfactors! for any linear-Gaussian model (pgbp_lg_families / pgbp_lg_params of include/pgbp.h; pgbp_lgfill.hip)
struct LgStatic {
  int32_t p, K, n_rates, n_rows;
  const int32_t *cl_off, *cl_fam;  // CSR cluster -> its families, in the reference's loop order
  const int32_t *n_parents, *child_pos, *data_row, *parent_pos;
  const double *length, *gamma;
  const int32_t* color;
  const double* data;
  const unsigned long long *child_mask, *parent_mask;  // null: complete data
  const double* data_sm;   // univariate batches: the tip data once more as [row][site] (rows padded: sm_row), lanes = sites read a line
  // univariate batches whose clusters all hold exactly ONE family with at most one parent and no scope mask (the clique tree of
  // a tree: cfg4): that family per cluster as one 48-byte record, or null (lg_fill_uni_sm_kernel then walks the general tables)
  const struct LgSimpleFam* simple;
};
struct LgSimpleFam {
  int32_t np, cpos, ppos, row, color, pad;   // parents (0: a root prior), positions of child / parent in the cluster (-1: fixed), data row, rate
  double length, gamma;
  double pad2[2];
};
static_assert(sizeof(LgSimpleFam) == 56 || sizeof(LgSimpleFam) == 64, "LgSimpleFam");
struct LgParams {
  int32_t model, per_site;
  const double *R, *alpha, *theta, *mu;  // device pointers
};
// one workgroup per (cluster, site); pool / fpool in the current layout (fpool may be null: beliefs only)
void launch_lg_fill(const LgStatic& F, const LgParams& M, double* pool, int64_t pool_stride, double* fpool,
                    int64_t fpool_stride, const int64_t* d_boff, const int32_t* d_dim, int bs16, int fast_p, int max_dim,
                    int n_clusters, int n_sites, hipStream_t st);
// univariate batches in the site-minor layout (every dimension <= 2): thread = (cluster, site)
void launch_lg_fill_uni_sm(const LgStatic& F, const LgParams& M, double* pool_sm, double* fpool_sm, const int64_t* d_poff,
                           const int32_t* d_dim, int n_clusters, int n_sites, hipStream_t st);

// free_energy (src/score.jl:162-182): per-belief terms + deterministic per-site sum -> out3[site] =
// (average energy, approximate entropy, free energy); info[site] (preset to INT_MAX) = first non-PD belief + 1
// lane-blocked version of the same fill (pgbp_fast.hip) for 2 <= p <= 16 when every cluster has dimension 0, p or 2p
// as its kind implies; fpool may be null (beliefs only: the score() body of src/calibration.jl:205 does not touch the
// factors).  Returns false if p has no instance.
bool launch_bm_tree_fill_fast(double* pool, int64_t pool_stride, double* fpool, int64_t fpool_stride,
                              const int64_t* d_boff, const int32_t* d_dim, const int32_t* d_kind, const double2* d_ithl,
                              const int32_t* d_row, const double* d_data, int n_rows, int p, const double* d_Rinv,
                              const double* d_logdetR, const double* d_mu, int per_site, int bs16, int n_clusters,
                              int n_sites, hipStream_t st);
void launch_bm_ithl(const double* d_length, const int32_t* d_kind, int p, double2* d_out, int n, hipStream_t st);

// d_big_idx / n_big: the beliefs of more than kFreeEnergyLdsMaxDim variables (their working matrix in d_ws: n_big * n_sites
// slabs of free_energy_ws_doubles(max_dim))
constexpr int kFreeEnergyLdsMaxDim = 96;    // [J | J_t | h] of a belief whole in 150 KB of LDS (one elimination; until round 4, second session, 139 with J_t in column blocks: a belief of 138 variables repeated its elimination 138 times, 250 ms)
int64_t free_energy_ws_doubles(int m);
void launch_free_energy(const double* pool, int64_t pool_stride, const double* fpool, int64_t fpool_stride,
                        const int64_t* d_boff, const int32_t* d_dim, int n_clusters, int n_beliefs, int max_dim, int bs16,
                        int fast_p, double* d_contrib, double* d_out3, int32_t* d_info, int n_sites, hipStream_t st,
                        const int32_t* d_big_idx = nullptr, int n_big = 0, double* d_ws = nullptr);

// sm != 0: the arrays are in the site-minor order [message][site]
void launch_reset_flags(const MsgDesc* msgs, int32_t* flags, int32_t* klflags, double* kldiv, int n_msgs, int n_sites,
                        int reset_kl, hipStream_t st, int sm = 0);
void launch_transpose_words_i32(const int32_t* src, int32_t* dst, int n, int n_sites, int to_sm, hipStream_t st);
void launch_transpose_words_f64(const double* src, double* dst, int n, int n_sites, int to_sm, hipStream_t st);

// residual_kldiv! (src/beliefs.jl:1060-1075) for the messages entries[e0 .. e0+n_entries) just sent by a level
// d_big_ent / n_big: the entries (absolute indices into d_entries) whose sepset has more than kKlLdsMaxS variables (their two
// systems in d_ws: n_big * n_sites slabs of kldiv_ws_doubles(max_s))
constexpr int kKlLdsMaxS = 96;   // [J0 | dJ | h0] of one sepset in a CU's LDS
int64_t kldiv_ws_doubles(int s);
void launch_residual_kldiv(const DevState& S, const Entry* d_entries, int e0, int n_entries, int max_s, double* d_kldiv,
                           int32_t* d_klflags, int n_sites, unsigned long long stop_below, hipStream_t st,
                           const int32_t* d_big_ent = nullptr, int n_big = 0, double* d_ws = nullptr);

// regularizebeliefs_bycluster! (src/clustergraphbeliefs.jl:235-275), plain layout; d_eps: [n_sites][n_clusters] scratch
void launch_regularize_bycluster(double* pool, int64_t pool_stride, const int64_t* d_boff, const int32_t* d_dim,
                                 const int32_t* d_nb_off, const int32_t* d_nb_msg, const MsgDesc* d_msgs,
                                 const int32_t* d_idx, const int32_t* d_sepcl, double* d_eps, int n_clusters,
                                 int n_sepsets, int n_sites, hipStream_t st);
void launch_reduce_flags(const int32_t* flags, int n_msgs, int n_sites, int32_t* d_iscal, hipStream_t st, int sm = 0);
// iscal[site] = notcal[site] == 0 (DevState::notcal)
void launch_iscal_from_notcal(const int32_t* d_notcal, int32_t* d_iscal, int n_sites, hipStream_t st);
// fail[site] = min(fail[site], key) where iscal[site] != 0 (key carries info = 0: a halt, not a failure)
void launch_halt_if_calibrated(const int32_t* d_iscal, unsigned long long* d_fail, unsigned long long key, int n_sites,
                               hipStream_t st);
// a fail word that reports a failed message (PosDefException.info in its low bits), as opposed to none / a halt key
__host__ __device__ inline bool is_failure_key(unsigned long long key) { return key != kNoFail && (key & ((1ull << kInfoBits) - 1)) != 0; }

// engine internals used by pgbp_dist.hip (defined in pgbp_engine.hip)
int engine_pack_gather_slot(pgbp_engine* e, int32_t slot_sites, double** d_slot, hipStream_t* st, int32_t* n_sites);
int engine_pack_records_device(pgbp_engine* e, int32_t site, int32_t n, const int32_t* beliefs, double* d_buf, int to_buf,
                               hipStream_t* st, int64_t* total);
int engine_fail(pgbp_engine* e, int code, const std::string& msg);
int engine_device(const pgbp_engine* e);
int engine_n_sites(const pgbp_engine* e);

}  // namespace pgbp
