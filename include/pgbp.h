/*
 * pgbp.h -- C ABI of the MI355X-native Gaussian belief-propagation calibration engine.
 *
 * Drop-in boundary for the message-passing hot path of
 * JuliaPhylo/PhyloGaussianBeliefProp.jl (reference paths relative to its repo root).
 * The reference has no FFI of its own; each entry point below replaces one Julia
 * function (cited), and is what a Julia `@ccall` shim binds.
 *
 * Conventions
 *   - all indices 0-based (the shim converts from Julia's 1-based);
 *   - a "belief" is a cluster (index < n_clusters) or a sepset (index >= n_clusters),
 *     clusters first, exactly like ClusterGraphBelief.belief (src/clustergraphbeliefs.jl:26-53);
 *   - "packed" belief storage = for each belief in index order: J (m*m doubles,
 *     column-major, full square), h (m), g (1); no padding; `pgbp_packed_size` doubles
 *     per site.  This is byte-for-byte what Julia's Matrix{Float64}/Vector{Float64}
 *     hold (src/beliefs.jl:122-127);
 *   - a directed message id is 2*k + dir for sepset k = (a, b): dir 0 is the message
 *     RECEIVED by a (sent by b), dir 1 the message received by b -- the (receiver, sender)
 *     key convention of ClusterGraphBelief.messageresidual (src/clustergraphbeliefs.jl:11-20);
 *   - n_sites >= 1 independent replicas ("sites": same graph, same scopes, different
 *     numbers) live in one engine; packed buffers are site-major;
 *   - every function returns PGBP_OK (0) or an error code; text via pgbp_last_error;
 *   - an engine owns one HIP stream; calls on one engine are not re-entrant
 *     (same as the reference: src/calibration.jl has no locking).
 *   - there is NO CPU fallback: pgbp_create fails with PGBP_ERR_NO_DEVICE without a GPU.
 */
#ifndef PGBP_H
#define PGBP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGBP_VERSION 1
#define PGBP_MAX_DIM 384 /* largest belief dimension the kernels accept (refused above).  Up to 128 variables the working
                            matrix of a message ([J | h]: 128 x 129 doubles = 132 KB) lives in a CU's 160 KB of LDS; above,
                            in a workspace in global memory that stays in the L2 (the 54-node clique of the reference's
                            documented clique tree, docs/src/man/clustergraphs.md:40-89, has 162 / 216 / 324 variables with
                            3 / 4 / 6 traits).  The scores and the KL residuals have the same two paths: in LDS up to 96
                            (residual_kldiv!: two systems side by side; free_energy) variables, in the workspace
                            above -- no size of the reference's is refused below PGBP_MAX_DIM (src/beliefs.jl:1060-1075 and
                            src/score.jl:162-182 have no bound). */

enum pgbp_status {
  PGBP_OK = 0,
  PGBP_ERR_INVALID = 1,    /* malformed description / argument (ErrorException in the reference: src/beliefs.jl:398-401) */
  PGBP_ERR_HIP = 2,        /* HIP runtime error */
  PGBP_ERR_NOT_TREE = 3,   /* a schedule entry is not a preorder edge list of a tree (src/clustergraph.jl:885-894) */
  PGBP_ERR_TOO_LARGE = 4,  /* a belief dimension exceeds PGBP_MAX_DIM, or more than 65535 sites of dimension > 2 */
  PGBP_ERR_NO_DEVICE = 5,  /* no HIP device: the engine has no CPU path */
  PGBP_ERR_STATE = 6       /* call out of order (e.g. calibrate before set_schedule) */
};

/* Static description of a cluster graph with allocated scopes: what
 * allocatebeliefs (src/beliefs.jl:478-594) + ClusterGraphBelief (src/clustergraphbeliefs.jl:89-109)
 * establish, with scopeindex(sepset, cluster) (src/beliefs.jl:389-405) precomputed once. */
typedef struct pgbp_desc {
  int32_t n_clusters;
  int32_t n_sepsets;
  const int32_t* dims;            /* [n_clusters + n_sepsets] dimension(belief) */
  const int32_t* sepset_clusters; /* [2*n_sepsets] the two incident clusters (a, b) of each sepset */
  const int64_t* scope_off;       /* [2*n_sepsets + 1] offsets into scope_idx; entry 2k+side */
  const int32_t* scope_idx;       /* scopeindex(sepset k, cluster a) then (sepset k, cluster b): positions
                                     of the sepset's variables inside the cluster, strictly increasing */
  int32_t n_sites;                /* >= 1 */
  int32_t device;                 /* HIP device ordinal */
} pgbp_desc;

/* calibrate! keyword arguments (src/calibration.jl:35-44) and the tolerance of
 * iscalibrated_residnorm! (src/beliefs.jl:994). */
typedef struct pgbp_opts {
  int32_t auto_stop;           /* auto */
  int32_t update_residualnorm; /* default 1 */
  int32_t update_residualkldiv;/* default 0 (src/calibration.jl:43); 1: residual_kldiv! after every message */
  int32_t reserved;
  double  atol;                /* 1e-5 */
} pgbp_opts;

/* Outcome of a traversal / calibration for one site. */
typedef struct pgbp_result {
  int32_t succ;         /* 1 unless a message failed (first tuple element of calibrate!: src/calibration.jl:59,82) */
  int32_t iscal;        /* iscalibrated_residnorm(beliefs) (src/clustergraphbeliefs.jl:168-169) */
  int32_t iter_reached; /* 1-based iteration / schedule tree at which calibration was first reached   */
  int32_t tree_reached; /*   ("calibration reached: iteration $i, schedule tree $j", src/calibration.jl:54); 0 if never */
  int32_t fail_iter;    /* 1-based iteration / tree / direction (0 post, 1 pre) / edge (0-based position in  */
  int32_t fail_tree;    /*   the tree's edge list) of the FIRST failing message in the reference's sequential */
  int32_t fail_dir;     /*   order (src/calibration.jl:121,147); fail_info = PosDefException.info             */
  int32_t fail_edge;    /*   (src/beliefupdates.jl:69-76). All 0 / -1 when succ == 1.                          */
  int32_t fail_info;
  int32_t reserved;
} pgbp_result;

typedef struct pgbp_engine pgbp_engine; /* opaque: device-resident ClusterGraphBelief */
typedef struct pgbp_plan pgbp_plan;     /* opaque: host-only layout + level schedule (no GPU needed) */

/* ---- host-only planning (no GPU): layout, message table, level schedule ------------- */
/* On failure *out still receives a plan object that only carries the message for pgbp_plan_last_error; destroy it. */
int  pgbp_plan_create(const pgbp_desc* desc, pgbp_plan** out);
void pgbp_plan_destroy(pgbp_plan* p);
/* schedule = vector of spanning trees, each the (pa_j, ch_j) index vectors of
 * spanningtree_clusterlist in preorder (src/clustergraph.jl:885-894); tree t owns
 * entries [tree_off[t], tree_off[t+1]). */
int  pgbp_plan_set_schedule(pgbp_plan* p, int32_t n_trees, const int32_t* tree_off,
                            const int32_t* pa_j, const int32_t* ch_j);
int64_t pgbp_plan_packed_size(const pgbp_plan* p);    /* doubles per site, beliefs */
int64_t pgbp_plan_residual_size(const pgbp_plan* p);  /* doubles per site, residuals: per message dJ (s*s) then dh (s) */
int32_t pgbp_plan_n_messages(const pgbp_plan* p);     /* 2 * n_sepsets */
/* level structure of one traversal (dir 0 = postorder, 1 = preorder): number of levels,
 * tasks and message entries; then the arrays (caller-allocated):
 * level_off[n_levels+1] -> tasks, task_off[n_tasks+1] -> entries,
 * entry_msg[n_entries] directed message id, entry_edge[n_entries] position in the tree's edge list,
 * entry_reuse[n_entries] 1 if the marginal of the previous entry of the task is reused; 2: the entry is the PROLOGUE of
 * the next entry of its task (pgbp_plan_prologues). */
int  pgbp_plan_traversal_sizes(const pgbp_plan* p, int32_t tree, int32_t dir,
                               int32_t* n_levels, int32_t* n_tasks, int32_t* n_entries);
int  pgbp_plan_traversal(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* level_off,
                         int32_t* task_off, int32_t* entry_msg, int32_t* entry_edge, int32_t* entry_reuse);
/* level_nfast[n_levels]: how many tasks of each level (they come first) run on the register-resident
 * kernel; the rest run on the generic in-LDS kernel. */
int  pgbp_plan_level_nfast(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* level_nfast);
/* Launch form of the fast-class tasks of one traversal (caller-allocated arrays, any may be NULL):
 * level_ngroups[n_levels]: workgroup passes ("groups" of 4 wavefront records) of each level;
 * *tail_levels: how many levels at the root end of the schedule tree (the last ones of a postorder, the first of a
 *   preorder) are walked by the single-workgroup tail launch (8 records per level);
 * records[6 * 4 * sum(level_ngroups)], tail_records[6 * 8 * tail_levels]: per record {valid, message id, first record of
 *   its task inside the group, records of the task, record that computes its marginal, mode bits (1 own receiver
 *   block, 2 accumulate task, 4 the accumulate task only touches the receiver's g, 8 the record has a prologue:
 *   pgbp_plan_prologues)}. */
int  pgbp_plan_groups(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* level_ngroups, int32_t* tail_levels,
                      int32_t* records, int32_t* tail_records);
/* Chunks of fused levels of one traversal: runs of narrow levels below the tail that go out as ONE launch each, one
 * workgroup per dependency-closed tree of tasks, a workgroup barrier between its levels.  *n_chunks; then (any may be
 * NULL) info[4 * n_chunks] = {first level, one past the last level, workgroups, groups} per chunk (groups < 0: a chunk
 * of generic-class tasks, |groups| groups of 8 TASK ids); wg_off[sum(workgroups + 1)]: per chunk, the group range of each of
 * its workgroups (relative to the chunk's first group); records[6 * 8 * sum(|groups|)]: chunk after chunk, 8 records per
 * group -- as for pgbp_plan_groups, or {1, task index in pgbp_plan_traversal's task_off, 0, 0, 0, 0} / all 0 for a generic
 * chunk. */
int  pgbp_plan_chunks(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* n_chunks, int32_t* info, int32_t* wg_off,
                      int32_t* records);
/* PROLOGUES of the records above (any may be NULL): one word per record of pgbp_plan_groups' records / tail_records and
 * of pgbp_plan_chunks' records, in the same order: the directed message X -> F the record's wavefront sends first -- F the
 * sender of the record's own message, the message lands on the block that one integrates out and integrates nothing
 * itself (a variable cluster of a Bethe graph into a factor cluster) -- or -1.  In pgbp_plan_traversal such a message is
 * the entry in front of the record's own, in the same task. */
int  pgbp_plan_prologues(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* level_pro, int32_t* tail_pro,
                         int32_t* chunk_pro);
/* CHAINS of the loop launches in the packed layout (csrc/pgbp_loop.hip), one word per record of the tail and of the chunks
 * (order as above; either may be NULL): kind | source record of the previous group << 8 | late << 16.  A workgroup loads
 * the operands of a group while the group before it still runs: kind 1 / 3 = the integrated block of the record's 2P-dim
 * sender / its P-dim sender's whole belief, 2 = the X of its prologue, was stored by that source record in the group
 * before and reaches this one through the workgroup's LDS; late = the group reads from memory something else the group
 * before it writes, or is the first of its walk: the workgroup waits for every store and loads at the group's top. */
int  pgbp_plan_chains(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* tail_chain, int32_t* chunk_chain);
/* The self-contained message records of the wave-per-task kernels (one 128-byte record per message of a generic-class
 * task; layout: struct GRec in csrc/pgbp_internal.hpp -- offsets of sender / receiver / sepset / residual inside a
 * site's pools, message id, sequence number, beliefs, index-pool offsets of the three maps, `next` = the record of the
 * task's next message or -1, dimensions {mf, mt, s, ni}, first kept / updated index when contiguous (255: not), reuse
 * flag, inline bits, perm[40] = the sender's variables, integrated first, kept last (bit 0), up[16] = the receiver's
 * positions (bit 1)).  *n_records; then (any may be NULL) level_first[n_levels] = the record of the level's first
 * generic-class task (its tasks follow in task order), task_first[n_tasks] = the first record of every task (-1: a
 * fast-class task), records[128 * n_records]. */
int  pgbp_plan_records(const pgbp_plan* p, int32_t tree, int32_t dir, int32_t* n_records, int32_t* level_first,
                       int32_t* task_first, uint8_t* records);
/* The ROW form of the postorder levels whose generic-class tasks are all small (at most 8 integrated and 8 kept variables)
 * and have at most four messages into their receiver: the level's messages one per row of 16 lanes (four rows = one
 * wavefront of bp_level_small4), the rows of a task consecutive, in the task's order, inside one wavefront.  *n_rows over
 * the traversal; then (any may be NULL) level_row0[n_levels], level_nrows[n_levels] (a multiple of four; 0: the level has
 * no row form) and rowmap[2 * n_rows] = pairs {record of pgbp_plan_records or -1 (an empty row), position in the task |
 * messages of the task << 8}.  mult! into a receiver happens in the order of the positions: the sequential task's sums. */
int  pgbp_plan_rows(const pgbp_plan* p, int32_t tree, int32_t dir, int64_t* n_rows, int64_t* level_row0,
                    int32_t* level_nrows, int32_t* rowmap);
const char* pgbp_plan_last_error(const pgbp_plan* p);

/* ---- engine lifetime ------------------------------------------------------------------ */
/* ClusterGraphBelief(beliefs, ...) constructor (src/clustergraphbeliefs.jl:89-109): allocates
 * beliefs, factors, message residuals (flags false / kldiv -1; empty messages born calibrated:
 * src/beliefs.jl:914-924) on the device. */
int  pgbp_create(const pgbp_desc* desc, pgbp_engine** out);
void pgbp_destroy(pgbp_engine* e);
const char* pgbp_last_error(const pgbp_engine* e); /* e == NULL: error of the last failed create */
int64_t pgbp_packed_size(const pgbp_engine* e);
int64_t pgbp_residual_size(const pgbp_engine* e);
int32_t pgbp_n_messages(const pgbp_engine* e);
int32_t pgbp_belief_dim(const pgbp_engine* e, int32_t belief); /* dimension of a cluster / sepset belief, -1: bad index */

/* ---- state transfer ------------------------------------------------------------------- */
/* Upload all beliefs of all sites (packed, site-major; host pointer). If snapshot_factors != 0 the
 * cluster part is also stored as the factors (init_factors_allocate, src/beliefs.jl:628-637). */
int  pgbp_set_beliefs(pgbp_engine* e, const double* packed, int32_t snapshot_factors);
int  pgbp_get_beliefs(pgbp_engine* e, double* packed);
int  pgbp_set_belief(pgbp_engine* e, int32_t site, int32_t belief, const double* rec); /* J,h,g of one belief */
int  pgbp_get_belief(pgbp_engine* e, int32_t site, int32_t belief, double* rec);
/* The EXCHANGE BUFFER of a cluster graph cut across devices (the roots of the subtrees a rank owns
 * before the top of a traversal, the records it owns after one): the records (J, h, g; the plain packing of
 * pgbp_get_belief) of `n` listed beliefs of one site, back to back in the order of the list -- gathered on the device
 * into one contiguous buffer and moved across the bus once (pack), or the reverse (unpack: overwrites those beliefs).
 * pgbp_packed_beliefs_size: doubles in that buffer (-1: a bad index).  This is the calibration loop's only exchange
 * when the traversals of src/calibration.jl:111-161 are cut by spanning-tree subtrees (sharding.py: NetworkCut). */
int64_t pgbp_packed_beliefs_size(pgbp_engine* e, int32_t n, const int32_t* beliefs);
int  pgbp_pack_beliefs(pgbp_engine* e, int32_t site, int32_t n, const int32_t* beliefs, double* buf);
int  pgbp_unpack_beliefs(pgbp_engine* e, int32_t site, int32_t n, const int32_t* beliefs, const double* buf);
/* all beliefs of ONE site (packed: pgbp_packed_size doubles): what a host reads back into the arrays of one
 * ClusterGraphBelief of a batch without downloading the other sites */
int  pgbp_get_site_beliefs(pgbp_engine* e, int32_t site, double* packed);
/* init_factors_frombeliefs! (src/beliefs.jl:746-761) */
int  pgbp_init_factors_frombeliefs(pgbp_engine* e);
/* init_beliefs_reset_fromfactors! (src/clustergraphbeliefs.jl:126-139) */
int  pgbp_reset_from_factors(pgbp_engine* e);
/* init_messagecalibrationflags_reset!(beliefs, reset_kl) (src/clustergraphbeliefs.jl:146-150) */
int  pgbp_reset_flags(pgbp_engine* e, int32_t reset_kl);
/* MessageResidual fields of every directed message, site-major: dJ,dh packed (pgbp_residual_size
 * doubles per site), iscalibrated_resid flags, kldiv and iscalibrated_kl flags (n_messages per site).
 * Any pointer may be NULL. */
int  pgbp_get_residuals(pgbp_engine* e, double* packed, int32_t* iscalibrated_resid, double* kldiv,
                        int32_t* iscalibrated_kl);
/* The same for ONE directed message of one site -- message 2k + dir is the one RECEIVED by end `dir` of sepset k
 * (messageresidual[(receiver, sender)], src/clustergraphbeliefs.jl:17-20): rec = dJ (s*s, column-major) then dh (s).
 * What a host that keeps the reference's objects fetches on first access after a calibrate! instead of downloading
 * every residual (the lazy write-back of INTEGRATION.md).  Any of the four output pointers may be NULL. */
int  pgbp_get_residual(pgbp_engine* e, int32_t site, int32_t msg, double* rec, int32_t* iscalibrated_resid, double* kldiv,
                       int32_t* iscalibrated_kl);

/* ---- the hot path ------------------------------------------------------------------------ */
int  pgbp_set_schedule(pgbp_engine* e, int32_t n_trees, const int32_t* tree_off,
                       const int32_t* pa_j, const int32_t* ch_j);
/* propagate_belief!(cluster_to, sepset, cluster_from, residual) (src/beliefupdates.jl:634-665) for
 * every site. info[site] = 0 ok, > 0 PosDefException.info (belief state of that site untouched,
 * exception "returned not thrown"); `sepset` is a belief index (>= n_clusters). */
int  pgbp_propagate(pgbp_engine* e, int32_t cluster_to, int32_t sepset, int32_t cluster_from,
                    const pgbp_opts* opts, int32_t* info);
/* residual_kldiv!(messageresidual[(to, from)], sepset) (src/beliefs.jl:1060-1075) for every site, for the
 * message last sent from cluster_from to cluster_to: updates that residual's kldiv and iscalibrated_kl
 * (left alone if the sepset belief, or the one before the message, is not positive definite).
 * iscalibrated_kl[n_sites] (may be NULL) receives the flag. */
int  pgbp_residual_kldiv(pgbp_engine* e, int32_t cluster_to, int32_t sepset, int32_t cluster_from,
                         const pgbp_opts* opts, int32_t* iscalibrated_kl);
/* regularizebeliefs_bycluster!(beliefs, clustergraph) (src/clustergraphbeliefs.jl:235-275) on the device:
 * per cluster eps = max(eps(Float64), max|J|); +eps on the cluster's diagonal at the scope of each
 * incident sepset, and on the sepset's diagonal.  Asynchronous on the engine's stream. */
int  pgbp_regularize_bycluster(pgbp_engine* e);
/* propagate_1traversal_postorder! / _preorder! (src/calibration.jl:111-161), dir 0 / 1.
 * results[n_sites]: succ, fail_* filled. */
int  pgbp_traverse(pgbp_engine* e, int32_t tree, int32_t dir, const pgbp_opts* opts, pgbp_result* results);
/* calibrate!(beliefs, schedule, niter; ...) (src/calibration.jl:35-84). results[n_sites].
 * With auto_stop EVERY SITE stops at the first schedule tree at which that site is calibrated, exactly the reference's
 * `auto` run on that site alone (the device skips the site's later traversals; results[s].iter_reached / tree_reached say
 * where it stopped); the call returns when every site has reached calibration or failed, or after niter iterations. */
int  pgbp_calibrate(pgbp_engine* e, int32_t niter, const pgbp_opts* opts, pgbp_result* results);
/* integratebelief!(obj, beliefindex) (src/clustergraphbeliefs.jl:194, src/beliefupdates.jl:168-200) for
 * every site: mu[n_sites * m] (may be NULL), norm[n_sites], info[n_sites] (0 ok, >0 not PD, may be NULL).
 * An all-zero belief gives mu = Inf, norm = g (src/beliefupdates.jl:189-191). */
int  pgbp_integrate(pgbp_engine* e, int32_t belief, double* mu, double* norm, int32_t* info);

/* ---- scores (second "next" row: SURVEY.md section 8(f)-2) -------------------------------------- */
/* free_energy(beliefs) (src/score.jl:162-182) for every site: out3[3*site + {0,1,2}] = (average energy,
 * approximate entropy, free energy = energy - entropy); factored_energy (src/score.jl:151-154) = the same
 * with the third value negated (the log-likelihood on a calibrated clique tree).  Uses the factors
 * (ClusterGraphBelief.factor) and the current beliefs.  info[site] (may be NULL) = 0, or 1-based index of the
 * first belief whose precision is not positive definite (its terms are NaN; the reference throws there). */
int  pgbp_free_energy(pgbp_engine* e, double* out3, int32_t* info);

/* ---- factor assignment on the device (first "next" row: SURVEY.md section 8(f)-1) ------------ */
/* assignfactors! (src/beliefs.jl:786-861) for a homogeneous Brownian motion with full rate matrix
 * (MvFullBrownianMotion: factor_treeedge src/evomodels/homogeneousbrownianmotion.jl:262-282,
 * absorbleaf!/absorbevidence! src/beliefupdates.jl:210-274) on a TREE with complete tip data and a fixed
 * root, when every cluster holds at most one node family {child, parent} (clique tree or Bethe graph of a
 * tree).  The static part is given once; every later call only moves the p*p + p + 1 model parameters. */
typedef struct pgbp_bm_tree {
  int32_t p;               /* traits */
  int32_t n_rows;          /* rows of `data` per site */
  const int32_t* kind;     /* [n_clusters] factor of the cluster:
                                0 edge child->parent, both in scope            J = [j -j; -j j], j = R^-1 / t
                                1 edge whose parent is the fixed root          mu absorbed on the parent's variables
                                2 edge whose child is a tip with data          data absorbed on the child's variables
                                3 tip attached to the fixed root               constant
                               -1 no factor (belief = 1), e.g. Bethe variable clusters */
  const double* length;    /* [n_clusters] branch length t > 0 (ignored for kind -1) */
  const int32_t* data_row; /* [n_clusters] row of the tip's data (kinds 2, 3), else -1 */
  const double* data;      /* [n_sites][n_rows][p] tip data, host pointer */
} pgbp_bm_tree;
int  pgbp_bm_tree_setup(pgbp_engine* e, const pgbp_bm_tree* t);
/* Fill every cluster belief AND the factors from (R^-1, log det R, mu), set sepsets to 1, reset the
 * calibration flags: init_beliefs_reset! + the factor loop of assignfactors! + what ClusterGraphBelief /
 * init_messagecalibrationflags_reset! do around it (src/calibration.jl:205-209).
 * Rinv: p*p (column-major, symmetric), mu: p.  per_site != 0: one parameter set per site
 * (Rinv [n_sites][p*p], logdetR [n_sites], mu [n_sites][p]), else one shared set.  Asynchronous. */
int  pgbp_bm_tree_assignfactors(pgbp_engine* e, const double* Rinv, const double* logdetR, const double* mu,
                                int32_t per_site);

/* ---- factor assignment on the device for every linear-Gaussian model of the reference, trees and networks ---- */
/* assignfactors! (src/beliefs.jl:786-861), all parent edges of positive length (no degenerate family); missing tip
 * values through the optional scope masks below (one missingness pattern for all sites of the engine): homogeneous / heterogeneous Brownian motion
 * (src/evomodels/homogeneousbrownianmotion.jl:222-351, heterogeneousmodels.jl:110-150), Ornstein-Uhlenbeck
 * (homogeneousornsteinuhlenbeck.jl:51-66), tree edges (factor_treeedge), hybrid nodes (factor_hybridnode,
 * evomodels.jl:314-330), root prior (factor_root, evomodels.jl:377-396), leaf data and fixed-root mean absorbed
 * (absorbleaf!, absorbevidence!: src/beliefupdates.jl:210-274).  Any cluster graph: a cluster may hold several
 * families (clique trees of networks, join graphs); they are added in the order given, the reference's loop order.
 * Static part (once): one entry per node family that carries a factor, in the order of the reference's loop
 * `for (ni, ci) in enumerate(node2cluster)` (preorder node index); per parent, in the order of node2family[ni][2:end]. */
typedef struct pgbp_lg_families {
  int32_t p;                 /* traits */
  int32_t n_families;
  int32_t max_parents;       /* K >= 1: row length of the per-parent arrays */
  int32_t n_rates;           /* number of p x p variance matrices in pgbp_lg_params.R */
  int32_t n_rows;            /* rows of `data` per site */
  const int32_t* cluster;    /* [n_families] node2cluster[ni] */
  const int32_t* n_parents;  /* [n_families] 0: root prior, 1: tree edge, >= 2: hybrid node */
  const int32_t* child_pos;  /* [n_families] position of the child's first variable in the cluster (its p traits are
                                contiguous: complete data), or -1: a tip, its data row is absorbed */
  const int32_t* data_row;   /* [n_families] row of the tip's data, else -1 */
  const int32_t* parent_pos; /* [n_families * K] position of parent k's first variable, or -1: the fixed root, whose
                                mean is absorbed */
  const double* length;      /* [n_families * K] length of the parent edge, > 0 */
  const double* gamma;       /* [n_families * K] inheritance of the parent edge (1 for a tree edge) */
  const int32_t* color;      /* [n_families * K] index into R of the parent edge's variance rate (heterogeneous models:
                                the edge's colour; homogeneous: 0).  Root prior family: entry [f*K] = index of the
                                prior variance among R */
  const double* data;        /* [n_sites][n_rows][p] tip data, host pointer; entries of traits outside a tip's child_mask
                                are ignored (may be NaN) */
  /* Missing data (both NULL: complete data, every in-scope node has all p traits).  Bit t of a mask = trait t.
   * child_mask[f]: an internal child's traits in scope (inscope column of the cluster belief: src/beliefs.jl:551-559) /
   * a tip's observed traits.  The factor keeps exactly these components of the residual: absorbleaf! marginalises
   * a tip's missing traits (src/beliefupdates.jl:266-274), assignfactors! an internal node's out-of-scope traits
   * (src/beliefs.jl:829-857; valid for the reference's models, whose q is a multiple of the identity).
   * parent_mask[f*K+k]: traits of parent k in scope (its block in the cluster holds popcount of them, in trait order);
   * must contain child_mask[f].  p <= 64. */
  const uint64_t* child_mask;  /* [n_families] or NULL */
  const uint64_t* parent_mask; /* [n_families * K] or NULL */
} pgbp_lg_families;
int  pgbp_lg_setup(pgbp_engine* e, const pgbp_lg_families* f);

enum pgbp_lg_model {
  PGBP_LG_BM = 0,  /* X_child | parents ~ N(sum_k gamma_k X_k, sum_k gamma_k^2 t_k R[color_k]) */
  PGBP_LG_OU = 1   /* a_k = exp(-alpha t_k): N(sum_k gamma_k (a_k X_k + (1 - a_k) theta), sum_k gamma_k^2 (1 - a_k^2) R[color_k]),
                      R = stationary variance sigma2 / (2 alpha); the reference has p = 1 (UnivariateOrnsteinUhlenbeck) */
};
/* Model parameters: everything that changes between two likelihood evaluations.  per_site != 0: one set per site
 * (R [n_sites][n_rates][p*p], alpha [n_sites], theta [n_sites][p], mu [n_sites][p]). */
typedef struct pgbp_lg_params {
  int32_t model;       /* enum pgbp_lg_model */
  int32_t per_site;
  const double* R;     /* [n_rates][p*p] column-major symmetric positive definite */
  const double* alpha; /* OU: [1] */
  const double* theta; /* OU: [p]; NULL for BM */
  const double* mu;    /* [p] root mean (fixed root: the value absorbed; random root: the prior mean) */
} pgbp_lg_params;
/* init_beliefs_reset! + the factor loop of assignfactors! + init_factors_frombeliefs! + flag reset.  A variance
 * sum_k vc_k R[color_k] that is not positive definite (the reference throws) makes that cluster's g NaN.  Asynchronous. */
int  pgbp_lg_assignfactors(pgbp_engine* e, const pgbp_lg_params* m);
/* The whole body of score(theta) (src/calibration.jl:195-221) on the device with the parameters of the LAST
 * pgbp_lg_assignfactors call: factor fill (beliefs only), postorder of tree 0, root integrate. */
int  pgbp_enqueue_loglik_lg(pgbp_engine* e, int32_t reps, const pgbp_opts* opts);

/* ---- device-side access for benchmarking / zero-copy callers ----------------------------- */
/* Enqueue `reps` full calibrate iterations (all trees, post+pre, flag reduction) without any host
 * synchronisation; the caller brackets with pgbp_sync. Resets beliefs from factors before each
 * repetition if reset_each != 0. */
int  pgbp_enqueue_calibrate(pgbp_engine* e, int32_t reps, int32_t reset_each, const pgbp_opts* opts);
/* Enqueue one log-likelihood evaluation body: reset from factors, postorder traversal of tree 0,
 * integrate the root cluster (src/calibration.jl:205-212 minus the host-side factor fill). The
 * per-site norms stay on the device until pgbp_fetch_loglik. */
/* (after an evaluation in which a message of some site failed -- pgbp_fetch_loglik's info[site] != 0 -- the beliefs of THAT
 * site are unspecified, as after a failed traversal of the reference, which stops at once and leaves its beliefs half
 * updated (src/calibration.jl:129-132); in particular the sepsets of that site which the stopped postorder did not reach
 * may still hold an earlier evaluation's values where the engine skipped their reset.  Other sites are unaffected.) */
int  pgbp_enqueue_loglik(pgbp_engine* e, int32_t reps, const pgbp_opts* opts);
int  pgbp_fetch_loglik(pgbp_engine* e, double* norm, int32_t* info);
/* The whole body of score(theta) (src/calibration.jl:195-221) on the device: pgbp_bm_tree_assignfactors with
 * the parameters uploaded by the LAST pgbp_bm_tree_assignfactors call, postorder of tree 0, root integrate. */
int  pgbp_enqueue_loglik_bm(pgbp_engine* e, int32_t reps, const pgbp_opts* opts);
int  pgbp_sync(pgbp_engine* e);
/* The comparison behind the residual-norm flags (iscalibrated_residnorm!, src/beliefs.jl:994-1003): the kernels do not
 * divide, they compare max|dh| (max|dJ|) with the largest x for which fl(x / divisor) <= atol -- divisor = fl(sqrt(s)) for dh,
 * s for dJ.  Pure host function (no device needed): returns that x; +inf for divisor 0 or atol = +inf, -1 (nothing passes)
 * for a NaN or negative atol. */
double pgbp_residual_threshold(double divisor, double atol);
/* Time `reps` repetitions of the enqueued work with HIP events on the engine's stream; returns the
 * total milliseconds in *ms_total and, per kernel family, accumulated device time is NOT measured here
 * (use rocprofv3). kind 0 = calibrate (reset_each honoured), 1 = loglik, 2 = loglik_bm, 3 = loglik_lg (2, 3: with the device factor fill). */
int  pgbp_time_enqueued(pgbp_engine* e, int32_t kind, int32_t reps, int32_t reset_each,
                        const pgbp_opts* opts, float* ms_total);
/* pgbp_enqueue_calibrate with HIP events on the engine's stream around the message launches, so that a caller who times
 * the whole region on the host gets the message kernels' share of exactly those repetitions: with reset_each = 0 ONE
 * event pair around all `reps` repetitions (they run back to back; an event record is a barrier packet, a pair per
 * repetition costs a few percent of a short calibrate), with reset_each != 0 a pair around the message launches of every
 * schedule tree.  pgbp_fetch_kernel_time waits for the stream and returns the sum of the event intervals (*ms_kernels)
 * and the number of message launches inside them. */
int  pgbp_enqueue_calibrate_timed(pgbp_engine* e, int32_t reps, int32_t reset_each, const pgbp_opts* opts);
int  pgbp_fetch_kernel_time(pgbp_engine* e, float* ms_kernels, int32_t* n_launches);
/* Time only the message-kernel launches of `reps` calibrate iterations (reset from factors before each):
 * one HIP event pair on the engine's stream brackets the back-to-back level launches of every traversal;
 * *ms_kernels = sum over traversals, *n_launches = number of level launches inside them. */
int  pgbp_time_message_kernels(pgbp_engine* e, int32_t reps, const pgbp_opts* opts,
                               float* ms_kernels, int32_t* n_launches);
/* Algorithmic bytes of one full calibrate iteration over all sites (SURVEY.md section 8(d) formula:
 * 8*[(mf^2+mf+1) + 4*(s^2+s+1) + (s^2+s)] per message) and message count. */
int  pgbp_traffic_model(const pgbp_engine* e, double* bytes_per_calibrate, int64_t* messages_per_calibrate);

/* ---- several GPUs ---------------------------------------------------------------------------
 * Independent sites are the dimension of the path that shards: calibrate!() of one site never reads another
 * (src/calibration.jl:35-60 works on one ClusterGraphBelief).  One big tree or network on one site does not shard
 * without an exchange step: run replicas (one engine per device).
 *
 * (1) ONE PROCESS, several devices.  A group = one engine (and stream) per listed device over contiguous site ranges
 * (shard i gets sites [first, first + count): the first n_sites % n_devices shards hold one site more).  Every call
 * fans out on one host thread per device and gathers in site order; buffers are the single-engine ones with
 * desc->n_sites = the TOTAL number of sites (desc->device is ignored).  No collective: the host owns every result.
 * A device may be listed more than once (two shards on one GPU: the rehearsal the tests run on a one-GPU box). */
typedef struct pgbp_group pgbp_group;
int  pgbp_group_create(const pgbp_desc* desc, int32_t n_devices, const int32_t* devices, pgbp_group** out);
void pgbp_group_destroy(pgbp_group* g);
const char* pgbp_group_last_error(const pgbp_group* g); /* g == NULL: error of the last failed create (this thread) */
int32_t pgbp_group_size(const pgbp_group* g);
pgbp_engine* pgbp_group_engine(pgbp_group* g, int32_t shard);  /* borrowed: any single-engine call on one shard */
int  pgbp_group_range(const pgbp_group* g, int32_t shard, int32_t* first_site, int32_t* n_sites);
int  pgbp_group_set_schedule(pgbp_group* g, int32_t n_trees, const int32_t* tree_off, const int32_t* pa_j, const int32_t* ch_j);
int  pgbp_group_set_beliefs(pgbp_group* g, const double* packed, int32_t snapshot_factors);
int  pgbp_group_get_beliefs(pgbp_group* g, double* packed);
int  pgbp_group_reset_from_factors(pgbp_group* g);
/* calibrate! on every site; results[n_sites_total].  With auto_stop every site stops at its own first calibrated tree
 * (pgbp_calibrate). */
int  pgbp_group_calibrate(pgbp_group* g, int32_t niter, const pgbp_opts* opts, pgbp_result* results);
int  pgbp_group_integrate(pgbp_group* g, int32_t belief, double* mu, double* norm, int32_t* info);
/* f->data = [n_sites_total][n_rows][p]; per-site parameter sets (m->per_site) = [n_sites_total][...]; each shard takes its
 * rows (n_rates, p: the strides of m->R). */
int  pgbp_group_lg_setup(pgbp_group* g, const pgbp_lg_families* f);
int  pgbp_group_lg_assignfactors(pgbp_group* g, const pgbp_lg_params* m, int32_t n_rates, int32_t p);
int  pgbp_group_enqueue_calibrate(pgbp_group* g, int32_t reps, int32_t reset_each, const pgbp_opts* opts);
int  pgbp_group_enqueue_loglik(pgbp_group* g, int32_t reps, const pgbp_opts* opts);
int  pgbp_group_enqueue_loglik_lg(pgbp_group* g, int32_t reps, const pgbp_opts* opts);
int  pgbp_group_fetch_loglik(pgbp_group* g, double* norm, int32_t* info); /* [n_sites_total] */
int  pgbp_group_sync(pgbp_group* g);

/* ---- several scope patterns behind one handle (src/beliefs.jl:551-559) ------------------------------------------------
 * A site whose data miss other traits at other tips than another site's has other scopes: an internal node has a trait in
 * scope iff some tip below it has a value for it, so belief dimensions and index maps -- the pgbp_desc -- differ.  One
 * engine per PATTERN, the sites that share it batched inside (descs[k]->n_sites of them), on the device descs[k] names;
 * every pattern describes the same cluster graph (clusters, sepsets, their order), so one schedule serves all.
 * sites[sum n_sites]: the caller's (global) index of every site, pattern after pattern: per-site results come back in the
 * caller's order.  Beliefs, factors and family tables have pattern-specific sizes: use pgbp_patterns_engine(k) with the
 * single-engine calls (pgbp_set_beliefs, pgbp_lg_setup, pgbp_lg_assignfactors, ...). */
typedef struct pgbp_patterns pgbp_patterns;
int  pgbp_patterns_create(int32_t n_patterns, const pgbp_desc* const* descs, const int32_t* sites, pgbp_patterns** out);
void pgbp_patterns_destroy(pgbp_patterns* g);
const char* pgbp_patterns_last_error(const pgbp_patterns* g);
int32_t pgbp_patterns_size(const pgbp_patterns* g);
pgbp_engine* pgbp_patterns_engine(pgbp_patterns* g, int32_t pattern);
int  pgbp_patterns_set_schedule(pgbp_patterns* g, int32_t n_trees, const int32_t* tree_off, const int32_t* pa_j,
                                const int32_t* ch_j);
int  pgbp_patterns_calibrate(pgbp_patterns* g, int32_t niter, const pgbp_opts* opts, pgbp_result* results); /* [n_sites] */
int  pgbp_patterns_enqueue_calibrate(pgbp_patterns* g, int32_t reps, int32_t reset_each, const pgbp_opts* opts);
int  pgbp_patterns_enqueue_loglik(pgbp_patterns* g, int32_t reps, const pgbp_opts* opts);
int  pgbp_patterns_enqueue_loglik_lg(pgbp_patterns* g, int32_t reps, const pgbp_opts* opts);
int  pgbp_patterns_fetch_loglik(pgbp_patterns* g, double* norm, int32_t* info);            /* [n_sites], caller's order */
int  pgbp_patterns_integrate(pgbp_patterns* g, int32_t belief, double* norm, int32_t* info); /* [n_sites], caller's order */
int  pgbp_patterns_sync(pgbp_patterns* g);

/* (2) ONE PROCESS PER GPU (torchrun / MPI / Distributed.jl).  Every rank owns an ordinary engine over its own sites;
 * the only exchange is ONE ncclAllGather (RCCL over xGMI) per pgbp_comm_gather_loglik call.  RCCL is bound at run time
 * (dlopen of librccl.so.1, or of the one path in the environment variable PGBP_RCCL_LIB); without it the calls return
 * PGBP_ERR_NO_DEVICE.
 * Rank 0 calls pgbp_comm_unique_id and hands the PGBP_COMM_ID_BYTES to the other ranks by whatever channel launched
 * them; every rank then calls pgbp_comm_create (ncclCommInitRank: collective, blocks until all ranks arrived). */
#define PGBP_COMM_ID_BYTES 128
typedef struct pgbp_comm pgbp_comm;
int  pgbp_comm_unique_id(uint8_t* id /* [PGBP_COMM_ID_BYTES] */);
int  pgbp_comm_create(const uint8_t* id, int32_t n_ranks, int32_t rank, int32_t device, pgbp_comm** out);
void pgbp_comm_destroy(pgbp_comm* c);
const char* pgbp_comm_last_error(const pgbp_comm* c);
/* Every rank contributes, from its engine e (same device as the communicator), the per-site log-likelihoods and info
 * words of its last pgbp_enqueue_loglik* / pgbp_integrate and its sites' (succ, iscal) of the last calibration, in a slot
 * of slot_sites sites (>= the largest number of sites on any rank; the same value on every rank).  On return, on EVERY
 * rank: norm_all[n_ranks * slot_sites] and info_all (may be NULL) hold rank r's sites at [r * slot_sites, ...) (unused
 * tail of a slot: 0), *all_succ / *all_iscal (may be NULL) the minimum over all sites of all ranks -- the
 * all-reduce(min) of the flags SURVEY.md section 8(e) asks for, carried by the same collective.  Enqueued on the engine's
 * stream behind the kernels that produce the values; returns after the result reached the host. */
int  pgbp_comm_gather_loglik(pgbp_comm* c, pgbp_engine* e, int32_t slot_sites, double* norm_all, int32_t* info_all,
                             int32_t* all_succ, int32_t* all_iscal);
/* What can fail on THIS rank before the collective pgbp_comm_create (RCCL not loadable, no such device), without
 * touching the other ranks: a launcher takes the minimum of (status == 0) over its ranks and only then lets every rank
 * enter pgbp_comm_create -- a rank that fails there alone would leave its peers blocked inside ncclCommInitRank. */
int  pgbp_comm_precheck(int32_t device);
/* The exchange step of a cluster graph CUT across ranks (the loop of src/calibration.jl:46-47 with each traversal,
 * src/calibration.jl:111-161, cut by spanning-tree subtrees: DESIGN.md section 6): rank r contributes the records (J, h, g;
 * the packing of pgbp_get_belief) of the beliefs lists[list_off[r] .. list_off[r + 1]) of `site`; ONE ncclAllGather carries
 * them, and every rank overwrites its copies of the other ranks' beliefs.  Every rank passes the same lists (the cut is a
 * function of the schedule).  Device to device: the payload never visits the host.  include_self != 0: a rank also
 * overwrites its own listed beliefs from its own slot (a self-test of the path on one rank). */
int  pgbp_comm_exchange_beliefs(pgbp_comm* c, pgbp_engine* e, int32_t site, const int32_t* list_off, const int32_t* lists,
                                int32_t include_self);
/* The host-side half of pgbp_comm_gather_loglik on its own (no GPU, no RCCL): recv = the gathered buffer, n_ranks slots
 * of 2 * slot_sites + 2 doubles each, a slot = [norm (slot_sites) | info (slot_sites) | succ | iscal] as a rank packs it;
 * outputs as for pgbp_comm_gather_loglik. */
int  pgbp_comm_unpack_slots(const double* recv, int32_t n_ranks, int32_t slot_sites, double* norm_all, int32_t* info_all,
                            int32_t* all_succ, int32_t* all_iscal);

#ifdef __cplusplus
}
#endif
#endif /* PGBP_H */
