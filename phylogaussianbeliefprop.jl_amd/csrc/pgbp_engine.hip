// Device-resident ClusterGraphBelief + calibration driver behind the C ABI of include/pgbp.h.
// One engine = one HIP stream; every level of a traversal is one kernel launch on it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <limits>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "pgbp_internal.hpp"
#include "pgbp_kernels.hpp"

using namespace pgbp;

namespace {

thread_local std::string g_create_error;  // (per thread: pgbp_group creates its engines concurrently)

// Kernel arguments in device memory: without it every launch of a narrow level pays a host-memory read for its
// 0x90-byte argument segment (measured: cfg3 +7 %, cfg5 +20 % per calibrate).  The HIP runtime reads the variable
// when it initialises, i.e. at the process's first HIP call; this constructor runs when the library is loaded
// (priority 101: ahead of the code object registration of this library), so a host that has not touched HIP yet --
// a Julia session that `dlopen`s libpgbp.so -- gets the setting without reading INTEGRATION.md.  A value the user
// exported is left alone; a host that initialised HIP earlier keeps whatever it had.
__attribute__((constructor(101))) void pgbp_default_environment() { setenv("HIP_FORCE_DEV_KERNARG", "1", 0); }

struct DevTraversal {
  int32_t* d_task_off = nullptr;
  Entry* d_entries = nullptr;
  URec* d_urecs = nullptr;           // the entries as self-contained records (plans of univariate site batches: bp_level_uni1)
  FEntry* d_fentries = nullptr;
  FPro* d_fpros = nullptr;           // Traversal::fpros / cpros (null: no record of the traversal has a prologue)
  FPro* d_cpros = nullptr;
  FEntry* d_centries = nullptr;      // Traversal::centries: the groups of the chunks of fused levels
  int32_t* d_chunk_wg_off = nullptr; // Traversal::chunk_wg_off
  int32_t* d_cgroups = nullptr;      // Traversal::cgroups as first records of the tasks (Traversal::task_grec)
  int32_t* d_cgroups_task = nullptr; // Traversal::cgroups as they are (task ids): the thread-per-site chunk kernel
  GRec* d_grecs = nullptr;           // Traversal::grecs
  int32_t* d_rowmap = nullptr;       // Traversal::rowmap
  // residual_kldiv! of sepsets beyond the LDS instance (kKlLdsMaxS): the entries concerned, ascending (device copy + host copy)
  int32_t* d_kl_big = nullptr;
  std::vector<int32_t> kl_big;
};

}  // namespace

struct pgbp_engine {
  Plan plan;
  hipStream_t st = nullptr;
  // the KL flags and divergences are as the last reset left them (nothing computed a KL residual since): the next reset of
  // the message flags leaves them alone (two of its three arrays; cfg4: 2.6 GB a call)
  bool kl_flags_clean = false, kl_div_clean = false;
  // device state
  double* d_pool = nullptr;    // [n_sites][pool_stride] beliefs
  double* d_fpool = nullptr;   // [n_sites][cluster_stride] factors
  double* d_rpool = nullptr;   // [n_sites][rpool_stride] residuals
  MsgDesc* d_msgs = nullptr;
  int32_t* d_idx = nullptr;
  int32_t* d_flags = nullptr;
  int32_t* d_status = nullptr;
  double* d_kldiv = nullptr;
  int32_t* d_klflags = nullptr;     // [n_sites][n_msgs] iscalibrated_kl
  int32_t* d_nb_off = nullptr;      // [n_clusters+1] -> d_nb_msg: the messages each cluster sends (regularisers)
  int32_t* d_nb_msg = nullptr;
  int32_t* d_sepcl = nullptr;       // [2*n_sepsets] sepset -> its two clusters
  double* d_eps = nullptr;          // [n_sites][n_clusters] regularisation scratch
  double* d_thr = nullptr;          // DevState::thr: residual-norm thresholds of the current tolerance
  double* d_logtab = nullptr;       // DevState::logtab
  std::vector<double> h_thr;
  double thr_atol = 0.0;
  bool thr_valid = false;
  unsigned long long* d_fail = nullptr;
  int32_t* d_poison = nullptr;      // [n_sites][n_clusters]
  int32_t* d_iscal = nullptr;       // [n_sites]
  int32_t* d_notcal = nullptr;      // [n_sites] DevState::notcal
  int32_t* d_iscal_hist = nullptr;  // [hist_cap][n_sites]
  int64_t hist_cap = 0;
  int64_t* d_boff = nullptr;        // record offset tables for pack/unpack
  int64_t* d_packed_off = nullptr;
  int64_t* d_roff = nullptr;
  int64_t* d_rpacked_off = nullptr;
  double* d_mu = nullptr;    // [n_sites][max_dim]
  double* d_norm = nullptr;  // [n_sites]
  int32_t* d_info = nullptr; // [n_sites]
  int32_t* d_bdim = nullptr;        // [n_beliefs] dims (layout conversion)
  int32_t* d_rdim = nullptr;        // [n_msgs] sepset dim of every directed message
  int32_t* d_symflag = nullptr;     // != 0: some precision matrix is not symmetric
  int32_t max_s = 0;                // largest sepset dimension
  // site-minor copies of the pools (univariate batches: every dimension <= 2), allocated at first use
  double *d_pool_sm = nullptr, *d_fpool_sm = nullptr, *d_rpool_sm = nullptr;
  // second copies of the per-message / per-cluster word arrays: a layout switch transposes into them and swaps
  int32_t *d_flags_alt = nullptr, *d_status_alt = nullptr, *d_klflags_alt = nullptr, *d_poison_alt = nullptr;
  double* d_kldiv_alt = nullptr;
  bool layout_sm = false;           // the live state is in the site-minor buffers
  bool layout_bs16 = false;         // current device layout of 16/32-dim beliefs and 16-dim residuals
  bool sym_known = false, sym_ok = false;
  int32_t* d_one_task_off = nullptr;  // single-message task for pgbp_propagate
  Entry* d_one_entry = nullptr;
  GRec* d_one_rec = nullptr;
  std::vector<DevTraversal> dpost, dpre;
  std::vector<FEntry*> d_tail;   // per tree: the tail groups of its postorder followed by those of its preorder
  std::vector<FPro*> d_tail_pros;  // ... and their prologues (null: none)
  // HIP events of the last pgbp_enqueue_calibrate_timed call (resolved by pgbp_fetch_kernel_time)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> kernel_events;
  int32_t kernel_launches = 0;
  double* d_ws = nullptr;        // workspace of the kernels whose working matrix exceeds the LDS (beliefs above kLdsMaxDim)
  int64_t ws_cap = 0;
  double* d_gather = nullptr;    // send slot of pgbp_comm_gather_loglik: [norm | info | succ, iscal]
  int64_t gather_cap = 0;
  double* d_xbuf = nullptr;      // exchange buffer of pgbp_pack_beliefs / pgbp_unpack_beliefs
  int64_t xbuf_cap = 0;
  int64_t* d_xoff = nullptr;     // ... its record offsets: [n] in the pool, [n + 1] in the buffer
  int64_t xoff_cap = 0;
  bool have_factors = false;
  // pgbp_bm_tree: static description + last parameters of the device factor fill
  int32_t bm_p = 0, bm_rows = 0, bm_per_site = 0;
  int32_t *d_bm_kind = nullptr, *d_bm_row = nullptr;
  double* d_bm_data_sm = nullptr;   // [row][site] copy of d_bm_data (p = 1)
  double *d_bm_length = nullptr, *d_bm_data = nullptr, *d_bm_Rinv = nullptr, *d_bm_logdet = nullptr, *d_bm_mu = nullptr;
  double2* d_bm_ithl = nullptr;    // [n_clusters] (1 / t, (p / 2) log t): launch_bm_ithl at set-up
  // pgbp_lg_families: static description + last parameters of the general linear-Gaussian factor fill
  LgStatic lg{};                  // device pointers (owned: lg_bufs)
  LgParams lgp{};
  std::vector<void*> lg_bufs;
  bool lg_ready = false, lg_have_params = false, lg_uni_ok = false;
  double *d_lg_R = nullptr, *d_lg_alpha = nullptr, *d_lg_theta = nullptr, *d_lg_mu = nullptr;
  std::string err;
  // a step of an asynchronous enqueue that could not be issued (a workspace that could not be allocated: ensure_ws has
  // set `err`): the launches that needed it were skipped; every entry point that enqueued, and the next pgbp_sync /
  // fetch, returns this code instead of results of messages that never ran
  int enqueue_rc = PGBP_OK;

  int fail(int code, const std::string& msg) {
    err = msg;
    return code;
  }
  int take_enqueue_rc() {
    const int rc = enqueue_rc;
    enqueue_rc = PGBP_OK;
    return rc;
  }
};

#define HIPCHK(e, call)                                                                          \
  do {                                                                                           \
    hipError_t _rc = (call);                                                                     \
    if (_rc != hipSuccess)                                                                       \
      return (e)->fail(PGBP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_rc));        \
  } while (0)

namespace {

// Every entry point makes its engine's device current first: a host that drives several engines on several devices (one
// host thread per engine: pgbp_group, pgbp_dist.hip) needs no hipSetDevice of its own.  The current device is per thread.
struct DeviceScope {
  explicit DeviceScope(const pgbp_engine* e) {
    if (e) (void)hipSetDevice(e->plan.device);
  }
};

// forget earlier failures: fail keys and the downstream-of-a-failure marks
int reset_fail(pgbp_engine* e);

template <class T>
int dev_alloc(pgbp_engine* e, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIPCHK(e, hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return PGBP_OK;
}

template <class T>
int upload(pgbp_engine* e, T** p, const std::vector<T>& v) {
  int rc = dev_alloc(e, p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(e, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return PGBP_OK;
}

// the largest x with fl(x / c) <= atol (x -> fl(x / c) is monotone; IEEE division on the host = the device's)
double quotient_threshold(double c, double atol) {
  if (!(atol >= 0.0)) return -1.0;                 // NaN or negative tolerance: nothing passes (the maxima are >= 0)
  if (std::isinf(atol) || c == 0.0) return INFINITY;
  double x = atol * c;
  if (std::isinf(x)) x = std::numeric_limits<double>::max();
  while (x / c > atol) x = std::nextafter(x, -INFINITY);
  while (std::nextafter(x, INFINITY) / c <= atol) x = std::nextafter(x, INFINITY);
  return x;
}

// DevState::thr for the tolerance of this call (rebuilt and uploaded only when the tolerance changes)
int ensure_thresholds(pgbp_engine* e, double atol) {
  if (e->thr_valid && std::memcmp(&atol, &e->thr_atol, sizeof(double)) == 0) return PGBP_OK;
  HIPCHK(e, hipStreamSynchronize(e->st));   // (an earlier upload may still read the host copy)
  e->h_thr.assign(2 * (PGBP_MAX_DIM + 1), INFINITY);
  for (int s = 1; s <= PGBP_MAX_DIM; ++s) {
    e->h_thr[s] = quotient_threshold(std::sqrt((double)s), atol);
    e->h_thr[PGBP_MAX_DIM + 1 + s] = quotient_threshold(std::sqrt((double)s * (double)s), atol);
  }
  HIPCHK(e, hipMemcpyAsync(e->d_thr, e->h_thr.data(), sizeof(double) * e->h_thr.size(), hipMemcpyHostToDevice, e->st));
  e->thr_atol = atol;
  e->thr_valid = true;
  return PGBP_OK;
}

DevState dev_state(pgbp_engine* e, const pgbp_opts* o) {
  DevState S;
  S.pool = e->d_pool;
  S.pool_stride = e->plan.pool_stride();
  S.rpool = e->d_rpool;
  S.rpool_stride = e->plan.rpool_stride();
  S.msgs = e->d_msgs;
  S.idx = e->d_idx;
  S.flags = e->d_flags;
  S.status = e->d_status;
  S.fail = e->d_fail;
  S.poison = e->d_poison;
  S.n_clusters = e->plan.n_clusters;
  S.n_msgs = e->plan.n_msgs();
  S.update_resnorm = o ? o->update_residualnorm : 1;
  S.atol = o ? o->atol : 1e-5;
  // (DevState::thr belongs to this tolerance: every entry point went through check_opts -> ensure_thresholds first)
  S.thr = e->d_thr;
  S.logtab = reinterpret_cast<const double2*>(e->d_logtab);
  const int sp = e->plan.fast_p > 0 ? e->plan.fast_p : 0;
  S.thr_h_p = e->h_thr.empty() ? 0.0 : e->h_thr[sp];
  S.thr_J_p = e->h_thr.empty() ? 0.0 : e->h_thr[PGBP_MAX_DIM + 1 + sp];
  S.bs16 = e->layout_bs16 ? 1 : 0;
  S.fast_p = e->plan.fast_p;
  S.sm = e->layout_sm ? 1 : 0;
  S.n_sites = e->plan.n_sites;
  S.packed_off = e->d_packed_off;
  S.rpacked_off = e->d_rpacked_off;
  S.sep_zero = 0;
  S.notcal = nullptr;
  if (e->layout_sm) {
    S.pool = e->d_pool_sm;
    S.rpool = e->d_rpool_sm;
  }
  return S;
}

// The downstream-of-a-failure marks are one word per (site, cluster) -- 1.3 GB for cfg4's 8 000 problems x 40 000 clusters,
// a memset of it in front of every enqueue call was 3 % of a sharded step -- and a mark is only ever set beside a failure key
// in the site's fail word (the message that failed, or one downstream of it), which stays until the next reset_fail: so the
// marks are cleared only where some site's fail word holds a failure (every workgroup looks at the fail words first).
__global__ __launch_bounds__(256) void clear_poison_if_failed(const unsigned long long* __restrict__ fail, int n_sites,
                                                              int32_t* __restrict__ poison, int64_t n_words) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  for (int s = threadIdx.x; s < n_sites; s += blockDim.x)
    if (is_failure_key(fail[s])) any = 1;   // (every writer writes the same value)
  __syncthreads();
  if (!any) return;
  int4* p4 = reinterpret_cast<int4*>(poison);
  const int64_t n4 = n_words / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    p4[i] = make_int4(0, 0, 0, 0);
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x)
    poison[i] = 0;
}

// init_messagecalibrationflags_reset! on the device (src/clustergraphbeliefs.jl:190-202); reset_kl: the KL divergences too
void reset_message_flags(pgbp_engine* e, int reset_kl) {
  const int mode = ((reset_kl && !e->kl_div_clean) ? 1 : 0) | (e->kl_flags_clean ? 0 : 2);
  launch_reset_flags(e->d_msgs, e->d_flags, e->d_klflags, e->d_kldiv, e->plan.n_msgs(), e->plan.n_sites, mode, e->st,
                     e->layout_sm ? 1 : 0);
  e->kl_flags_clean = true;
  if (reset_kl) e->kl_div_clean = true;
}

int reset_fail(pgbp_engine* e) {
  const size_t ns = (size_t)e->plan.n_sites;
  // (either layout: the site-minor one has padded rows)
  const int64_t n_words = (int64_t)sm_row(e->plan.n_sites) * (int64_t)e->plan.n_clusters;
  const int64_t want = (n_words / 4 + 255) / 256;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, 2048));
  hipLaunchKernelGGL(clear_poison_if_failed, dim3(grid), dim3(256), 0, e->st, e->d_fail, (int)ns, e->d_poison, n_words);
  HIPCHK(e, hipMemsetAsync(e->d_fail, 0xFF, sizeof(unsigned long long) * ns, e->st));
  return PGBP_OK;
}

// Switch the device layout of every 16/32-dimensional belief (beliefs + factors) and every 16-dimensional
// residual between plain (ABI order, what the generic kernel and all transfers use) and BS16 (symmetric
// block-packed, what the register-resident kernel streams).  plain -> BS16 keeps the upper triangle, so it
// is only taken when every precision matrix is symmetric (checked once per upload).
// Site-minor layout of univariate batches (DevState::sm): the state moves between the plain pools and the site-minor
// buffers as a whole; only the traversals of bp_level_uni, the root integrate, reset-from-factors and the univariate
// factor fill work on the site-minor buffers, every other entry point asks for the plain layout first.
int ensure_site_minor(pgbp_engine* e, bool want) {
  const Plan& p = e->plan;
  if (want == e->layout_sm) return PGBP_OK;
  const size_t ns = (size_t)sm_row(p.n_sites);   // (rows of the site-minor buffers: padded)
  if (want && !e->d_pool_sm) {
    int rc;
    if ((rc = dev_alloc(e, &e->d_pool_sm, ns * (size_t)p.packed_off.back()))) return rc;
    if ((rc = dev_alloc(e, &e->d_fpool_sm, ns * (size_t)p.packed_off[p.n_clusters]))) return rc;
    if ((rc = dev_alloc(e, &e->d_rpool_sm, ns * (size_t)p.rpacked_off.back()))) return rc;
    const size_t nm = (size_t)p.n_msgs();
    if ((rc = dev_alloc(e, &e->d_flags_alt, ns * nm))) return rc;
    if ((rc = dev_alloc(e, &e->d_status_alt, ns * nm))) return rc;
    if ((rc = dev_alloc(e, &e->d_klflags_alt, ns * nm))) return rc;
    if ((rc = dev_alloc(e, &e->d_kldiv_alt, ns * nm))) return rc;
    if ((rc = dev_alloc(e, &e->d_poison_alt, ns * (size_t)std::max(1, p.n_clusters)))) return rc;
  }
  const int to = want ? 1 : 0;
  // flags, status, KL words and poison marks: [site][index] <-> [index][site]
  launch_transpose_words_i32(e->d_flags, e->d_flags_alt, p.n_msgs(), p.n_sites, to, e->st);
  launch_transpose_words_i32(e->d_status, e->d_status_alt, p.n_msgs(), p.n_sites, to, e->st);
  launch_transpose_words_i32(e->d_klflags, e->d_klflags_alt, p.n_msgs(), p.n_sites, to, e->st);
  launch_transpose_words_f64(e->d_kldiv, e->d_kldiv_alt, p.n_msgs(), p.n_sites, to, e->st);
  launch_transpose_words_i32(e->d_poison, e->d_poison_alt, p.n_clusters, p.n_sites, to, e->st);
  std::swap(e->d_flags, e->d_flags_alt);
  std::swap(e->d_status, e->d_status_alt);
  std::swap(e->d_klflags, e->d_klflags_alt);
  std::swap(e->d_kldiv, e->d_kldiv_alt);
  std::swap(e->d_poison, e->d_poison_alt);
  launch_site_minor(e->d_pool, p.pool_stride(), e->d_pool_sm, e->d_boff, e->d_packed_off, p.n_beliefs(), p.n_sites, to, e->st);
  launch_site_minor(e->d_fpool, p.cluster_stride(), e->d_fpool_sm, e->d_boff, e->d_packed_off, p.n_clusters, p.n_sites, to, e->st);
  launch_site_minor(e->d_rpool, p.rpool_stride(), e->d_rpool_sm, e->d_roff, e->d_rpacked_off, p.n_msgs(), p.n_sites, to, e->st);
  e->layout_sm = want;
  return PGBP_OK;
}

bool want_site_minor(const pgbp_engine* e) {
  return e->plan.tune.packed_layouts && e->plan.max_dim <= 2 && e->plan.n_sites >= 64;
}

int ensure_layout(pgbp_engine* e, bool want_bs16, bool want_sm = false) {
  const Plan& p = e->plan;
  if (want_sm != e->layout_sm) {
    const int rc = ensure_site_minor(e, want_sm);
    if (rc) return rc;
  }
  if (want_bs16 == e->layout_bs16) return PGBP_OK;
  if (want_bs16) {
    if (!e->sym_known) {
      HIPCHK(e, hipMemsetAsync(e->d_symflag, 0, sizeof(int32_t), e->st));
      launch_check_symmetry(e->d_pool, p.pool_stride(), e->d_boff, e->d_bdim, p.n_beliefs(), p.n_sites, e->d_symflag, p.fast_p, e->st);
      launch_check_symmetry(e->d_fpool, p.cluster_stride(), e->d_boff, e->d_bdim, p.n_clusters, p.n_sites, e->d_symflag, p.fast_p, e->st);
      int32_t flag = 0;
      HIPCHK(e, hipMemcpyAsync(&flag, e->d_symflag, sizeof(flag), hipMemcpyDeviceToHost, e->st));
      HIPCHK(e, hipStreamSynchronize(e->st));
      e->sym_known = true;
      e->sym_ok = flag == 0;
    }
    if (!e->sym_ok) return PGBP_OK;  // stay plain: exact reference semantics for asymmetric input
  }
  const int to = want_bs16 ? 1 : 0;
  launch_convert_layout(e->d_pool, p.pool_stride(), e->d_boff, e->d_bdim, p.n_beliefs(), p.n_sites, to, 0, p.fast_p, e->st);
  launch_convert_layout(e->d_fpool, p.cluster_stride(), e->d_boff, e->d_bdim, p.n_clusters, p.n_sites, to, 0, p.fast_p, e->st);
  launch_convert_layout(e->d_rpool, p.rpool_stride(), e->d_roff, e->d_rdim, p.n_msgs(), p.n_sites, to, 1, p.fast_p, e->st);
  e->layout_bs16 = want_bs16;
  return PGBP_OK;
}

// layout wanted by the traversals of the current schedule
// The postorder of tree 0 writes every sepset and, straight after a reset, would read only zeros from them: true when
// that traversal runs entirely on kernels that honour DevState::sep_zero (the register-resident ones; the thread-per-site
// ones of the site-minor layout) and the schedule tree spans every sepset (a clique tree, the Bethe graph of a tree).
bool fresh_sepsets_shortcut(const pgbp_engine* e) {
  const Plan& p = e->plan;
  // (the site-minor layout belongs to the thread-per-site kernels, which honour it too)
  return (e->layout_sm || p.all_fast) && !p.trees.empty() && (int)p.trees[0].pa.size() == p.n_sepsets;
}

bool want_bs16(const pgbp_engine* e) {
  return e->plan.tune.packed_layouts && e->plan.all_fast && e->plan.fast_p > 0 && e->plan.fast_p % 2 == 0;  // packed tiles: even P
}

void free_traversals(pgbp_engine* e) {
  for (auto* v : {&e->dpost, &e->dpre}) {
    for (auto& d : *v) {
      if (d.d_task_off) (void)hipFree(d.d_task_off);
      if (d.d_entries) (void)hipFree(d.d_entries);
      if (d.d_urecs) (void)hipFree(d.d_urecs);
      if (d.d_fentries) (void)hipFree(d.d_fentries);
      if (d.d_fpros) (void)hipFree(d.d_fpros);
      if (d.d_cpros) (void)hipFree(d.d_cpros);
      if (d.d_centries) (void)hipFree(d.d_centries);
      if (d.d_chunk_wg_off) (void)hipFree(d.d_chunk_wg_off);
      if (d.d_cgroups) (void)hipFree(d.d_cgroups);
      if (d.d_cgroups_task) (void)hipFree(d.d_cgroups_task);
      if (d.d_grecs) (void)hipFree(d.d_grecs);
      if (d.d_rowmap) (void)hipFree(d.d_rowmap);
      if (d.d_kl_big) (void)hipFree(d.d_kl_big);
    }
    v->clear();
  }
  for (FEntry* t : e->d_tail)
    if (t) (void)hipFree(t);
  e->d_tail.clear();
  for (FPro* t : e->d_tail_pros)
    if (t) (void)hipFree(t);
  e->d_tail_pros.clear();
}

// at least `doubles` of workspace (grown when a launch needs more: the stream is drained first, kernels in flight use it)
int ensure_ws(pgbp_engine* e, int64_t doubles) {
  if (doubles <= e->ws_cap) return PGBP_OK;
  HIPCHK(e, hipStreamSynchronize(e->st));
  if (e->d_ws) (void)hipFree(e->d_ws);
  e->d_ws = nullptr;
  e->ws_cap = 0;
  HIPCHK(e, hipMalloc(reinterpret_cast<void**>(&e->d_ws), sizeof(double) * (size_t)doubles));
  e->ws_cap = doubles;
  return PGBP_OK;
}

// Every entry point that launches a message kernel comes through here BEFORE its first launch: the options are sane and
// the threshold table of DevState::thr is the one of this tolerance (a failed upload is this call's error, not a launch
// with whatever the table held).
int check_opts(pgbp_engine* e, const pgbp_opts* o) {
  if (o && !(o->atol >= 0.0)) return e->fail(PGBP_ERR_INVALID, "pgbp_opts.atol must be >= 0");
  const int rc = ensure_thresholds(e, o ? o->atol : 1e-5);
  if (rc != PGBP_OK) e->thr_valid = false;
  return rc;
}

unsigned long long seq_stride(const pgbp_engine* e) {
  size_t mx = 0;
  for (const Tree& t : e->plan.trees) mx = std::max(mx, t.pa.size());
  return 2ull * (mx + 1);
}

// how many levels at the root end of a traversal the tail launch takes over (0: none)
int tail_levels(const pgbp_engine* e, const Traversal& tr, bool kl) {
  // residual_kldiv! runs between levels; the site-minor layout belongs to the thread-per-site kernel
  if (kl || e->layout_sm || !e->plan.tune.tail) return 0;
  return tr.tail_levels;
}

// a loop launch (the tail, a chunk of fused levels): pgbp_loop.hip in the packed layout, pgbp_fast.hip's loop mode otherwise
void launch_loop_or_tail(pgbp_engine* e, const DevState& S, const FEntry* recs, const FPro* pros, int ngroups, int split,
                         unsigned long long seq_base, unsigned long long stop_a, unsigned long long stop_b,
                         const int32_t* d_wg_off, int n_wg) {
  if (e->plan.tune.loop2 && S.bs16 && e->plan.fast_p % 2 == 0)
    launch_loop16(S, recs, pros, ngroups, split, e->plan.n_sites, seq_base, stop_a, stop_b, e->st, d_wg_off, n_wg);
  else
    launch_fast16(S, recs, pros, kFastTail, ngroups, split, e->plan.n_sites, seq_base, stop_a, stop_b, e->st, d_wg_off, n_wg);
}

// levels [L0, L1) of one traversal: one launch per level (two where a level mixes fast-class and generic tasks)
void enqueue_levels(pgbp_engine* e, const DevState& S, const Traversal& tr, const DevTraversal& d, int L0, int L1,
                    unsigned long long seq_base, unsigned long long stop_below, bool kl, int* launches) {
  const bool uni = e->plan.max_dim <= 2 && e->plan.n_sites >= 8;  // many tiny problems: lanes = sites (bp_level_uni / uni1)
  // (the loop mode of the thread-per-site kernels exists for sepsets of at most one variable: bp_chunk_uni1)
  const bool chunks_on = !kl && e->plan.tune.tail && (uni ? e->max_s <= 1 : !e->layout_sm);
  size_t next_chunk = 0;
  for (int L = L0; L < L1; ++L) {
    if (chunks_on) {
      // a chunk of fused levels starting here (and ending inside the range): one launch, one workgroup per tree of tasks
      while (next_chunk < tr.chunks.size() && tr.chunks[next_chunk].level0 < L) ++next_chunk;
      if (next_chunk < tr.chunks.size() && tr.chunks[next_chunk].level0 == L && tr.chunks[next_chunk].level1 <= L1) {
        const Traversal::Chunk& ch = tr.chunks[next_chunk];
        if (uni && d.d_urecs)   // (the planner builds chunks of tasks for such plans: pgbp_plan.cpp, plan_uni)
          launch_chunk_uni1(S, d.d_task_off, d.d_urecs, d.d_cgroups_task + ch.group0 * kTailWaves, d.d_chunk_wg_off + ch.wg0,
                            ch.n_wg, e->plan.n_sites, seq_base, stop_below, e->st);
        else if (ch.generic)
          launch_chunk_generic(S, d.d_grecs, d.d_cgroups + ch.group0 * kTailWaves, d.d_chunk_wg_off + ch.wg0, ch.n_wg,
                               e->plan.n_sites, seq_base, stop_below, ch.max_mf, ch.small_only != 0, e->plan.tune.pair2, e->st);
        else
          launch_loop_or_tail(e, S, d.d_centries + ch.group0 * kTailWaves, d.d_cpros ? d.d_cpros + ch.group0 * kTailWaves : nullptr,
                              ch.n_groups, INT32_MAX, seq_base, stop_below, stop_below, d.d_chunk_wg_off + ch.wg0, ch.n_wg);
        if (launches) *launches += 1;
        L = ch.level1 - 1;
        continue;
      }
    }
    const int t0 = tr.level_off[L], nt = tr.level_off[L + 1] - t0;
    const int nf = tr.level_nfast[L], ng = tr.level_ngroups[L];
    launch_fast16(S, d.d_fentries + tr.level_fbase[L], d.d_fpros ? d.d_fpros + tr.level_fbase[L] : nullptr, kFastLevel, ng, ng,
                  e->plan.n_sites, seq_base, stop_below, stop_below, e->st);
    const int nbig = tr.level_nbig[L];
    if (uni)
      launch_level_uni(S, d.d_task_off, d.d_entries, d.d_urecs, t0 + nf, nt - nf, e->plan.n_sites, seq_base, stop_below, e->max_s, e->st);
    else {
      const bool rows = d.d_rowmap && !tr.level_nrows.empty() && tr.level_nrows[L] > 0;
      launch_level_generic(S, d.d_grecs, tr.level_gbase[L], nt - nf - nbig, e->plan.n_sites, seq_base, stop_below,
                           tr.max_mf, nbig == 0 && tr.level_small[L] != 0, e->st,
                           rows ? d.d_rowmap + 2 * tr.level_rowbase[L] : nullptr, rows ? tr.level_nrows[L] : 0,
                           e->plan.tune.small4_min);
      if (nbig > 0) {
        if (ensure_ws(e, (int64_t)nbig * e->plan.n_sites * big_ws_doubles(tr.max_mf_big)) == PGBP_OK)
          launch_level_big(S, d.d_task_off, d.d_entries, t0 + nt - nbig, nbig, e->plan.n_sites, seq_base, stop_below,
                           tr.max_mf_big, e->d_ws, e->st);
        else
          e->enqueue_rc = PGBP_ERR_HIP;   // (these messages never ran: the entry point reports it)
      }
    }
    if (launches) *launches += (nf > 0) + (nt - nf - nbig > 0) + (nbig > 0);
    if (kl) {  // residual_kldiv! right after the messages of the level (src/calibration.jl:128,154)
      const int e0 = tr.task_off[t0], e1 = tr.task_off[t0 + nt];
      int level_s = 0;   // the largest sepset of THIS level's messages sizes the launch's LDS, not the graph's largest
      for (int q = e0; q < e1; ++q) level_s = std::max(level_s, (int)e->plan.msgs[tr.entries[q].msg].s);
      // the level's entries whose sepset is beyond the LDS instance: a second launch with a workspace slab each
      const auto b0 = std::lower_bound(d.kl_big.begin(), d.kl_big.end(), e0), b1 = std::lower_bound(d.kl_big.begin(), d.kl_big.end(), e1);
      const int n_big = (int)(b1 - b0);
      e->kl_flags_clean = e->kl_div_clean = false;
      if (n_big == 0 || ensure_ws(e, (int64_t)n_big * e->plan.n_sites * kldiv_ws_doubles(level_s)) == PGBP_OK)
        launch_residual_kldiv(S, d.d_entries, e0, e1 - e0, level_s, e->d_kldiv, e->d_klflags, e->plan.n_sites, stop_below, e->st,
                              d.d_kl_big + (b0 - d.kl_big.begin()), n_big, e->d_ws);
      else
        e->enqueue_rc = PGBP_ERR_HIP;
    }
  }
}

// Traversals of one schedule tree.  dirs: 1 = postorder only, 2 = preorder only, 3 = postorder then preorder (one
// iteration of calibrate! on this tree: src/calibration.jl:72-84).  The narrow levels at the root end of the tree go out
// as ONE single-workgroup launch (Traversal::tail_levels); in a postorder + preorder pair the postorder's last levels and
// the preorder's first ones share it.
// Once the postorder of a tree failed, its preorder does not run here.  (The reference still walks it, on beliefs the
// sequential postorder left half-updated, may log a second failure, and returns (false, false): src/calibration.jl:80-82.
// The first failure of the reference's order is what the engine reports; the state after a failure is unspecified.)
void enqueue_tree(pgbp_engine* e, const DevState& S, int tree, int dirs, unsigned long long pair_index, bool kl = false,
                  int* n_launches = nullptr) {
  const Tree& T = e->plan.trees[tree];
  const unsigned long long seq_base = pair_index * seq_stride(e);
  const unsigned long long stop_post = seq_base, stop_pre = seq_base + (unsigned long long)T.pa.size();
  const int np = (dirs & 1) ? tail_levels(e, T.post, kl) : 0, nq = (dirs & 2) ? tail_levels(e, T.pre, kl) : 0;
  const int nlev_post = (int)T.post.level_off.size() - 1, nlev_pre = (int)T.pre.level_off.size() - 1;
  if (dirs & 1) enqueue_levels(e, S, T.post, e->dpost[tree], 0, nlev_post - np, seq_base, stop_post, kl, n_launches);
  if (np + nq > 0) {
    // d_tail = the postorder's tail groups followed by the preorder's
    const size_t first = (size_t)(np > 0 ? 0 : T.post.tail_levels) * kTailWaves;
    launch_loop_or_tail(e, S, e->d_tail[tree] + first, e->d_tail_pros[tree] ? e->d_tail_pros[tree] + first : nullptr, np + nq,
                        np, seq_base, stop_post, stop_pre, nullptr, 0);
    if (n_launches) *n_launches += 1;
  }
  if (dirs & 2) enqueue_levels(e, S, T.pre, e->dpre[tree], nq, nlev_pre, seq_base, stop_pre, kl, n_launches);
}

// One iteration of calibrate! on one schedule tree (postorder + preorder) followed by iscalibrated_residnorm(beliefs) into
// `d_out` [n_sites].  Thread-per-site engines whose tree holds every sepset: the message kernels mark the sites with a false
// flag themselves (DevState::notcal) and the all-flags reduction is a kernel of n_sites threads instead of one over
// n_sites x n_messages words.
void enqueue_pair_and_iscal(pgbp_engine* e, const DevState& S, int tree, unsigned long long pair, bool kl, int32_t* d_out,
                            int* n_launches = nullptr) {
  const Plan& p = e->plan;
  const bool fused = e->layout_sm && !kl && S.update_resnorm != 0 && (int)p.trees[tree].pa.size() == p.n_sepsets;
  if (fused) {
    (void)hipMemsetAsync(e->d_notcal, 0, sizeof(int32_t) * (size_t)p.n_sites, e->st);
    DevState S1 = S;
    S1.notcal = e->d_notcal;
    enqueue_tree(e, S1, tree, 3, pair, kl, n_launches);
    launch_iscal_from_notcal(e->d_notcal, d_out, p.n_sites, e->st);
  } else {
    enqueue_tree(e, S, tree, 3, pair, kl, n_launches);
    launch_reduce_flags(e->d_flags, p.n_msgs(), p.n_sites, d_out, e->st, e->layout_sm ? 1 : 0);
  }
}

// integratebelief! of one belief in whatever layout the state is in
void integrate_async(pgbp_engine* e, int belief, double* d_mu) {
  const Plan& p = e->plan;
  if (e->layout_sm)
    launch_integrate_sm(e->d_pool_sm, p.packed_off[belief], p.dims[belief], d_mu, std::max(1, p.max_dim), e->d_norm,
                        e->d_info, p.n_sites, e->st);
  else if (ensure_ws(e, (int64_t)p.n_sites * big_ws_doubles(p.dims[belief])) == PGBP_OK)
    launch_integrate(e->d_pool, p.pool_stride(), p.boff[belief], p.dims[belief], e->layout_bs16 ? 1 : 0, p.fast_p, d_mu,
                     std::max(1, p.max_dim), e->d_norm, e->d_info, p.n_sites, e->d_ws, e->st);
  else
    e->enqueue_rc = PGBP_ERR_HIP;   // (e->err is set; the entry point returns the code)
}

// skip_sepsets: the caller's next traversal overwrites every sepset without reading it (DevState::sep_zero)
int reset_from_factors_async(pgbp_engine* e, bool skip_sepsets = false) {
  if (!e->have_factors) return e->fail(PGBP_ERR_STATE, "no factors: call pgbp_set_beliefs(snapshot) or pgbp_init_factors_frombeliefs first");
  const Plan& p = e->plan;
  if (e->layout_sm) {  // cluster elements come first and are contiguous over sites
    const int64_t nc = p.packed_off[p.n_clusters] * sm_row(p.n_sites), nall = p.packed_off.back() * sm_row(p.n_sites);
    HIPCHK(e, hipMemcpyAsync(e->d_pool_sm, e->d_fpool_sm, sizeof(double) * (size_t)nc, hipMemcpyDeviceToDevice, e->st));
    if (!skip_sepsets) HIPCHK(e, hipMemsetAsync(e->d_pool_sm + nc, 0, sizeof(double) * (size_t)(nall - nc), e->st));
    return PGBP_OK;
  }
  if (e->layout_bs16)  // packed records use about half of their slots: copy what is in use
    launch_copy_records(e->d_fpool, p.cluster_stride(), e->d_pool, p.pool_stride(), e->d_boff, e->d_bdim, p.n_clusters, 1,
                        p.fast_p, p.n_sites, e->st);
  else
    launch_copy_strided(e->d_fpool, p.cluster_stride(), e->d_pool, p.pool_stride(), p.cluster_stride(), p.n_sites, e->st);
  if (!skip_sepsets)
    launch_zero_strided(e->d_pool + p.cluster_stride(), p.pool_stride(), p.pool_stride() - p.cluster_stride(),
                        p.n_sites, e->st);
  return PGBP_OK;
}

void decode_fail(const pgbp_engine* e, unsigned long long key, pgbp_result& r) {
  const unsigned long long q = key >> kInfoBits;
  r.fail_info = (int32_t)(key & ((1ull << kInfoBits) - 1));
  const unsigned long long stride = seq_stride(e);
  const unsigned long long pair = q / stride, seq = q % stride;
  const int nt = (int)e->plan.trees.size();
  r.fail_iter = (int32_t)(pair / nt) + 1;
  r.fail_tree = (int32_t)(pair % nt) + 1;
  const int n = (int)e->plan.trees[pair % nt].pa.size();
  if ((int)seq < n) {
    r.fail_dir = 0;
    r.fail_edge = n - 1 - (int)seq;
  } else {
    r.fail_dir = 1;
    r.fail_edge = (int)seq - n;
  }
}

}  // namespace

extern "C" {

const char* pgbp_last_error(const pgbp_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

void pgbp_destroy(pgbp_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->plan.device);
  free_traversals(e);
  for (auto& pr : e->kernel_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  for (void* p : {(void*)e->d_pool, (void*)e->d_fpool, (void*)e->d_rpool, (void*)e->d_msgs, (void*)e->d_idx,
                  (void*)e->d_flags, (void*)e->d_status, (void*)e->d_kldiv, (void*)e->d_klflags, (void*)e->d_nb_off, (void*)e->d_nb_msg, (void*)e->d_sepcl, (void*)e->d_eps, (void*)e->d_thr, (void*)e->d_logtab, (void*)e->d_pool_sm, (void*)e->d_fpool_sm, (void*)e->d_rpool_sm, (void*)e->d_flags_alt, (void*)e->d_status_alt,
                  (void*)e->d_klflags_alt, (void*)e->d_poison_alt, (void*)e->d_kldiv_alt, (void*)e->d_fail, (void*)e->d_poison,
                  (void*)e->d_iscal, (void*)e->d_notcal,
                  (void*)e->d_iscal_hist, (void*)e->d_boff, (void*)e->d_packed_off, (void*)e->d_roff,
                  (void*)e->d_rpacked_off, (void*)e->d_mu, (void*)e->d_norm, (void*)e->d_info,
                  (void*)e->d_one_task_off, (void*)e->d_one_entry, (void*)e->d_one_rec, (void*)e->d_bdim, (void*)e->d_rdim,
                  (void*)e->d_symflag, (void*)e->d_bm_kind, (void*)e->d_bm_row, (void*)e->d_bm_length, (void*)e->d_bm_ithl,
                  (void*)e->d_bm_data, (void*)e->d_bm_Rinv, (void*)e->d_bm_logdet, (void*)e->d_bm_mu})
    if (p) (void)hipFree(p);
  for (void* p : e->lg_bufs)
    if (p) (void)hipFree(p);
  for (void* p : {(void*)e->d_lg_R, (void*)e->d_lg_alpha, (void*)e->d_lg_theta, (void*)e->d_lg_mu})
    if (p) (void)hipFree(p);
  if (e->d_gather) (void)hipFree(e->d_gather);
  if (e->d_ws) (void)hipFree(e->d_ws);
  if (e->d_xbuf) (void)hipFree(e->d_xbuf);
  if (e->d_xoff) (void)hipFree(e->d_xoff);
  if (e->d_bm_data_sm) (void)hipFree(e->d_bm_data_sm);
  if (e->st) (void)hipStreamDestroy(e->st);
  delete e;
}

int pgbp_create(const pgbp_desc* desc, pgbp_engine** out) {
  if (!out) return PGBP_ERR_INVALID;
  *out = nullptr;
  std::unique_ptr<pgbp_engine> up(new pgbp_engine());
  pgbp_engine* e = up.get();
  int rc = plan_build(e->plan, desc);
  if (rc) {
    g_create_error = e->plan.err;
    return rc;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || desc->device < 0 || desc->device >= ndev) {
    g_create_error = "no usable HIP device (device " + std::to_string(desc->device) + " of " + std::to_string(ndev) +
                     "): the engine has no CPU path";
    return PGBP_ERR_NO_DEVICE;
  }
  auto bail = [&](int code) {
    g_create_error = e->err;
    pgbp_destroy(up.release());
    return code;
  };
  if (hipSetDevice(desc->device) != hipSuccess) {
    e->err = "hipSetDevice failed";
    return bail(PGBP_ERR_HIP);
  }
  if (hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking) != hipSuccess) {
    e->err = "hipStreamCreate failed";
    return bail(PGBP_ERR_HIP);
  }
  const Plan& p = e->plan;
  const size_t ns = (size_t)p.n_sites;
  const size_t nm = (size_t)p.n_msgs();
  const size_t nsw = (size_t)sm_row(p.n_sites);   // the per-message / per-cluster word arrays serve both layouts (a switch swaps
                                                   // them with their twins): sized for the padded rows of the site-minor one
  if ((rc = dev_alloc(e, &e->d_pool, ns * p.pool_stride()))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_fpool, ns * p.cluster_stride()))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_rpool, ns * p.rpool_stride()))) return bail(rc);
  if ((rc = upload(e, &e->d_msgs, p.msgs))) return bail(rc);
  if ((rc = upload(e, &e->d_idx, p.idxpool))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_flags, nsw * nm))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_status, nsw * nm))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_kldiv, nsw * nm))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_klflags, nsw * nm))) return bail(rc);
  {
    std::vector<int32_t> nb_off(p.n_clusters + 1, 0), nb_msg(nm);
    for (int k = 0; k < p.n_sepsets; ++k) {
      ++nb_off[p.sepset_clusters[2 * k] + 1];
      ++nb_off[p.sepset_clusters[2 * k + 1] + 1];
    }
    for (int c = 0; c < p.n_clusters; ++c) nb_off[c + 1] += nb_off[c];
    std::vector<int32_t> fill(nb_off.begin(), nb_off.end() - 1);
    for (int k = 0; k < p.n_sepsets; ++k) {  // message 2k is received by cluster a (sent by b), 2k+1 the reverse
      nb_msg[fill[p.sepset_clusters[2 * k]]++] = 2 * k + 1;
      nb_msg[fill[p.sepset_clusters[2 * k + 1]]++] = 2 * k;
    }
    if ((rc = upload(e, &e->d_nb_off, nb_off))) return bail(rc);
    if ((rc = upload(e, &e->d_nb_msg, nb_msg))) return bail(rc);
    if ((rc = upload(e, &e->d_sepcl, p.sepset_clusters))) return bail(rc);
    if ((rc = dev_alloc(e, &e->d_eps, ns * (size_t)std::max(1, p.n_clusters)))) return bail(rc);
    if ((rc = dev_alloc(e, &e->d_thr, 2 * (size_t)(PGBP_MAX_DIM + 1)))) return bail(rc);
    if ((rc = dev_alloc(e, &e->d_logtab, 256))) return bail(rc);
    {
      // DevState::logtab, in the host's extended precision
      double tab[256];
      for (int i = 0; i < 128; ++i) {
        const long double c = 0.5L + ((long double)i + 0.5L) / 256.0L;
        const double inv = (double)(1.0L / c);
        tab[2 * i] = inv;
        tab[2 * i + 1] = (double)(-logl((long double)inv));
      }
      if (hipMemcpy(e->d_logtab, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) return bail(PGBP_ERR_HIP);
    }
  }
  if ((rc = dev_alloc(e, &e->d_fail, ns))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_poison, nsw * (size_t)p.n_clusters))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_iscal, ns))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_notcal, ns))) return bail(rc);
  if ((rc = upload(e, &e->d_boff, p.boff))) return bail(rc);
  if ((rc = upload(e, &e->d_packed_off, p.packed_off))) return bail(rc);
  if ((rc = upload(e, &e->d_roff, p.roff))) return bail(rc);
  if ((rc = upload(e, &e->d_rpacked_off, p.rpacked_off))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_mu, ns * (size_t)std::max(1, p.max_dim)))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_norm, ns))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_info, ns))) return bail(rc);
  {
    std::vector<int32_t> rdim(nm);
    for (size_t d = 0; d < nm; ++d) rdim[d] = p.dims[p.n_clusters + d / 2];
    for (int32_t v : rdim) e->max_s = std::max(e->max_s, v);
    if ((rc = upload(e, &e->d_bdim, p.dims))) return bail(rc);
    if ((rc = upload(e, &e->d_rdim, rdim))) return bail(rc);
    if ((rc = dev_alloc(e, &e->d_symflag, 1))) return bail(rc);
  }
  if ((rc = dev_alloc(e, &e->d_one_task_off, 2))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_one_entry, 1))) return bail(rc);
  if ((rc = dev_alloc(e, &e->d_one_rec, 1))) return bail(rc);
  // beliefs = constant function 1 (h, J, g all 0: src/beliefs.jl:108-132); residuals 0;
  // flags false / kldiv -1, empty messages born calibrated (src/beliefs.jl:914-924)
  if (hipMemsetAsync(e->d_pool, 0, ns * p.pool_stride() * sizeof(double), e->st) != hipSuccess ||
      hipMemsetAsync(e->d_fpool, 0, ns * p.cluster_stride() * sizeof(double), e->st) != hipSuccess ||
      hipMemsetAsync(e->d_rpool, 0, ns * p.rpool_stride() * sizeof(double), e->st) != hipSuccess ||
      hipMemsetAsync(e->d_status, 0, nsw * nm * sizeof(int32_t), e->st) != hipSuccess ||
      hipMemsetAsync(e->d_flags, 0, nsw * nm * sizeof(int32_t), e->st) != hipSuccess ||
      hipMemsetAsync(e->d_klflags, 0, nsw * nm * sizeof(int32_t), e->st) != hipSuccess ||
      hipMemsetAsync(e->d_kldiv, 0, nsw * nm * sizeof(double), e->st) != hipSuccess ||
      hipMemsetAsync(e->d_poison, 0, nsw * (size_t)p.n_clusters * sizeof(int32_t), e->st) != hipSuccess ||
      hipMemsetAsync(e->d_fail, 0xFF, ns * sizeof(unsigned long long), e->st) != hipSuccess) {
    e->err = "hipMemsetAsync failed";
    return bail(PGBP_ERR_HIP);
  }
  reset_message_flags(e, 1);
  if (hipStreamSynchronize(e->st) != hipSuccess) {
    e->err = "initialisation kernels failed";
    return bail(PGBP_ERR_HIP);
  }
  *out = up.release();
  return PGBP_OK;
}

int64_t pgbp_packed_size(const pgbp_engine* e) { return e ? e->plan.packed_off.back() : -1; }
int64_t pgbp_residual_size(const pgbp_engine* e) { return e ? e->plan.rpacked_off.back() : -1; }
int32_t pgbp_n_messages(const pgbp_engine* e) { return e ? e->plan.n_msgs() : -1; }
int32_t pgbp_belief_dim(const pgbp_engine* e, int32_t belief) {
  return (e && belief >= 0 && belief < e->plan.n_beliefs()) ? e->plan.dims[belief] : -1;
}

double pgbp_residual_threshold(double divisor, double atol) { return quotient_threshold(divisor, atol); }

int pgbp_sync(pgbp_engine* e) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  HIPCHK(e, hipStreamSynchronize(e->st));
  HIPCHK(e, hipGetLastError());  // a launch that failed since the last check (bad grid, LDS size) surfaces here
  return e->take_enqueue_rc();   // ... and so does an enqueue step that was skipped for want of a workspace
}

int pgbp_init_factors_frombeliefs(pgbp_engine* e) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  const Plan& p = e->plan;
  if (e->layout_sm)
    HIPCHK(e, hipMemcpyAsync(e->d_fpool_sm, e->d_pool_sm, sizeof(double) * (size_t)p.packed_off[p.n_clusters] * (size_t)sm_row(p.n_sites),
                             hipMemcpyDeviceToDevice, e->st));
  else
    launch_copy_strided(e->d_pool, p.pool_stride(), e->d_fpool, p.cluster_stride(), p.cluster_stride(), p.n_sites,
                        e->st);
  e->have_factors = true;
  return pgbp_sync(e);
}

int pgbp_set_beliefs(pgbp_engine* e, const double* packed, int32_t snapshot_factors) {
  DeviceScope device_scope(e);
  if (!e || !packed) return PGBP_ERR_INVALID;
  const Plan& p = e->plan;
  {
    int rc0 = ensure_layout(e, false);  // factors / residuals that stay must be in the layout the upload uses
    if (rc0) return rc0;
    e->sym_known = false;
  }
  const int64_t psz = p.packed_off.back();
  double* stage = nullptr;
  HIPCHK(e, hipMalloc((void**)&stage, sizeof(double) * (size_t)psz * p.n_sites));
  hipError_t rc = hipMemcpyAsync(stage, packed, sizeof(double) * (size_t)psz * p.n_sites, hipMemcpyHostToDevice, e->st);
  if (rc == hipSuccess) {
    launch_records(stage, psz, e->d_packed_off, e->d_pool, p.pool_stride(), e->d_boff, e->d_packed_off,
                   p.n_beliefs(), p.n_sites, e->st);
    rc = hipStreamSynchronize(e->st);
  }
  (void)hipFree(stage);
  if (rc != hipSuccess) return e->fail(PGBP_ERR_HIP, std::string("pgbp_set_beliefs: ") + hipGetErrorString(rc));
  if (snapshot_factors) return pgbp_init_factors_frombeliefs(e);
  return PGBP_OK;
}

int pgbp_get_beliefs(pgbp_engine* e, double* packed) {
  DeviceScope device_scope(e);
  if (!e || !packed) return PGBP_ERR_INVALID;
  const Plan& p = e->plan;
  {
    int rc0 = ensure_layout(e, false);
    if (rc0) return rc0;
  }
  const int64_t psz = p.packed_off.back();
  double* stage = nullptr;
  HIPCHK(e, hipMalloc((void**)&stage, sizeof(double) * (size_t)psz * p.n_sites));
  launch_records(e->d_pool, p.pool_stride(), e->d_boff, stage, psz, e->d_packed_off, e->d_packed_off, p.n_beliefs(),
                 p.n_sites, e->st);
  hipError_t rc = hipMemcpyAsync(packed, stage, sizeof(double) * (size_t)psz * p.n_sites, hipMemcpyDeviceToHost, e->st);
  if (rc == hipSuccess) rc = hipStreamSynchronize(e->st);
  (void)hipFree(stage);
  if (rc != hipSuccess) return e->fail(PGBP_ERR_HIP, std::string("pgbp_get_beliefs: ") + hipGetErrorString(rc));
  return PGBP_OK;
}

int pgbp_get_site_beliefs(pgbp_engine* e, int32_t site, double* packed) {
  DeviceScope device_scope(e);
  if (!e || !packed) return PGBP_ERR_INVALID;
  const Plan& p = e->plan;
  if (site < 0 || site >= p.n_sites) return e->fail(PGBP_ERR_INVALID, "site index out of range");
  {
    int rc0 = ensure_layout(e, false);
    if (rc0) return rc0;
  }
  const int64_t psz = p.packed_off.back();
  double* stage = nullptr;
  HIPCHK(e, hipMalloc((void**)&stage, sizeof(double) * (size_t)std::max<int64_t>(1, psz)));
  launch_records(e->d_pool + (int64_t)site * p.pool_stride(), p.pool_stride(), e->d_boff, stage, psz, e->d_packed_off,
                 e->d_packed_off, p.n_beliefs(), 1, e->st);
  hipError_t rc = hipMemcpyAsync(packed, stage, sizeof(double) * (size_t)psz, hipMemcpyDeviceToHost, e->st);
  if (rc == hipSuccess) rc = hipStreamSynchronize(e->st);
  (void)hipFree(stage);
  if (rc != hipSuccess) return e->fail(PGBP_ERR_HIP, std::string("pgbp_get_site_beliefs: ") + hipGetErrorString(rc));
  return PGBP_OK;
}

static int belief_rec(pgbp_engine* e, int32_t site, int32_t b, double** dptr, int64_t* len) {
  const Plan& p = e->plan;
  if (site < 0 || site >= p.n_sites || b < 0 || b >= p.n_beliefs())
    return e->fail(PGBP_ERR_INVALID, "site or belief index out of range");
  *dptr = e->d_pool + (int64_t)site * p.pool_stride() + p.boff[b];
  *len = p.packed_off[b + 1] - p.packed_off[b];
  return PGBP_OK;
}

int pgbp_set_belief(pgbp_engine* e, int32_t site, int32_t belief, const double* rec) {
  DeviceScope device_scope(e);
  if (!e || !rec) return PGBP_ERR_INVALID;
  double* d;
  int64_t len;
  int rc = belief_rec(e, site, belief, &d, &len);
  if (rc) return rc;
  if ((rc = ensure_layout(e, false))) return rc;
  e->sym_known = false;
  HIPCHK(e, hipMemcpyAsync(d, rec, sizeof(double) * len, hipMemcpyHostToDevice, e->st));
  return pgbp_sync(e);
}

int pgbp_get_belief(pgbp_engine* e, int32_t site, int32_t belief, double* rec) {
  DeviceScope device_scope(e);
  if (!e || !rec) return PGBP_ERR_INVALID;
  double* d;
  int64_t len;
  int rc = belief_rec(e, site, belief, &d, &len);
  if (rc) return rc;
  if ((rc = ensure_layout(e, false))) return rc;
  HIPCHK(e, hipMemcpyAsync(rec, d, sizeof(double) * len, hipMemcpyDeviceToHost, e->st));
  return pgbp_sync(e);
}

// The exchange buffer of a cut cluster graph: records gathered on the device into one contiguous buffer (what a
// collective would carry), one copy across the bus.
static int xbuf_prepare(pgbp_engine* e, int32_t site, int32_t n, const int32_t* beliefs, int64_t* total) {
  const Plan& p = e->plan;
  if (site < 0 || site >= p.n_sites || n < 0 || (n > 0 && !beliefs)) return e->fail(PGBP_ERR_INVALID, "site or belief list");
  std::vector<int64_t> off((size_t)2 * n + 1);
  int64_t at = 0;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t b = beliefs[i];
    if (b < 0 || b >= p.n_beliefs()) return e->fail(PGBP_ERR_INVALID, "belief index out of range");
    off[i] = p.boff[b];
    off[(size_t)n + i] = at;
    at += p.packed_off[b + 1] - p.packed_off[b];
  }
  off[(size_t)2 * n] = at;
  *total = at;
  int rc = ensure_layout(e, false);
  if (rc) return rc;
  if ((int64_t)off.size() > e->xoff_cap || at > e->xbuf_cap) HIPCHK(e, hipStreamSynchronize(e->st));
  if ((int64_t)off.size() > e->xoff_cap) {
    if (e->d_xoff) (void)hipFree(e->d_xoff);
    e->d_xoff = nullptr;
    e->xoff_cap = 0;
    HIPCHK(e, hipMalloc(reinterpret_cast<void**>(&e->d_xoff), sizeof(int64_t) * off.size()));
    e->xoff_cap = (int64_t)off.size();
  }
  if (at > e->xbuf_cap) {
    if (e->d_xbuf) (void)hipFree(e->d_xbuf);
    e->d_xbuf = nullptr;
    e->xbuf_cap = 0;
    HIPCHK(e, hipMalloc(reinterpret_cast<void**>(&e->d_xbuf), sizeof(double) * (size_t)std::max<int64_t>(at, 1)));
    e->xbuf_cap = std::max<int64_t>(at, 1);
  }
  HIPCHK(e, hipMemcpyAsync(e->d_xoff, off.data(), sizeof(int64_t) * off.size(), hipMemcpyHostToDevice, e->st));
  HIPCHK(e, hipStreamSynchronize(e->st));   // (`off` is a local)
  return PGBP_OK;
}

int64_t pgbp_packed_beliefs_size(pgbp_engine* e, int32_t n, const int32_t* beliefs) {
  if (!e || n < 0 || (n > 0 && !beliefs)) return -1;
  const Plan& p = e->plan;
  int64_t at = 0;
  for (int32_t i = 0; i < n; ++i) {
    if (beliefs[i] < 0 || beliefs[i] >= p.n_beliefs()) return -1;
    at += p.packed_off[beliefs[i] + 1] - p.packed_off[beliefs[i]];
  }
  return at;
}

int pgbp_pack_beliefs(pgbp_engine* e, int32_t site, int32_t n, const int32_t* beliefs, double* buf) {
  DeviceScope device_scope(e);
  if (!e || (n > 0 && !buf)) return PGBP_ERR_INVALID;
  int64_t total = 0;
  int rc = xbuf_prepare(e, site, n, beliefs, &total);
  if (rc || n == 0) return rc;
  launch_pack_records(e->d_pool + (int64_t)site * e->plan.pool_stride(), e->d_xoff, e->d_xoff + n, n, e->d_xbuf, 1, e->st);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(buf, e->d_xbuf, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, e->st));
  return pgbp_sync(e);
}

int pgbp_unpack_beliefs(pgbp_engine* e, int32_t site, int32_t n, const int32_t* beliefs, const double* buf) {
  DeviceScope device_scope(e);
  if (!e || (n > 0 && !buf)) return PGBP_ERR_INVALID;
  int64_t total = 0;
  int rc = xbuf_prepare(e, site, n, beliefs, &total);
  if (rc || n == 0) return rc;
  e->sym_known = false;
  HIPCHK(e, hipMemcpyAsync(e->d_xbuf, buf, sizeof(double) * (size_t)total, hipMemcpyHostToDevice, e->st));
  launch_pack_records(e->d_pool + (int64_t)site * e->plan.pool_stride(), e->d_xoff, e->d_xoff + n, n, e->d_xbuf, 0, e->st);
  HIPCHK(e, hipGetLastError());
  return pgbp_sync(e);
}

int pgbp_reset_from_factors(pgbp_engine* e) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  int rc = reset_from_factors_async(e);
  if (rc) return rc;
  return pgbp_sync(e);
}

int pgbp_reset_flags(pgbp_engine* e, int32_t reset_kl) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  reset_message_flags(e, reset_kl);
  return pgbp_sync(e);
}

int pgbp_get_residuals(pgbp_engine* e, double* packed, int32_t* iscalibrated_resid, double* kldiv,
                       int32_t* iscalibrated_kl) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  const Plan& p = e->plan;
  if (packed || e->layout_sm) {  // the word arrays are transposed in the site-minor layout
    int rc0 = ensure_layout(e, false);
    if (rc0) return rc0;
  }
  const size_t ns = (size_t)p.n_sites, nm = (size_t)p.n_msgs();
  if (packed) {
    const int64_t rsz = p.rpacked_off.back();
    double* stage = nullptr;
    HIPCHK(e, hipMalloc((void**)&stage, sizeof(double) * (size_t)std::max<int64_t>(1, rsz) * ns));
    launch_records(e->d_rpool, p.rpool_stride(), e->d_roff, stage, rsz, e->d_rpacked_off, e->d_rpacked_off,
                   (int)nm, p.n_sites, e->st);
    hipError_t rc = hipMemcpyAsync(packed, stage, sizeof(double) * (size_t)rsz * ns, hipMemcpyDeviceToHost, e->st);
    if (rc == hipSuccess) rc = hipStreamSynchronize(e->st);
    (void)hipFree(stage);
    if (rc != hipSuccess) return e->fail(PGBP_ERR_HIP, std::string("pgbp_get_residuals: ") + hipGetErrorString(rc));
  }
  if (iscalibrated_resid)
    HIPCHK(e, hipMemcpyAsync(iscalibrated_resid, e->d_flags, sizeof(int32_t) * ns * nm, hipMemcpyDeviceToHost, e->st));
  if (kldiv) HIPCHK(e, hipMemcpyAsync(kldiv, e->d_kldiv, sizeof(double) * ns * nm, hipMemcpyDeviceToHost, e->st));
  if (iscalibrated_kl)
    HIPCHK(e, hipMemcpyAsync(iscalibrated_kl, e->d_klflags, sizeof(int32_t) * ns * nm, hipMemcpyDeviceToHost, e->st));
  return pgbp_sync(e);
}

int pgbp_get_residual(pgbp_engine* e, int32_t site, int32_t msg, double* rec, int32_t* iscalibrated_resid, double* kldiv,
                      int32_t* iscalibrated_kl) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  const Plan& p = e->plan;
  if (site < 0 || site >= p.n_sites || msg < 0 || msg >= p.n_msgs())
    return e->fail(PGBP_ERR_INVALID, "pgbp_get_residual: site or message index out of range");
  int rc = ensure_layout(e, false);   // the record in the ABI's order; the word arrays as [site][message]
  if (rc) return rc;
  const size_t w = (size_t)site * (size_t)p.n_msgs() + (size_t)msg;
  const int64_t len = p.rpacked_off[msg + 1] - p.rpacked_off[msg];
  if (rec && len > 0)
    HIPCHK(e, hipMemcpyAsync(rec, e->d_rpool + (int64_t)site * p.rpool_stride() + p.roff[msg], sizeof(double) * (size_t)len,
                             hipMemcpyDeviceToHost, e->st));
  if (iscalibrated_resid) HIPCHK(e, hipMemcpyAsync(iscalibrated_resid, e->d_flags + w, sizeof(int32_t), hipMemcpyDeviceToHost, e->st));
  if (kldiv) HIPCHK(e, hipMemcpyAsync(kldiv, e->d_kldiv + w, sizeof(double), hipMemcpyDeviceToHost, e->st));
  if (iscalibrated_kl) HIPCHK(e, hipMemcpyAsync(iscalibrated_kl, e->d_klflags + w, sizeof(int32_t), hipMemcpyDeviceToHost, e->st));
  return pgbp_sync(e);
}

int pgbp_set_schedule(pgbp_engine* e, int32_t n_trees, const int32_t* tree_off, const int32_t* pa_j,
                      const int32_t* ch_j) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  int rc = plan_set_schedule(e->plan, n_trees, tree_off, pa_j, ch_j);
  if (rc) return e->fail(rc, e->plan.err);  // the previous schedule (if any) stays in force
  HIPCHK(e, hipStreamSynchronize(e->st));
  free_traversals(e);
  e->dpost.resize(n_trees);
  e->dpre.resize(n_trees);
  for (int t = 0; t < n_trees && rc == PGBP_OK; ++t) {
    for (int dir = 0; dir < 2 && rc == PGBP_OK; ++dir) {
      const Traversal& tr = dir == 0 ? e->plan.trees[t].post : e->plan.trees[t].pre;
      DevTraversal& d = dir == 0 ? e->dpost[t] : e->dpre[t];
      if ((rc = upload(e, &d.d_task_off, tr.task_off))) break;
      if ((rc = upload(e, &d.d_entries, tr.entries))) break;
      if (e->plan.max_dim <= 2 && e->plan.n_sites >= 8 && e->max_s <= 1) {
        // thread-per-site kernel for sepsets of at most one variable: every entry as ONE record (URec)
        const Plan& pl = e->plan;
        std::vector<URec> ur(tr.entries.size());
        for (size_t q = 0; q < tr.entries.size(); ++q) {
          const Entry& en = tr.entries[q];
          const MsgDesc& m = pl.msgs[en.msg];
          URec& r = ur[q];
          r = URec{};
          r.msg = en.msg; r.seq = en.seq; r.reuse = en.reuse; r.from_b = m.from_b; r.to_b = m.to_b;
          r.mf = m.mf; r.s = m.s; r.mt = m.mt; r.ni = m.ni;
          r.k = m.s >= 1 ? pl.idxpool[m.keep_map] : 0;
          r.i0 = m.ni >= 1 ? pl.idxpool[m.int_map] : 0;
          r.i1 = m.ni >= 2 ? pl.idxpool[m.int_map + 1] : 0;
          r.u = m.s >= 1 ? pl.idxpool[m.up_map] : 0;
          r.from_off = m.from_off; r.sep_off = m.sep_off; r.to_off = m.to_off; r.res_off = m.res_off;
          r.from_p = pl.packed_off[m.from_b]; r.sep_p = pl.packed_off[m.sep_b]; r.to_p = pl.packed_off[m.to_b];
          r.res_p = pl.rpacked_off[en.msg];
        }
        if ((rc = upload(e, &d.d_urecs, ur))) break;
      }
      if ((rc = upload(e, &d.d_fentries, tr.fentries))) break;
      if (tr.has_pro) {
        if ((rc = upload(e, &d.d_fpros, tr.fpros))) break;
        if ((rc = upload(e, &d.d_cpros, tr.cpros))) break;
      }
      if ((rc = upload(e, &d.d_centries, tr.centries))) break;
      if ((rc = upload(e, &d.d_chunk_wg_off, tr.chunk_wg_off))) break;
      std::vector<int32_t> grp_recs(tr.cgroups);
      for (int32_t& t : grp_recs)
        if (t >= 0) t = tr.task_grec[t];
      if ((rc = upload(e, &d.d_cgroups, grp_recs))) break;
      if ((rc = upload(e, &d.d_cgroups_task, tr.cgroups))) break;
      if ((rc = upload(e, &d.d_grecs, tr.grecs))) break;
      if ((rc = upload(e, &d.d_rowmap, tr.rowmap))) break;
      d.kl_big.clear();
      for (size_t q = 0; q < tr.entries.size(); ++q)
        if (e->plan.msgs[tr.entries[q].msg].s > kKlLdsMaxS) d.kl_big.push_back((int32_t)q);
      if ((rc = upload(e, &d.d_kl_big, d.kl_big))) break;
    }
    if (rc == PGBP_OK) {
      const Tree& T = e->plan.trees[t];
      FEntry* dt = nullptr;
      if ((rc = upload(e, &dt, T.tail)) == PGBP_OK) e->d_tail.push_back(dt);
      FPro* dp = nullptr;   // (null unless a traversal of this tree has prologues)
      if (rc == PGBP_OK && (T.post.has_pro || T.pre.has_pro)) rc = upload(e, &dp, T.tail_pros);
      if (rc == PGBP_OK) e->d_tail_pros.push_back(dp);
    }
  }
  if (rc) {  // out of device memory half way: leave the engine without a schedule rather than with half of one
    free_traversals(e);
    e->plan.trees.clear();
    e->plan.all_fast = false;
  }
  return rc;
}

int pgbp_propagate(pgbp_engine* e, int32_t cluster_to, int32_t sepset, int32_t cluster_from, const pgbp_opts* opts,
                   int32_t* info) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  int rc = check_opts(e, opts);
  if (rc) return rc;
  const Plan& p = e->plan;
  const int k = sepset - p.n_clusters;
  if (k < 0 || k >= p.n_sepsets) return e->fail(PGBP_ERR_INVALID, "pgbp_propagate: not a sepset index");
  const int a = p.sepset_clusters[2 * k], b = p.sepset_clusters[2 * k + 1];
  int dir;
  if (cluster_to == a && cluster_from == b)
    dir = 0;
  else if (cluster_to == b && cluster_from == a)
    dir = 1;
  else
    return e->fail(PGBP_ERR_INVALID, "pgbp_propagate: the sepset does not connect these two clusters");
  if ((rc = ensure_layout(e, false))) return rc;  // single messages run on the generic kernel
  const int32_t toff[2] = {0, 1};
  Entry en{2 * k + dir, 0, 0, 0};
  HIPCHK(e, hipMemcpyAsync(e->d_one_task_off, toff, sizeof(toff), hipMemcpyHostToDevice, e->st));
  HIPCHK(e, hipMemcpyAsync(e->d_one_entry, &en, sizeof(en), hipMemcpyHostToDevice, e->st));
  if ((rc = reset_fail(e))) return rc;
  DevState S = dev_state(e, opts);
  if (big_msg(p.msgs[en.msg])) {
    if ((rc = ensure_ws(e, (int64_t)p.n_sites * big_ws_doubles(p.msgs[en.msg].mf)))) return rc;
    launch_level_big(S, e->d_one_task_off, e->d_one_entry, 0, 1, p.n_sites, 0, 0, p.msgs[en.msg].mf, e->d_ws, e->st);
  } else {
    const GRec rec = make_grec(p, en, -1);
    HIPCHK(e, hipMemcpyAsync(e->d_one_rec, &rec, sizeof(rec), hipMemcpyHostToDevice, e->st));
    launch_level_generic(S, e->d_one_rec, 0, 1, p.n_sites, 0, 0, p.msgs[en.msg].mf, false, e->st);
  }
  std::vector<unsigned long long> keys(p.n_sites);
  HIPCHK(e, hipMemcpyAsync(keys.data(), e->d_fail, sizeof(unsigned long long) * p.n_sites, hipMemcpyDeviceToHost, e->st));
  HIPCHK(e, hipStreamSynchronize(e->st));
  HIPCHK(e, hipGetLastError());
  if (info)
    for (int s = 0; s < p.n_sites; ++s)
      info[s] = !is_failure_key(keys[s]) ? 0 : (int32_t)(keys[s] & ((1ull << kInfoBits) - 1));
  return PGBP_OK;
}

int pgbp_residual_kldiv(pgbp_engine* e, int32_t cluster_to, int32_t sepset, int32_t cluster_from, const pgbp_opts* opts,
                        int32_t* iscalibrated_kl) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  const Plan& p = e->plan;
  int rc = check_opts(e, opts);
  if (rc) return rc;
  const int k = sepset - p.n_clusters;
  if (k < 0 || k >= p.n_sepsets) return e->fail(PGBP_ERR_INVALID, "pgbp_residual_kldiv: not a sepset index");
  const int a = p.sepset_clusters[2 * k], b = p.sepset_clusters[2 * k + 1];
  int dir;
  if (cluster_to == a && cluster_from == b)
    dir = 0;
  else if (cluster_to == b && cluster_from == a)
    dir = 1;
  else
    return e->fail(PGBP_ERR_INVALID, "pgbp_residual_kldiv: the sepset does not connect these two clusters");
  Entry en{2 * k + dir, 0, 0, 0};
  const int s_msg = p.msgs[en.msg].s;
  const int32_t toff[2] = {0, 1};   // (d_one_task_off[0] = 0 doubles as the list "entry 0" of the workspace instance)
  HIPCHK(e, hipMemcpyAsync(e->d_one_task_off, toff, sizeof(toff), hipMemcpyHostToDevice, e->st));
  HIPCHK(e, hipMemcpyAsync(e->d_one_entry, &en, sizeof(en), hipMemcpyHostToDevice, e->st));
  HIPCHK(e, hipStreamSynchronize(e->st));   // (toff, en: locals)
  const bool kl_ws = s_msg > kKlLdsMaxS;
  if (kl_ws && (rc = ensure_ws(e, (int64_t)p.n_sites * kldiv_ws_doubles(s_msg)))) return rc;
  if (e->layout_sm && (rc = ensure_site_minor(e, false))) return rc;
  DevState S = dev_state(e, opts);
  // a standalone call always computes (stop_below = 0; the status of the last attempt of this message still gates)
  e->kl_flags_clean = e->kl_div_clean = false;
  launch_residual_kldiv(S, e->d_one_entry, 0, 1, s_msg, e->d_kldiv, e->d_klflags, p.n_sites, 0, e->st,
                        kl_ws ? e->d_one_task_off : nullptr, kl_ws ? 1 : 0, e->d_ws);
  if (iscalibrated_kl) {
    std::vector<int32_t> all((size_t)p.n_sites * std::max(1, p.n_msgs()));
    HIPCHK(e, hipMemcpyAsync(all.data(), e->d_klflags, sizeof(int32_t) * (size_t)p.n_sites * p.n_msgs(),
                             hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    for (int s = 0; s < p.n_sites; ++s) iscalibrated_kl[s] = all[(size_t)s * p.n_msgs() + en.msg];
  }
  HIPCHK(e, hipStreamSynchronize(e->st));
  HIPCHK(e, hipGetLastError());
  return PGBP_OK;
}

int pgbp_regularize_bycluster(pgbp_engine* e) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  const Plan& p = e->plan;
  int rc = ensure_layout(e, false);
  if (rc) return rc;
  launch_regularize_bycluster(e->d_pool, p.pool_stride(), e->d_boff, e->d_bdim, e->d_nb_off, e->d_nb_msg, e->d_msgs,
                              e->d_idx, e->d_sepcl, e->d_eps, p.n_clusters, p.n_sepsets, p.n_sites, e->st);
  e->sym_known = false;
  HIPCHK(e, hipGetLastError());
  return PGBP_OK;
}

static int need_schedule(pgbp_engine* e, int tree) {
  if (e->plan.trees.empty()) return e->fail(PGBP_ERR_STATE, "no schedule: call pgbp_set_schedule first");
  if (tree < 0 || tree >= (int)e->plan.trees.size()) return e->fail(PGBP_ERR_INVALID, "schedule tree index out of range");
  return PGBP_OK;
}

static int collect_results(pgbp_engine* e, pgbp_result* results, const std::vector<int32_t>* hist, int n_pairs) {
  const Plan& p = e->plan;
  const int ns = p.n_sites;
  std::vector<unsigned long long> keys(ns);
  std::vector<int32_t> iscal(ns);
  HIPCHK(e, hipMemcpyAsync(keys.data(), e->d_fail, sizeof(unsigned long long) * ns, hipMemcpyDeviceToHost, e->st));
  HIPCHK(e, hipMemcpyAsync(iscal.data(), e->d_iscal, sizeof(int32_t) * ns, hipMemcpyDeviceToHost, e->st));
  HIPCHK(e, hipStreamSynchronize(e->st));
  HIPCHK(e, hipGetLastError());
  if (const int erc = e->take_enqueue_rc()) return erc;
  const int nt = std::max<int>(1, (int)p.trees.size());
  for (int s = 0; s < ns; ++s) {
    pgbp_result r;
    std::memset(&r, 0, sizeof(r));
    r.fail_edge = -1;
    if (is_failure_key(keys[s])) {
      r.succ = 0;
      r.iscal = 0;  // (false, false): src/calibration.jl:82
      decode_fail(e, keys[s], r);
    } else {
      r.succ = 1;
      r.iscal = iscal[s] != 0;
    }
    if (hist && r.succ) {
      for (int q = 0; q < n_pairs; ++q)
        if ((*hist)[(size_t)q * ns + s]) {
          r.iter_reached = q / nt + 1;
          r.tree_reached = q % nt + 1;
          break;
        }
    }
    results[s] = r;
  }
  return PGBP_OK;
}

int pgbp_traverse(pgbp_engine* e, int32_t tree, int32_t dir, const pgbp_opts* opts, pgbp_result* results) {
  DeviceScope device_scope(e);
  if (!e || !results || dir < 0 || dir > 1) return PGBP_ERR_INVALID;
  int rc = check_opts(e, opts);
  if (rc) return rc;
  if ((rc = need_schedule(e, tree))) return rc;
  const Plan& p = e->plan;
  if ((rc = reset_fail(e))) return rc;
  if ((rc = ensure_layout(e, want_bs16(e), want_site_minor(e) && !(opts && opts->update_residualkldiv)))) return rc;
  DevState S = dev_state(e, opts);
  enqueue_tree(e, S, tree, dir == 0 ? 1 : 2, (unsigned long long)tree, opts && opts->update_residualkldiv);
  launch_reduce_flags(e->d_flags, p.n_msgs(), p.n_sites, e->d_iscal, e->st, e->layout_sm ? 1 : 0);
  return collect_results(e, results, nullptr, 0);
}

int pgbp_calibrate(pgbp_engine* e, int32_t niter, const pgbp_opts* opts, pgbp_result* results) {
  DeviceScope device_scope(e);
  if (!e || !results || niter < 0) return PGBP_ERR_INVALID;
  int rc = check_opts(e, opts);
  if (rc) return rc;
  if ((rc = need_schedule(e, 0))) return rc;
  const Plan& p = e->plan;
  const int ns = p.n_sites, nt = (int)p.trees.size();
  if (niter == 0) {  // the loop of src/calibration.jl:46 does not run: (succ, iscal) = (false, false)
    for (int s = 0; s < ns; ++s) {
      std::memset(&results[s], 0, sizeof(pgbp_result));
      results[s].fail_edge = -1;
    }
    return PGBP_OK;
  }
  const bool auto_stop = opts && opts->auto_stop;
  const bool kl = opts && opts->update_residualkldiv;
  const int64_t n_pairs_max = (int64_t)niter * nt;
  if (n_pairs_max > e->hist_cap) {
    if (e->d_iscal_hist) (void)hipFree(e->d_iscal_hist);
    e->d_iscal_hist = nullptr;
    e->hist_cap = 0;
    if ((rc = dev_alloc(e, &e->d_iscal_hist, (size_t)n_pairs_max * ns))) return rc;
    e->hist_cap = n_pairs_max;
  }
  if ((rc = reset_fail(e))) return rc;
  HIPCHK(e, hipMemsetAsync(e->d_iscal, 0, sizeof(int32_t) * ns, e->st));
  if ((rc = ensure_layout(e, want_bs16(e), want_site_minor(e) && !(opts && opts->update_residualkldiv)))) return rc;
  DevState S = dev_state(e, opts);
  int pairs_done = 0;
  bool stop = false;
  std::vector<int32_t> now;
  std::vector<unsigned long long> keys(ns);
  // `auto`: kAhead schedule trees are enqueued per host round trip.  The device halts EACH SITE at the first tree at which
  // that site is calibrated (a key without failure information is min-ed into the site's fail word right behind that
  // tree's flag reduction, so every traversal enqueued after it returns at its first instruction for that site) and the
  // host reads back which tree that was: every site's beliefs are those of src/calibration.jl:53-56 run on that site
  // alone, at a fraction of the round trips.  The loop ends when every site has reached calibration or failed.
  constexpr int kAhead = 4;
  const unsigned long long stride = seq_stride(e);
  int batch_first = 0;
  std::vector<char> site_done(ns, 0);
  int n_done = 0, last_needed = 0;   // last_needed: one past the last tree at which some site was still running
  for (int i = 0; i < niter && !stop; ++i) {
    for (int j = 0; j < nt && !stop; ++j) {
      const unsigned long long pair = (unsigned long long)i * nt + j;
      enqueue_pair_and_iscal(e, S, j, pair, kl, e->d_iscal_hist + pair * ns);
      ++pairs_done;
      if (!auto_stop) continue;
      launch_halt_if_calibrated(e->d_iscal_hist + pair * ns, e->d_fail, ((pair + 1) * stride - 1) << kInfoBits, ns, e->st);
      const bool last = (i == niter - 1 && j == nt - 1);
      if (pairs_done - batch_first < kAhead && !last) continue;
      const int nb = pairs_done - batch_first;
      now.resize((size_t)nb * ns);
      HIPCHK(e, hipMemcpyAsync(now.data(), e->d_iscal_hist + (size_t)batch_first * ns, sizeof(int32_t) * now.size(), hipMemcpyDeviceToHost, e->st));
      HIPCHK(e, hipMemcpyAsync(keys.data(), e->d_fail, sizeof(unsigned long long) * ns, hipMemcpyDeviceToHost, e->st));
      HIPCHK(e, hipStreamSynchronize(e->st));
      for (int s = 0; s < ns; ++s) {
        if (site_done[s]) continue;
        int reached = -1;
        for (int q = 0; q < nb && reached < 0; ++q)
          if (now[(size_t)q * ns + s] != 0) reached = batch_first + q;
        if (reached >= 0 || is_failure_key(keys[s])) {
          site_done[s] = 1;
          ++n_done;
          // (a failed site: its traversals stopped on their own somewhere in this batch)
          last_needed = std::max(last_needed, reached >= 0 ? reached + 1 : pairs_done);
        }
      }
      if (n_done == ns) {
        pairs_done = last_needed;   // the trees enqueued behind it did nothing, for any site
        stop = true;
      }
      batch_first = pairs_done;
    }
  }
  std::vector<int32_t> hist((size_t)std::max(1, pairs_done) * ns, 0);
  if (pairs_done > 0) {
    HIPCHK(e, hipMemcpyAsync(hist.data(), e->d_iscal_hist, sizeof(int32_t) * (size_t)pairs_done * ns, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipMemcpyAsync(e->d_iscal, e->d_iscal_hist + (size_t)(pairs_done - 1) * ns, sizeof(int32_t) * ns, hipMemcpyDeviceToDevice, e->st));
  }
  return collect_results(e, results, &hist, pairs_done);
}

int pgbp_integrate(pgbp_engine* e, int32_t belief, double* mu, double* norm, int32_t* info) {
  DeviceScope device_scope(e);
  if (!e || !norm) return PGBP_ERR_INVALID;
  const Plan& p = e->plan;
  if (belief < 0 || belief >= p.n_beliefs()) return e->fail(PGBP_ERR_INVALID, "belief index out of range");
  const int m = p.dims[belief];
  const int ns = p.n_sites;
  integrate_async(e, belief, mu ? e->d_mu : nullptr);
  HIPCHK(e, hipMemcpyAsync(norm, e->d_norm, sizeof(double) * ns, hipMemcpyDeviceToHost, e->st));
  std::vector<double> mus;
  if (mu && m > 0) {
    mus.resize((size_t)ns * std::max(1, p.max_dim));
    HIPCHK(e, hipMemcpyAsync(mus.data(), e->d_mu, sizeof(double) * mus.size(), hipMemcpyDeviceToHost, e->st));
  }
  std::vector<int32_t> inf(ns);
  HIPCHK(e, hipMemcpyAsync(inf.data(), e->d_info, sizeof(int32_t) * ns, hipMemcpyDeviceToHost, e->st));
  HIPCHK(e, hipStreamSynchronize(e->st));
  HIPCHK(e, hipGetLastError());
  if (const int erc = e->take_enqueue_rc()) return erc;
  if (mu && m > 0)
    for (int s = 0; s < ns; ++s)
      std::memcpy(mu + (size_t)s * m, mus.data() + (size_t)s * std::max(1, p.max_dim), sizeof(double) * m);
  if (info) std::copy(inf.begin(), inf.end(), info);
  return PGBP_OK;
}

// ---- scores ---------------------------------------------------------------------------------------------

int pgbp_free_energy(pgbp_engine* e, double* out3, int32_t* info) {
  DeviceScope device_scope(e);
  if (!e || !out3) return PGBP_ERR_INVALID;
  if (!e->have_factors) return e->fail(PGBP_ERR_STATE, "pgbp_free_energy: no factors (pgbp_set_beliefs with snapshot, or pgbp_init_factors_frombeliefs)");
  if (e->layout_sm) {
    const int rc0 = ensure_site_minor(e, false);
    if (rc0) return rc0;
  }
  const Plan& p = e->plan;
  const int ns = p.n_sites;
  // beliefs whose [J | J_t | h] does not fit a CU's LDS whole (more than kFreeEnergyLdsMaxDim variables): a second
  // launch, their working matrix in the workspace
  std::vector<int32_t> big;
  for (int b = 0; b < p.n_beliefs(); ++b)
    if (p.dims[b] > kFreeEnergyLdsMaxDim) big.push_back(b);
  double *d_contrib = nullptr, *d_out = nullptr;
  int32_t *d_inf = nullptr, *d_big = nullptr;
  int rc;
  if (!big.empty()) {
    if ((rc = ensure_ws(e, (int64_t)big.size() * ns * free_energy_ws_doubles(p.max_dim)))) return rc;
    if ((rc = upload(e, &d_big, big))) return rc;
  }
  if ((rc = dev_alloc(e, &d_contrib, (size_t)2 * ns * p.n_beliefs()))) { if (d_big) (void)hipFree(d_big); return rc; }
  if ((rc = dev_alloc(e, &d_out, (size_t)3 * ns))) { (void)hipFree(d_contrib); if (d_big) (void)hipFree(d_big); return rc; }
  if ((rc = dev_alloc(e, &d_inf, (size_t)ns))) { (void)hipFree(d_contrib); (void)hipFree(d_out); if (d_big) (void)hipFree(d_big); return rc; }
  std::vector<int32_t> inf(ns, 0x7fffffff);
  hipError_t herr = hipMemcpyAsync(d_inf, inf.data(), sizeof(int32_t) * ns, hipMemcpyHostToDevice, e->st);
  if (herr == hipSuccess) {
    launch_free_energy(e->d_pool, p.pool_stride(), e->d_fpool, p.cluster_stride(), e->d_boff, e->d_bdim, p.n_clusters,
                       p.n_beliefs(), p.max_dim, e->layout_bs16 ? 1 : 0, p.fast_p, d_contrib, d_out, d_inf, ns, e->st, d_big,
                       (int)big.size(), e->d_ws);
    herr = hipMemcpyAsync(out3, d_out, sizeof(double) * 3 * ns, hipMemcpyDeviceToHost, e->st);
  }
  if (herr == hipSuccess) herr = hipMemcpyAsync(inf.data(), d_inf, sizeof(int32_t) * ns, hipMemcpyDeviceToHost, e->st);
  if (herr == hipSuccess) herr = hipStreamSynchronize(e->st);
  if (herr == hipSuccess) herr = hipGetLastError();
  (void)hipFree(d_contrib); (void)hipFree(d_out); (void)hipFree(d_inf);
  if (d_big) (void)hipFree(d_big);
  if (herr != hipSuccess) return e->fail(PGBP_ERR_HIP, std::string("pgbp_free_energy: ") + hipGetErrorString(herr));
  if (info)
    for (int s = 0; s < ns; ++s) info[s] = inf[s] == 0x7fffffff ? 0 : inf[s];
  return PGBP_OK;
}

// ---- device factor assignment (homogeneous BM on a tree) ----------------------------------------

int pgbp_bm_tree_setup(pgbp_engine* e, const pgbp_bm_tree* t) {
  DeviceScope device_scope(e);
  if (!e || !t || !t->kind || !t->length || !t->data_row || t->p <= 0 || t->p > PGBP_MAX_DIM || t->n_rows < 0)
    return PGBP_ERR_INVALID;
  const Plan& p = e->plan;
  const int nc = p.n_clusters;
  // the cluster dimensions must be what the factor kinds imply
  for (int c = 0; c < nc; ++c) {
    const int k = t->kind[c];
    const int want = k == 0 ? 2 * t->p : (k == 1 || k == 2) ? t->p : (k == 3 ? 0 : p.dims[c]);
    if (k < -1 || k > 3 || p.dims[c] != want || (k >= 0 && !(t->length[c] > 0.0)) ||
        (k >= 2 && (t->data_row[c] < 0 || t->data_row[c] >= t->n_rows)))
      return e->fail(PGBP_ERR_INVALID, "pgbp_bm_tree_setup: cluster " + std::to_string(c) + ": kind, dimension, branch length or data row inconsistent");
  }
  if (t->n_rows > 0 && !t->data) return e->fail(PGBP_ERR_INVALID, "pgbp_bm_tree_setup: data missing");
  for (void* q : {(void*)e->d_bm_kind, (void*)e->d_bm_row, (void*)e->d_bm_length, (void*)e->d_bm_data, (void*)e->d_bm_Rinv,
                  (void*)e->d_bm_logdet, (void*)e->d_bm_mu})
    if (q) (void)hipFree(q);
  e->d_bm_kind = e->d_bm_row = nullptr;
  e->d_bm_length = e->d_bm_data = e->d_bm_Rinv = e->d_bm_logdet = e->d_bm_mu = nullptr;
  int rc;
  if ((rc = upload(e, &e->d_bm_kind, std::vector<int32_t>(t->kind, t->kind + nc)))) return rc;
  if ((rc = upload(e, &e->d_bm_row, std::vector<int32_t>(t->data_row, t->data_row + nc)))) return rc;
  if ((rc = upload(e, &e->d_bm_length, std::vector<double>(t->length, t->length + nc)))) return rc;
  if (e->d_bm_ithl) (void)hipFree(e->d_bm_ithl);
  e->d_bm_ithl = nullptr;
  if ((rc = dev_alloc(e, &e->d_bm_ithl, (size_t)nc))) return rc;
  launch_bm_ithl(e->d_bm_length, e->d_bm_kind, t->p, e->d_bm_ithl, nc, e->st);
  const size_t nd = (size_t)p.n_sites * t->n_rows * t->p;
  if ((rc = dev_alloc(e, &e->d_bm_data, nd))) return rc;
  if (nd) HIPCHK(e, hipMemcpy(e->d_bm_data, t->data, nd * sizeof(double), hipMemcpyHostToDevice));
  if (e->d_bm_data_sm) (void)hipFree(e->d_bm_data_sm);
  e->d_bm_data_sm = nullptr;
  if (t->p == 1 && t->n_rows > 0) {   // univariate batches: [row][site] copy for the thread-per-site fill
    if ((rc = dev_alloc(e, &e->d_bm_data_sm, (size_t)t->n_rows * (size_t)sm_row(p.n_sites)))) return rc;
    launch_transpose_words_f64(e->d_bm_data, e->d_bm_data_sm, t->n_rows, p.n_sites, 1, e->st);
    HIPCHK(e, hipStreamSynchronize(e->st));
  }
  if ((rc = dev_alloc(e, &e->d_bm_Rinv, (size_t)p.n_sites * t->p * t->p))) return rc;
  if ((rc = dev_alloc(e, &e->d_bm_logdet, (size_t)p.n_sites))) return rc;
  if ((rc = dev_alloc(e, &e->d_bm_mu, (size_t)p.n_sites * t->p))) return rc;
  e->bm_p = t->p;
  e->bm_rows = t->n_rows;
  return PGBP_OK;
}

// also_factors: assignfactors! followed by init_factors_frombeliefs! (what the callers of the optimisers do once);
// without it only the beliefs are written, as in the body of score() (src/calibration.jl:205-209).
static int bm_fill_async(pgbp_engine* e, bool also_factors, bool skip_sepsets = false) {
  const Plan& p = e->plan;
  if (e->layout_sm && e->bm_p == 1) {  // univariate batch, site-minor state
    launch_bm_tree_fill_uni_sm(e->d_pool_sm, also_factors ? e->d_fpool_sm : nullptr, e->d_packed_off, e->d_bdim,
                               e->d_bm_kind, e->d_bm_length, e->d_bm_row, e->d_bm_data_sm, e->bm_rows, e->d_bm_Rinv,
                               e->d_bm_logdet, e->d_bm_mu, e->bm_per_site, p.n_clusters, p.n_sites, e->st);
    const int64_t nc = p.packed_off[p.n_clusters] * sm_row(p.n_sites), nall = p.packed_off.back() * sm_row(p.n_sites);
    if (!skip_sepsets) HIPCHK(e, hipMemsetAsync(e->d_pool_sm + nc, 0, sizeof(double) * (size_t)(nall - nc), e->st));  // sepsets = 1
    reset_message_flags(e, also_factors ? 1 : 0);
    if (also_factors) e->have_factors = true;
    return PGBP_OK;
  }
  if (e->layout_sm) {
    const int rc0 = ensure_site_minor(e, false);
    if (rc0) return rc0;
  }
  double* fp = also_factors ? e->d_fpool : nullptr;
  bool done = false;
  if (e->bm_p == p.fast_p)  // lane-blocked instance: every cluster has dimension 0, p or 2p (checked at setup)
    done = launch_bm_tree_fill_fast(e->d_pool, p.pool_stride(), fp, p.cluster_stride(), e->d_boff, e->d_bdim, e->d_bm_kind,
                                    e->d_bm_ithl, e->d_bm_row, e->d_bm_data, e->bm_rows, e->bm_p, e->d_bm_Rinv,
                                    e->d_bm_logdet, e->d_bm_mu, e->bm_per_site, e->layout_bs16 ? 1 : 0, p.n_clusters,
                                    p.n_sites, e->st);
  if (!done) {
    launch_bm_tree_fill(e->d_pool, p.pool_stride(), e->d_fpool, p.cluster_stride(), e->d_boff, e->d_bdim, e->d_bm_kind,
                        e->d_bm_length, e->d_bm_row, e->d_bm_data, e->bm_rows, e->bm_p, e->d_bm_Rinv, e->d_bm_logdet,
                        e->d_bm_mu, e->bm_per_site, e->layout_bs16 ? 1 : 0, p.fast_p, p.n_clusters, p.n_sites, e->st);
    also_factors = true;  // the general kernel always writes both
  }
  if (!skip_sepsets)
    launch_zero_strided(e->d_pool + p.cluster_stride(), p.pool_stride(), p.pool_stride() - p.cluster_stride(),
                        p.n_sites, e->st);  // sepsets = 1 (init_beliefs_reset!)
  reset_message_flags(e, also_factors ? 1 : 0);
  if (also_factors) e->have_factors = true;
  return PGBP_OK;
}

int pgbp_bm_tree_assignfactors(pgbp_engine* e, const double* Rinv, const double* logdetR, const double* mu,
                               int32_t per_site) {
  DeviceScope device_scope(e);
  if (!e || !Rinv || !logdetR || !mu) return PGBP_ERR_INVALID;
  if (!e->d_bm_kind) return e->fail(PGBP_ERR_STATE, "pgbp_bm_tree_assignfactors: call pgbp_bm_tree_setup first");
  const size_t n = per_site ? (size_t)e->plan.n_sites : 1, pp = (size_t)e->bm_p;
  HIPCHK(e, hipMemcpyAsync(e->d_bm_Rinv, Rinv, n * pp * pp * sizeof(double), hipMemcpyHostToDevice, e->st));
  HIPCHK(e, hipMemcpyAsync(e->d_bm_logdet, logdetR, n * sizeof(double), hipMemcpyHostToDevice, e->st));
  HIPCHK(e, hipMemcpyAsync(e->d_bm_mu, mu, n * pp * sizeof(double), hipMemcpyHostToDevice, e->st));
  e->bm_per_site = per_site ? 1 : 0;
  // R^-1 is symmetric and the fill writes symmetric blocks: the layout the traversals want can be kept
  e->sym_known = true;
  e->sym_ok = true;
  if (want_site_minor(e) && e->bm_p == 1 && !e->layout_sm) {  // univariate batch: fill straight in the site-minor layout
    const int rc = ensure_layout(e, false, true);
    if (rc) return rc;
  }
  return bm_fill_async(e, true);
}

int pgbp_enqueue_loglik_bm(pgbp_engine* e, int32_t reps, const pgbp_opts* opts) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  int rc = check_opts(e, opts);
  if (rc) return rc;
  if ((rc = need_schedule(e, 0))) return rc;
  if (!e->d_bm_kind) return e->fail(PGBP_ERR_STATE, "pgbp_enqueue_loglik_bm: call pgbp_bm_tree_setup / assignfactors first");
  if ((rc = reset_fail(e))) return rc;
  if ((rc = ensure_layout(e, want_bs16(e), want_site_minor(e)))) return rc;
  DevState S = dev_state(e, opts);
  const Plan& p = e->plan;
  for (int r = 0; r < reps; ++r) {
    DevState S1 = S;
    S1.sep_zero = fresh_sepsets_shortcut(e) ? 1 : 0;
    if ((rc = bm_fill_async(e, false, S1.sep_zero != 0))) return rc;   // assignfactors!        calibration.jl:205-209
    enqueue_tree(e, S1, 0, 1, 0);                                // postorder             :210
    const int root = p.trees[0].pa.empty() ? 0 : p.trees[0].pa[0];
    integrate_async(e, root, nullptr);  // :212
  }
  return e->take_enqueue_rc();
}

// ---- device factor assignment (any linear-Gaussian model, trees and networks) ---------------------------

int pgbp_lg_setup(pgbp_engine* e, const pgbp_lg_families* f) {
  DeviceScope device_scope(e);
  if (!e || !f || f->p <= 0 || f->p > PGBP_MAX_DIM || f->n_families < 0 || f->max_parents < 1 || f->n_rates < 1 ||
      f->n_rows < 0)
    return PGBP_ERR_INVALID;
  if (f->n_families > 0 && (!f->cluster || !f->n_parents || !f->child_pos || !f->data_row || !f->parent_pos || !f->length ||
                            !f->gamma || !f->color))
    return PGBP_ERR_INVALID;
  if (f->n_rows > 0 && !f->data) return e->fail(PGBP_ERR_INVALID, "pgbp_lg_setup: data missing");
  const Plan& p = e->plan;
  const int nc = p.n_clusters, K = f->max_parents, pp = f->p;
  if ((f->child_mask || f->parent_mask) && pp > 64) return e->fail(PGBP_ERR_INVALID, "pgbp_lg_setup: scope masks need p <= 64");
  const unsigned long long full = pp >= 64 ? ~0ull : ((1ull << pp) - 1ull);
  auto cmask = [&](int i) { return f->child_mask ? (unsigned long long)f->child_mask[i] & full : full; };
  auto pmask = [&](int i, int k) { return f->parent_mask ? (unsigned long long)f->parent_mask[(size_t)i * f->max_parents + k] & full : full; };
  // every family fits its cluster; the blocks of one family do not overlap
  std::vector<int32_t> count(nc + 1, 0);
  bool uni_ok = p.max_dim <= 2 && pp == 1;
  if (f->n_rows > 0 && !f->child_mask)
    for (size_t i = 0, n = (size_t)p.n_sites * f->n_rows * pp; i < n; ++i)
      if (!std::isfinite(f->data[i]))
        return e->fail(PGBP_ERR_INVALID, "pgbp_lg_setup: a tip value is missing or not finite (missing values need child_mask / parent_mask)");
  for (int i = 0; i < f->n_families; ++i) {
    const std::string where = "pgbp_lg_setup: family " + std::to_string(i) + ": ";
    const int c = f->cluster[i], np = f->n_parents[i];
    if (c < 0 || c >= nc) return e->fail(PGBP_ERR_INVALID, where + "cluster out of range");
    if (np < 0 || np > K) return e->fail(PGBP_ERR_INVALID, where + "number of parents out of range");
    const int m = p.dims[c];
    std::vector<std::pair<int, int>> pos;  // (first variable, number of variables) of every in-scope block
    const int cp = f->child_pos[i];
    const unsigned long long O = cmask(i);
    if (cp < 0 && O == 0 && f->child_mask) {
      // a node with nothing in scope (no data below it) or a tip with every trait missing: the factor integrates to 1
    } else if (cp < 0) {
      if (np == 0) return e->fail(PGBP_ERR_INVALID, where + "a root prior needs the root in scope");
      if (f->data_row[i] < 0 || f->data_row[i] >= f->n_rows) return e->fail(PGBP_ERR_INVALID, where + "data row out of range");
      for (int s2 = 0; s2 < p.n_sites; ++s2)
        for (int t = 0; t < pp; ++t)
          if (((O >> t) & 1ull) && !std::isfinite(f->data[((size_t)s2 * f->n_rows + f->data_row[i]) * pp + t]))
            return e->fail(PGBP_ERR_INVALID, where + "a tip value is missing or not finite where child_mask says observed "
                                                     "(missing values: clear the trait's bit in child_mask; one pattern for all sites)");
    } else {
      pos.push_back({cp, __builtin_popcountll(O)});
    }
    if (np == 0 && (f->color[(size_t)i * K] < 0 || f->color[(size_t)i * K] >= f->n_rates))
      return e->fail(PGBP_ERR_INVALID, where + "rate index out of range");
    for (int k = 0; k < np; ++k) {
      const size_t o = (size_t)i * K + k;
      if (!(f->length[o] > 0.0) || !std::isfinite(f->length[o]))
        return e->fail(PGBP_ERR_INVALID, where + "parent edge length must be positive (degenerate families are out of scope)");
      if (!std::isfinite(f->gamma[o])) return e->fail(PGBP_ERR_INVALID, where + "inheritance not finite");
      if (f->color[o] < 0 || f->color[o] >= f->n_rates) return e->fail(PGBP_ERR_INVALID, where + "rate index out of range");
      if (f->parent_pos[o] >= 0) {
        if ((O & ~pmask(i, k)) != 0)
          return e->fail(PGBP_ERR_INVALID, where + "a trait kept by the child is out of the parent's scope");
        pos.push_back({f->parent_pos[o], __builtin_popcountll(pmask(i, k))});
      }
    }
    std::sort(pos.begin(), pos.end());
    for (size_t a = 0; a < pos.size(); ++a)
      if (pos[a].first + pos[a].second > m || (a > 0 && pos[a].first < pos[a - 1].first + pos[a - 1].second))
        return e->fail(PGBP_ERR_INVALID, where + "variable blocks overlap or leave the cluster (dimension " + std::to_string(m) + ")");
    ++count[c + 1];
  }
  for (int c = 0; c < nc; ++c) count[c + 1] += count[c];
  std::vector<int32_t> fam(std::max(1, f->n_families));
  {
    std::vector<int32_t> at(count.begin(), count.end() - 1);
    for (int i = 0; i < f->n_families; ++i) fam[at[f->cluster[i]]++] = i;  // stable: the reference's loop order
  }
  for (void* q : e->lg_bufs)
    if (q) (void)hipFree(q);
  e->lg_bufs.clear();
  for (double** q : {&e->d_lg_R, &e->d_lg_alpha, &e->d_lg_theta, &e->d_lg_mu}) {
    if (*q) (void)hipFree(*q);
    *q = nullptr;
  }
  e->lg_ready = e->lg_have_params = false;
  const size_t nf = (size_t)f->n_families, nfk = nf * K;
  int rc;
  int32_t *d_off = nullptr, *d_fam = nullptr, *d_np = nullptr, *d_cp = nullptr, *d_row = nullptr, *d_pp = nullptr, *d_col = nullptr;
  double *d_len = nullptr, *d_gam = nullptr, *d_data = nullptr;
  auto keep = [&](void* q) { e->lg_bufs.push_back(q); };
  if ((rc = upload(e, &d_off, count))) return rc; keep(d_off);
  if ((rc = upload(e, &d_fam, fam))) return rc; keep(d_fam);
  if ((rc = upload(e, &d_np, std::vector<int32_t>(f->n_parents, f->n_parents + nf)))) return rc; keep(d_np);
  if ((rc = upload(e, &d_cp, std::vector<int32_t>(f->child_pos, f->child_pos + nf)))) return rc; keep(d_cp);
  if ((rc = upload(e, &d_row, std::vector<int32_t>(f->data_row, f->data_row + nf)))) return rc; keep(d_row);
  if ((rc = upload(e, &d_pp, std::vector<int32_t>(f->parent_pos, f->parent_pos + nfk)))) return rc; keep(d_pp);
  if ((rc = upload(e, &d_col, std::vector<int32_t>(f->color, f->color + nfk)))) return rc; keep(d_col);
  if ((rc = upload(e, &d_len, std::vector<double>(f->length, f->length + nfk)))) return rc; keep(d_len);
  if ((rc = upload(e, &d_gam, std::vector<double>(f->gamma, f->gamma + nfk)))) return rc; keep(d_gam);
  const size_t nd = (size_t)p.n_sites * f->n_rows * pp;
  if ((rc = dev_alloc(e, &d_data, nd))) return rc; keep(d_data);
  unsigned long long *d_cm = nullptr, *d_pm = nullptr;
  if (f->child_mask) {
    std::vector<unsigned long long> v(nf);
    for (size_t i = 0; i < nf; ++i) v[i] = cmask((int)i);
    if ((rc = upload(e, &d_cm, v))) return rc; keep(d_cm);
  }
  if (f->parent_mask) {
    std::vector<unsigned long long> v(nfk);
    for (size_t i = 0; i < nf; ++i)
      for (int k = 0; k < K; ++k) v[i * K + k] = pmask((int)i, k);
    if ((rc = upload(e, &d_pm, v))) return rc; keep(d_pm);
  }
  if (d_data && nd) {
    if (!f->child_mask) {  // complete data: checked finite above
      HIPCHK(e, hipMemcpy(d_data, f->data, nd * sizeof(double), hipMemcpyHostToDevice));
    } else {               // masked-out entries may be NaN on the host: never let them reach arithmetic
      std::vector<double> clean(f->data, f->data + nd);
      for (double& x : clean) if (!std::isfinite(x)) x = 0.0;
      HIPCHK(e, hipMemcpy(d_data, clean.data(), nd * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  const size_t ns = (size_t)p.n_sites;
  if ((rc = dev_alloc(e, &e->d_lg_R, ns * f->n_rates * pp * pp))) return rc;
  if ((rc = dev_alloc(e, &e->d_lg_alpha, ns))) return rc;
  if ((rc = dev_alloc(e, &e->d_lg_theta, ns * pp))) return rc;
  if ((rc = dev_alloc(e, &e->d_lg_mu, ns * pp))) return rc;
  double* d_data_sm = nullptr;
  if (uni_ok && f->n_rows > 0) {   // the thread-per-site fill reads the tip data with lanes = sites: keep a [row][site] copy
    if ((rc = dev_alloc(e, &d_data_sm, (size_t)f->n_rows * (size_t)sm_row(p.n_sites)))) return rc;
    keep(d_data_sm);
    launch_transpose_words_f64(d_data, d_data_sm, f->n_rows, p.n_sites, 1, e->st);
    HIPCHK(e, hipStreamSynchronize(e->st));
  }
  // one record per cluster where every cluster holds exactly one family with at most one parent and there are no scope masks
  LgSimpleFam* d_simple = nullptr;
  if (uni_ok && !f->child_mask && !f->parent_mask && f->n_families == nc) {
    std::vector<LgSimpleFam> sf(nc);
    bool simple = true;
    for (int c = 0; c < nc && simple; ++c) {
      if (count[c + 1] - count[c] != 1) { simple = false; break; }
      const int i = fam[count[c]];
      const int np = f->n_parents[i];
      if (np > 1) { simple = false; break; }
      LgSimpleFam r{};
      r.np = np; r.cpos = f->child_pos[i]; r.row = f->data_row[i];
      r.ppos = np > 0 ? f->parent_pos[(size_t)i * K] : -1;
      r.color = f->color[(size_t)i * K];
      r.length = np > 0 ? f->length[(size_t)i * K] : 0.0;
      r.gamma = np > 0 ? f->gamma[(size_t)i * K] : 0.0;
      if (r.cpos > 1 || r.ppos > 1) { simple = false; break; }
      sf[c] = r;
    }
    if (simple) {
      if ((rc = upload(e, &d_simple, sf))) return rc;
      keep(d_simple);
    }
  }
  e->lg = LgStatic{pp, K, f->n_rates, f->n_rows, d_off, d_fam, d_np, d_cp, d_row, d_pp, d_len, d_gam, d_col, d_data, d_cm, d_pm,
                   d_data_sm, d_simple};
  e->lg_ready = true;
  e->lg_uni_ok = uni_ok;
  return PGBP_OK;
}

// also_factors: assignfactors! followed by init_factors_frombeliefs!; without it only the beliefs are written, as in
// the body of score() (src/calibration.jl:205-209)
static int lg_fill_async(pgbp_engine* e, bool also_factors, bool skip_sepsets = false) {
  const Plan& p = e->plan;
  if (e->layout_sm && e->lg_uni_ok) {
    launch_lg_fill_uni_sm(e->lg, e->lgp, e->d_pool_sm, also_factors ? e->d_fpool_sm : nullptr, e->d_packed_off, e->d_bdim,
                          p.n_clusters, p.n_sites, e->st);
    const int64_t nc = p.packed_off[p.n_clusters] * sm_row(p.n_sites), nall = p.packed_off.back() * sm_row(p.n_sites);
    if (!skip_sepsets) HIPCHK(e, hipMemsetAsync(e->d_pool_sm + nc, 0, sizeof(double) * (size_t)(nall - nc), e->st));  // sepsets = 1
  } else {
    if (e->layout_sm) {
      const int rc0 = ensure_site_minor(e, false);
      if (rc0) return rc0;
    }
    launch_lg_fill(e->lg, e->lgp, e->d_pool, p.pool_stride(), also_factors ? e->d_fpool : nullptr, p.cluster_stride(),
                   e->d_boff, e->d_bdim, e->layout_bs16 ? 1 : 0, p.fast_p, p.max_dim, p.n_clusters, p.n_sites, e->st);
    if (!skip_sepsets)
      launch_zero_strided(e->d_pool + p.cluster_stride(), p.pool_stride(), p.pool_stride() - p.cluster_stride(),
                          p.n_sites, e->st);  // sepsets = 1 (init_beliefs_reset!)
  }
  reset_message_flags(e, also_factors ? 1 : 0);
  if (also_factors) e->have_factors = true;
  return PGBP_OK;
}

int pgbp_lg_assignfactors(pgbp_engine* e, const pgbp_lg_params* m) {
  DeviceScope device_scope(e);
  if (!e || !m || !m->R || !m->mu) return PGBP_ERR_INVALID;
  if (!e->lg_ready) return e->fail(PGBP_ERR_STATE, "pgbp_lg_assignfactors: call pgbp_lg_setup first");
  if (m->model != PGBP_LG_BM && m->model != PGBP_LG_OU) return e->fail(PGBP_ERR_INVALID, "pgbp_lg_assignfactors: unknown model");
  if (m->model == PGBP_LG_OU && (!m->alpha || !m->theta))
    return e->fail(PGBP_ERR_INVALID, "pgbp_lg_assignfactors: the OU model needs alpha and theta");
  const size_t n = m->per_site ? (size_t)e->plan.n_sites : 1, pp = (size_t)e->lg.p;
  HIPCHK(e, hipMemcpyAsync(e->d_lg_R, m->R, n * e->lg.n_rates * pp * pp * sizeof(double), hipMemcpyHostToDevice, e->st));
  HIPCHK(e, hipMemcpyAsync(e->d_lg_mu, m->mu, n * pp * sizeof(double), hipMemcpyHostToDevice, e->st));
  if (m->alpha) HIPCHK(e, hipMemcpyAsync(e->d_lg_alpha, m->alpha, n * sizeof(double), hipMemcpyHostToDevice, e->st));
  if (m->theta) HIPCHK(e, hipMemcpyAsync(e->d_lg_theta, m->theta, n * pp * sizeof(double), hipMemcpyHostToDevice, e->st));
  HIPCHK(e, hipStreamSynchronize(e->st));  // the host buffers may go away
  e->lgp = LgParams{m->model, m->per_site ? 1 : 0, e->d_lg_R, e->d_lg_alpha, m->theta ? e->d_lg_theta : nullptr, e->d_lg_mu};
  e->lg_have_params = true;
  // the fill writes exactly symmetric blocks: the layout the traversals want can be kept
  e->sym_known = true;
  e->sym_ok = true;
  // a univariate batch is filled straight in the site-minor layout its traversals use (the wavefront-per-cluster
  // kernel would spend one workgroup per (cluster, site))
  if (want_site_minor(e) && e->lg_uni_ok && !e->layout_sm) {
    const int rc = ensure_layout(e, false, true);
    if (rc) return rc;
  }
  return lg_fill_async(e, true);
}

int pgbp_enqueue_loglik_lg(pgbp_engine* e, int32_t reps, const pgbp_opts* opts) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  int rc = check_opts(e, opts);
  if (rc) return rc;
  if ((rc = need_schedule(e, 0))) return rc;
  if (!e->lg_ready || !e->lg_have_params)
    return e->fail(PGBP_ERR_STATE, "pgbp_enqueue_loglik_lg: call pgbp_lg_setup / pgbp_lg_assignfactors first");
  if ((rc = reset_fail(e))) return rc;
  if ((rc = ensure_layout(e, want_bs16(e), want_site_minor(e)))) return rc;
  DevState S = dev_state(e, opts);
  const Plan& p = e->plan;
  for (int r = 0; r < reps; ++r) {
    DevState S1 = S;
    S1.sep_zero = fresh_sepsets_shortcut(e) ? 1 : 0;
    if ((rc = lg_fill_async(e, false, S1.sep_zero != 0))) return rc;  // assignfactors!        calibration.jl:205-209
    enqueue_tree(e, S1, 0, 1, 0);                                     // postorder             :210
    const int root = p.trees[0].pa.empty() ? 0 : p.trees[0].pa[0];
    integrate_async(e, root, nullptr);                                // :212
  }
  return e->take_enqueue_rc();
}

// ---- benchmarking / zero-copy entry points ---------------------------------------------------

// one iteration of calibrate! over every schedule tree.  ev != null: one HIP event pair around the message launches of
// each tree (they run back to back on the stream), *n_launches += their number
static int enqueue_calibrate_once(pgbp_engine* e, const DevState& S, int reset_each,
                                  std::vector<std::pair<hipEvent_t, hipEvent_t>>* ev, int* n_launches = nullptr) {
  const Plan& p = e->plan;
  if (reset_each) {
    int rc = reset_from_factors_async(e);
    if (rc) return rc;
    reset_message_flags(e, 1);
  }
  for (int j = 0; j < (int)p.trees.size(); ++j) {
    hipEvent_t a = nullptr, b = nullptr;
    if (ev) {
      HIPCHK(e, hipEventCreate(&a));
      HIPCHK(e, hipEventCreate(&b));
      HIPCHK(e, hipEventRecord(a, e->st));
    }
    enqueue_pair_and_iscal(e, S, j, (unsigned long long)j, false, e->d_iscal, n_launches);
    if (ev) {
      HIPCHK(e, hipEventRecord(b, e->st));
      ev->push_back({a, b});
    }
  }
  return PGBP_OK;
}

int pgbp_enqueue_calibrate(pgbp_engine* e, int32_t reps, int32_t reset_each, const pgbp_opts* opts) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  int rc = check_opts(e, opts);
  if (rc) return rc;
  if ((rc = need_schedule(e, 0))) return rc;
  if ((rc = reset_fail(e))) return rc;
  if ((rc = ensure_layout(e, want_bs16(e), want_site_minor(e)))) return rc;
  DevState S = dev_state(e, opts);
  for (int r = 0; r < reps; ++r)
    if ((rc = enqueue_calibrate_once(e, S, reset_each, nullptr))) return rc;
  return e->take_enqueue_rc();
}

static void drop_kernel_events(pgbp_engine* e) {
  for (auto& pr : e->kernel_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  e->kernel_events.clear();
  e->kernel_launches = 0;
}

int pgbp_enqueue_calibrate_timed(pgbp_engine* e, int32_t reps, int32_t reset_each, const pgbp_opts* opts) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  int rc = check_opts(e, opts);
  if (rc) return rc;
  if ((rc = need_schedule(e, 0))) return rc;
  if ((rc = reset_fail(e))) return rc;
  if ((rc = ensure_layout(e, want_bs16(e), want_site_minor(e)))) return rc;
  drop_kernel_events(e);
  DevState S = dev_state(e, opts);
  int launches = 0;
  if (reset_each) {
    // resets between the repetitions: an event pair around the message launches of every schedule tree
    for (int r = 0; r < reps; ++r)
      if ((rc = enqueue_calibrate_once(e, S, reset_each, &e->kernel_events, &launches))) return rc;
  } else {
    // nothing but message launches (and the flag reduction behind each tree) in the region: ONE pair around all of it
    // -- an event record is a barrier packet on the stream, and two of them per repetition cost 4 % of a cfg3 calibrate
    hipEvent_t a = nullptr, b = nullptr;
    HIPCHK(e, hipEventCreate(&a));
    HIPCHK(e, hipEventCreate(&b));
    e->kernel_events.push_back({a, b});
    HIPCHK(e, hipEventRecord(a, e->st));
    for (int r = 0; r < reps; ++r)
      if ((rc = enqueue_calibrate_once(e, S, 0, nullptr, &launches))) return rc;
    HIPCHK(e, hipEventRecord(b, e->st));
  }
  e->kernel_launches = launches;
  return e->take_enqueue_rc();
}

int pgbp_fetch_kernel_time(pgbp_engine* e, float* ms_kernels, int32_t* n_launches) {
  DeviceScope device_scope(e);
  if (!e || !ms_kernels) return PGBP_ERR_INVALID;
  HIPCHK(e, hipStreamSynchronize(e->st));
  HIPCHK(e, hipGetLastError());
  double total = 0;
  for (auto& pr : e->kernel_events) {
    float ms = 0;
    HIPCHK(e, hipEventElapsedTime(&ms, pr.first, pr.second));
    total += ms;
  }
  *ms_kernels = (float)total;
  if (n_launches) *n_launches = e->kernel_launches;
  drop_kernel_events(e);
  return PGBP_OK;
}

static int enqueue_loglik_once(pgbp_engine* e, const DevState& S0) {
  const Plan& p = e->plan;
  DevState S = S0;
  S.sep_zero = fresh_sepsets_shortcut(e) ? 1 : 0;
  int rc = reset_from_factors_async(e, S.sep_zero != 0);
  if (rc) return rc;
  reset_message_flags(e, 1);  // calibration.jl:209
  enqueue_tree(e, S, 0, 1, 0);                                                              // :210
  const int root = p.trees[0].pa.empty() ? 0 : p.trees[0].pa[0];
  integrate_async(e, root, nullptr);         // :212
  return PGBP_OK;
}

int pgbp_enqueue_loglik(pgbp_engine* e, int32_t reps, const pgbp_opts* opts) {
  DeviceScope device_scope(e);
  if (!e) return PGBP_ERR_INVALID;
  int rc = check_opts(e, opts);
  if (rc) return rc;
  if ((rc = need_schedule(e, 0))) return rc;
  if ((rc = reset_fail(e))) return rc;
  if ((rc = ensure_layout(e, want_bs16(e), want_site_minor(e)))) return rc;
  DevState S = dev_state(e, opts);
  for (int r = 0; r < reps; ++r)
    if ((rc = enqueue_loglik_once(e, S))) return rc;
  return e->take_enqueue_rc();
}

int pgbp_fetch_loglik(pgbp_engine* e, double* norm, int32_t* info) {
  DeviceScope device_scope(e);
  if (!e || !norm) return PGBP_ERR_INVALID;
  const int ns = e->plan.n_sites;
  HIPCHK(e, hipMemcpyAsync(norm, e->d_norm, sizeof(double) * ns, hipMemcpyDeviceToHost, e->st));
  if (info) HIPCHK(e, hipMemcpyAsync(info, e->d_info, sizeof(int32_t) * ns, hipMemcpyDeviceToHost, e->st));
  HIPCHK(e, hipStreamSynchronize(e->st));
  HIPCHK(e, hipGetLastError());
  if (const int erc = e->take_enqueue_rc()) return erc;
  if (info) {
    // a failed postorder message also invalidates the likelihood
    std::vector<unsigned long long> keys(ns);
    HIPCHK(e, hipMemcpy(keys.data(), e->d_fail, sizeof(unsigned long long) * ns, hipMemcpyDeviceToHost));
    for (int s = 0; s < ns; ++s)
      if (is_failure_key(keys[s]) && info[s] == 0) info[s] = (int32_t)(keys[s] & ((1ull << kInfoBits) - 1));
  }
  return PGBP_OK;
}

int pgbp_time_enqueued(pgbp_engine* e, int32_t kind, int32_t reps, int32_t reset_each, const pgbp_opts* opts,
                       float* ms_total) {
  DeviceScope device_scope(e);
  if (!e || !ms_total || kind < 0 || kind > 3) return PGBP_ERR_INVALID;
  hipEvent_t a, b;
  HIPCHK(e, hipEventCreate(&a));
  HIPCHK(e, hipEventCreate(&b));
  HIPCHK(e, hipStreamSynchronize(e->st));
  HIPCHK(e, hipEventRecord(a, e->st));
  int rc = kind == 0 ? pgbp_enqueue_calibrate(e, reps, reset_each, opts)
                     : (kind == 3 ? pgbp_enqueue_loglik_lg(e, reps, opts)
                                  : (kind == 2 ? pgbp_enqueue_loglik_bm(e, reps, opts) : pgbp_enqueue_loglik(e, reps, opts)));
  if (rc == PGBP_OK) {
    HIPCHK(e, hipEventRecord(b, e->st));
    HIPCHK(e, hipEventSynchronize(b));
    HIPCHK(e, hipEventElapsedTime(ms_total, a, b));
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return rc;
}

int pgbp_time_message_kernels(pgbp_engine* e, int32_t reps, const pgbp_opts* opts, float* ms_kernels,
                              int32_t* n_launches) {
  DeviceScope device_scope(e);
  if (!e || !ms_kernels) return PGBP_ERR_INVALID;
  int rc = pgbp_enqueue_calibrate_timed(e, reps, 1, opts);
  if (rc) return rc;
  return pgbp_fetch_kernel_time(e, ms_kernels, n_launches);
}

int pgbp_traffic_model(const pgbp_engine* e, double* bytes_per_calibrate, int64_t* messages_per_calibrate) {
  if (!e) return PGBP_ERR_INVALID;
  int64_t nm = 0;
  double b = plan_bytes_per_calibrate(e->plan, &nm);
  if (bytes_per_calibrate) *bytes_per_calibrate = b;
  if (messages_per_calibrate) *messages_per_calibrate = nm;
  return PGBP_OK;
}

}  // extern "C"

// ---- internal (pgbp_dist.hip): one rank's contribution to the all-gather of pgbp_comm_gather_loglik -----------------------
namespace {
__global__ void pack_gather_slot(const double* __restrict__ norm, const int32_t* __restrict__ info,
                                 const unsigned long long* __restrict__ fail, const int32_t* __restrict__ iscal, int ns,
                                 int slot_sites, double* __restrict__ out) {
  // out[0 .. slot): log-likelihoods; out[slot .. 2 slot): info words; out[2 slot], out[2 slot + 1]: min over this rank's
  // sites of succ (no failed message) and iscal.  One workgroup.
  __shared__ int s_succ, s_iscal;
  if (threadIdx.x == 0) { s_succ = 1; s_iscal = 1; }
  __syncthreads();
  for (int i = threadIdx.x; i < slot_sites; i += blockDim.x) {
    const bool in = i < ns;
    const bool failed = in && pgbp::is_failure_key(fail[i]);
    out[i] = in ? norm[i] : 0.0;
    out[slot_sites + i] = in ? (double)(failed && info[i] == 0 ? (int)(fail[i] & ((1ull << pgbp::kInfoBits) - 1)) : info[i]) : 0.0;
    if (failed) atomicMin(&s_succ, 0);
    if (in && iscal[i] == 0) atomicMin(&s_iscal, 0);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * slot_sites] = (double)s_succ;
    out[2 * slot_sites + 1] = (double)s_iscal;
  }
}
}  // namespace

namespace pgbp {
int engine_pack_gather_slot(pgbp_engine* e, int32_t slot_sites, double** d_slot, hipStream_t* st, int32_t* n_sites) {
  if (!e || !d_slot || !st) return PGBP_ERR_INVALID;
  DeviceScope device_scope(e);
  const int ns = e->plan.n_sites;
  if (slot_sites < ns) return e->fail(PGBP_ERR_INVALID, "gather slot smaller than this rank's number of sites");
  const int64_t need = 2 * (int64_t)slot_sites + 2;
  if (need > e->gather_cap) {
    if (e->d_gather) (void)hipFree(e->d_gather);
    e->d_gather = nullptr;
    e->gather_cap = 0;
    int rc = dev_alloc(e, &e->d_gather, (size_t)need);
    if (rc) return rc;
    e->gather_cap = need;
  }
  hipLaunchKernelGGL(pack_gather_slot, dim3(1), dim3(256), 0, e->st, e->d_norm, e->d_info, e->d_fail, e->d_iscal, ns,
                     slot_sites, e->d_gather);
  HIPCHK(e, hipGetLastError());
  *d_slot = e->d_gather;
  *st = e->st;
  if (n_sites) *n_sites = ns;
  return PGBP_OK;
}
// the records of the listed beliefs of one site <-> a device buffer of the CALLER (pgbp_comm's send slot / a rank's part of
// its receive buffer), on the engine's stream: the device halves of pgbp_pack_beliefs / pgbp_unpack_beliefs
int engine_pack_records_device(pgbp_engine* e, int32_t site, int32_t n, const int32_t* beliefs, double* d_buf, int to_buf,
                               hipStream_t* st, int64_t* total) {
  if (!e || n < 0 || (n > 0 && (!beliefs || !d_buf))) return PGBP_ERR_INVALID;
  DeviceScope device_scope(e);
  int64_t tot = 0;
  int rc = xbuf_prepare(e, site, n, beliefs, &tot);
  if (rc) return rc;
  if (n > 0) {
    if (!to_buf) e->sym_known = false;
    launch_pack_records(e->d_pool + (int64_t)site * e->plan.pool_stride(), e->d_xoff, e->d_xoff + n, n, d_buf, to_buf, e->st);
    HIPCHK(e, hipGetLastError());
  }
  if (st) *st = e->st;
  if (total) *total = tot;
  return PGBP_OK;
}
int engine_fail(pgbp_engine* e, int code, const std::string& msg) { return e->fail(code, msg); }
int engine_device(const pgbp_engine* e) { return e->plan.device; }
int engine_n_sites(const pgbp_engine* e) { return e->plan.n_sites; }
}  // namespace pgbp
