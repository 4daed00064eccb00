// Several GPUs behind the C ABI (include/pgbp.h, "several GPUs").
//
// Independent sites are the one dimension of the path that shards without an exchange step: calibrate!() of one site
// never reads another site (src/calibration.jl:35-60 runs on one ClusterGraphBelief).  Two forms:
//   * pgbp_group  -- ONE process, several devices: one engine + stream per device over contiguous site ranges, every call
//     fans out on one host thread per device (HIP's current device is per thread; the engines share nothing) and gathers
//     results in site order.  No collective at all: the host already owns every result buffer.
//   * pgbp_comm   -- ONE PROCESS PER GPU (torchrun, MPI, Julia's Distributed): ranks hold their own engine; the only
//     exchange is ONE ncclAllGather (RCCL over xGMI) per call carrying every rank's per-site log-likelihoods, info words
//     and its (succ, iscal) pair -- a few KB, latency-bound, so one fused collective instead of an all-gather plus an
//     all-reduce(min); the minimum is taken on the host from the gathered pairs.
// RCCL is bound at run time (dlopen of librccl.so.1: the copy the host process already loaded, e.g. PyTorch's, if any), so
// libpgbp.so itself has no link-time dependency on it and loads on machines without RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pgbp_internal.hpp"
#include "pgbp_kernels.hpp"

// ------------------------------------------------------------------------------------------------ one process, many devices
struct pgbp_group {
  std::vector<pgbp_engine*> eng;
  std::vector<int32_t> first, count;  // site range of each shard
  int32_t n_sites = 0;
  int64_t packed = 0;  // doubles per site
  int32_t max_dim = 1;
  std::string err;
};

namespace {

// contiguous, balanced: the first n % k shards get one site more (same rule as pgbp_amd/sharding.py:shard_range)
void shard_range(int n, int k, int i, int32_t* lo, int32_t* cnt) {
  const int base = n / k, extra = n % k;
  *lo = i * base + std::min(i, extra);
  *cnt = base + (i < extra ? 1 : 0);
}

// fn(shard) on one host thread per shard; returns the first non-zero status (by shard index) and keeps its message
template <class F>
int for_shards(pgbp_group* g, F fn) {
  const int k = (int)g->eng.size();
  std::vector<int> rc(k, PGBP_OK);
  if (k == 1) {
    rc[0] = fn(0);
  } else {
    std::vector<std::thread> th;
    th.reserve(k);
    for (int i = 0; i < k; ++i) th.emplace_back([&, i] { rc[i] = fn(i); });
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < k; ++i)
    if (rc[i] != PGBP_OK) {
      g->err = "shard " + std::to_string(i) + " (device " + std::to_string(pgbp::engine_device(g->eng[i])) +
               "): " + pgbp_last_error(g->eng[i]);
      return rc[i];
    }
  return PGBP_OK;
}

thread_local std::string g_group_create_error;

}  // namespace

extern "C" {

const char* pgbp_group_last_error(const pgbp_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

void pgbp_group_destroy(pgbp_group* g) {
  if (!g) return;
  for (pgbp_engine* e : g->eng) pgbp_destroy(e);
  delete g;
}

int pgbp_group_create(const pgbp_desc* desc, int32_t n_devices, const int32_t* devices, pgbp_group** out) {
  if (!out) return PGBP_ERR_INVALID;
  *out = nullptr;
  if (!desc || n_devices < 1 || !devices || desc->n_sites < n_devices) {
    g_group_create_error = "pgbp_group_create: need 1 <= n_devices <= n_sites and a device list";
    return PGBP_ERR_INVALID;
  }
  std::unique_ptr<pgbp_group> g(new pgbp_group());
  g->n_sites = desc->n_sites;
  g->eng.assign(n_devices, nullptr);
  g->first.resize(n_devices);
  g->count.resize(n_devices);
  for (int i = 0; i < n_devices; ++i) shard_range(desc->n_sites, n_devices, i, &g->first[i], &g->count[i]);
  std::vector<int> rc(n_devices, PGBP_OK);
  std::vector<std::string> msg(n_devices);
  {
    std::vector<std::thread> th;
    for (int i = 0; i < n_devices; ++i)
      th.emplace_back([&, i] {
        pgbp_desc d = *desc;
        d.n_sites = g->count[i];
        d.device = devices[i];
        rc[i] = pgbp_create(&d, &g->eng[i]);
        if (rc[i]) msg[i] = pgbp_last_error(nullptr);  // (thread-local in the engine: read on the creating thread)
      });
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < n_devices; ++i)
    if (rc[i]) {
      g_group_create_error = "shard " + std::to_string(i) + " (device " + std::to_string(devices[i]) + "): " + msg[i];
      const int code = rc[i];
      pgbp_group_destroy(g.release());
      return code;
    }
  g->packed = pgbp_packed_size(g->eng[0]);
  for (int b = 0; b < desc->n_clusters + desc->n_sepsets; ++b) g->max_dim = std::max(g->max_dim, desc->dims[b]);
  *out = g.release();
  return PGBP_OK;
}

int32_t pgbp_group_size(const pgbp_group* g) { return g ? (int32_t)g->eng.size() : -1; }

pgbp_engine* pgbp_group_engine(pgbp_group* g, int32_t shard) {
  return (g && shard >= 0 && shard < (int)g->eng.size()) ? g->eng[shard] : nullptr;
}

int pgbp_group_range(const pgbp_group* g, int32_t shard, int32_t* first_site, int32_t* n_sites) {
  if (!g || shard < 0 || shard >= (int)g->eng.size()) return PGBP_ERR_INVALID;
  if (first_site) *first_site = g->first[shard];
  if (n_sites) *n_sites = g->count[shard];
  return PGBP_OK;
}

int pgbp_group_set_schedule(pgbp_group* g, int32_t n_trees, const int32_t* tree_off, const int32_t* pa_j,
                            const int32_t* ch_j) {
  if (!g) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) { return pgbp_set_schedule(g->eng[i], n_trees, tree_off, pa_j, ch_j); });
}

int pgbp_group_set_beliefs(pgbp_group* g, const double* packed, int32_t snapshot_factors) {
  if (!g || !packed) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) {
    return pgbp_set_beliefs(g->eng[i], packed + (int64_t)g->first[i] * g->packed, snapshot_factors);
  });
}

int pgbp_group_get_beliefs(pgbp_group* g, double* packed) {
  if (!g || !packed) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) { return pgbp_get_beliefs(g->eng[i], packed + (int64_t)g->first[i] * g->packed); });
}

int pgbp_group_reset_from_factors(pgbp_group* g) {
  if (!g) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) { return pgbp_reset_from_factors(g->eng[i]); });
}

int pgbp_group_calibrate(pgbp_group* g, int32_t niter, const pgbp_opts* opts, pgbp_result* results) {
  if (!g || !results) return PGBP_ERR_INVALID;
  // every shard runs the reference's loop on its own sites; with auto_stop every site stops at its own first calibrated
  // schedule tree (pgbp_calibrate), whichever device holds it
  return for_shards(g, [&](int i) { return pgbp_calibrate(g->eng[i], niter, opts, results + g->first[i]); });
}

int pgbp_group_integrate(pgbp_group* g, int32_t belief, double* mu, double* norm, int32_t* info) {
  if (!g || !norm) return PGBP_ERR_INVALID;
  const int64_t m = pgbp_belief_dim(g->eng[0], belief);  // mu has one row of the belief's dimension per site
  if (m < 0) {
    g->err = "belief index out of range";
    return PGBP_ERR_INVALID;
  }
  return for_shards(g, [&](int i) {
    return pgbp_integrate(g->eng[i], belief, mu ? mu + (int64_t)g->first[i] * m : nullptr, norm + g->first[i],
                          info ? info + g->first[i] : nullptr);
  });
}

int pgbp_group_lg_setup(pgbp_group* g, const pgbp_lg_families* f) {
  if (!g || !f) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) {
    pgbp_lg_families fi = *f;  // the static tables are shared; the tip data is [n_sites][n_rows][p]
    if (f->data) fi.data = f->data + (int64_t)g->first[i] * f->n_rows * f->p;
    return pgbp_lg_setup(g->eng[i], &fi);
  });
}

int pgbp_group_lg_assignfactors(pgbp_group* g, const pgbp_lg_params* m, int32_t n_rates, int32_t p) {
  if (!g || !m) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) {
    pgbp_lg_params mi = *m;
    if (m->per_site) {  // one parameter set per site: each shard takes its rows
      const int64_t s0 = g->first[i];
      if (m->R) mi.R = m->R + s0 * n_rates * p * p;
      if (m->alpha) mi.alpha = m->alpha + s0;
      if (m->theta) mi.theta = m->theta + s0 * p;
      if (m->mu) mi.mu = m->mu + s0 * p;
    }
    return pgbp_lg_assignfactors(g->eng[i], &mi);
  });
}

int pgbp_group_enqueue_calibrate(pgbp_group* g, int32_t reps, int32_t reset_each, const pgbp_opts* opts) {
  if (!g) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) { return pgbp_enqueue_calibrate(g->eng[i], reps, reset_each, opts); });
}

int pgbp_group_enqueue_loglik_lg(pgbp_group* g, int32_t reps, const pgbp_opts* opts) {
  if (!g) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) { return pgbp_enqueue_loglik_lg(g->eng[i], reps, opts); });
}

int pgbp_group_enqueue_loglik(pgbp_group* g, int32_t reps, const pgbp_opts* opts) {
  if (!g) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) { return pgbp_enqueue_loglik(g->eng[i], reps, opts); });
}

int pgbp_group_fetch_loglik(pgbp_group* g, double* norm, int32_t* info) {
  if (!g || !norm) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) {
    return pgbp_fetch_loglik(g->eng[i], norm + g->first[i], info ? info + g->first[i] : nullptr);
  });
}

int pgbp_group_sync(pgbp_group* g) {
  if (!g) return PGBP_ERR_INVALID;
  return for_shards(g, [&](int i) { return pgbp_sync(g->eng[i]); });
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ several scope patterns
// Sites whose data miss different traits at different tips have different scopes (allocatebeliefs, src/beliefs.jl:551-559:
// an internal node has a trait in scope iff some tip below it has a value for it): different belief dimensions, different
// index maps -- a different pgbp_desc.  One engine per PATTERN, the sites of a pattern batched inside it, every call fanned
// out over the engines (their streams run side by side on the device) and the per-site results scattered back into the
// caller's site order.
struct pgbp_patterns {
  std::vector<pgbp_engine*> eng;
  std::vector<int32_t> first, count;  // position of each pattern's sites in `sites`
  std::vector<int32_t> sites;         // global site index of every site, pattern after pattern
  int32_t n_sites = 0;
  std::string err;
};

namespace {

thread_local std::string g_patterns_create_error;

template <class F>
int for_patterns(pgbp_patterns* g, F fn) {
  const int k = (int)g->eng.size();
  std::vector<int> rc(k, PGBP_OK);
  if (k == 1) {
    rc[0] = fn(0);
  } else {
    std::vector<std::thread> th;
    th.reserve(k);
    for (int i = 0; i < k; ++i) th.emplace_back([&, i] { rc[i] = fn(i); });
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < k; ++i)
    if (rc[i] != PGBP_OK) {
      g->err = "pattern " + std::to_string(i) + ": " + pgbp_last_error(g->eng[i]);
      return rc[i];
    }
  return PGBP_OK;
}

}  // namespace

extern "C" {

const char* pgbp_patterns_last_error(const pgbp_patterns* g) { return g ? g->err.c_str() : g_patterns_create_error.c_str(); }

void pgbp_patterns_destroy(pgbp_patterns* g) {
  if (!g) return;
  for (pgbp_engine* e : g->eng) pgbp_destroy(e);
  delete g;
}

int pgbp_patterns_create(int32_t n_patterns, const pgbp_desc* const* descs, const int32_t* sites, pgbp_patterns** out) {
  if (!out) return PGBP_ERR_INVALID;
  *out = nullptr;
  if (n_patterns < 1 || !descs || !sites) {
    g_patterns_create_error = "pgbp_patterns_create: need at least one pattern, its descriptions and the site list";
    return PGBP_ERR_INVALID;
  }
  std::unique_ptr<pgbp_patterns> g(new pgbp_patterns());
  for (int k = 0; k < n_patterns; ++k) {
    if (!descs[k] || descs[k]->n_sites < 1 || descs[k]->n_clusters != descs[0]->n_clusters ||
        descs[k]->n_sepsets != descs[0]->n_sepsets) {
      g_patterns_create_error = "pgbp_patterns_create: pattern " + std::to_string(k) +
                                " is not the same cluster graph (clusters, sepsets) as pattern 0, or has no site";
      return PGBP_ERR_INVALID;
    }
    g->first.push_back(g->n_sites);
    g->count.push_back(descs[k]->n_sites);
    g->n_sites += descs[k]->n_sites;
  }
  g->sites.assign(sites, sites + g->n_sites);
  {  // a permutation of 0 .. n_sites - 1
    std::vector<char> seen(g->n_sites, 0);
    for (int32_t v : g->sites)
      if (v < 0 || v >= g->n_sites || seen[v]++) {
        g_patterns_create_error = "pgbp_patterns_create: `sites` is not a permutation of 0 .. n_sites - 1";
        return PGBP_ERR_INVALID;
      }
  }
  g->eng.assign(n_patterns, nullptr);
  for (int k = 0; k < n_patterns; ++k) {
    const int rc = pgbp_create(descs[k], &g->eng[k]);
    if (rc) {
      g_patterns_create_error = "pattern " + std::to_string(k) + ": " + pgbp_last_error(nullptr);
      pgbp_patterns_destroy(g.release());
      return rc;
    }
  }
  *out = g.release();
  return PGBP_OK;
}

int32_t pgbp_patterns_size(const pgbp_patterns* g) { return g ? (int32_t)g->eng.size() : -1; }

pgbp_engine* pgbp_patterns_engine(pgbp_patterns* g, int32_t pattern) {
  return (g && pattern >= 0 && pattern < (int)g->eng.size()) ? g->eng[pattern] : nullptr;
}

int pgbp_patterns_set_schedule(pgbp_patterns* g, int32_t n_trees, const int32_t* tree_off, const int32_t* pa_j,
                               const int32_t* ch_j) {
  if (!g) return PGBP_ERR_INVALID;
  return for_patterns(g, [&](int i) { return pgbp_set_schedule(g->eng[i], n_trees, tree_off, pa_j, ch_j); });
}

int pgbp_patterns_calibrate(pgbp_patterns* g, int32_t niter, const pgbp_opts* opts, pgbp_result* results) {
  if (!g || !results) return PGBP_ERR_INVALID;
  std::vector<pgbp_result> tmp(g->n_sites);
  const int rc = for_patterns(g, [&](int i) { return pgbp_calibrate(g->eng[i], niter, opts, tmp.data() + g->first[i]); });
  if (rc) return rc;
  for (int q = 0; q < g->n_sites; ++q) results[g->sites[q]] = tmp[q];
  return PGBP_OK;
}

int pgbp_patterns_enqueue_calibrate(pgbp_patterns* g, int32_t reps, int32_t reset_each, const pgbp_opts* opts) {
  if (!g) return PGBP_ERR_INVALID;
  return for_patterns(g, [&](int i) { return pgbp_enqueue_calibrate(g->eng[i], reps, reset_each, opts); });
}

int pgbp_patterns_enqueue_loglik(pgbp_patterns* g, int32_t reps, const pgbp_opts* opts) {
  if (!g) return PGBP_ERR_INVALID;
  return for_patterns(g, [&](int i) { return pgbp_enqueue_loglik(g->eng[i], reps, opts); });
}

int pgbp_patterns_enqueue_loglik_lg(pgbp_patterns* g, int32_t reps, const pgbp_opts* opts) {
  if (!g) return PGBP_ERR_INVALID;
  return for_patterns(g, [&](int i) { return pgbp_enqueue_loglik_lg(g->eng[i], reps, opts); });
}

int pgbp_patterns_fetch_loglik(pgbp_patterns* g, double* norm, int32_t* info) {
  if (!g || !norm) return PGBP_ERR_INVALID;
  std::vector<double> tn(g->n_sites);
  std::vector<int32_t> ti(g->n_sites);
  const int rc = for_patterns(g, [&](int i) { return pgbp_fetch_loglik(g->eng[i], tn.data() + g->first[i], ti.data() + g->first[i]); });
  if (rc) return rc;
  for (int q = 0; q < g->n_sites; ++q) {
    norm[g->sites[q]] = tn[q];
    if (info) info[g->sites[q]] = ti[q];
  }
  return PGBP_OK;
}

int pgbp_patterns_integrate(pgbp_patterns* g, int32_t belief, double* norm, int32_t* info) {
  if (!g || !norm) return PGBP_ERR_INVALID;
  std::vector<double> tn(g->n_sites);
  std::vector<int32_t> ti(g->n_sites);
  const int rc = for_patterns(g, [&](int i) {
    return pgbp_integrate(g->eng[i], belief, nullptr, tn.data() + g->first[i], ti.data() + g->first[i]);
  });
  if (rc) return rc;
  for (int q = 0; q < g->n_sites; ++q) {
    norm[g->sites[q]] = tn[q];
    if (info) info[g->sites[q]] = ti[q];
  }
  return PGBP_OK;
}

int pgbp_patterns_sync(pgbp_patterns* g) {
  if (!g) return PGBP_ERR_INVALID;
  return for_patterns(g, [&](int i) { return pgbp_sync(g->eng[i]); });
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ one process per GPU: RCCL
namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  std::string err;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // by soname: the copy already mapped into the process (PyTorch ships one) is returned if there is one.
    // PGBP_RCCL_LIB=<path> names the one library to try instead (a site with its own build; the test of this very path)
    std::string last;
    auto try_open = [&](const char* name) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!r.lib) {
        const char* m = dlerror();   // (read ONCE: glibc clears the message on the first call)
        last = m ? m : "";
      }
      return r.lib != nullptr;
    };
    if (const char* forced = getenv("PGBP_RCCL_LIB")) {
      (void)try_open(forced);
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if (try_open(name)) break;
    }
    if (!r.lib) {
      r.err = std::string("RCCL not found (dlopen librccl.so.1): ") + last;
      return;
    }
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.lib, "ncclCommInitRank");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.lib, "ncclCommDestroy");
    r.all_gather = (decltype(r.all_gather))dlsym(r.lib, "ncclAllGather");
    r.error_string = (decltype(r.error_string))dlsym(r.lib, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.error_string)
      r.err = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
  });
  return r;
}

thread_local std::string g_comm_create_error;

}  // namespace

struct pgbp_comm {
  ncclComm_t comm = nullptr;
  int32_t n_ranks = 1, rank = 0, device = 0;
  double* d_recv = nullptr;
  int64_t recv_cap = 0;
  double* d_send = nullptr;    // send slot of pgbp_comm_exchange_beliefs
  int64_t send_cap = 0;
  std::vector<double> h_recv;
  std::string err;
};

extern "C" {

const char* pgbp_comm_last_error(const pgbp_comm* c) { return c ? c->err.c_str() : g_comm_create_error.c_str(); }

int pgbp_comm_unique_id(uint8_t* id128) {
  static_assert(sizeof(ncclUniqueId) == PGBP_COMM_ID_BYTES, "pgbp.h: PGBP_COMM_ID_BYTES");
  if (!id128) return PGBP_ERR_INVALID;
  Rccl& r = rccl();
  if (!r.err.empty()) {
    g_comm_create_error = r.err;
    return PGBP_ERR_NO_DEVICE;
  }
  ncclUniqueId id;
  const ncclResult_t rc = r.get_unique_id(&id);
  if (rc != ncclSuccess) {
    g_comm_create_error = std::string("ncclGetUniqueId: ") + r.error_string(rc);
    return PGBP_ERR_HIP;
  }
  std::memcpy(id128, &id, sizeof(id));
  return PGBP_OK;
}

int pgbp_comm_precheck(int32_t device) {
  Rccl& r = rccl();
  if (!r.err.empty()) {
    g_comm_create_error = r.err;
    return PGBP_ERR_NO_DEVICE;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    g_comm_create_error = "pgbp_comm_precheck: no device " + std::to_string(device) + " (" + std::to_string(n) + " visible)";
    return PGBP_ERR_NO_DEVICE;
  }
  return PGBP_OK;
}

int pgbp_comm_unpack_slots(const double* recv, int32_t n_ranks, int32_t slot_sites, double* norm_all, int32_t* info_all,
                           int32_t* all_succ, int32_t* all_iscal) {
  if (!recv || n_ranks < 1 || slot_sites < 1 || !norm_all) return PGBP_ERR_INVALID;
  const int64_t slot = 2 * (int64_t)slot_sites + 2;
  int succ = 1, iscal = 1;
  for (int r = 0; r < n_ranks; ++r) {
    const double* s = recv + (size_t)r * slot;
    for (int i = 0; i < slot_sites; ++i) {
      norm_all[(size_t)r * slot_sites + i] = s[i];
      if (info_all) info_all[(size_t)r * slot_sites + i] = (int32_t)s[slot_sites + i];
    }
    succ = std::min(succ, (int)s[2 * slot_sites]);
    iscal = std::min(iscal, (int)s[2 * slot_sites + 1]);
  }
  if (all_succ) *all_succ = succ;
  if (all_iscal) *all_iscal = iscal;
  return PGBP_OK;
}

int pgbp_comm_create(const uint8_t* id128, int32_t n_ranks, int32_t rank, int32_t device, pgbp_comm** out) {
  if (!out) return PGBP_ERR_INVALID;
  *out = nullptr;
  if (!id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
    g_comm_create_error = "pgbp_comm_create: bad rank / n_ranks / id";
    return PGBP_ERR_INVALID;
  }
  Rccl& r = rccl();
  if (!r.err.empty()) {
    g_comm_create_error = r.err;
    return PGBP_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_comm_create_error = "hipSetDevice(" + std::to_string(device) + ") failed";
    return PGBP_ERR_NO_DEVICE;
  }
  std::unique_ptr<pgbp_comm> c(new pgbp_comm());
  c->n_ranks = n_ranks;
  c->rank = rank;
  c->device = device;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  const ncclResult_t rc = r.comm_init_rank(&c->comm, n_ranks, id, rank);
  if (rc != ncclSuccess) {
    g_comm_create_error = std::string("ncclCommInitRank: ") + r.error_string(rc);
    return PGBP_ERR_HIP;
  }
  *out = c.release();
  return PGBP_OK;
}

void pgbp_comm_destroy(pgbp_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->comm) (void)rccl().comm_destroy(c->comm);
  delete c;
}

int pgbp_comm_gather_loglik(pgbp_comm* c, pgbp_engine* e, int32_t slot_sites, double* norm_all, int32_t* info_all,
                            int32_t* all_succ, int32_t* all_iscal) {
  if (!c || !e || slot_sites < 1 || !norm_all) return PGBP_ERR_INVALID;
  if (pgbp::engine_device(e) != c->device) {
    c->err = "the engine lives on device " + std::to_string(pgbp::engine_device(e)) + ", the communicator on " +
             std::to_string(c->device);
    return PGBP_ERR_INVALID;
  }
  double* d_slot = nullptr;
  hipStream_t st = nullptr;
  int rc = pgbp::engine_pack_gather_slot(e, slot_sites, &d_slot, &st, nullptr);
  if (rc) {
    c->err = pgbp_last_error(e);
    return rc;
  }
  const int64_t slot = 2 * (int64_t)slot_sites + 2, total = slot * c->n_ranks;
  if (total > c->recv_cap) {
    if (c->d_recv) (void)hipFree(c->d_recv);
    c->d_recv = nullptr;
    c->recv_cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(&c->d_recv), sizeof(double) * (size_t)total) != hipSuccess) {
      c->err = "hipMalloc of the gather buffer failed";
      return PGBP_ERR_HIP;
    }
    c->recv_cap = total;
  }
  // THE collective of the site-sharded path: one all-gather on the engine's stream, behind the kernels that produced
  // the log-likelihoods (no host synchronisation in between)
  const ncclResult_t nrc = rccl().all_gather(d_slot, c->d_recv, (size_t)slot, ncclFloat64, c->comm, st);
  if (nrc != ncclSuccess) {
    c->err = std::string("ncclAllGather: ") + rccl().error_string(nrc);
    return PGBP_ERR_HIP;
  }
  c->h_recv.resize((size_t)total);
  if (hipMemcpyAsync(c->h_recv.data(), c->d_recv, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    c->err = "copy of the gathered log-likelihoods failed";
    return PGBP_ERR_HIP;
  }
  const int urc = pgbp_comm_unpack_slots(c->h_recv.data(), c->n_ranks, slot_sites, norm_all, info_all, all_succ, all_iscal);
  if (urc) return urc;
  return PGBP_OK;
}

// The exchange step of a cluster graph CUT across ranks (sharding.py: NetworkCut): rank r contributes
// the records of the beliefs lists[list_off[r] .. list_off[r + 1]) of `site` -- gathered on ITS device into the send slot --,
// ONE ncclAllGather (slot = the largest contribution) puts every rank's slot on every rank, and each rank scatters the other
// ranks' records into its own engine.  No host copy of the payload; the engine's stream orders everything behind the
// traversals that produced the records.  include_self != 0: a rank also scatters its own slot back (the single-rank test).
int pgbp_comm_exchange_beliefs(pgbp_comm* c, pgbp_engine* e, int32_t site, const int32_t* list_off, const int32_t* lists,
                               int32_t include_self) {
  if (!c || !e || !list_off || (list_off[c->n_ranks] > 0 && !lists)) return PGBP_ERR_INVALID;
  if (pgbp::engine_device(e) != c->device) {
    c->err = "the engine lives on device " + std::to_string(pgbp::engine_device(e)) + ", the communicator on " +
             std::to_string(c->device);
    return PGBP_ERR_INVALID;
  }
  int64_t slot = 0;
  std::vector<int64_t> size(c->n_ranks, 0);
  for (int r = 0; r < c->n_ranks; ++r) {
    const int32_t n = list_off[r + 1] - list_off[r];
    if (n < 0) return PGBP_ERR_INVALID;
    size[r] = n > 0 ? pgbp_packed_beliefs_size(e, n, lists + list_off[r]) : 0;
    if (size[r] < 0) {
      c->err = "pgbp_comm_exchange_beliefs: belief index out of range";
      return PGBP_ERR_INVALID;
    }
    slot = std::max(slot, size[r]);
  }
  if (slot == 0) return PGBP_OK;
  (void)hipSetDevice(c->device);
  auto grow = [&](double** buf, int64_t* cap, int64_t need) {
    if (need <= *cap) return true;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(buf), sizeof(double) * (size_t)need) != hipSuccess) return false;
    *cap = need;
    return true;
  };
  if (!grow(&c->d_send, &c->send_cap, slot) || !grow(&c->d_recv, &c->recv_cap, slot * c->n_ranks)) {
    c->err = "hipMalloc of the exchange buffers failed";
    return PGBP_ERR_HIP;
  }
  hipStream_t st = nullptr;
  int rc = pgbp::engine_pack_records_device(e, site, list_off[c->rank + 1] - list_off[c->rank], lists + list_off[c->rank],
                                            c->d_send, 1, &st, nullptr);
  if (rc) {
    c->err = pgbp_last_error(e);
    return rc;
  }
  const ncclResult_t nrc = rccl().all_gather(c->d_send, c->d_recv, (size_t)slot, ncclFloat64, c->comm, st);
  if (nrc != ncclSuccess) {
    c->err = std::string("ncclAllGather: ") + rccl().error_string(nrc);
    return PGBP_ERR_HIP;
  }
  for (int r = 0; r < c->n_ranks; ++r) {
    if ((r == c->rank && !include_self) || size[r] == 0) continue;
    rc = pgbp::engine_pack_records_device(e, site, list_off[r + 1] - list_off[r], lists + list_off[r],
                                          c->d_recv + (int64_t)r * slot, 0, nullptr, nullptr);
    if (rc) {
      c->err = pgbp_last_error(e);
      return rc;
    }
  }
  return PGBP_OK;
}

}  // extern "C"
