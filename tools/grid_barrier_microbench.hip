// What does it cost to keep a dependent step inside ONE launch instead of cutting the launch there?  (gfx950, MI355X)
//
// A phase = every workgroup publishes 4 KB (the size of a pass's receiver blocks) and then reads the 4 KB another
// workgroup published in the phase before, checking every word.  G workgroups of 1024 threads, one per CU -- the geometry
// of bp_loop16.  Variants of the seam between two phases:
//   boundary    : one kernel launch per phase (what the engine's level launches do)
//   fence_flat  : one counter; lane 0: agent release fence -> add -> poll (sc1) -> agent acquire fence; plain loads / stores
//   fence_xcd   : the same, hierarchical: per-XCD counter, the XCD's last arriver adds to the top counter and publishes the
//                 generation word of its XCD
//   sc1_flat    : write-through (sc1) stores, drained; one counter; sc1 loads; no fence
//   sc1_xcd     : the same with the hierarchical counter
//   team_xcd    : 8 independent teams (the workgroups that find themselves on one XCD: HW_REG_XCC_ID); a team only reads
//                 what its own members wrote: plain stores (they stay in the XCD's L2), one counter per team, sc1 loads
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier tools/grid_barrier_microbench.hip ; run: ./grid_barrier [G] [phases]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 1024, kPayload = 512;   // doubles per workgroup and phase
constexpr unsigned kSpinMax = 1u << 18;

typedef unsigned int u32;
struct Sync {
  u32 top;        u32 pad0[31];
  u32 xcd[8][32];            // one counter per XCD, each on a line of its own
  u32 gen[8][32];            // generation word per XCD
  u32 team_n[8][32];         // members registered per XCD
  u32 reg_total;  u32 pad1[31];
  u32 timeout;    u32 pad2[31];
  u32 errors;     u32 pad3[31];
};

__device__ __forceinline__ u32 ld_sc1(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ldd_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void std_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 add_agent(u32* p, u32 v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xcc_id() {
  u32 v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return (int)(v & 15);
}
__device__ __forceinline__ bool wait_ge(const u32* p, u32 target, Sync* s) {
  for (u32 spins = 0; spins < kSpinMax; ++spins) {
    if ((int)(ld_sc1(p) - target) >= 0) return true;
    if ((spins & 1023u) == 1023u && ld_sc1(&s->timeout) != 0u) return false;   // somebody gave up: so do we
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(&s->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

enum { kBoundary = 0, kFenceFlat, kFenceXcd, kSc1Flat, kSc1Xcd, kTeamXcd, kNone };

// one phase's publish (stores) of workgroup `me` and the check of what `src` published the phase before
template <bool SC1ST>
__device__ __forceinline__ void publish(double* buf, int G, int me, int phase) {
  double* out = buf + ((size_t)(phase & 1) * G + me) * kPayload;
  if (threadIdx.x < kPayload) {
    const double v = (double)phase * 4096.0 + (double)me * 2.0 + (double)threadIdx.x * 1e-3;
    if (SC1ST) std_sc1(out + threadIdx.x, v); else out[threadIdx.x] = v;
  }
}
template <bool SC1LD>
__device__ __forceinline__ void check(const double* buf, int G, int src, int phase, Sync* s) {
  if (phase == 0) return;
  const double* in = buf + ((size_t)((phase - 1) & 1) * G + src) * kPayload;
  if (threadIdx.x < kPayload) {
    const double want = (double)(phase - 1) * 4096.0 + (double)src * 2.0 + (double)threadIdx.x * 1e-3;
    const double got = SC1LD ? ldd_sc1(in + threadIdx.x) : in[threadIdx.x];
    if (got != want) atomicAdd(&s->errors, 1u);
  }
}

__global__ __launch_bounds__(kThreads) void phase_kernel(double* buf, int G, int phase, Sync* s) {
  const int me = blockIdx.x;
  check<false>(buf, G, (me + 37) % G, phase, s);
  publish<false>(buf, G, me, phase);
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void persistent(double* buf, int G, int phases, Sync* s, u32* roster) {
  __shared__ u32 sh[4];
  const int me = blockIdx.x;
  const int xcc = xcc_id() & 7;
  constexpr bool SC1 = MODE == kSc1Flat || MODE == kSc1Xcd;
  constexpr bool SC1LD = SC1 || MODE == kTeamXcd;
  u32 slot = 0, M = 0;
  if (threadIdx.x == 0) sh[2] = 0u;
  __syncthreads();
  if (MODE == kFenceXcd || MODE == kSc1Xcd || MODE == kTeamXcd) {
    // registration: who shares an XCD with whom (once per launch)
    if (threadIdx.x == 0) {
      slot = add_agent(&s->team_n[xcc][0], 1u);
      roster[xcc * 1024 + slot] = (u32)me;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      add_agent(&s->reg_total, 1u);
      wait_ge(&s->reg_total, (u32)G, s);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      sh[0] = slot;
      sh[1] = ld_sc1(&s->team_n[xcc][0]);
    }
    __syncthreads();
    slot = sh[0];
    M = sh[1];
  }
  int src = (me + 37) % G, mine = me;
  if (MODE == kTeamXcd) {   // the neighbour inside the team; payload slots indexed by (xcc, slot)
    mine = xcc * 64 + (int)slot;
    src = xcc * 64 + (int)((slot + 1) % M);
  }
  const int GP = MODE == kTeamXcd ? 8 * 64 : G;
  for (int phase = 0; phase < phases; ++phase) {
    check<SC1LD>(buf, GP, src, phase, s);
    publish<SC1>(buf, GP, mine, phase);
    if (MODE == kNone) continue;
    // ---- the seam
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its loads and stores are complete
    __syncthreads();
    if (threadIdx.x == 0) {
      const u32 ph1 = (u32)phase + 1u;
      if (MODE == kFenceFlat || MODE == kFenceXcd) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (MODE == kFenceFlat || MODE == kSc1Flat) {
        add_agent(&s->top, 1u);
        if (!wait_ge(&s->top, ph1 * (u32)G, s)) sh[2] = 1u;
      } else if (MODE == kTeamXcd) {
        add_agent(&s->xcd[xcc][0], 1u);
        if (!wait_ge(&s->xcd[xcc][0], ph1 * M, s)) sh[2] = 1u;
      } else {
        const u32 old = add_agent(&s->xcd[xcc][0], 1u);
        if (old + 1u == ph1 * M) {   // the XCD's last arriver
          add_agent(&s->top, 1u);
          u32 nx = 0;
          for (int x = 0; x < 8; ++x) nx += ld_sc1(&s->team_n[x][0]) != 0u;
          if (!wait_ge(&s->top, ph1 * nx, s)) sh[2] = 1u;
          __hip_atomic_store(&s->gen[xcc][0], ph1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          if (!wait_ge(&s->gen[xcc][0], ph1, s)) sh[2] = 1u;
        }
      }
      if (MODE == kFenceFlat || MODE == kFenceXcd) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (sh[2] != 0u) break;   // a wait of this workgroup gave up
  }
}

int main(int argc, char** argv) {
  int G = argc > 1 ? atoi(argv[1]) : 0, phases = argc > 2 ? atoi(argv[2]) : 200;
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  if (G <= 0) G = prop.multiProcessorCount;
  printf("device %s, %d CUs; G = %d workgroups x %d threads, %d phases, payload %d B per workgroup and phase\n", prop.name,
         prop.multiProcessorCount, G, kThreads, phases, (int)(kPayload * sizeof(double)));
  double* buf;
  Sync* s;
  u32* roster;
  CHK(hipMalloc(&buf, sizeof(double) * 2 * (size_t)(G > 512 ? G : 512) * kPayload));
  CHK(hipMalloc(&s, sizeof(Sync)));
  CHK(hipMalloc(&roster, sizeof(u32) * 8 * 1024));
  hipStream_t st;
  CHK(hipStreamCreate(&st));
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  auto report = [&](const char* name, float ms, float base_ms) {
    Sync h;
    CHK(hipMemcpy(&h, s, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-12s %8.3f us per phase   (seam alone: %7.3f us)   errors %u  timeout %u  teams", name, 1e3f * ms / phases,
           1e3f * (ms - base_ms) / phases, h.errors, h.timeout);
    for (int x = 0; x < 8; ++x) printf(" %u", h.team_n[x][0]);
    printf("\n");
  };
  float base = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    // launches
    CHK(hipMemsetAsync(s, 0, sizeof(Sync), st));
    CHK(hipEventRecord(a, st));
    for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(kThreads), 0, st, buf, G, p, s);
    CHK(hipEventRecord(b, st));
    CHK(hipEventSynchronize(b));
    float ms_b;
    CHK(hipEventElapsedTime(&ms_b, a, b));
#define RUN(MODE, NAME)                                                                                              \
  {                                                                                                                  \
    CHK(hipMemsetAsync(s, 0, sizeof(Sync), st));                                                                     \
    CHK(hipEventRecord(a, st));                                                                                      \
    hipLaunchKernelGGL((persistent<MODE>), dim3(G), dim3(kThreads), 0, st, buf, G, phases, s, roster);               \
    CHK(hipEventRecord(b, st));                                                                                      \
    CHK(hipEventSynchronize(b));                                                                                     \
    float ms;                                                                                                        \
    CHK(hipEventElapsedTime(&ms, a, b));                                                                             \
    if (MODE == kNone) base = ms;                                                                                    \
    report(NAME, ms, base);                                                                                          \
  }
    RUN(kNone, "no seam");
    report("boundary", ms_b, base);
    RUN(kFenceFlat, "fence_flat");
    RUN(kFenceXcd, "fence_xcd");
    RUN(kSc1Flat, "sc1_flat");
    RUN(kSc1Xcd, "sc1_xcd");
    RUN(kTeamXcd, "team_xcd");
    printf("\n");
  }
  return 0;
}
