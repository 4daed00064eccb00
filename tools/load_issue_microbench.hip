// Microbenchmark (round 3): what does ISSUING a batch of independent 8-byte-per-lane global loads cost one wavefront that
// is alone on its CU?  (the operand phase of a small message of the wave-per-task kernels: 12 - 14 loads in 1 400 - 2 000
// clocks by tools/stamp_generic.py).  Variants: lanes active (8 / 16 / 64), address form (saddr + voffset / 64-bit vaddr),
// loads per batch.  Prints shader clocks from before the first load to after the last is issued, and to data arrival.
//   hipcc --offload-arch=gfx950 -O3 -o load_issue tools/load_issue_microbench.hip && ./load_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N, int LANES, bool VADDR>
__global__ void k(const double* __restrict__ base, const int* __restrict__ idx, double* out, unsigned int* clk, int stride) {
  const int lane = threadIdx.x;
  double v[N];
  int off[N];
#pragma unroll
  for (int j = 0; j < N; ++j) off[j] = idx[j] * stride + lane;   // column j, row = lane: contiguous over the lanes
  const double* p[N];
#pragma unroll
  for (int j = 0; j < N; ++j) p[j] = base + off[j];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane < LANES) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if constexpr (VADDR) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v[j]) : "v"(p[j]) : "memory");
      else asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v[j]) : "v"(off[j] * 8), "s"(base) : "memory");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  double s = 0;
  if (lane < LANES)
#pragma unroll
    for (int j = 0; j < N; ++j) s += v[j];
  out[lane] = s;
  if (lane == 0) { clk[0] = (unsigned int)(t1 - t0); clk[1] = (unsigned int)(t2 - t0); }
}

// the same for STORES: N independent 8-byte-per-lane global stores, LANES active
template <int N, int LANES>
__global__ void ks(double* __restrict__ base, const int* __restrict__ idx, unsigned int* clk, int stride) {
  const int lane = threadIdx.x;
  double* p[N];
#pragma unroll
  for (int j = 0; j < N; ++j) p[j] = base + idx[j] * stride + lane;
  const double v = (double)lane;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane < LANES) {
#pragma unroll
    for (int j = 0; j < N; ++j) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p[j]), "v"(v) : "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) { clk[0] = (unsigned int)(t1 - t0); clk[1] = (unsigned int)(t2 - t0); }
}
template <int N, int LANES>
void run_store(double* d, const int* di, unsigned int* c) {
  unsigned int h[2], best[2] = {~0u, ~0u};
  for (int r = 0; r < 20; ++r) {
    hipLaunchKernelGGL((ks<N, LANES>), dim3(1), dim3(64), 0, 0, d, di, c, 16);
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    if (r >= 5) { best[0] = h[0] < best[0] ? h[0] : best[0]; best[1] = h[1] < best[1] ? h[1] : best[1]; }
  }
  printf("%-28s stores %2d lanes %2d: issued after %5u clocks (%4u per store), acknowledged after %5u\n", "64-bit vaddr", N, LANES, best[0], best[0] / N, best[1]);
}

template <int N, int LANES, bool VADDR>
void run(const double* d, const int* di, double* o, unsigned int* c, const char* name) {
  unsigned int h[2], best[2] = {~0u, ~0u};
  for (int r = 0; r < 20; ++r) {
    hipLaunchKernelGGL((k<N, LANES, VADDR>), dim3(1), dim3(64), 0, 0, d, di, o, c, 16);
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    if (r >= 5) { best[0] = h[0] < best[0] ? h[0] : best[0]; best[1] = h[1] < best[1] ? h[1] : best[1]; }
  }
  printf("%-28s loads %2d lanes %2d: issued after %5u clocks (%4u per load), data after %5u\n", name, N, LANES, best[0], best[0] / N, best[1]);
}

int main() {
  double *d, *o; int* di; unsigned int* c;
  hipMalloc(&d, 1 << 20); hipMalloc(&o, 4096); hipMalloc(&di, 256); hipMalloc(&c, 64);
  hipMemset(d, 0, 1 << 20);
  std::vector<int> idx(64); for (int i = 0; i < 64; ++i) idx[i] = (i * 7) % 61;
  hipMemcpy(di, idx.data(), 256, hipMemcpyHostToDevice);
  run<1, 8, false>(d, di, o, c, "saddr+voffset");
  run<4, 8, false>(d, di, o, c, "saddr+voffset");
  run<14, 8, false>(d, di, o, c, "saddr+voffset");
  run<14, 16, false>(d, di, o, c, "saddr+voffset");
  run<14, 64, false>(d, di, o, c, "saddr+voffset");
  run<14, 8, true>(d, di, o, c, "64-bit vaddr");
  run<14, 64, true>(d, di, o, c, "64-bit vaddr");
  run<28, 8, true>(d, di, o, c, "64-bit vaddr");
  run_store<1, 8>(d, di, c);
  run_store<7, 8>(d, di, c);
  run_store<19, 8>(d, di, c);
  run_store<19, 64>(d, di, c);
  run_store<38, 8>(d, di, c);
  return 0;
}
