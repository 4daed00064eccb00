// Issue rate of INDEPENDENT fp64 / fp32 FMAs on gfx950, per wavefront and per SIMD: 16 accumulators per lane, 64 FMAs per
// loop iteration, 1 / 2 / 4 wavefronts per SIMD (workgroups of 256 / 512 / 1024 threads, one per CU).  Prints shader clocks
// per FMA instruction as seen by one wavefront, and per SIMD (= that / wavefronts per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o build/exp/fp64_issue tools/fp64_issue_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ void issue(T* out, unsigned long long* t, int n) {
  T acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = out[threadIdx.x] + (T)i;
  const T m = (T)1.0000001, c = (T)0.5;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], m, c);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  T s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}

template <typename T>
void run(const char* name, int threads, int wgs) {
  T* out;
  unsigned long long* t;
  hipMalloc(&out, sizeof(T) * 1024 * 1024);
  hipMalloc(&t, 8);
  hipMemset(out, 0, sizeof(T) * 1024 * 1024);
  const int n = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(issue<T>, dim3(wgs), dim3(threads), 0, 0, out, t, n);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(issue<T>, dim3(wgs), dim3(threads), 0, 0, out, t, n);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h = 0;
  hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  const double per = (double)h / ((double)n * 64.0);
  printf("%s  %4d threads/workgroup (%d waves/SIMD), %4d workgroups: %.2f clocks per FMA per wavefront, %.2f per SIMD\n", name,
         threads, threads / 256, wgs, per, per / (threads / 256));
  printf("      wall clock %.1f us: %.2f TFLOP/s, memtime ticks at %.0f MHz if the kernel is its loop\n", ms * 1e3,
         2.0 * n * 64.0 * threads * wgs / (ms * 1e-3) * 1e-12, (double)h / (ms * 1e3));
  hipFree(out);
  hipFree(t);
}

int main() {
  for (int threads : {64, 256, 512, 1024}) {
    run<double>("fp64", threads, 1);
    run<float>("fp32", threads, 1);
  }
  run<double>("fp64", 1024, 256);
  run<double>("fp64", 1024, 512);
  return 0;
}
