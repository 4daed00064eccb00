// Lone-wave latency of dependent fp64 operations on gfx950 (one wavefront on an otherwise idle chip: the regime of
// the narrow levels of a schedule).  hipcc --offload-arch=gfx950 -O3 -o build/exp/fp64_latency tools/fp64_latency_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void chain(double* out, unsigned long long* t, int n, int mode) {
  double x = out[threadIdx.x];
  double y = x + 1.0;
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  if (mode == 0) {
    for (int i = 0; i < n; ++i) x = fma(x, 1.0000001, 0.5);                       // n dependent FMAs
  } else if (mode == 1) {
    for (int i = 0; i < n; ++i) x = __builtin_amdgcn_rcp(x) + 1.5;                 // rcp + add
  } else if (mode == 2) {
    for (int i = 0; i < n; ++i) {                                                  // rcp + 2 Newton steps + look-ahead: 7 ops
      double rd = __builtin_amdgcn_rcp(x);
      rd = fma(fma(-x, rd, 1.0), rd, rd);
      rd = fma(fma(-x, rd, 1.0), rd, rd);
      x = y - (1.25 * rd) * 0.75;
    }
  } else if (mode == 3) {
    for (int i = 0; i < n; ++i) { x = fma(x, 1.0000001, 0.5); y = fma(y, 1.0000002, 0.25); }   // 2 independent chains
  } else {
    float xf = (float)x;
    for (int i = 0; i < n; ++i) xf = fmaf(xf, 1.0000001f, 0.5f);                   // fp32 dependent chain
    x = xf;
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x] = x + y;
  if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}

int main() {
  double* d; unsigned long long* t;
  hipMalloc(&d, 64 * sizeof(double)); hipMalloc(&t, 16);
  double h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.0 + i * 0.01;
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  const char* names[] = {"dependent fp64 fma", "rcp_f64 + add", "rcp + 2 Newton + update (7 ops)", "2 independent fp64 fma chains", "dependent fp32 fma"};
  const int n = 4096;
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, d, t, n, mode);
    hipDeviceSynchronize();
    unsigned long long ht[2]; hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost);
    printf("%-34s: %7.1f ns per iteration (%6.1f s_memtime ticks; realtime 100 MHz)\n", names[mode], ht[1] * 10.0 / n, (double)ht[0] / n);
  }
  return 0;
}
