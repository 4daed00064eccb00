// Probe (round 4): which SIMD does wavefront w of a 1 024-thread workgroup run on?  (the helper assignment of the loop
// launches -- pgbp_plan.cpp: assign_helpers -- assumes w mod 4).  Prints HW_ID's SIMD_ID / CU_ID per wavefront of 3 workgroups.
//   hipcc --offload-arch=gfx950 -O2 -o build/exp/simd_probe tools/simd_probe.hip && build/exp/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 3 * 16 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(3), dim3(1024), 0, 0, d);
  unsigned h[48];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 3; ++b) {
    printf("workgroup %d: simd of wavefront 0..15:", b);
    for (int w = 0; w < 16; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
    printf("   (cu %u, wave slots:", (h[b * 16] >> 8) & 15);
    for (int w = 0; w < 16; ++w) printf(" %u", h[b * 16 + w] & 15);
    printf(")\n");
  }
  return 0;
}
