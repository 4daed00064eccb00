// Should the kept block of the elimination (S <- S - sum_R X_R D_R^-1 X_R', a 16 x 16 x 16 product) go to the matrix pipe?
// Compares, per message and wavefront, on gfx950:
//   A. what the message kernel does today for the kept block: per round D^-1 x for two kept columns (8 fp64 ops) and the
//      rank-2 update of the lane's 2 x 2 patch (8 FMAs): 8 rounds x 16 = 128 vector fp64 instructions;
//   B. the same as TWO products on v_mfma_f64_16x16x4_f64 after the last round: T = X E (E = the 2 x 2 blocks D_R^-1 on the
//      diagonal), S -= T X': operands gathered from the per-round LDS strips with per-lane addresses, T turned from the
//      C/D layout into the A layout through LDS, S turned from the C/D layout into the kernel's 2 x 2-per-lane layout
//      through LDS: 8 dependent-in-fours MFMAs + 28 LDS accesses + their address arithmetic.
// Reports shader clocks per message for ONE wavefront (latency: the narrow levels) and wall-clock throughput with 4
// wavefronts per SIMD on every CU (the wide levels), each for A and B.
//   hipcc --offload-arch=gfx950 -O3 -o build/exp/mfma_schur tools/mfma_f64_schur_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int kStride = 10, kStrip = 84;   // the kernel's strip geometry (pgbp_fast_dev.hpp)

__global__ __launch_bounds__(256) void kept_valu(double* out, unsigned long long* t, int reps) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, a = lane & 7, b = lane >> 3;
  double* strips = lds + wave * 8 * kStrip;
  for (int i = lane; i < 8 * kStrip; i += 64) strips[i] = 1.0 / (3.0 + i);
  __syncthreads();
  double w[2][2] = {{1.0 + lane, 0.5}, {0.5, 2.0 + lane}};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    asm volatile("" ::: "memory");   // (the strips of another message: nothing of the round is loop-invariant)
#pragma unroll
    for (int R = 0; R < 8; ++R) {
      const double* strip = strips + R * kStrip;
      const double2 xr0 = *reinterpret_cast<const double2*>(strip + a * kStride + 2);
      const double2 xr1 = *reinterpret_cast<const double2*>(strip + a * kStride + 6);
      const double2 xc0 = *reinterpret_cast<const double2*>(strip + b * kStride + 2);
      const double2 xc1 = *reinterpret_cast<const double2*>(strip + b * kStride + 6);
      const double e00 = strip[80], e01 = strip[81], e11 = strip[82];
      const double y00 = fma(e00, xc0.x, e01 * xc1.x), y10 = fma(e01, xc0.x, e11 * xc1.x);
      const double y01 = fma(e00, xc0.y, e01 * xc1.y), y11 = fma(e01, xc0.y, e11 * xc1.y);
      w[0][0] = fma(-xr0.x, y00, fma(-xr1.x, y10, w[0][0]));
      w[1][0] = fma(-xr0.y, y00, fma(-xr1.y, y10, w[1][0]));
      w[0][1] = fma(-xr0.x, y01, fma(-xr1.x, y11, w[0][1]));
      w[1][1] = fma(-xr0.y, y01, fma(-xr1.y, y11, w[1][1]));
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = w[0][0] + w[1][0] + w[0][1] + w[1][1];
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}

__global__ __launch_bounds__(256) void kept_mfma(double* out, unsigned long long* t, int reps) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, a = lane & 7, b = lane >> 3;
  double* strips = lds + wave * (8 * kStrip + 2 * 256);
  double* tbuf = strips + 8 * kStrip;        // T in the C/D layout -> A layout
  double* sbuf = tbuf + 256;                  // S in the C/D layout -> 2 x 2 per lane
  for (int i = lane; i < 8 * kStrip; i += 64) strips[i] = 1.0 / (3.0 + i);
  __syncthreads();
  const int i16 = lane & 15, k4 = lane >> 4;
  double w[2][2] = {{1.0 + lane, 0.5}, {0.5, 2.0 + lane}};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    asm volatile("" ::: "memory");
    // T = X E: A = X[:, 4kc .. 4kc+3] (kept row i16, pivot column 4kc + k4), B = E[4kc + k4][j = i16]
    d4 T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int c = 4 * kc + k4, R = c >> 1, ik = c & 1;
      const double xa = strips[R * kStrip + (i16 >> 1) * kStride + ik * 4 + 2 + (i16 & 1)];
      const double* e = strips + R * kStrip + 80;
      const int jj = i16;   // column of E
      const double eb = (jj >> 1) == R ? ((ik == 0) ? ((jj & 1) ? e[1] : e[0]) : ((jj & 1) ? e[2] : e[1])) : 0.0;
      T = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, eb, T, 0, 0, 0);
    }
    // T: C/D layout (col = lane & 15, row = (lane >> 4) + 4 reg) -> LDS -> A layout of the second product
#pragma unroll
    for (int q = 0; q < 4; ++q) tbuf[(k4 + 4 * q) * 16 + i16] = T[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    d4 Sacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int c = 4 * kc + k4, R = c >> 1, ik = c & 1;
      const double ta = tbuf[i16 * 16 + c];                                                        // T[i16][c]
      const double xb = strips[R * kStrip + (i16 >> 1) * kStride + ik * 4 + 2 + (i16 & 1)];      // X'[c][j = i16]
      Sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, xb, Sacc, 0, 0, 0);
    }
    // S - T X' from the C/D layout into the kernel's 2 x 2 patch per lane (rows 2a, 2a+1; columns 2b, 2b+1)
#pragma unroll
    for (int q = 0; q < 4; ++q) sbuf[(k4 + 4 * q) * 16 + i16] = Sacc[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double2 r0 = *reinterpret_cast<const double2*>(sbuf + (2 * a) * 16 + 2 * b);
    const double2 r1 = *reinterpret_cast<const double2*>(sbuf + (2 * a + 1) * 16 + 2 * b);
    w[0][0] -= r0.x; w[0][1] -= r0.y; w[1][0] -= r1.x; w[1][1] -= r1.y;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = w[0][0] + w[1][0] + w[0][1] + w[1][1];
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}

template <class K>
void run(const char* name, K kernel, int threads, int wgs, size_t lds_bytes) {
  double* out;
  unsigned long long* t;
  hipMalloc(&out, sizeof(double) * 1024 * 2048);
  hipMalloc(&t, 8);
  const int reps = 500;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(wgs), dim3(threads), lds_bytes, 0, out, t, reps);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kernel, dim3(wgs), dim3(threads), lds_bytes, 0, out, t, reps);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h = 0;
  hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  const double msgs = (double)wgs * (threads / 64) * reps;
  const double simds = wgs >= 256 ? 1024.0 : (double)wgs * (threads / 64 > 4 ? 4 : threads / 64);
  printf("%-28s %4d threads x %5d workgroups: %7.0f clocks per message (first wavefront), %8.1f ns per message per SIMD (wall)\n",
         name, threads, wgs, (double)h / reps, ms * 1e6 / (msgs / simds));
  hipFree(out);
  hipFree(t);
}

int main() {
  const size_t la = sizeof(double) * 4 * 8 * kStrip, lb = sizeof(double) * 4 * (8 * kStrip + 512);
  run("kept block, vector fp64", kept_valu, 64, 1, la);
  run("kept block, 2 x 4 MFMA f64", kept_mfma, 64, 1, lb);
  run("kept block, vector fp64", kept_valu, 256, 1024, la);
  run("kept block, 2 x 4 MFMA f64", kept_mfma, 256, 1024, lb);
  return 0;
}
