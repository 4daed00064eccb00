// Diagnostic microbenchmark (not part of the product): cycles of ONE wavefront's marginalisation chain.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/elim tools/elim_microbench.hip && /tmp/elim
#include "../phylogaussianbeliefprop.jl_amd/csrc/pgbp_fast.hip"

#include <cstdio>
#include <vector>

namespace pgbp {

// 16 rank-1 steps, column broadcast by ds_bpermute (the first version of the kernel)
template <int K, bool FASTDIV>
__device__ __forceinline__ int eliminate1(Frag& f, const int a, const int b, double& mant, int& expo, double& quad) {
  if constexpr (K == P) {
    return 0;
  } else {
    constexpr int kk = K >> 1, ik = K & 1;
    const int src_r = kk * 8 + a, src_c = kk * 8 + b;
    double xr[4], xc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xr[i] = __shfl(f.w[i][ik], src_r);
      xc[i] = __shfl(f.w[i][ik], src_c);
    }
    const double d = readlane_f64(f.w[ik][ik], kk * 8 + kk);
    const double hk = readlane_f64(f.h[ik], kk);
    if (!(d > 0.0)) return K + 1;
    double rd;
    if constexpr (FASTDIV) {
      rd = __builtin_amdgcn_rcp(d);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
    } else {
      rd = 1.0 / d;
    }
    int e;
    mant *= frexp(d, &e);
    expo += e;
    quad = fma(hk * hk, rd, quad);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(i < 2 && j >= 2)) f.w[i][j] = fma(-(xr[i] * xc[j]), rd, f.w[i][j]);
      f.h[i] = fma(-(xr[i] * hk), rd, f.h[i]);
    }
    return eliminate1<K + 1, FASTDIV>(f, a, b, mant, expo, quad);
  }
}

template <int VARIANT>
__global__ __launch_bounds__(64) void micro(const double* __restrict__ J, double* __restrict__ out,
                                            long long* __restrict__ cycles, int reps) {
  extern __shared__ double lds_[];
  const int lane = threadIdx.x, a = lane & 7, b = lane >> 3;
  Frag f0;
  const int r0 = 2 * a, r1 = 2 * a + 16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cl = 2 * b + (j & 1) + (j >> 1) * 16;
    f0.w[0][j] = J[r0 + 32 * cl]; f0.w[1][j] = J[r0 + 1 + 32 * cl];
    f0.w[2][j] = J[r1 + 32 * cl]; f0.w[3][j] = J[r1 + 1 + 32 * cl];
  }
  f0.h[0] = 0.1 * r0; f0.h[1] = 0.1 * (r0 + 1); f0.h[2] = 0.1 * r1; f0.h[3] = 0.1 * (r1 + 1);
  double acc = 0.0;
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  for (int r = 0; r < reps; ++r) {
    Frag f = f0;
    f.w[2][2] += acc * 1e-300;  // serialise repetitions
    double mant = 1.0, quad = 0.0;
    int expo = 0, info;
    if constexpr (VARIANT == 0) info = eliminate1<0, false>(f, a, b, mant, expo, quad);
    else if constexpr (VARIANT == 1) info = eliminate1<0, true>(f, a, b, mant, expo, quad);
    else info = eliminate2<0>(f, a, b, lds_, mant, expo, quad);
    acc += f.w[2][2] + f.w[3][3] + f.h[2] + quad + mant + expo + info;
  }
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  out[lane] = acc;
  if (lane == 0) { cycles[0] = c1 - c0; cycles[1] = t1 - t0; }
}

}  // namespace pgbp

int main() {
  using namespace pgbp;
  std::vector<double> J(32 * 32);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) J[i + 32 * j] = (i == j ? 40.0 : 0.0) + 1.0 / (1.0 + abs(i - j));
  double *dJ, *dout;
  long long* dc;
  hipMalloc(&dJ, sizeof(double) * 1024);
  hipMalloc(&dout, sizeof(double) * 64);
  hipMalloc(&dc, sizeof(long long) * 2);
  hipMemcpy(dJ, J.data(), sizeof(double) * 1024, hipMemcpyHostToDevice);
  const int reps = 2000;
  const char* names[3] = {"16 steps, bpermute, IEEE div", "16 steps, bpermute, rcp+2NR", "8 rounds 2x2, LDS strip, rcp+2NR"};
  for (int v = 0; v < 3; ++v) {
    for (int it = 0; it < 2; ++it) {
      if (v == 0) hipLaunchKernelGGL(micro<0>, dim3(1), dim3(64), 8192, 0, dJ, dout, dc, reps);
      if (v == 1) hipLaunchKernelGGL(micro<1>, dim3(1), dim3(64), 8192, 0, dJ, dout, dc, reps);
      if (v == 2) hipLaunchKernelGGL(micro<2>, dim3(1), dim3(64), 8192, 0, dJ, dout, dc, reps);
      hipDeviceSynchronize();
    }
    long long c[2];
    double o[64];
    hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("%-36s: %8.0f shader cycles, %7.3f us (100 MHz clock) per elimination; check %.10g\n", names[v],
           (double)c[0] / reps, (double)c[1] / reps / 100.0, o[0]);
  }
  return 0;
}
