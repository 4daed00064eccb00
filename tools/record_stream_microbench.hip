// Microbenchmark: how fast can the chip move belief-record-sized chunks at random places of a large pool?
// Each wave reads `rd` contiguous bytes at a pseudo-random record and writes `wr` bytes at another one, `iters`
// times; waves per SIMD are set by the (dummy) register budget.  Compare with the message kernel's ~2.7 TB/s of
// measured HBM traffic in wide levels: if this reaches much more, the message kernel is latency/occupancy-bound,
// not DRAM-efficiency-bound.
//   hipcc --offload-arch=gfx950 -O3 -o build/exp/record_stream tools/record_stream_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int WAVES_PER_EU>
__attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
__global__ __launch_bounds__(64) void stream_records(const double4* __restrict__ src, double4* __restrict__ dst,
                                                     long n_rec, int rec_d4, int rd_d4, int wr_d4, int iters,
                                                     double* sink) {
  const int lane = threadIdx.x;
  unsigned long long s = (blockIdx.x + 1) * 0x9E3779B97F4A7C15ull;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const long r0 = (long)((s >> 17) % (unsigned long long)n_rec);
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const long r1 = (long)((s >> 17) % (unsigned long long)n_rec);
    const double4* p = src + r0 * rec_d4;
    double4 v[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (lane + 64 * k < rd_d4) v[k] = p[lane + 64 * k];
    double4* q = dst + r1 * rec_d4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (lane + 64 * k < wr_d4) q[lane + 64 * k] = v[k & 1];
    acc += v[0].x + v[1].y + v[2].z + v[3].w;
  }
  if (acc == 1.2345) sink[0] = acc;
}

int main(int argc, char** argv) {
  const long pool_mb = argc > 1 ? atol(argv[1]) : 1024;
  const int rec_bytes = 4736, rd_bytes = argc > 2 ? atoi(argv[2]) : 7168, wr_bytes = argc > 3 ? atoi(argv[3]) : 3712;
  const long n_rec = pool_mb * 1024 * 1024 / rec_bytes - 2;
  double4 *src, *dst;
  double* sink;
  hipMalloc(&src, pool_mb * 1024 * 1024);
  hipMalloc(&dst, pool_mb * 1024 * 1024);
  hipMalloc(&sink, 8);
  hipMemset(src, 0, pool_mb * 1024 * 1024);
  hipMemset(dst, 0, pool_mb * 1024 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 4;
  for (int occ = 0; occ < 3; ++occ) {
    for (int nblk : {4096, 16384, 65536}) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        if (occ == 0)
          hipLaunchKernelGGL(stream_records<4>, dim3(nblk), dim3(64), 0, 0, src, dst, n_rec, rec_bytes / 32, rd_bytes / 32,
                             wr_bytes / 32, iters, sink);
        else if (occ == 1)
          hipLaunchKernelGGL(stream_records<6>, dim3(nblk), dim3(64), 0, 0, src, dst, n_rec, rec_bytes / 32, rd_bytes / 32,
                             wr_bytes / 32, iters, sink);
        else
          hipLaunchKernelGGL(stream_records<8>, dim3(nblk), dim3(64), 0, 0, src, dst, n_rec, rec_bytes / 32, rd_bytes / 32,
                             wr_bytes / 32, iters, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
      }
      const double bytes = (double)nblk * iters * (rd_bytes + wr_bytes);
      printf("waves/SIMD %d  waves %6d x %d records  rd %d wr %d B: %.3f ms  %.2f TB/s\n", occ == 0 ? 4 : (occ == 1 ? 6 : 8), nblk,
             iters, rd_bytes, wr_bytes, best, bytes / best / 1e9);
    }
  }
  return 0;
}
