// Calibration of the rocprofv3 HBM counters (FETCH_SIZE, WRITE_SIZE) for 8-BYTE-PER-LANE accesses: the univariate
// message kernel (bp_level_uni, site-minor layout: consecutive lanes = consecutive sites, one double each) reads and
// writes 8 B per lane, for which MI355X_MICROARCH.md's gfx950 corrections (stated for 16 B per lane) are not given.
// copy8 moves a known byte count with exactly that access shape; copy16 is the 16-B-per-lane control.
//   hipcc --offload-arch=gfx950 -O3 -o build/exp/copy8 tools/copy8_microbench.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -- build/exp/copy8
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out_w -- build/exp/copy8
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void copy8(double* __restrict__ y, const double* __restrict__ x, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}
__global__ void copy16(double2* __restrict__ y, const double2* __restrict__ x, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}
// the message kernel's shape: each lane reads 13 and writes 8 doubles that sit `stride` doubles apart (element rows
// of the site-minor layout), lanes along the contiguous axis
__global__ void rows8(double* __restrict__ y, const double* __restrict__ x, size_t stride, size_t nrec) {
  const size_t site = blockIdx.y * (size_t)blockDim.x + threadIdx.x;
  if (site >= stride) return;
  const size_t rec = blockIdx.x;
  if (rec >= nrec) return;
  double s = 0.0;
  for (int t = 0; t < 13; ++t) s += x[(rec * 13 + t) * stride + site];
  for (int t = 0; t < 8; ++t) y[(rec * 8 + t) * stride + site] = s + t;
}

// the same rows with two consecutive sites per lane (16 B per lane)
__global__ void rows16(double2* __restrict__ y, const double2* __restrict__ x, size_t stride2, size_t nrec) {
  const size_t site = blockIdx.y * (size_t)blockDim.x + threadIdx.x;
  if (site >= stride2) return;
  const size_t rec = blockIdx.x;
  if (rec >= nrec) return;
  double2 s = {0.0, 0.0};
  for (int t = 0; t < 13; ++t) {
    const double2 v = x[(rec * 13 + t) * stride2 + site];
    s.x += v.x;
    s.y += v.y;
  }
  for (int t = 0; t < 8; ++t) y[(rec * 8 + t) * stride2 + site] = double2{s.x + t, s.y + t};
}

// timing of the four shapes (GB/s moved, reads + writes): `copy8 time`
static void time_all(double* x, double* y, size_t n) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const size_t stride = 8000, nrec = n / 13 / stride;
  auto run = [&](const char* name, double bytes, auto&& launch) {
    launch();
    (void)hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) launch();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%-8s %8.1f GB/s\n", name, bytes * 5 / (ms * 1e-3) / 1e9);
  };
  run("copy8", 2.0 * n * 8, [&] { hipLaunchKernelGGL(copy8, dim3(8192), dim3(256), 0, 0, y, x, n); });
  run("copy16", 2.0 * n * 8, [&] { hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (double2*)y, (const double2*)x, n / 2); });
  run("rows8", (13.0 + 8.0) * nrec * stride * 8, [&] {
    hipLaunchKernelGGL(rows8, dim3((unsigned)nrec, (unsigned)((stride + 255) / 256)), dim3(256), 0, 0, y, x, stride, nrec);
  });
  run("rows16", (13.0 + 8.0) * nrec * stride * 8, [&] {
    hipLaunchKernelGGL(rows16, dim3((unsigned)nrec, (unsigned)((stride / 2 + 255) / 256)), dim3(256), 0, 0, (double2*)y,
                       (const double2*)x, stride / 2, nrec);
  });
  run("rows16/128", (13.0 + 8.0) * nrec * stride * 8, [&] {
    hipLaunchKernelGGL(rows16, dim3((unsigned)nrec, (unsigned)((stride / 2 + 127) / 128)), dim3(128), 0, 0, (double2*)y,
                       (const double2*)x, stride / 2, nrec);
  });
}

int main(int argc, char** argv) {
  const size_t n = (size_t)1 << 27;  // 1 GiB each way
  double *x, *y;
  if (hipMalloc(&x, n * 8) != hipSuccess || hipMalloc(&y, n * 8) != hipSuccess) return 1;
  (void)hipMemset(x, 0, n * 8);
  (void)hipMemset(y, 0, n * 8);
  if (argc > 1) {
    time_all(x, y, n);
    return 0;
  }
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(copy8, dim3(8192), dim3(256), 0, 0, y, x, n);
    hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (double2*)y, (const double2*)x, n / 2);
    const size_t stride = 8000, nrec = n / 13 / stride;  // reads nrec*13*stride doubles, writes nrec*8*stride
    hipLaunchKernelGGL(rows8, dim3((unsigned)nrec, (unsigned)((stride + 255) / 256)), dim3(256), 0, 0, y, x, stride, nrec);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  const size_t stride = 8000, nrec = n / 13 / stride;
  printf("copy8 bytes each way %zu\ncopy16 bytes each way %zu\nrows8 read %zu write %zu\n", n * 8, n * 8, nrec * 13 * stride * 8,
         nrec * 8 * stride * 8);
  return 0;
}
