// Launch floor of small dependent kernels on one stream: period per launch for grids of 2048 x 64, 512 x 256 and 128 x 1024
// threads, for an empty kernel, one that loads + stores 8 bytes per thread, and one that also writes `kb` KB per wave.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(double* p) {}
__global__ void k_touch(double* p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  p[i] = p[i] + 1.0;
}
__global__ void k_write(double* p, double* out, int doubles_per_thread) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double v = p[i] + 1.0;
  p[i] = v;
  const size_t total = (size_t)gridDim.x * blockDim.x;
  for (int k = 0; k < doubles_per_thread; ++k) __builtin_nontemporal_store(v, out + (size_t)k * total + i);
}
template <class F> static double period_us(F launch, int n) {
  for (int i = 0; i < 50; ++i) launch();
  (void)hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) launch();
  (void)hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
  const size_t threads = 2048 * 64;
  double *p, *out;
  (void)hipMalloc(&p, threads * 8);
  (void)hipMalloc(&out, threads * 8 * 32);
  (void)hipMemset(p, 0, threads * 8);
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int shapes[3][2] = {{2048, 64}, {512, 256}, {128, 1024}};
  for (auto& sh : shapes) {
    const dim3 g(sh[0]), b(sh[1]);
    const double e = period_us([&] { hipLaunchKernelGGL(k_empty, g, b, 0, s, p); }, 2000);
    const double t = period_us([&] { hipLaunchKernelGGL(k_touch, g, b, 0, s, p); }, 2000);
    const double w = period_us([&] { hipLaunchKernelGGL(k_write, g, b, 0, s, p, out, 16); }, 2000);  // 16 MB per launch
    printf("grid %4d x %4d threads: empty %.2f us, load+store 8 B/thread %.2f us, + 16 MB of streaming stores %.2f us per launch\n", sh[0], sh[1], e, t, w);
  }
  return 0;
}
