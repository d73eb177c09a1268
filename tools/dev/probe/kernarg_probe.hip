// Probe: cost of a kernel launch as a function of the size of its by-value arguments.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Blob { int v[N]; };
template <int N> __global__ void k(Blob<N> b, int* out) { if (b.v[N - 1] == 12345 && threadIdx.x == 999) *out = 1; }
template <int N> void run(int* d) {
  Blob<N> b{}; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<N>, dim3(256), dim3(256), 0, 0, b, d);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  hipEventRecord(e0, 0);
  const int R = 2000;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k<N>, dim3(256), dim3(256), 0, 0, b, d);
  hipEventRecord(e1, 0);
  auto t1 = std::chrono::steady_clock::now();
  hipDeviceSynchronize();
  auto t2 = std::chrono::steady_clock::now();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("args %5d B: host enqueue %.2f us/launch, gpu span %.2f us/launch, wall %.2f us/launch\n", N * 4,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / R, ms * 1e3 / R,
         std::chrono::duration<double, std::micro>(t2 - t0).count() / R);
}
int main() {
  int* d; hipMalloc(&d, 4);
  run<16>(d); run<64>(d); run<128>(d); run<256>(d); run<360>(d); run<512>(d); run<900>(d);
  return 0;
}
