// Dev probe: what streaming rate do 1R+1W / 2R+1W / 3R+1W elementwise passes reach on this GPU at the BatchNorm apply's
// largest shape (218 MB per bf16 tensor, beyond the 256 MiB MALL), by launch shape and cache policy?
// build: hipcc --offload-arch=gfx950 -O3 triad_probe.hip -o triad_probe.bin ; run: ./triad_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
struct V16 { u32 a, b, c, d; };
__device__ __forceinline__ V16 ld(const V16* p, int nt) {
  if (nt) { V16 v; v.a = __builtin_nontemporal_load(&p->a); v.b = __builtin_nontemporal_load(&p->b);
            v.c = __builtin_nontemporal_load(&p->c); v.d = __builtin_nontemporal_load(&p->d); return v; }
  return *p;
}
__device__ __forceinline__ void st(V16* p, V16 v, int nt) {
  if (nt) { __builtin_nontemporal_store(v.a, &p->a); __builtin_nontemporal_store(v.b, &p->b);
            __builtin_nontemporal_store(v.c, &p->c); __builtin_nontemporal_store(v.d, &p->d); }
  else *p = v;
}
__device__ __forceinline__ V16 mix(V16 x, V16 y) { x.a ^= y.a; x.b += y.b; x.c ^= y.c; x.d += y.d; return x; }
template <int NR, int VPT, int NT>
__global__ __launch_bounds__(256) void pass(const V16* __restrict__ a, const V16* __restrict__ b, const V16* __restrict__ c,
                                            V16* __restrict__ o, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += stride * VPT) {
    V16 va[VPT], vb[VPT], vc[VPT];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
      const long long i = i0 + u * stride;
      if (i < n) { va[u] = ld(a + i, NT); if (NR > 1) vb[u] = ld(b + i, NT); if (NR > 2) vc[u] = ld(c + i, NT); }
    }
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
      const long long i = i0 + u * stride;
      if (i < n) { V16 v = va[u]; if (NR > 1) v = mix(v, vb[u]); if (NR > 2) v = mix(v, vc[u]); st(o + i, v, NT); }
    }
  }
}
template <int NR, int VPT, int NT>
void run(const char* name, V16* a, V16* b, V16* c, V16* o, long long n, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (blocks == 0) blocks = (int)((n + 256LL * VPT - 1) / (256LL * VPT));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((pass<NR, VPT, NT>), dim3(blocks), dim3(256), 0, 0, a, b, c, o, n);
  hipEventRecord(e0);
  const int R = 10;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL((pass<NR, VPT, NT>), dim3(blocks), dim3(256), 0, 0, a, b, c, o, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)n * 16 * (NR + 1);
  printf("%-28s blocks %7d  %7.1f us  %6.2f TB/s\n", name, blocks, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e12);
}
int main() {
  const long long n = 16LL * 128 * 208 * 256 * 2 / 16;   // 218 MB per tensor
  V16 *a, *b, *c, *o;
  hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&o, n * 16);
  hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16); hipMemset(c, 3, n * 16);
  for (int blocks : {0, 256 * 8, 256 * 16, 256 * 32}) {
    run<1, 4, 0>("1R1W vpt4", a, b, c, o, n, blocks);
    run<2, 4, 0>("2R1W vpt4", a, b, c, o, n, blocks);
    run<2, 8, 0>("2R1W vpt8", a, b, c, o, n, blocks);
    run<2, 4, 1>("2R1W vpt4 nontemporal", a, b, c, o, n, blocks);
    run<2, 8, 1>("2R1W vpt8 nontemporal", a, b, c, o, n, blocks);
    run<3, 4, 0>("3R1W vpt4", a, b, c, o, n, blocks);
    run<3, 4, 1>("3R1W vpt4 nontemporal", a, b, c, o, n, blocks);
  }
  return 0;
}
