// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which 4 elements does each lane get?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void k(short* c) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, q = l & 15;
  const int row = g * 4 + (q >> 2), col = (q & 3) * 4;
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(lds + row * 128 + col));
  for (int j = 0; j < 4; ++j) c[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] / 128, h[l * 4 + j] % 128);
    printf("\n");
  }
  return 0;
}
