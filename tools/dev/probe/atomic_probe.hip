// Probe: cost of the weight-gradient epilogue's f32 atomics on gfx950. 252 workgroups (9 tiles x 28 splits) each
// add a 256 x 256 f32 tile into a [256][2304] array; the 28 splits of a tile hit the same words.
//   pattern 0: what the MFMA accumulator layout gives — per wave-instruction 4 rows x 16 consecutive floats
//   pattern 1: one row x 64 consecutive floats per wave-instruction (after a transpose through LDS)
//   pattern 2: like 0 but the splits of a tile start at rotated positions (less same-word contention in time)
//   pattern 3: plain stores into a per-split workspace (two-pass reduction, first pass only)
// build: hipcc --offload-arch=gfx950 -O2 atomic_probe.hip -o atomic_probe.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int CO = 256, K = 2304, TILES = 9, SPLITS = 28;
template <int PAT>
__global__ __launch_bounds__(512) void k(float* dw, float* ws) {
  const int tile = blockIdx.x % TILES, split = blockIdx.x / TILES;
  const int n0 = tile * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g4 = lane >> 4, q = lane & 15;
  if (PAT == 0 || PAT == 2 || PAT == 3) {
    // wave tile 128 x 64: acc[8][4][4]
    const int wo = (wave & 1) * 128, wn = (wave >> 1) * 64;
    for (int i = 0; i < 32; ++i) {
      const int ii = PAT == 2 ? (i + split) & 31 : i;
      const int a = ii >> 2, b = ii & 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = wo + a * 16 + g4 * 4 + j, n = n0 + wn + b * 16 + q;
        if (PAT == 3) ws[((size_t)split * CO + o) * K + n] = 1.0f;
        else atomicAdd(dw + (size_t)o * K + n, 1.0f);
      }
    }
  } else {
    // 8 waves x 32 rows each, 4 instructions of 64 consecutive floats per row
    for (int r = 0; r < 32; ++r) {
      const int o = wave * 32 + r;
#pragma unroll
      for (int c = 0; c < 4; ++c) atomicAdd(dw + (size_t)o * K + n0 + c * 64 + lane, 1.0f);
    }
  }
}
template <int PAT>
float run(float* dw, float* ws) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<PAT>, dim3(TILES * SPLITS), dim3(512), 0, 0, dw, ws);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<PAT>, dim3(TILES * SPLITS), dim3(512), 0, 0, dw, ws);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3f;
}
int main() {
  float *dw, *ws;
  hipMalloc(&dw, sizeof(float) * CO * K);
  hipMalloc(&ws, sizeof(float) * CO * K * SPLITS);
  hipMemset(dw, 0, sizeof(float) * CO * K);
  printf("pattern 0 (4 rows x 16): %.1f us\n", run<0>(dw, ws));
  printf("pattern 1 (1 row x 64):  %.1f us\n", run<1>(dw, ws));
  printf("pattern 2 (rotated):     %.1f us\n", run<2>(dw, ws));
  printf("pattern 3 (plain stores to a workspace): %.1f us\n", run<3>(dw, ws));
  return 0;
}
