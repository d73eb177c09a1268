// Probe: `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) on gfx950:
//   * LDS image is lane-linear (lane l -> M0 base + 16*l),
//   * a lane whose voffset is out of range (>= num_records) gets ZEROS written (not skipped).
// build: hipcc --offload-arch=gfx950 -O2 buf_lds_probe.hip -o buf_lds_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_buf(unsigned voff, v4i rsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__global__ void k(const char* x, unsigned bytes, unsigned* out) {
  extern __shared__ char sm[];
  for (int i = threadIdx.x; i < 2048 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(sm)[i] = 0xABABABABu;
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm;
  unsigned long long a = (unsigned long long)x;
  v4i r;
  r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r.y = __builtin_amdgcn_readfirstlane((int)(a >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // lanes 0..31: reversed 16-byte pieces; lanes 32..47: far out of range; lanes 48..63: straddling the end
  unsigned voff = lane < 32 ? (31 - lane) * 16u + wave * 512u : (lane < 48 ? 0xFFFFFFF0u : bytes - 8u);
  dma16_buf(voff, r, lds0 + wave * 1024);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = threadIdx.x; i < 2048 / 4; i += blockDim.x) out[i] = reinterpret_cast<unsigned*>(sm)[i];
}
int main() {
  const unsigned bytes = 4096;
  std::vector<unsigned> h(bytes / 4);
  for (unsigned i = 0; i < bytes / 4; ++i) h[i] = 0x1000u + i;
  char* dx; unsigned* dout;
  hipMalloc(&dx, bytes + 4096); hipMalloc(&dout, 2048);
  hipMemset(dx, 0x77, bytes + 4096);
  hipMemcpy(dx, h.data(), bytes, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 2048, 0, dx, bytes, dout);
  std::vector<unsigned> o(512);
  hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; ++w)
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 4; ++j) {
        unsigned got = o[w * 256 + lane * 4 + j], want;
        if (lane < 32) want = 0x1000u + ((31 - lane) * 16 + w * 512) / 4 + j;
        else if (lane < 48) want = 0;
        else want = j < 2 ? 0x1000u + (bytes - 8) / 4 + j : 0;   // last 8 bytes valid, rest out of range
        if (got != want) { if (bad < 12) printf("wave %d lane %d dword %d: got %08x want %08x\n", w, lane, j, got, want); ++bad; }
      }
  printf("buffer_load lds probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
  return bad != 0;
}
