// Persistent 256 x 128 tile convolution (included by conv_igemm.hip inside its anonymous namespace). OFF by default
// (das_tuning key conv.pt3_mintiles): measured within +-5 % of the one-tile kernels on the step's shapes, DESIGN 2.2f.
//
// conv_glds3_kernel<PP> runs its K loop at the rate of the guide's 8-phase template, but a launch of it is a sequence
// of ROUNDS of one-tile workgroups, and per tile nothing overlaps the first-stage latency (4-5 us), the C tile's trip
// through LDS + the write burst (3.6 us) or the under-filled last round (416 tiles on 256 CUs). This kernel keeps the
// same ping-pong K loop (two wave groups half a step apart, three 48-KiB stages, LDS-DMA two steps ahead: the loop body
// is the one-tile kernel's, step for step) and makes the workgroup PERSISTENT:
//   * one workgroup per usable CU walks tiles logical + i * gridDim.x (XCD-remapped: the 32 workgroups of an XCD work on
//     32 neighbouring tiles at any time, so the 3x3 halo rows and the weights they share come from that XCD's L2);
//   * at a tile boundary — both wave groups aligned, every stage free — the next tile's address state is rebuilt, its
//     first two K steps are staged, and THEN the finished tile is stored: the DMA latency overlaps the epilogue;
//   * the epilogue goes straight from the accumulators to global memory — no C tile in LDS, no barrier: the weight rows
//     are STAGED in a permuted order (wperm) that makes an accumulator lane own EIGHT consecutive channels of a pixel =
//     one 16-byte store (the permutation conv1x1_stream_kernel applies to its register-resident weights). The stores
//     are fire-and-forget: the next tile's K loop starts while they drain;
//   * the epilogue's global operands (second gradient, raw, y) are hand-issued into the dead fragment registers, all
//     loads of a chunk in flight together, and waited for with vmcnt(0) BEFORE the next tile's DMAs are issued (a
//     counted wait must not span LDS-DMA loads and register loads: they do not retire in order with respect to each
//     other); every load is unconditional (a branch around a hand-issued load makes the compiler copy a register that is
//     still in flight);
//   * BatchNorm statistics / the fused BatchNorm-backward sums are carried in registers through the tiles (a
//     workgroup keeps its column block: gridDim.x is a multiple of the column-block count) and leave once per launch.
// Inside the K loop the counted vmcnt waits see DMAs only (same kind: in order); stores still draining from the previous
// epilogue only make them conservative.
// MODE as conv1x1_stream_kernel: 0 plain / BatchNorm statistics, 1 scale / shift (+ReLU), 2 residual (+ReLU),
// 3 fused BatchNorm-backward reduction with the mask from the saved output (optional residual, no y = no mask; NOT
// dispatched: see try_launch_pt3), 4 the same with the mask recomputed from raw, 5 scale / shift + residual (+ReLU).

// LDS weight row r holds output channel (column block base) + wperm(r): MFMA row R of A tile a = 2 qd + a1 is channel
// qd * 32 + (R >> 2) * 8 + a1 * 4 + (R & 3), so accumulator lane (g4 = R >> 2, j = R & 3) owns channels
// qd * 32 + g4 * 8 + a1 * 4 + j of the tile pair: eight consecutive ones.
__device__ __forceinline__ int wperm(int r) { return (r & ~31) | (((r & 15) >> 2) << 3) | (((r >> 4) & 1) << 2) | (r & 3); }

// n / d for 0 <= n < 2^22 (exact in float), d >= 1, with rd = 1.0f / d: the estimate is off by at most one
__device__ __forceinline__ int fdiv_small(int n, int d, float rd) {
  int qv = (int)((float)n * rd);
  const int r = n - qv * d;
  qv += (r >= d) ? 1 : 0;
  qv -= (r < 0) ? 1 : 0;
  return qv;
}
__device__ __forceinline__ RowGeom row_geom_f(const ConvP& p, int m, float rHoWo, float rWo) {
  RowGeom g;
  if (m >= p.M) {
    g.pix0 = 0; g.hi0 = -(1 << 28); g.wi0 = 0; g.H = 0; g.W = 0;
    return g;
  }
  const int b = fdiv_small(m, p.HoWo, rHoWo), rem = m - b * p.HoWo;
  const int ho = fdiv_small(rem, p.Wo, rWo), wo = rem - ho * p.Wo;
  g.pix0 = (long long)b * p.H * p.W;
  g.hi0 = ho * p.stride - p.pad;
  g.wi0 = wo * p.stride - p.pad;
  g.H = p.H; g.W = p.W;
  return g;
}

constexpr int PT3_STAGES = 3 * (256 + 128) * 128;          // bytes of the three operand stages
constexpr int PT3_CST = 6 * 128 * 4, PT3_RED = 2 * 128 * 4;   // per-channel constants, statistic sums
constexpr int PT3_SMEM = PT3_STAGES + PT3_CST + PT3_RED;

template <int MODE>
__global__ __launch_bounds__(512) void conv_pt3_kernel(ConvP p, int total) {
  using T = bf16_t;
  constexpr bool STATS = MODE == 0, AFF = MODE == 1 || MODE == 5, RES = MODE >= 2;
  constexpr bool BNB = MODE == 3 || MODE == 4;
  constexpr int BN = 128, BMT = 256, NBUF = 3, BK = 64, EPV = 8;
  constexpr int A_BYTES = BMT * 128, W_BYTES = BN * 128, BUF = A_BYTES + W_BYTES;
  constexpr int A_INSTR = 4, W_INSTR = 2, DPS = A_INSTR + W_INSTR;
  constexpr unsigned E = 2, OOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cst = reinterpret_cast<float*>(smem + PT3_STAGES);   // [6][128]: scale, shift, mean, invstd, gamma, beta
  float* sred = cst + 6 * 128;                                // [2][128]

  DAS_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, ntl = p.ntiles;
  const int logical = xcd_remap(blockIdx.x, G);
  const int n0 = (logical % ntl) * BN;          // (the same for every tile of this workgroup: G % ntl == 0)
  const int wave_m0 = (wave & 3) * 64, wave_n0 = (wave >> 2) * 64;
  const int grp = wave >> 2;
  const int q = lane & 15, g4 = lane >> 4;
  const int nk = p.K / BK;

  // ---- per-channel constants -> LDS (read back as float4 pairs in the epilogue), statistic sums zeroed
  if (tid < 128) {
    const int c = n0 + tid;
    const bool ok = c < p.Cout;
    cst[tid] = (AFF && p.scale && ok) ? p.scale[c] : 1.f;
    cst[128 + tid] = (AFF && p.shift && ok) ? p.shift[c] : 0.f;
    if (BNB) {
      cst[256 + tid] = ok ? p.bnb_mean[c] : 0.f;
      cst[384 + tid] = ok ? p.bnb_invstd[c] : 0.f;
      cst[512 + tid] = (MODE == 4 && ok) ? p.bnb_gamma[c] : 0.f;
      cst[640 + tid] = (MODE == 4 && ok) ? p.bnb_beta[c] : 0.f;
    }
    sred[tid] = 0.f;
    sred[128 + tid] = 0.f;
  }
  __syncthreads();

  // ---- the DMA stream's address state (runs up to two K steps ahead of the multiplying waves, across tiles)
  const int lrow = lane >> 3, pslot = lane & 7;
  const v4i_t xrs = make_rsrc(p.x, p.xbytes), wrs = make_rsrc(p.w, (unsigned)((long long)p.Cout * p.K * E));
  unsigned wbase[W_INSTR];
#pragma unroll
  for (int j = 0; j < W_INSTR; ++j) {
    const int row = (wave * W_INSTR + j) * 8 + lrow;
    wbase[j] = (unsigned)(((long long)(n0 + wperm(row)) * p.K + (pslot ^ ((row >> 1) & 7)) * EPV) * E);  // rows >= Cout: out of range
  }
  // (single-level tensors of fewer than 2^22 rows: the row decode by float reciprocals + one correction step instead of
  // two integer divisions per row — the stream rebuilds its state once per tile, in the middle of the K loop)
  const bool fastdiv = p.nlev <= 1 && p.M < (1 << 22);
  const float rHoWo = 1.0f / (float)p.HoWo, rWo = 1.0f / (float)p.Wo;
  unsigned acur[A_INSTR], arowstep[A_INSTR], wcur[W_INSTR];
  int ahi[A_INSTR], awi[A_INSTR], aH[A_INSTR], aW[A_INSTR];
  bool aok[A_INSTR];
  int f_kh = 0, f_kw = 0, f_ci = 0;
  auto setup = [&](int tile) {
    const int m0 = (tile / ntl) * BMT;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const int row = (wave * A_INSTR + j) * 8 + lrow;
      const RowGeom g = fastdiv ? row_geom_f(p, m0 + row, rHoWo, rWo) : row_geom(p, m0 + row);
      const int akg = (pslot ^ ((row >> 1) & 7)) * EPV;
      acur[j] = (unsigned)(((g.pix0 + (long long)g.hi0 * g.W + g.wi0) * p.xps + akg) * (long long)E);
      arowstep[j] = (unsigned)g.W * (unsigned)p.xps * E;
      ahi[j] = g.hi0; awi[j] = g.wi0; aH[j] = g.H; aW[j] = g.W;
      aok[j] = (unsigned)g.hi0 < (unsigned)g.H && (unsigned)g.wi0 < (unsigned)g.W;
    }
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) wcur[j] = wbase[j];
    f_kh = 0; f_kw = 0; f_ci = 0;
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned dA = BK * E, dB = (unsigned)(p.xps - p.Cin + BK) * E;
  const unsigned dC = 0u - (unsigned)(p.KW - 1) * (unsigned)p.xps * E - (unsigned)(p.Cin - BK) * E;  // + rowstep
  auto issue_step = [&](int buf) {   // stage the stream's next K step into `buf`, advance the tap walk
    const unsigned sA = lds0 + buf * BUF, sW = sA + A_BYTES;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) dma16_buf(aok[j] ? acur[j] : OOB, xrs, sA + (wave * A_INSTR + j) * 1024);
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
      dma16_buf(wcur[j], wrs, sW + (wave * W_INSTR + j) * 1024);
      wcur[j] += BK * E;
    }
    f_ci += BK;
    if (f_ci < p.Cin) {
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) acur[j] += dA;
    } else {  // next tap: move the pointer, re-evaluate the bounds
      f_ci = 0;
      const bool wrap = ++f_kw == p.KW;
      if (wrap) { f_kw = 0; ++f_kh; }
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        acur[j] += wrap ? arowstep[j] + dC : dB;
        aok[j] = (unsigned)(ahi[j] + f_kh) < (unsigned)aH[j] && (unsigned)(awi[j] + f_kw) < (unsigned)aW[j];
      }
    }
  };

  // ---- epilogue operands of this lane: pixels m0 + wave_m0 + b * 16 + q, channels n0 + wave_n0 + qd * 32 + g4 * 8 .. + 7
  const int cl0 = wave_n0 + g4 * 8;                       // + qd * 32: channel index inside the column block
  const bool cok0 = n0 + cl0 < p.Cout, cok1 = n0 + cl0 + 32 < p.Cout;
  T* yg = reinterpret_cast<T*>(p.y);
  const T* rg = RES ? reinterpret_cast<const T*>(p.res) : nullptr;
  const T* bx = BNB ? reinterpret_cast<const T*>(p.bnb_raw) : nullptr;
  const T* by = MODE == 3 ? reinterpret_cast<const T*>(p.bnb_y) : nullptr;
  float carry[(STATS || BNB) ? 32 : 1];   // [qd][8][sum | second sum]: kept through the tiles
#pragma unroll
  for (int i = 0; i < ((STATS || BNB) ? 32 : 1); ++i) carry[i] = 0.f;
  auto orow = [&](int m) -> long long {
    if (!p.osub) return m;
    const int b = m / p.HoWo, rem = m - b * p.HoWo;
    const int i = rem / p.Wo, j = rem - i * p.Wo;
    return ((long long)b * p.oH + 2 * i + p.oph) * p.oW + 2 * j + p.opw;
  };

  f32x4_t acc[4][4];
  // The epilogue's global operands (second gradient, raw, y) are requested by hand-issued loads — tracked loads would
  // be sunk next to their uses and waited for one by one with vmcnt(0) (loads, stores and the DMAs share the counter:
  // cdna guide 5.7) — for CHB pixel blocks at a time: the fragment registers of the K loop are free by then.
  constexpr int NOPS = (RES ? 1 : 0) + (BNB ? 1 : 0) + (MODE == 3 ? 1 : 0);   // operand tensors
  constexpr int CHB = NOPS >= 2 ? 2 : 4;                                       // pixel blocks per chunk
  v4i_t lr[CHB][2], lx[CHB][2], ly[CHB][2];
  long long om[4];
  bool mok[4];
  auto tile_rows = [&](int tile) {
    const int m0 = (tile / ntl) * BMT + wave_m0 + q;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int m = m0 + b * 16;
      mok[b] = m < p.M;
      om[b] = orow(mok[b] ? m : p.M - 1);   // (rows past M re-read the last row: the instruction count stays uniform)
    }
  };
  constexpr bool HAS_R = RES, HAS_X = BNB, HAS_Y = MODE == 3;     // operand arrays this MODE can touch
  // (sources of the unconditional loads: the tensor itself, or a valid stand-in whose values are not used)
  const T* r_src = (RES && rg) ? rg : bx;
  const long long r_ps = (RES && rg) ? p.rps : p.bnb_ps;
  const T* y_src = by ? by : bx;
  auto request = [&](int b0) {
    if (NOPS == 0) return;
#pragma unroll
    for (int u = 0; u < CHB; ++u)
#pragma unroll
      for (int qd = 0; qd < 2; ++qd) {
        const int c = (qd ? cok1 : cok0) ? n0 + cl0 + qd * 32 : 0;
        // Every array of this MODE is loaded UNCONDITIONALLY (an absent tensor is replaced by a valid one — raw, or the
        // residual — and its values are ignored): a branch around a hand-issued load makes the compiler merge two
        // definitions of the destination, i.e. COPY a register whose load is still in flight (measured: stale y masks
        // on the trailing wave group).
        if constexpr (HAS_R)
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lr[u][qd]) : "v"(r_src + om[b0 + u] * r_ps + c) : "memory");
        if constexpr (HAS_X)
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lx[u][qd]) : "v"(bx + om[b0 + u] * p.bnb_ps + c) : "memory");
        if constexpr (HAS_Y)
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ly[u][qd]) : "v"(y_src + om[b0 + u] * p.bnb_ps + c) : "memory");
      }
  };
  auto landed = [&]() {   // wait for the chunk requested last (the operands name ITS registers only)
    if (NOPS == 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < CHB; ++u)
#pragma unroll
      for (int qd = 0; qd < 2; ++qd) {   // (empty asm: ties the loaded registers to this point of the program)
        if constexpr (HAS_R) asm volatile("" : "+v"(lr[u][qd])::"memory");
        if constexpr (HAS_X) asm volatile("" : "+v"(lx[u][qd])::"memory");
        if constexpr (HAS_Y) asm volatile("" : "+v"(ly[u][qd])::"memory");
      }
  };
  auto finish = [&](int b0) {
#pragma unroll
    for (int u = 0; u < CHB; ++u)
#pragma unroll
      for (int qd = 0; qd < 2; ++qd) {
        const int b = b0 + u;
        if (!(mok[b] && (qd ? cok1 : cok0))) continue;
        const int cl = cl0 + qd * 32;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = acc[2 * qd][b][j]; v[4 + j] = acc[2 * qd + 1][b][j]; }
        if (AFF) {
          const float4 s0 = *reinterpret_cast<const float4*>(cst + cl), s1 = *reinterpret_cast<const float4*>(cst + cl + 4);
          const float4 h0 = *reinterpret_cast<const float4*>(cst + 128 + cl), h1 = *reinterpret_cast<const float4*>(cst + 128 + cl + 4);
          v[0] = v[0] * s0.x + h0.x; v[1] = v[1] * s0.y + h0.y; v[2] = v[2] * s0.z + h0.z; v[3] = v[3] * s0.w + h0.w;
          v[4] = v[4] * s1.x + h1.x; v[5] = v[5] * s1.y + h1.y; v[6] = v[6] * s1.z + h1.z; v[7] = v[7] * s1.w + h1.w;
        }
        uint4 o = Elem<T>::pack(v);
        if (STATS) {
          if (p.stats) {
            Elem<T>::unpack(o, v);   // the values as stored
#pragma unroll
            for (int j = 0; j < 8; ++j) { carry[qd * 16 + j] += v[j]; carry[qd * 16 + 8 + j] += v[j] * v[j]; }
          }
        } else if (BNB) {
          float x[8];
          Elem<T>::unpack(o, v);     // the conv result as a tile kernel would have staged it (bf16)
          if (RES && rg) {
            float r[8];
            Elem<T>::unpack(__builtin_bit_cast(uint4, lr[u][qd]), r);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += r[j];
          }
          Elem<T>::unpack(__builtin_bit_cast(uint4, lx[u][qd]), x);
          if (MODE == 4) {
            const float* mu = cst + 256 + cl;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = bn_affine(x[j], mu[j], mu[128 + j], mu[256 + j], mu[384 + j]) > 0.f ? v[j] : 0.f;
          } else if (by && p.bnb_relu) {
            float yo[8];
            Elem<T>::unpack(__builtin_bit_cast(uint4, ly[u][qd]), yo);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = yo[j] > 0.f ? v[j] : 0.f;
          }
          o = Elem<T>::pack(v);
          Elem<T>::unpack(o, v);     // dZ as stored
#pragma unroll
          for (int j = 0; j < 8; ++j) { carry[qd * 16 + j] += v[j]; carry[qd * 16 + 8 + j] += v[j] * x[j]; }   // (sum dZ * raw: centred at the end)
        } else if (RES || AFF) {
          if ((RES && rg) || p.relu) {
            Elem<T>::unpack(o, v);
            if (RES && rg) {
              float r[8];
              Elem<T>::unpack(__builtin_bit_cast(uint4, lr[u][qd]), r);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] += r[j];
            }
            if (p.relu) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            o = Elem<T>::pack(v);
          }
        }
        *reinterpret_cast<uint4*>(yg + om[b] * p.yps + n0 + cl) = o;
      }
  };

  // ---- prologue: the first two K steps in flight
  setup(logical);
  issue_step(0);
  if (nk > 1) issue_step(1);
  DAS_STAMP(1);
  const int frow = q, fkg = g4;
  int buf = 0, nbuf = 2;
  for (int tile = logical; tile < total; tile += G) {
    if (nk > 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");   // (conservative while the previous tile's stores drain)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                 // step 0 of this tile landed
#ifdef DAS_STAMPS
    if (tile == logical) DAS_STAMP(2);
#endif
    // The trailing group runs one interval behind inside a tile's K loop and catches up at its end: both groups then
    // stage the next tile and run their epilogue side by side (one after the other — each waiting for the other's at
    // the next barrier — costs an epilogue per tile more).
    if (grp == 1) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto wait_next = [&](int kt) {   // this wave's DMAs of step kt + 1 (step kt + 2 may still fly)
      if (kt + 2 < nk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    for (int kt = 0; kt < nk; ++kt) {
      // ---- R: stage step kt + 2, read this step's fragments
      if (kt + 2 < nk) issue_step(nbuf);
      const char* sA = smem + buf * BUF;
      const char* sW = sA + A_BYTES;
      uint4 fb[2][4], fa[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[t][i] = *reinterpret_cast<const uint4*>(sW + slot128(wave_n0 + i * 16 + frow, t * 4 + fkg));
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[t][i] = *reinterpret_cast<const uint4*>(sA + slot128(wave_m0 + i * 16 + frow, t * 4 + fkg));
      }
      if (grp == 1) wait_next(kt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- M
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) mma<T>(fa[t][a], fb[t][b], acc[a][b]);
      __builtin_amdgcn_s_setprio(0);
      if (grp == 0) wait_next(kt);
      __builtin_amdgcn_s_barrier();
      buf = buf == NBUF - 1 ? 0 : buf + 1;
      nbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // leading group: match the trailing group's extra interval
#ifdef DAS_STAMPS
    if (tile == logical) DAS_STAMP(3);
    if (tile == logical + G) DAS_STAMP(5);
#endif
    // ---- tile boundary: every stage is free (all fragment reads retired before the last barrier). Request the
    // epilogue's operands, stage the next tile's first two K steps, then store while those DMAs fly.
    tile_rows(tile);
    request(0);
    const bool more = tile + G < total;
    if (more) setup(tile + G);          // (address arithmetic of the next tile: runs while the operands fly)
    // The operands are waited for with vmcnt(0) BEFORE the next tile's DMAs go out: a counted wait across the two kinds
    // is not safe — LDS-DMA loads and loads into registers do not retire in order with respect to EACH OTHER (measured:
    // with 12 operand loads followed by 12 DMAs, "vmcnt(12)" let stale registers through on the trailing wave group).
    landed();
    if (more) {
      issue_step(buf);
      if (nk > 1) issue_step(buf == NBUF - 1 ? 0 : buf + 1);
      nbuf = buf >= 1 ? buf - 1 : NBUF - 1;    // (= buf + 2 mod 3: where step 2 of the next tile goes)
    }
    finish(0);
    if (CHB < 4) {
      request(CHB);
      landed();       // (also drains the DMAs just issued: they are needed at the next barrier anyway)
      finish(CHB);
    }
#ifdef DAS_STAMPS
    if (tile == logical) DAS_STAMP(4);
    if (tile == logical + G) DAS_STAMP(6);
#endif
  }
  DAS_STAMP(7);

  // ---- the carried sums: over the 16 pixels (lanes) of a DPP row, over the four pixel-quarter waves (LDS), one
  // round of atomics per workgroup
  if ((STATS && p.stats) || BNB) {
#pragma unroll
    for (int qd = 0; qd < 2; ++qd)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int cl = cl0 + qd * 32 + j;
        float s = carry[qd * 16 + j], t = carry[qd * 16 + 8 + j];
        if (BNB) t = cst[384 + cl] * (t - cst[256 + cl] * s);   // sum dZ * xhat = invstd * (sum dZ * raw - mean * sum dZ)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          float v = w ? t : s;
          v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
          v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));
          v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));
          v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));
          if (q == 15) atomicAdd(&sred[w * 128 + cl], v);
        }
      }
    __syncthreads();
    if (tid < 256) {
      const int w = tid >> 7, c = tid & 127;
      if (n0 + c < p.Cout) {
        const int slot = p.stat_slots > 1 ? (int)(blockIdx.x % (unsigned)p.stat_slots) : 0;
        atomicAdd(p.stats + (slot * 2 + w) * p.Cout + n0 + c, sred[w * 128 + c]);
      }
    }
  }
}
