// 3x3 / stride 1 / pad 1 convolution of a NEAREST-x2-UPSAMPLED BF16_C8 source, polyphase form (round 5).
//
// conv3x3(nearest_up2(z)) reads every low-resolution pixel of z through up to four of its nine taps with the same value: for the
// output parity class (a, b) = (y & 1, x & 1) the nine products collapse to a 2 x 2 filter on z,
//     out[2i + a][2j + b] = sum_{ty, tx in {0, 1}} E_ab[ty][tx] . z[i + a - 1 + ty][j + b - 1 + tx],
//     E_ab[ty][tx] = sum of w[ky][kx] over ky in R(a, ty), kx in R(b, tx),   R(0,0) = {0}, R(0,1) = {1,2}, R(1,0) = {0,1}, R(1,1) = {2}
// -- 16 tap products per low-resolution pixel for four outputs instead of 36: 2.25x fewer matrix instructions, a quarter of the
// staged input positions, and EXACT at the image border (the convolution's zero padding of the upsampled image is the zero padding of
// z: up[-1] is z[-1]).  The decoder's last 3x3 layer (64^ -> 32 @ 480 x 640, models/style_networks.py:95-97 `decoder_scale_4`) was
// the slowest layer of a decoder pass per FLOP (7.5 rounds of 32 x 640 tiles, 850 TFLOP/s).
//
// With so few matrix instructions per staged byte the chunk pipeline of the other kernels does not fit (a 16-channel chunk is 32
// MFMAs = 0.55 us per wave, shorter than a memory round trip: the first version of this file, chunk by chunk with two LDS stages and
// the effective weights re-summed per tile, ran 100 us where the direct kernel needs 106).  So this kernel is WEIGHTS-STATIONARY and
// pipelined per TILE:
//   * a workgroup owns ONE 32-channel output tile for its whole life: the effective weights of all (<= 4) chunks -- 16 (class, tap)
//     pairs x 2 x 32 vectors = 16 KB per chunk -- are summed once from the layer's standard bf16 pack ([tile][chunk][tap][c/8][cout][8]:
//     1, 2 or 4 of its vectors per effective one, added in fp32, rounded once: no second weight layout, plans / packs / the
//     optimiser's re-pack unchanged) and stay in LDS;
//   * a tile is 8 x 32 LOW-resolution pixels (= 16 x 64 outputs x 32 channels x 4 classes); its whole 10 x 34 halo tile, all chunks,
//     is one LDS stage (43.5 KB at 64 input channels), two stages; waves 4-7 load tile t + 1 (16-byte pixel vectors, 11 per thread)
//     while waves 0-3 contract tile t -- the K loop runs through all chunks without a barrier, ONE workgroup barrier per tile;
//   * matrix waves: two pixel blocks of 1 row x 32 columns each (consecutive lanes read consecutive vectors: any row pitch is
//     conflict-free, the tile rows are dense), 4 classes x 2 blocks x 16 = 128 accumulator registers; a frame tap's pixel fragment
//     feeds the 1 / 2 / 4 classes that use it (the centre tap all four); fragment reads one tap ahead, counted lgkmcnt;
//   * epilogue: conv_epilogue_c8 per class at output pixel (2 y + a, 2 x + b) of the full-resolution tensor.
#include "conv_bf16_common.h"

namespace {

using namespace essconv;

constexpr int P_TW = 32, P_TH = 8, P_IH = P_TH + 2, P_IW = P_TW + 2, P_RP = P_IW;
constexpr int P_PLANE = P_IH * P_RP;       // one 8-channel block of the input tile (16-byte vectors): 340
constexpr int P_NPOS = P_IH * P_IW;
constexpr int P_MAXCH = 4;                 // 16-channel chunks a workgroup can keep (C0 <= 64)
constexpr int P_WSZ = 16 * 2 * 32;         // effective weights of a chunk: [class * 4 + ty * 2 + tx][c/8][32 rows]
constexpr int P_WTOT = P_MAXCH * P_WSZ;    // resident weights (vectors)
constexpr int P_STAGE = P_MAXCH * 2 * P_PLANE;  // one tile stage (vectors)
constexpr int P_KPT = (P_MAXCH * 2 * P_NPOS + 255) / 256;  // vectors a staging thread loads per tile: 11

// frame tap t = ky * 3 + kx of the 3 x 3 window over z[i-1 .. i+1][j-1 .. j+1]; class c = a * 2 + b
constexpr bool p_uses(int t, int c) {
  const int ty = t / 3 - (c >> 1), tx = t % 3 - (c & 1);
  return ty >= 0 && ty <= 1 && tx >= 0 && tx <= 1;
}
constexpr int p_pair(int t, int c) { return c * 4 + (t / 3 - (c >> 1)) * 2 + (t % 3 - (c & 1)); }
constexpr int p_na(int t) { return t > 8 ? 0 : (int)p_uses(t, 0) + (int)p_uses(t, 1) + (int)p_uses(t, 2) + (int)p_uses(t, 3); }

// H: IEEE-half operands and output (ESS_COMPUTE_F16)
template <bool H = false>
__global__ __launch_bounds__(512, 2) void conv_bf16_poly_up2_kernel(const ConvKArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  // a.Hin / a.Win: the LOW-resolution (stored) source extent; a.Hout / a.Wout: the full-resolution output's.  Workgroup b owns output
  // tile ct = b % n_cout_tiles and walks the pixel tiles (sample-major) b / n_cout_tiles, + gridDim.x / n_cout_tiles, ...
  const int nct = a.n_cout_tiles, ct = (int)blockIdx.x % nct;
  const int p_first = (int)blockIdx.x / nct, p_step = (int)gridDim.x / nct, p_total = a.n_tiles * a.N;
  const int nch = a.n_chunks;
  u32x4* w_l = smem16;               // resident effective weights [chunk][pair][c/8][32]
  u32x4* in_l = smem16 + P_WTOT;     // two tile stages [stage][chunk][c/8][IH][IW]
#define ESS_P_TILE_DECODE(PT_)                                               \
  const int tile = (PT_) % a.n_tiles, n = (PT_) / a.n_tiles;                 \
  const int ty_ = tile / a.tiles_x, tx_ = tile - ty_ * a.tiles_x;            \
  const int y0 = ty_ * P_TH, x0 = tx_ * P_TW;

  // ---- effective weights, once (all eight waves).  LDS slot j = ((ch * 16 + pair) * 2 + cb) * 32 + row; sources: taps (ky, kx) in
  // R(a, ty) x R(b, tx) of the standard pack, rows r0 + row of slab tile T (slab = the plan's cout_tile, 32 or 64)
  {
    const int SLAB = a.slab;
    const int T = (ct * 32) / SLAB, r0 = (ct * 32) % SLAB;
    const u32x4* wbase = (const u32x4*)a.wpk + (size_t)T * nch * (9 * 2 * SLAB);
    for (int j = (int)threadIdx.x; j < nch * P_WSZ; j += 512) {
      const int ch = j / P_WSZ, jj = j - ch * P_WSZ;
      const int pair = jj >> 6, cb = (jj >> 5) & 1, row = jj & 31;
      const int c = pair >> 2, ca = c >> 1, cbb = c & 1, ty = (pair >> 1) & 1, tx = pair & 1;
      const int ky0 = ca == 0 ? (ty ? 1 : 0) : (ty ? 2 : 0), ky1 = ca == 0 ? (ty ? 2 : 0) : (ty ? 2 : 1);
      const int kx0 = cbb == 0 ? (tx ? 1 : 0) : (tx ? 2 : 0), kx1 = cbb == 0 ? (tx ? 2 : 0) : (tx ? 2 : 1);
      const u32x4* wsrc = wbase + (size_t)ch * (9 * 2 * SLAB) + cb * SLAB + r0 + row;
      float s[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = 0.f;
      for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) {
          const u32x4 v = wsrc[(ky * 3 + kx) * 2 * SLAB];
          if constexpr (H) {
            const f16x8m hv = __builtin_bit_cast(f16x8m, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)hv[e];
          } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s[2 * e] += __builtin_bit_cast(float, v[e] << 16);
            s[2 * e + 1] += __builtin_bit_cast(float, v[e] & 0xffff0000u);
          }
          }
        }
      w_l[j] = H ? pack8h(s) : pack8(s);
    }
  }

  if (role == 1) {
    const int tid = threadIdx.x & 255;
    const size_t hw = (size_t)a.Hin * a.Win;
    const int nb0 = (a.C0 + 7) >> 3;
    const int nvec = nch * 2 * P_NPOS;  // vectors of a tile stage that exist
    // this thread's vectors of a tile: index v = tid + k * 256 -> (block = v / NPOS, position); fixed per thread
    int v_blk[P_KPT], v_iy[P_KPT], v_ix[P_KPT];
#pragma unroll
    for (int k = 0; k < P_KPT; ++k) {
      const int v = tid + k * 256;
      const int blk = v / P_NPOS, pos = v - blk * P_NPOS;
      v_blk[k] = v < nvec ? blk : -1;
      v_iy[k] = pos / P_IW;
      v_ix[k] = pos - v_iy[k] * P_IW;
    }
    u32x4 pre[P_KPT];
    unsigned keep[P_KPT];
    auto load_tile = [&](int pt) {
      ESS_P_TILE_DECODE(pt)
      const u32x4* s0 = (const u32x4*)a.src0 + (size_t)n * nb0 * hw;
#pragma unroll
      for (int k = 0; k < P_KPT; ++k) {
        const int gy = y0 - 1 + v_iy[k], gx = x0 - 1 + v_ix[k];
        const bool in = v_blk[k] >= 0 && v_blk[k] < nb0 && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        keep[k] = in ? 0xffffffffu : 0u;
        pre[k] = s0[in ? (size_t)v_blk[k] * hw + (size_t)(gy * a.Win + gx) : 0];
      }
    };
    auto commit = [&](int stage) {
      u32x4* dst = in_l + stage * P_STAGE;
#pragma unroll
      for (int k = 0; k < P_KPT; ++k) {
        u32x4 v = pre[k];
        const unsigned m = keep[k];
        v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
        if (v_blk[k] >= 0) dst[v_blk[k] * P_PLANE + v_iy[k] * P_RP + v_ix[k]] = v;
      }
    };
    int it = 0;
    if (p_first < p_total) { load_tile(p_first); commit(0); }
    __syncthreads();  // B_0: weights + the first tile are in LDS
    for (int pt = p_first; pt < p_total; pt += p_step, ++it) {
      const int nxt = pt + p_step;
      if (nxt < p_total) { load_tile(nxt); commit((it + 1) & 1); }  // (its stage was last read for tile it - 1: finished before B_it)
      __syncthreads();  // B_{it+1}
    }
    return;
  }
  // --------------------------------------------------------------------------------------------- matrix waves
  __syncthreads();  // B_0
  const unsigned lds0 = (unsigned)(size_t)(smem16);
  int it = 0;
  for (int pt = p_first; pt < p_total; pt += p_step, ++it) {
  ESS_P_TILE_DECODE(pt)
  int tid_t = (int)(threadIdx.x & 255);
  asm volatile("" : "+v"(tid_t));  // (keeps the lane addressing out of the tile loop's live range: see conv_bf16_ws.hip)
  const int tid = tid_t, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  int ly[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) ly[nb] = wave * NBW + nb;
  f32x16 acc[4][NBW];
  const bool biased = a.scale == nullptr && a.shift != nullptr;
  if (biased) {  // (every class starts from the same bias rows)
    conv_bias_init<1, NBW>(a, (f32x16(&)[1][NBW])acc[0], ct, half);
#pragma unroll
    for (int c = 1; c < 4; ++c)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) acc[c][nb] = acc[0][nb];
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][nb][r] = 0.f;
  }
  __builtin_amdgcn_s_setprio(1);
  const unsigned stage_b = lds0 + (unsigned)((P_WTOT + (it & 1) * P_STAGE) * 16);
  const unsigned a_lane = lds0 + (unsigned)((half * 32 + p) * 16);
  unsigned b_lane[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) b_lane[nb] = stage_b + (unsigned)((half * P_PLANE + ly[nb] * P_RP + p) * 16);
  struct Frags { u32x4 a[4]; u32x4 b[NBW]; };
#define ESS_P_READ_A(F_, T_, C_)                                                                                                 \
    if constexpr (p_uses((T_), (C_)))                                                                                            \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.a[C_]) : "v"(wa), "n"(p_uses((T_), (C_)) ? p_pair((T_), (C_)) * 64 * 16 : 0));
#define ESS_P_READ(F_, T_)                                                                                                       \
    {                                                                                                                            \
      ESS_P_READ_A(F_, T_, 0) ESS_P_READ_A(F_, T_, 1) ESS_P_READ_A(F_, T_, 2) ESS_P_READ_A(F_, T_, 3)                             \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.b[0]) : "v"(ba0), "n"((((T_) / 3) * P_RP + (T_) % 3) * 16));        \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F_.b[1]) : "v"(ba1), "n"((((T_) / 3) * P_RP + (T_) % 3) * 16));        \
    }
#define ESS_P_WAIT(F_, N_)                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(F_.a[0]), "+v"(F_.a[1]), "+v"(F_.a[2]), "+v"(F_.a[3]), "+v"(F_.b[0]), "+v"(F_.b[1]) : "n"(N_));
#define ESS_P_MMA_C(F_, T_, C_)                                                                                                  \
    if constexpr (p_uses((T_), (C_))) {                                                                                          \
      _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                                                         \
        acc[C_][nb] = ess_mfma16<H>(F_.a[C_], F_.b[nb], acc[C_][nb]); \
    }
#define ESS_P_MMA(F_, T_) { ESS_P_MMA_C(F_, T_, 0) ESS_P_MMA_C(F_, T_, 1) ESS_P_MMA_C(F_, T_, 2) ESS_P_MMA_C(F_, T_, 3) }
  // ---- K loop over the tile's chunks, no barrier inside: per frame tap the two pixel fragments and the weight fragments of the
  // classes that use it; the reads of tap t + 1 are issued before the MFMAs of tap t (counted lgkmcnt; volatile asm as conv_bf16_ws.hip)
  for (int ch = 0; ch < nch; ++ch) {
    const unsigned wa = a_lane + (unsigned)(ch * P_WSZ * 16);
    const unsigned ba0 = b_lane[0] + (unsigned)(ch * 2 * P_PLANE * 16), ba1 = b_lane[1] + (unsigned)(ch * 2 * P_PLANE * 16);
    Frags f0{}, f1{};  // (defined: a tap leaves the slots of the classes that do not use it untouched)
    ESS_P_READ(f0, 0)
    ESS_P_READ(f1, 1) ESS_P_WAIT(f0, NBW + p_na(1)) ESS_P_MMA(f0, 0)
    ESS_P_READ(f0, 2) ESS_P_WAIT(f1, NBW + p_na(2)) ESS_P_MMA(f1, 1)
    ESS_P_READ(f1, 3) ESS_P_WAIT(f0, NBW + p_na(3)) ESS_P_MMA(f0, 2)
    ESS_P_READ(f0, 4) ESS_P_WAIT(f1, NBW + p_na(4)) ESS_P_MMA(f1, 3)
    ESS_P_READ(f1, 5) ESS_P_WAIT(f0, NBW + p_na(5)) ESS_P_MMA(f0, 4)
    ESS_P_READ(f0, 6) ESS_P_WAIT(f1, NBW + p_na(6)) ESS_P_MMA(f1, 5)
    ESS_P_READ(f1, 7) ESS_P_WAIT(f0, NBW + p_na(7)) ESS_P_MMA(f0, 6)
    ESS_P_READ(f0, 8) ESS_P_WAIT(f1, NBW + p_na(8)) ESS_P_MMA(f1, 7)
    ESS_P_WAIT(f0, 0) ESS_P_MMA(f0, 8)
  }
#undef ESS_P_MMA
#undef ESS_P_MMA_C
#undef ESS_P_WAIT
#undef ESS_P_READ
#undef ESS_P_READ_A
  __builtin_amdgcn_s_setprio(0);
  // every fragment read of this stage has returned (the last wait drained the queue): the staging waves may overwrite it after the
  // NEXT barrier only, which this wave reaches behind its epilogue
  // ---- epilogue: class (ca, cb) of low-resolution pixel (y, x) is output pixel (2 y + ca, 2 x + cb); a.Hout / a.Wout are the
  // full-resolution extent (bounds and pixel indices)
  int ly2[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) ly2[nb] = 2 * ly[nb];
  // (spelled out per class: left as a loop, hipcc keeps the four inlined epilogues rolled and the accumulators in scratch)
  conv_epilogue_c8<1, H>(a, (f32x16(&)[1][NBW])acc[0], ct, n, half, 2 * (x0 + p), 2 * y0, ly2, biased);
  conv_epilogue_c8<1, H>(a, (f32x16(&)[1][NBW])acc[1], ct, n, half, 2 * (x0 + p) + 1, 2 * y0, ly2, biased);
  conv_epilogue_c8<1, H>(a, (f32x16(&)[1][NBW])acc[2], ct, n, half, 2 * (x0 + p), 2 * y0 + 1, ly2, biased);
  conv_epilogue_c8<1, H>(a, (f32x16(&)[1][NBW])acc[3], ct, n, half, 2 * (x0 + p) + 1, 2 * y0 + 1, ly2, biased);
  __syncthreads();  // B_{it+1}: the next tile is staged
  }  // tile loop
#undef ESS_P_TILE_DECODE
}

}  // namespace

namespace essconv {

// low-resolution tile of a workgroup, and the input channels it can keep resident weights for
void conv_bf16_poly_tile(int* th, int* tw, int* max_cin) { *th = P_TH; *tw = P_TW; *max_cin = P_MAXCH * 16; }

void conv_bf16_launch_poly(dim3 grid, hipStream_t st, const ConvKArgs& a) {
  constexpr size_t lds = (size_t)(P_WTOT + 2 * P_STAGE) * 16;
  static_assert(lds <= 160 * 1024, "resident weights + two tile stages must fit the LDS");
  if (a.f16) {  // ESS_COMPUTE_F16
    ess_allow_lds(conv_bf16_poly_up2_kernel<true>, lds);
    hipLaunchKernelGGL(conv_bf16_poly_up2_kernel<true>, grid, dim3(512), lds, st, a);
    return;
  }
  ess_allow_lds(conv_bf16_poly_up2_kernel<false>, lds);
  hipLaunchKernelGGL(conv_bf16_poly_up2_kernel<false>, grid, dim3(512), lds, st, a);
}

}  // namespace essconv
