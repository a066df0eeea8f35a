// Weight gradients from BF16_C8 tensors (the stored form of the trainable networks' activations X and output gradients dY in
// the bf16 configuration: bfloat16 [N][C/8][H][W][8]).
//
// GEMM view as in conv_wgrad.hip:  dW[tap][co][ci] = sum_px dY[co][px] * X[ci][px (+) tap], contraction over PIXELS on
// v_mfma_f32_32x32x16_bf16 -- both operands want 8 consecutive pixels of ONE channel per lane, while BF16_C8 keeps the 8
// channels of one pixel together.
//   3x3 / stride 1 / pad 1 (95 % of the weight-gradient FLOPs): wgrad_c8_ws_kernel -- the tiles go global -> LDS by LDS-DMA in
//     the tensors' own layout and are transposed by the LDS read itself (ds_read_b64_tr_b16); loader waves issue the DMA, MFMA
//     waves contract.  See the comment block in front of it.
//   1x1 (ResNet downsample, stride 1 or 2): wgrad_c8_kernel<1, TWL> -- register staging: a thread loads a "quad" = 4 consecutive
//     pixel vectors of one 8-channel block, regroups the 16-bit elements with 16 v_perm_b32 into 8 half-vectors (4 pixels of one
//     channel each) and writes them to a channel-major LDS tile (ds_write_b64); one centre tap of the 3x3 tile geometry.
//   1x1 head (32 -> K classes, fp32 dY): wgrad_small1x1_c8_kernel, exact-fp32 matrix core.
// Split-K slabs and the reduce kernel are conv_wgrad.hip's.
#include "conv_wgrad_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct Quad { u32x4w v[4]; };

// 4 pixel vectors (8 channels each) -> for channel 2d / 2d+1: the 4 pixels as two dwords
__device__ __forceinline__ void quad_transpose(const Quad& q, uint2 (&out)[8]) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    out[2 * d].x = __builtin_amdgcn_perm(q.v[1][d], q.v[0][d], 0x05040100u);
    out[2 * d].y = __builtin_amdgcn_perm(q.v[3][d], q.v[2][d], 0x05040100u);
    out[2 * d + 1].x = __builtin_amdgcn_perm(q.v[1][d], q.v[0][d], 0x07060302u);
    out[2 * d + 1].y = __builtin_amdgcn_perm(q.v[3][d], q.v[2][d], 0x07060302u);
  }
}

template <int OFF>
__device__ __forceinline__ void lds_read128(u32x4w& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

// TWL = log2 of the pixel-tile width (4: 16 x 8 pixels, 5: 32 x 4): fixes every LDS offset of the fragment reads at compile time
template <int TAPS, int TWL>
__global__ __launch_bounds__(256) void wgrad_c8_kernel(const WgradBArgs b, int sx_abl) {
  extern __shared__ __attribute__((aligned(16))) u32x4w smemv[];
  const int sx = sx_abl & 0xff, abl = sx_abl >> 8;  // (abl: diagnostic ablation switches ESS_WG_ABL: 1 no MFMA, 2 no loads, 4 no LDS staging, 8 no slab write)
  const WgradArgs& a = b.w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int pair = logical % a.npairs, split = logical / a.npairs, nsplit = a.nsplit;
  const int cot = pair / a.ci_tiles, cit = pair - cot * a.ci_tiles;
  const int cb = wave >> 1, ib = wave & 1;
  const int TWp = 1 << a.twl, THp = 128 >> a.twl;
  const int TV = TWp >> 3;
  const int IH = THp + 2;
  const int Cin = a.C0 + a.C1;
  u32x4w* dy_t = smemv;
  u32x4w* x_t = smemv + 64 * b.pyv;
  const int stage = 64 * (b.pyv + b.pxv);  // 16-byte vectors per LDS stage
  constexpr int XQ = 3;                    // X quads per thread: 8 blocks x IH x 2*rv quads = 576 (twl 5) / 640 (twl 4) <= 768
  const int QW = TWp >> 2;                 // dY quads per tile row
  const int XW = 2 * b.rv;                 // X quads per tile row
  const int nxq = 8 * IH * XW;
  const size_t HWo = (size_t)a.Hout * a.Wout;
  const int sh0 = a.mode0 != ESS_SRC_DIRECT ? 1 : 0, sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;
  const int nbo = (a.Cout + 7) >> 3, nb0 = (a.C0 + 7) >> 3, nb1 = (a.C1 + 7) >> 3;

  // ---- tile-independent staging plan
  // dY: quad = (block cbk of this 64-channel tile, tile row qy, quad column xq): one per thread
  const int d_xq = tid % QW, d_r = tid / QW;
  const int d_qy = d_r % THp, d_cbk = d_r / THp;
  const int d_blk = cot * 8 + d_cbk;
  const bool d_ok = d_blk < nbo;
  const int d_lds2 = 2 * ((d_cbk * 8) * b.pyv + d_qy * TV + (d_xq >> 1)) + (d_xq & 1);  // 8-byte units, channel j adds 2 * pyv * j
  int x_lds2[XQ], x_iy[XQ], x_xq[XQ], x_sh[XQ], x_w[XQ], x_zero[XQ];
  const u32x4w* x_base[XQ];
  size_t x_ns[XQ];
  bool x_ok[XQ];
#pragma unroll
  for (int i = 0; i < XQ; ++i) {
    const int q = tid + i * 256;
    const int xq = q % XW, r = q / XW;
    const int iy = r % IH, cbk = r / IH;
    const int c0 = (cit * 8 + cbk) * 8;            // first channel of the block in the concatenated input
    const bool first = c0 < a.C0 || a.C1 == 0;
    const int bi = (first ? c0 : c0 - a.C0) >> 3, nbs = first ? nb0 : nb1;
    const int sh = first ? sh0 : sh1;
    const int Hp = a.Hin >> sh, Wp = a.Win >> sh;
    x_ok[i] = q < nxq && c0 < Cin && bi < nbs;
    x_base[i] = (const u32x4w*)(first ? a.src0 : a.src1) + (size_t)(bi < nbs ? bi : 0) * Hp * Wp;
    x_ns[i] = (size_t)nbs * Hp * Wp;
    x_w[i] = Wp; x_sh[i] = sh; x_iy[i] = iy; x_xq[i] = xq;
    x_zero[i] = (first ? a.mode0 : a.mode1) == ESS_SRC_ZERO_UP2;
    x_lds2[i] = q < nxq ? 2 * ((cbk * 8) * b.pxv + iy * b.rv + (xq >> 1)) + (xq & 1) : -1;
  }

  Quad dq, xq_[XQ];
  unsigned dmask, xmask[XQ];  // bit i: pixel i of the quad is real data
  auto issue = [&](int tile) {
    if (abl & 2) return;
    const int n = tile / (a.tiles_x * a.tiles_y);
    const int tr = tile - n * a.tiles_x * a.tiles_y;
    const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int y0 = ty * THp, x0 = tx * TWp;
    {
      const int y = y0 + d_qy, x = x0 + 4 * d_xq;
      const u32x4w* src = (const u32x4w*)a.dy + ((size_t)n * nbo + (d_ok ? d_blk : 0)) * HWo + (size_t)min(y, a.Hout - 1) * a.Wout;
      dmask = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (d_ok && y < a.Hout && x + i < a.Wout) dmask |= 1u << i;
        dq.v[i] = src[min(x + i, a.Wout - 1)];
      }
    }
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
      const int gy = (y0 - a.pad + x_iy[k]) * sx;
      const int cy = min(max(gy, 0), a.Hin - 1) >> x_sh[k];
      const u32x4w* src = x_base[k] + (size_t)n * x_ns[k] + (size_t)cy * x_w[k];
      const bool rok = x_ok[k] && gy >= 0 && gy < a.Hin && !(x_zero[k] && (gy & 1));
      xmask[k] = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gx = (x0 - 8 + 4 * x_xq[k] + i) * sx;
        if (rok && gx >= 0 && gx < a.Win && !(x_zero[k] && (gx & 1))) xmask[k] |= 1u << i;
        xq_[k].v[i] = src[min(max(gx, 0), a.Win - 1) >> x_sh[k]];
      }
    }
  };
  auto commit = [&](const Quad& qin, unsigned mask, uint2* base2, int lds2, int pitch2) {
    if (abl & 4) return;
    Quad q = qin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned m = (mask >> i) & 1u ? 0xffffffffu : 0u;
      q.v[i][0] &= m; q.v[i][1] &= m; q.v[i][2] &= m; q.v[i][3] &= m;
    }
    uint2 o[8];
    quad_transpose(q, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) base2[lds2 + j * pitch2] = o[j];
  };
  auto commit_d = [&](int st) { commit(dq, dmask, (uint2*)(dy_t + st * stage), d_lds2, 2 * b.pyv); };
  auto commit_x = [&](int k, int st) {
    if (x_lds2[k] >= 0) commit(xq_[k], xmask[k], (uint2*)(x_t + st * stage), x_lds2[k], 2 * b.pxv);
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const bool want_b0 = a.ws_b && cit == 0 && ib == 0;
  bool want_b = want_b0;  // (several passes: the bias gradient takes the passes of bias_mask -- split operands contract dY_hi twice)

  // K loop over this workgroup's pixel tiles: 8 k-steps of 16 pixels x TAPS taps per tile, two LDS stages; the quads of tile
  // t+1 (loads issued behind the first k-step) are transposed and written into the other stage on k-steps 4-7
  struct Frag { u32x4w a; u32x4w v[3][3]; };
  if (split < a.ntiles) {
    issue(split);
    commit_d(0);
#pragma unroll
    for (int k = 0; k < XQ; ++k) commit_x(k, 0);
  }
  __syncthreads();
  int st = 0;
  for (int tile = split; tile < a.ntiles; tile += nsplit, st ^= 1) {
    const bool more = tile + nsplit < a.ntiles;
    // ---- fragment reads / MFMAs of the 8 k-steps.  One wave per SIMD: nothing else hides an LDS round trip, and hipcc sinks
    // plain fragment loads to just above their first MFMA with an `s_waitcnt lgkmcnt(0)` (every 2-3 MFMAs in the first version's
    // ISA).  So: volatile-asm reads of k-step ks+1 are issued BEFORE the MFMAs of k-step ks, and the wait that releases k-step
    // ks+1 is a counted lgkmcnt taking the fragments as in/out operands (which ties the MFMAs behind it).  LDS operations
    // retire in order; between the reads of ks+1 and their wait only the NR reads of ks+2 are issued, the staging writes of a
    // k-step come before its reads.
    constexpr int KSH = TWL - 4, KMASK = (1 << KSH) - 1, RV = (1 << TWL) / 8 + 2;
    constexpr int NR = TAPS == 1 ? 2 : 10;
    const unsigned lds0 = (unsigned)(size_t)smemv;
    const unsigned a_addr = lds0 + (unsigned)((st * stage + (cb * 32 + p) * b.pyv + half) * 16);
    const unsigned x_addr = lds0 + (unsigned)((64 * b.pyv + st * stage + (ib * 32 + p) * b.pxv + half) * 16);
#define ESS_ROFF(KS_) ((((KS_) >> KSH) * RV + (((KS_) & KMASK) << 1)) * 16)
#define ESS_RD(F_, KS_)                                                                               \
    {                                                                                                 \
      lds_read128<(KS_) * 32>(F_.a, a_addr);                                                          \
      if constexpr (TAPS == 1) {                                                                      \
        lds_read128<ESS_ROFF(KS_) + (RV + 1) * 16>(F_.v[1][1], x_addr);                               \
      } else {                                                                                        \
        lds_read128<ESS_ROFF(KS_) + (0 * RV + 0) * 16>(F_.v[0][0], x_addr);                           \
        lds_read128<ESS_ROFF(KS_) + (0 * RV + 1) * 16>(F_.v[0][1], x_addr);                           \
        lds_read128<ESS_ROFF(KS_) + (0 * RV + 2) * 16>(F_.v[0][2], x_addr);                           \
        lds_read128<ESS_ROFF(KS_) + (1 * RV + 0) * 16>(F_.v[1][0], x_addr);                           \
        lds_read128<ESS_ROFF(KS_) + (1 * RV + 1) * 16>(F_.v[1][1], x_addr);                           \
        lds_read128<ESS_ROFF(KS_) + (1 * RV + 2) * 16>(F_.v[1][2], x_addr);                           \
        lds_read128<ESS_ROFF(KS_) + (2 * RV + 0) * 16>(F_.v[2][0], x_addr);                           \
        lds_read128<ESS_ROFF(KS_) + (2 * RV + 1) * 16>(F_.v[2][1], x_addr);                           \
        lds_read128<ESS_ROFF(KS_) + (2 * RV + 2) * 16>(F_.v[2][2], x_addr);                           \
      }                                                                                               \
    }
#define ESS_WT(F_, N_)                                                                                \
    {                                                                                                 \
      if constexpr (TAPS == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(F_.a), "+v"(F_.v[1][1]) : "n"(N_)); \
      else asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(F_.a), "+v"(F_.v[0][0]), "+v"(F_.v[0][1]), "+v"(F_.v[0][2]), "+v"(F_.v[1][0]), \
                        "+v"(F_.v[1][1]), "+v"(F_.v[1][2]), "+v"(F_.v[2][0]), "+v"(F_.v[2][1]), "+v"(F_.v[2][2]) : "n"(N_)); \
    }
    auto mma = [&](const Frag& f) {
      if (abl & 1) return;
      const bf16x8w af = __builtin_bit_cast(bf16x8w, f.a);
      if (want_b) {  // (wave-uniform: the bias gradient rides on one wave per output-channel half)
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += (float)af[j];
      }
      if constexpr (TAPS == 1) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, f.v[1][1]), acc[0], 0, 0, 0);
      } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, shift_left1(f.v[ky][0], f.v[ky][1])),
                                                                    acc[ky * 3 + 0], 0, 0, 0);
          acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, f.v[ky][1]), acc[ky * 3 + 1], 0, 0, 0);
          acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, shift_right1(f.v[ky][1], f.v[ky][2])),
                                                                    acc[ky * 3 + 2], 0, 0, 0);
        }
      }
    };
    Frag f0, f1;
    ESS_RD(f0, 0)
    ESS_RD(f1, 1) ESS_WT(f0, NR) mma(f0);
    if (more) issue(tile + nsplit);  // address arithmetic + 16 loads, behind the first k-step's MFMAs
    ESS_RD(f0, 2) ESS_WT(f1, NR) mma(f1);
    ESS_RD(f1, 3) ESS_WT(f0, NR) mma(f0);
    ESS_RD(f0, 4) ESS_WT(f1, NR) mma(f1);
    if (more) commit_d(st ^ 1);      // (staging writes of a k-step are issued before its fragment reads: see the wait counts)
    ESS_RD(f1, 5) ESS_WT(f0, NR) mma(f0);
    if (more) commit_x(0, st ^ 1);
    ESS_RD(f0, 6) ESS_WT(f1, NR) mma(f1);
    if (more) commit_x(1, st ^ 1);
    ESS_RD(f1, 7) ESS_WT(f0, NR) mma(f0);
    if (more) commit_x(2, st ^ 1);
    ESS_WT(f1, 0) mma(f1);
#undef ESS_RD
#undef ESS_WT
#undef ESS_ROFF
    __syncthreads();
  }
  const int ci = cit * 64 + ib * 32 + p;
  if (abl & 8) return;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cot * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < a.Cout && ci < Cin) a.ws[(((size_t)split * TAPS + t) * a.Cout + co) * Cin + ci] = acc[t][r];
    }
  if (want_b) {
    bsum += __shfl_xor(bsum, 32, 64);
    const int co = cot * 64 + cb * 32 + p;
    if (half == 0 && co < a.Cout) a.ws_b[(size_t)split * a.Cout + co] = bsum;
  }
}

// ---- 3x3 / stride 1 / pad 1: no staging registers, no transposition code.
// gfx950 has the two instructions that make BF16_C8 the NATIVE operand format of a pixel-contraction:
//   * buffer_load_dwordx4 ... lds  (LDS-DMA): 64 lanes x 16 bytes go from per-lane global addresses straight into 1 KiB of LDS
//     (lane-linear), out-of-range lanes write zeros -- one instruction stages 64 pixel vectors (8 channels each) of a tile plane,
//     image borders / tile overhang / absent channel blocks cost nothing (their lanes get an out-of-range offset);
//   * ds_read_b64_tr_b16: a 16-lane group fetches 16 x 8 bytes from per-lane addresses and receives them transposed
//     (lane c of the group gets element c%4 of pieces c/4, 4 + c/4, 8 + c/4, 12 + c/4).  With lane i of a group pointing at
//     (pixel k0 + i/4, channels 4*(i%4) ..+3) the group ends up with lane = channel, elements = 4 consecutive pixels: two such
//     reads are exactly one v_mfma_f32_32x32x16_bf16 operand (8 pixels of one channel per lane).
// So the LDS tile is the tensor's own layout -- [8-channel block][tile pixel][8 channels], one plane per block, plane pitch =
// 64 B mod 256 B so that the four blocks a 32-lane read group touches sit in different bank quarters -- and a filter tap is an
// immediate offset of the X read ((ky * row + kx) pixels x 16 B): no shifts, no masks, no VALU in the loop at all.
// Every LDS access of the kernel is inline asm: hipcc orders a ds_read it can see behind ALL outstanding LDS-DMA (vmcnt(0)), which
// would serialise the stages.
template <int OFF>
__device__ __forceinline__ void lds_read_tr(uint2& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
#define ESS_LDS_PTR(off_) ((__attribute__((address_space(3))) void*)(size_t)(off_))

struct DmaGeom {  // 16 x 8 pixel tiles
  static constexpr int TWL = 4, TW = 16, TH = 8, RW = TW + 2, IH = TH + 2, XPX = IH * RW;  // 180 X pixels incl. the halo
  static constexpr int XP = (XPX + 63) / 64, XTAIL = XPX - (XP - 1) * 64;  // LDS-DMA pieces (64 pixels) per X plane; lanes of the last
  static constexpr int DPL = 2 * 1024 + 64, XPL = XPX * 16;               // plane pitches, both = 64 B mod 256 B
  static_assert(XPL % 256 == 64, "X plane pitch");
  static constexpr int XREG = 8 * DPL, STAGE = 8 * (DPL + XPL);
};

#ifdef ESS_WG_TRACE
__device__ unsigned long long g_wg_trace[1024 * 8 * 40];
#define ESS_TR(i_) do { if (lane == 0 && blockIdx.x < 1024) g_wg_trace[(blockIdx.x * 8 + wave) * 40 + (i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define ESS_TR(i_) do { } while (0)
#endif
// ---- loader waves.  What cycle stamps (s_memtime per tile, -DESS_WG_TRACE) said about the first arrangements of this kernel, in
// which the MFMA waves issued their own DMA: a tile's 72 MFMAs need 2304 cycles, a wave that issues no DMA finishes a tile in
// ~2900 -- and every buffer_load ... lds costs the ISSUING wave 110 - 150 cycles (20 per tile in a one-tile-per-wave variant:
// 5070 cycles per tile; 10 per tile with four waves sharing a tile: 4200).  LDS-DMA is asynchronous for the data, not for the
// instruction.  So the instruction goes to waves that have nothing else to do: eight waves, 0-3 contract (one per SIMD,
// 32 x 32 x 9 accumulators each, a 64 x 64 (co, ci) pair tile per workgroup), 4-7 only issue DMA (10 per tile each), two tiles
// ahead, into four LDS stages.  One workgroup barrier per tile: the loaders arrive when their share of the NEXT tile has landed
// (counted vmcnt: the younger tile stays in flight), the MFMA waves two units before the end of the current one -- their fragment
// reads run two units ahead of the MFMAs (counted lgkmcnt) and continue into the next tile behind that barrier.
// Measured (B = 8): 256 -> 256 @ 60x80 101 us (register-staged kernel of this file's first version) -> 56 us incl. the 8 us
// reduce; 2930 cycles per tile = 79 % of the MFMA rate in the loop.  What bounds it is the L2 -> LDS FILL RATE: a tile is 40 KB
// of DMA per 2304 MFMA cycles, i.e. 26-30 GB/s per CU, and the chip's LDS-DMA fill rate is ~6.4-6.8 TB/s = 25-26 GB/s per CU
// (MI355X_MICROARCH.md, ldsdma-fill).  Three changes to the MFMA waves' side that all measured +-0 say the same: fragment reads
// pipelined across tile boundaries (kept), two MFMA waves per SIMD on 16x16x32 tiles (142 registers, 12 waves), 45 % fewer LDS
// reads by assembling the three filter columns of a row from 12 pixels read once (v_alignbyte; note: hipcc realigns an odd
// register quad through SCRATCH unless it is moved explicitly -- 3.5x slower until it was).  More flops per DMA byte needs a
// 128 x 64 pair tile = 295 KB of accumulators, i.e. no registers left for loader waves.  Two K-groups of four waves issuing their
// own DMA (1.37 us per tile, but 34 us of tile-independent time) and 32 x 32 pair tiles with a wave-private pixel range (4x
// smaller slabs, 2x the L2 -> LDS traffic, 20 DMA issues per tile and wave) were measured and dropped as well.
template <int NST>
__global__ __launch_bounds__(512) void wgrad_c8_ws_kernel(const WgradBArgs b) {
  using G = DmaGeom;
  constexpr int PER_WAVE = 2 * (2 + G::XP);  // DMA instructions per loader wave and tile
  constexpr int AHEAD = 2;                   // tiles the loaders run ahead of the barrier
  static_assert(NST == AHEAD + 2, "stages: tile k - 1 (last two units), k, k + 1 (landing), k + 2 (being issued)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dma[];
  const WgradArgs& a = b.w;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, p = lane & 31;
  const bool loader = wave >= 4;
  const int w4 = wave & 3;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int pair = logical % a.npairs, split = logical / a.npairs, nsplit = a.nsplit;
  const int cot = pair / a.ci_tiles, cit = pair - cot * a.ci_tiles;
  const int Cin = a.C0 + a.C1;
  const unsigned lds0 = (unsigned)(size_t)smem_dma;
  const int ntt = b.npass > 1 ? b.npass * a.ntiles : a.ntiles;  // (several (dY, X) sets: npass passes over the tile list, see WgradBArgs)
  int ntl = 0;  // this workgroup's tiles: split, split + nsplit, ...
  if (split < ntt) ntl = (ntt - 1 - split) / nsplit + 1;

  ESS_TR(0);
  if (loader) {
    const int sh0 = a.mode0 != ESS_SRC_DIRECT ? 1 : 0, sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;
    const bool z0 = a.mode0 == ESS_SRC_ZERO_UP2, z1 = a.mode1 == ESS_SRC_ZERO_UP2;
    const int nbo = (a.Cout + 7) >> 3, nb0 = (a.C0 + 7) >> 3, nb1 = (a.C1 + 7) >> 3;
    const unsigned HWo16 = (unsigned)a.Hout * a.Wout * 16u;
    const int W0 = (a.Win << a.ps) >> sh0, W1 = a.Win >> sh1;  // (ps: parity-phase gather of the first source, see WgradArgs)
    const unsigned HW0_16 = (unsigned)((a.Hin << a.ps) >> sh0) * W0 * 16u, HW1_16 = (unsigned)(a.Hin >> sh1) * W1 * 16u;
    int d_r[2], d_c[2], x_r[G::XP], x_c[G::XP];
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int pi = q * 64 + lane; d_r[q] = pi >> G::TWL; d_c[q] = pi & (G::TW - 1); }
#pragma unroll
    for (int q = 0; q < G::XP; ++q) { const int pi = q * 64 + lane; x_r[q] = pi / G::RW; x_c[q] = pi % G::RW; }
    unsigned dpl_off[2], xpl_off[2];
    bool xpl_first[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pl = 2 * w4 + j;
      const int ob = cot * 8 + pl;
      dpl_off[j] = ob < nbo ? (unsigned)ob * HWo16 : OOBW;
      const int c0 = (cit * 8 + pl) * 8;
      const bool first = c0 < a.C0 || a.C1 == 0;
      const int bi = (first ? c0 : c0 - a.C0) >> 3;
      xpl_first[j] = first;
      xpl_off[j] = (c0 < Cin && bi < (first ? nb0 : nb1)) ? (unsigned)bi * (first ? HW0_16 : HW1_16) : OOBW;
    }
    const int tpi = a.tiles_x * a.tiles_y;
    auto issue = [&](int tile, int st) {
      const char* p_dy = (const char*)a.dy;
      const char* p_x0 = (const char*)a.src0;
      const char* p_x1 = (const char*)a.src1;
      if (b.npass > 1) {  // (wave-uniform: the pass of this tile picks its tensors)
        const int pass = tile / a.ntiles;
        tile -= pass * a.ntiles;
        p_dy = (const char*)(pass == 0 ? b.p_dy[0] : pass == 1 ? b.p_dy[1] : b.p_dy[2]);
        p_x0 = (const char*)(pass == 0 ? b.p_x0[0] : pass == 1 ? b.p_x0[1] : b.p_x0[2]);
        p_x1 = (const char*)(pass == 0 ? b.p_x1[0] : pass == 1 ? b.p_x1[1] : b.p_x1[2]);
      }
      if (!a.C1) p_x1 = p_x0;
      const int n = tile / tpi;
      const int tr = tile - n * tpi;
      const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
      const int y0 = ty * G::TH, x0 = tx * G::TW;
      const __amdgpu_buffer_rsrc_t r_dy =
          __builtin_amdgcn_make_buffer_rsrc((void*)(p_dy + (size_t)n * nbo * HWo16), 0, (int)(nbo * HWo16), 0x00020000);
      const __amdgpu_buffer_rsrc_t r_x0 =
          __builtin_amdgcn_make_buffer_rsrc((void*)(p_x0 + (size_t)n * nb0 * HW0_16), 0, (int)(nb0 * HW0_16), 0x00020000);
      const __amdgpu_buffer_rsrc_t r_x1 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p_x1 + (size_t)n * (a.C1 ? nb1 * HW1_16 : 0u)), 0, (int)(a.C1 ? nb1 * HW1_16 : 0u), 0x00020000);
      const unsigned sbase = lds0 + (unsigned)st * G::STAGE;
      unsigned vd[2], vx0[G::XP], vx1[G::XP];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int y = y0 + d_r[q], x = x0 + d_c[q];
        const bool in = (y < a.Hout) & (x < a.Wout);
        vd[q] = in ? (unsigned)(y * a.Wout + x) * 16u : OOBW;
      }
#pragma unroll
      for (int q = 0; q < G::XP; ++q) {
        const int gy = y0 - 1 + x_r[q], gx = x0 - 1 + x_c[q];
        const bool in = ((unsigned)gy < (unsigned)a.Hin) & ((unsigned)gx < (unsigned)a.Win);
        const bool odd = (gy | gx) & 1;
        vx0[q] = (in & !(z0 & odd)) ? (unsigned)((((gy << a.ps) + a.pp) >> sh0) * W0 + (((gx << a.ps) + a.pq) >> sh0)) * 16u : OOBW;
        vx1[q] = (in & !(z1 & odd)) ? (unsigned)((gy >> sh1) * W1 + (gx >> sh1)) * 16u : OOBW;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pl = 2 * w4 + j;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const unsigned vo = (dpl_off[j] | vd[q]) & OOBW ? OOBW : dpl_off[j] + vd[q];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_dy, ESS_LDS_PTR(sbase + pl * G::DPL + q * 1024), 16, (int)vo, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < G::XP; ++q) {
          const unsigned vq = xpl_first[j] ? vx0[q] : vx1[q];
          const unsigned vo = (xpl_off[j] | vq) & OOBW ? OOBW : xpl_off[j] + vq;
          if (q < G::XP - 1 || lane < G::XTAIL) {  // (lanes switched off by EXEC write nothing: the last piece stops at the plane's end)
            if (xpl_first[j])
              __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x0, ESS_LDS_PTR(sbase + G::XREG + pl * G::XPL + q * 1024), 16, (int)vo, 0, 0, 0);
            else
              __builtin_amdgcn_raw_ptr_buffer_load_lds(r_x1, ESS_LDS_PTR(sbase + G::XREG + pl * G::XPL + q * 1024), 16, (int)vo, 0, 0, 0);
          }
        }
      }
    };
    ESS_TR(1);
    for (int k = 0; k < AHEAD && k < ntl; ++k) issue(split + k * nsplit, k);
    ESS_TR(2);
    for (int k = 0; k < ntl; ++k) {
      if (k < 30) ESS_TR(3 + k);
      // tile k has landed (this wave's share); the younger one may still be in flight
      if (k + 1 < ntl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // B_k: tile k is complete in LDS.  The MFMA waves arrive with two units of tile k - 1 left to do (their reads of those are
      // already issued, but may not have returned), so the stage that is free to overwrite is tile k - 2's: four stages
      __builtin_amdgcn_s_barrier();
      if (k + AHEAD < ntl) issue(split + (k + AHEAD) * nsplit, (k + AHEAD) % NST);
    }
    ESS_TR(37);
    return;
  }

  // ---- MFMA waves
  const int cb = w4 >> 1, ib = w4 & 1;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const bool want_b0 = a.ws_b && cit == 0 && ib == 0;
  bool want_b = want_b0;  // (several passes: the bias gradient takes the passes of bias_mask -- split operands contract dY_hi twice)
  const int g = lane >> 4, i16 = lane & 15;
  const unsigned pix_off = (unsigned)((8 * (g >> 1) + (i16 >> 2)) * 16 + (i16 & 1) * 8);
  const unsigned pl_sel = (unsigned)(2 * (g & 1) + ((i16 & 3) >> 1));
  const unsigned a_lane = (unsigned)(cb * 4 + pl_sel) * G::DPL + pix_off;
  const unsigned x_lane = G::XREG + (unsigned)(ib * 4 + pl_sel) * G::XPL + pix_off;
  struct FA { uint2 lo, hi; };
  struct FB { uint2 lo[3], hi[3]; };
  FA fa[2];
  FB fb[3];
  // unit u = (k-step ks = u / 3: the 16 pixels of tile row ks, filter row ky = u % 3): 6 X reads (+2 dY reads on ky = 0), 3 MFMAs.
  // Reads run two units ahead of their MFMAs -- across the tile boundary too: units 22 / 23 of a tile fetch units 0 / 1 of the next
  // one (NEXT = true), which the barrier in front of unit 22 has released.
  unsigned a_addr = lds0 + a_lane, x_addr = lds0 + x_lane, a_next = 0, x_next = 0;
  auto rd = [&](auto uc, unsigned aad, unsigned xad) {
    constexpr int U = decltype(uc)::value, KS = U / 3, KY = U % 3;
    if constexpr (KY == 0) {
      lds_read_tr<KS * 256>(fa[KS & 1].lo, aad);
      lds_read_tr<KS * 256 + 64>(fa[KS & 1].hi, aad);
    }
    constexpr int XO = (KS + KY) * G::RW * 16;
    lds_read_tr<XO + 0>(fb[U % 3].lo[0], xad);
    lds_read_tr<XO + 64>(fb[U % 3].hi[0], xad);
    lds_read_tr<XO + 16>(fb[U % 3].lo[1], xad);
    lds_read_tr<XO + 80>(fb[U % 3].hi[1], xad);
    lds_read_tr<XO + 32>(fb[U % 3].lo[2], xad);
    lds_read_tr<XO + 96>(fb[U % 3].hi[2], xad);
  };
  auto unit = [&](auto uc, auto nextc) {
    constexpr int U = decltype(uc)::value, KS = U / 3, KY = U % 3;
    constexpr bool NEXT = decltype(nextc)::value;
    if constexpr (U + 2 < 24) rd(std::integral_constant<int, U + 2>{}, a_addr, x_addr);
    else if constexpr (NEXT) rd(std::integral_constant<int, U + 2 - 24>{}, a_next, x_next);
    constexpr int N1 = (U + 1 < 24 || NEXT) ? 6 + ((U + 1) % 3 == 0 ? 2 : 0) : 0, N2 = (U + 2 < 24 || NEXT) ? 6 + ((U + 2) % 3 == 0 ? 2 : 0) : 0;
    constexpr int LEFT = N1 + N2;  // reads issued behind this unit's own
    FA& A_ = fa[KS & 1];
    FB& B_ = fb[U % 3];
    if constexpr (KY == 0)
      asm volatile("s_waitcnt lgkmcnt(%8)"
                   : "+v"(A_.lo), "+v"(A_.hi), "+v"(B_.lo[0]), "+v"(B_.hi[0]), "+v"(B_.lo[1]), "+v"(B_.hi[1]), "+v"(B_.lo[2]), "+v"(B_.hi[2])
                   : "n"(LEFT));
    else
      asm volatile("s_waitcnt lgkmcnt(%6)"
                   : "+v"(B_.lo[0]), "+v"(B_.hi[0]), "+v"(B_.lo[1]), "+v"(B_.hi[1]), "+v"(B_.lo[2]), "+v"(B_.hi[2])
                   : "n"(LEFT));
    const u32x4w av = {A_.lo.x, A_.lo.y, A_.hi.x, A_.hi.y};
    const bf16x8w af = __builtin_bit_cast(bf16x8w, av);
    if constexpr (KY == 0) {
      if (want_b) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) bsum += (float)af[jj];
      }
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const u32x4w xv = {B_.lo[kx].x, B_.lo[kx].y, B_.hi[kx].x, B_.hi[kx].y};
      acc[KY * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8w, xv), acc[KY * 3 + kx], 0, 0, 0);
    }
  };
  using Yes = std::true_type;
  using No = std::false_type;
  ESS_TR(1);
  if (ntl > 0) {
    __builtin_amdgcn_s_barrier();  // B_0
    rd(std::integral_constant<int, 0>{}, a_addr, x_addr);
    rd(std::integral_constant<int, 1>{}, a_addr, x_addr);
  }
  for (int k = 0; k + 1 < ntl; ++k) {  // every tile but the last
    if (k < 30) ESS_TR(3 + k);
    if (b.npass > 1) want_b = want_b0 && ((b.bias_mask >> ((split + k * nsplit) / a.ntiles)) & 1);
    const unsigned nst = (unsigned)((k + 1) % NST) * G::STAGE;
    a_next = lds0 + nst + a_lane;
    x_next = lds0 + nst + x_lane;
    static_for<0, 22>([&](auto uc) { unit(uc, No{}); });
    __builtin_amdgcn_s_barrier();  // B_{k+1}: tile k + 1 is complete in LDS; this wave has no read of tile k - 1 left
    unit(std::integral_constant<int, 22>{}, Yes{});
    unit(std::integral_constant<int, 23>{}, Yes{});
    a_addr = a_next;
    x_addr = x_next;
  }
  if (ntl > 0) {
    if (ntl - 1 < 30) ESS_TR(3 + ntl - 1);
    if (b.npass > 1) want_b = want_b0 && ((b.bias_mask >> ((split + (ntl - 1) * nsplit) / a.ntiles)) & 1);
    static_for<0, 24>([&](auto uc) { unit(uc, No{}); });
  }
  ESS_TR(34);
  // ---- slab: ws[split][tap][co][ci]; one base pointer, 32-bit element offsets.  (Round 5: 16-byte stores through an in-quad 4 x 4
  // DPP transpose -- 36 dwordx4 instead of 144 dword stores per lane, same layout -- measured 56.1 -> 58.2 us for 256 -> 256 @ 60 x 80 and
  // +-0 in the step: the store instructions are not what the epilogue waits for.  Removed.)
  const int ci = cit * 64 + ib * 32 + p, co0 = cot * 64 + cb * 32 + 4 * half;
  float* wsp = a.ws + ((size_t)split * 9 * a.Cout + co0) * Cin + ci;
  const int tap_stride = a.Cout * Cin;
  if (ci < Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dco = (r & 3) + 8 * (r >> 2);
      if (co0 + dco < a.Cout) {
        float* q = wsp + dco * Cin;
#pragma unroll
        for (int t = 0; t < 9; ++t) q[t * tap_stride] = acc[t][r];
      }
    }
  }
  if (want_b) {
    bsum += __shfl_xor(bsum, 32, 64);
    const int co = cot * 64 + cb * 32 + p;
    if (half == 0 && co < a.Cout) a.ws_b[(size_t)split * a.Cout + co] = bsum;
  }
  ESS_TR(37);
}

// ---- the K-class 1x1 head (32 -> K at full resolution): X is BF16_C8 (C_in <= 32), dY fp32 NCHW (the loss kernels' logit
// gradients, C_out <= 32).  HBM-bound.  A workgroup walks chunks of 128 pixels: X and dY are staged to LDS as fp32
// [channel][pixel] rows, each wave contracts its 32 pixels of the chunk on the exact-fp32 matrix core
// (v_mfma_f32_32x32x2_f32, one 32x32 (co, ci) tile), the four waves are combined at the end; one slab per workgroup.
__global__ __launch_bounds__(256) void wgrad_small1x1_c8_kernel(const WgradArgs a, int chunks_per_img, int total_chunks) {
  constexpr int P = 128, PP = P + 1;
  __shared__ float dy_s[32 * PP];
  __shared__ float x_s[32 * PP];
  float* red = dy_s;  // reused after the loop: [4][16 * 64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int Cin = a.C0, Cout = a.Cout, nbi = (Cin + 7) >> 3;
  const size_t HW = (size_t)a.Hout * a.Wout;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  for (int i = tid; i < 32 * PP; i += 256) { dy_s[i] = 0.f; x_s[i] = 0.f; }  // rows of absent channels stay zero
  for (int chunk = blockIdx.x; chunk < total_chunks; chunk += gridDim.x) {
    const int n = chunk / chunks_per_img;
    const size_t px0 = (size_t)(chunk - n * chunks_per_img) * P;
    __syncthreads();
    // X: nbi blocks x 128 pixel vectors
    for (int i = tid; i < nbi * P; i += 256) {
      const int blk = i / P, j = i - blk * P;
      const size_t px = px0 + j;
      const u32x4w v = ((const u32x4w*)a.src0)[((size_t)n * nbi + blk) * HW + (px < HW ? px : HW - 1)];
      const unsigned m = px < HW ? 0xffffffffu : 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        x_s[(blk * 8 + 2 * q) * PP + j] = __builtin_bit_cast(float, (v[q] << 16) & m);
        x_s[(blk * 8 + 2 * q + 1) * PP + j] = __builtin_bit_cast(float, v[q] & 0xffff0000u & m);
      }
    }
    for (int i = tid; i < Cout * P; i += 256) {
      const int r = i / P, j = i - r * P;
      const size_t px = px0 + j;
      const float t = a.dy[((size_t)n * Cout + r) * HW + (px < HW ? px : HW - 1)];
      dy_s[r * PP + j] = px < HW ? t : 0.f;
    }
    __syncthreads();
    const float* dp = dy_s + p * PP + wave * 32 + half;
    const float* xp = x_s + p * PP + wave * 32 + half;
#pragma unroll 8
    for (int kk = 0; kk < 16; ++kk) {
      const float d1 = dp[2 * kk], x1 = xp[2 * kk];
      bsum += d1;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, x1, acc, 0, 0, 0);
    }
  }
  __syncthreads();
  float* redb = x_s;  // [4][32]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[r];
  bsum += __shfl_xor(bsum, 32, 64);
  if (half == 0) redb[wave * 32 + p] = bsum;
  __syncthreads();
  for (int i = tid; i < 16 * 64; i += 256) {
    const int r = i >> 6, l = i & 63;
    const int co = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ci = l & 31;
    if (co < Cout && ci < Cin)
      a.ws[(size_t)blockIdx.x * Cout * Cin + co * Cin + ci] = (red[i] + red[1024 + i]) + (red[2048 + i] + red[3072 + i]);
  }
  if (a.ws_b && tid < Cout) a.ws_b[(size_t)blockIdx.x * Cout + tid] = (redb[tid] + redb[32 + tid]) + (redb[64 + tid] + redb[96 + tid]);
}

}  // namespace

template <int TAPS, int TWL>
static void wgrad_c8_go(const WgradBArgs& b, int sx, int lds_bytes, dim3 grid, hipStream_t st) {
  if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)wgrad_c8_kernel<TAPS, TWL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL((wgrad_c8_kernel<TAPS, TWL>), grid, dim3(256), lds_bytes, st, b, sx);
}

int wgrad_c8_launch(const WgradBArgs& b, int taps, int sx, int lds_bytes, dim3 grid, hipStream_t st) {
  if (taps == 9) {  // 3x3 / stride 1: the LDS-DMA kernel with loader waves (16 x 8 pixel tiles)
    if (b.w.twl != 4 || sx != 1) { ess_set_error("wgrad(BF16_C8, 3x3): stride 1, 16-wide pixel tiles"); return ESS_EINVAL; }
    constexpr int NST = 4, lds = NST * DmaGeom::STAGE;
    static_assert(lds <= 160 * 1024, "LDS stages of the weight-gradient kernel");
    (void)hipFuncSetAttribute((const void*)wgrad_c8_ws_kernel<NST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((wgrad_c8_ws_kernel<NST>), grid, dim3(512), lds, st, b);
    return ess_launch_status("conv2d_wgrad(BF16_C8, LDS-DMA, loader waves)");
  }
  if (b.w.twl != 4 && b.w.twl != 5) { ess_set_error("wgrad(BF16_C8): pixel tiles are 16 or 32 wide"); return ESS_EINVAL; }
  if (b.w.twl == 5) wgrad_c8_go<1, 5>(b, sx, lds_bytes, grid, st); else wgrad_c8_go<1, 4>(b, sx, lds_bytes, grid, st);
  return ess_launch_status("conv2d_wgrad(BF16_C8)");
}

int wgrad_small1x1_c8_launch(const WgradArgs& a, int nsplit, hipStream_t st) {
  const int per_img = ceil_div(a.Hout * a.Wout, 128);
  hipLaunchKernelGGL(wgrad_small1x1_c8_kernel, dim3(nsplit), dim3(256), 0, st, a, per_img, a.N * per_img);
  return ess_launch_status("conv2d_wgrad(1x1 head, BF16_C8)");
}

#ifdef ESS_WG_TRACE
extern "C" int ess_debug_wgrad_trace(unsigned long long* host_out, size_t n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wg_trace), n * sizeof(unsigned long long));
}
#endif
