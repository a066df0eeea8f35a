// 5x5 / stride-1 convolution of a 1- or 2-channel fp32 image on the bf16 matrix cores: the head of the recurrent encoder
// (reference e2vid/model/unet.py:118 -- ConvLayer(num_bins, 32, kernel_size=5, padding=2) on every time step's voxel grid).
//
// The tap-paired kernel (conv_bf16_pair.hip) runs this layer as ONE 8-channel chunk with 6 of the 8 channels zero: all prologue
// and epilogue, 167 us at B=8 / 480x640 against a ~35 us floor of its 157 MB output.  Here the K dimension is the filter itself:
// row r = c * 5 + ky (10 rows for 2 channels) x 8 column slots (kx = 0..4, three zero weights), i.e. five 32x32x16 MFMAs per 32
// pixels with K-step s, lane half h <-> row 2s + h.  A lane's B operand for one K-step is 8 CONSECUTIVE pixels of one input row
// starting at its own pixel: four dwords of a 16-bit tile staged in two copies one element apart (round 6; rounds 3-5 read an fp32 tile
// and converted in registers, in the K loop).
// The weights are read from the tap-paired pack the plan already prescribes for this descriptor (no format of its own) and
// rearranged once per workgroup.  Epilogue: scale / shift / ReLU, BF16_C8 vectors (16-byte stores) and / or fp32 NCHW planes.
#include "conv_bf16_common.h"

namespace {

using namespace essconv;

constexpr int HT_W = 32, HT_H = 32, HT_IW = 40, HT_IH = HT_H + 4;

// NC: channel capacity of the instance (2: the BASELINE voxel grids; 5: the reference's own default, nr_temporal_bins = 5,
// config/settings_DSEC.yaml:15) -- R = 5 NC filter rows = NS = ceil(R / 2) K-steps; a missing last row carries zero weights.
// H: IEEE-half operands (the fp32 image rounded to half in registers, half weights from the pack) and half 16-bit output (ESS_COMPUTE_F16)
template <bool SC, int NC, bool H = false>
__global__ __launch_bounds__(256) void conv_bf16_head5_kernel(const ConvKArgs a, int tiles_x, int tiles_y) {
  constexpr int R = 5 * NC, NS = (R + 1) / 2;
  typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
  // the input tile as 16-bit operand elements, converted ONCE while it is staged (a lane's B operand overlaps its neighbours' in 7 of 8
  // pixels: converting in the K loop cost 8 conversions per lane and K-step -- 62 of the kernel's 71 us were that VALU work, the
  // 157 MB output alone takes 33).  Two copies: copy q holds element i + q at halfword i, so the 8 consecutive elements starting at
  // ANY element e are four aligned dwords of copy e & 1.
  constexpr int TOT = NC * HT_IH * HT_IW, TPITCH = TOT + 8;
  __shared__ __attribute__((aligned(16))) unsigned short tile16[2 * TPITCH];
  __shared__ __attribute__((aligned(16))) u32x4 wfrag[NS * 2 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = logical % a.n_cout_tiles;
  const int sp = logical / a.n_cout_tiles;
  const int n_tiles = tiles_x * tiles_y;
  const int t = sp % n_tiles, n = sp / n_tiles;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int y0 = ty * HT_H, x0 = tx * HT_W;
  const unsigned HW = (unsigned)(a.Hout * a.Wout);

  // ---- weights: tap-paired pack [tile][chunk 0][pair][tap parity][cout 32][channel 8] -> A fragments [K-step][half][cout][8 slots]
  const unsigned short* wp = (const unsigned short*)a.wpk + (size_t)ct * (13 * 2 * 32 * 8);
  for (int i = tid; i < NS * 2 * 32; i += 256) {
    const int m = i & 31, r = i >> 5;
    const int c = r / 5, ky = r - 5 * c;
    unsigned v[5];
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) v[kx] = r < R ? wp[(size_t)((ky * 5 + kx) * 32 + m) * 8 + c] : 0u;  // (pair * 2 + parity = tap)
    const u32x4 f = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4], 0u};
    wfrag[r * 32 + m] = f;  // row r = 2 s + h
  }
  // ---- input tile: both channels, 2-pixel halo, zero outside the image (and for an absent second channel)
  {
    const ess_rsrc r_in = ess_make_rsrc(a.src0 + (size_t)n * a.C0 * a.Hin * a.Win, (size_t)a.C0 * a.Hin * a.Win * 4);
    constexpr int NLD = (NC * HT_IH * HT_IW + 255) / 256;
    float ld[NLD];  // every load is in flight before the first LDS write
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 256;
      const int c = i / (HT_IH * HT_IW), rem = i - c * (HT_IH * HT_IW);
      const int iy = rem / HT_IW, ix = rem - iy * HT_IW;
      const int gy = y0 - 2 + iy, gx = x0 - 2 + ix;
      const bool ok = (i < NC * HT_IH * HT_IW) & (c < a.C0) & (gy >= 0) & (gy < a.Hin) & (gx >= 0) & (gx < a.Win);
      ld[k] = ess_bload(r_in, ok ? (unsigned)((c * a.Hin + gy) * a.Win + gx) * 4u : ESS_OOB, 0);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 256;
      if (i < TOT) {
        unsigned short h;
        if constexpr (H) h = __builtin_bit_cast(unsigned short, ess_f16_sat(ld[k]));
        else h = __builtin_bit_cast(unsigned short, (__bf16)ld[k]);
        tile16[i] = h;
        if (i > 0) tile16[TPITCH + i - 1] = h;
      }
    }
    if (tid == 0) tile16[TPITCH + TOT - 1] = 0;
  }
  // per-channel scale / shift of this lane's rows (the packed vectors are padded to the 32-row tile)
  float sc[16], sh[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c0 = ct * 32 + j * 8 + 4 * half;
    const float4 s4 = SC ? *(const float4*)(a.scale + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 h4 = a.shift ? *(const float4*)(a.shift + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    sc[4 * j] = s4.x; sc[4 * j + 1] = s4.y; sc[4 * j + 2] = s4.z; sc[4 * j + 3] = s4.w;
    sh[4 * j] = h4.x; sh[4 * j + 1] = h4.y; sh[4 * j + 2] = h4.z; sh[4 * j + 3] = h4.w;
  }
  __syncthreads();
  u32x4 af[NS];
  int roff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int r = 2 * s + half, c = r / 5, ky = r - 5 * c;
    af[s] = wfrag[r * 32 + p];
    // dword offset inside copy p & 1 of the lane's first element (rows start at even elements: HT_IW is even)
    roff[s] = ((r < R ? (c * HT_IH + ky) * HT_IW : 0) + p - (p & 1)) >> 1;  // (a row past the filter: zero weights on finite data)
  }
  const unsigned* t32 = (const unsigned*)(tile16 + (p & 1) * TPITCH);
  static_assert(HT_IW % 2 == 0 && TPITCH % 2 == 0, "dword-aligned rows and copies");
  const bool relu = a.act == ESS_ACT_RELU;
  const bool out8 = a.fmt_out == ESS_FMT_BF16_C8;
  const int nb_all = (a.Cout + 7) >> 3;
  void* dst8 = out8 ? (void*)a.out : a.out_bf;
  float* dst32 = out8 ? nullptr : a.out;
  const ess_rsrc r_8 = ess_make_rsrc(dst8 ? (const char*)dst8 + (size_t)n * nb_all * HW * 16 : (const char*)a.src0, dst8 ? (size_t)nb_all * HW * 16 : 0);
  const ess_rsrc r_32 = ess_make_rsrc(dst32 ? (const char*)(dst32 + (size_t)n * a.Cout * HW) : (const char*)a.src0, dst32 ? (size_t)a.Cout * HW * 4 : 0);
  const int x = x0 + p;
#pragma unroll 2
  for (int rr = 0; rr < HT_H / 4; ++rr) {
    const int ly = wave * (HT_H / 4) + rr;
    const int y = y0 + ly;
    if (y >= a.Hout) break;  // (wave-uniform)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const unsigned* src = t32 + ly * (HT_IW / 2) + roff[s];
      const u32x4 b = {src[0], src[1], src[2], src[3]};
      acc = ess_mfma16<H>(af[s], b, acc);
    }
    const bool inb = x < a.Wout;
    const unsigned pix = (unsigned)(y * a.Wout + x);
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[r] = SC ? acc[r] * sc[r] + sh[r] : acc[r] + sh[r];
      if (relu) v[r] = fmaxf(v[r], 0.f);
    }
    if (dst32) {  // (uniform) accumulator register r = 4 j + i <-> channel ct*32 + 8 j + 4 half + i
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = ct * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        ess_bstore(v[r], r_32, (inb && c < a.Cout) ? ((unsigned)c * HW + pix) * 4u : ESS_OOB, 0);
      }
    }
    if (dst8) {  // (uniform)
#pragma unroll
      for (int jp = 0; jp < 4; jp += 2) {
        uint2 pk[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          float q[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = ct * 32 + 8 * (jp + jj) + 4 * half + i;
            q[i] = c < a.Cout ? v[4 * (jp + jj) + i] : 0.f;
          }
          pk[jj] = ess_cvt4<H>(q[0], q[1], q[2], q[3]);
        }
        // exchange halves: lanes 0-31 end up with the whole vector of block jp, lanes 32-63 with that of block jp + 1
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
        const u32x4e vec = {s0[0], s1[0], s0[1], s1[1]};
        const int myblk = ct * 4 + jp + half;
        __builtin_amdgcn_raw_buffer_store_b128(vec, r_8, (int)((inb && myblk < nb_all) ? ((unsigned)myblk * HW + pix) * 16u : ESS_OOB), 0, 0);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// 7x7 / stride 2 / pad 3 on ONE fp32 channel: the image encoder's stem (reference models/style_networks.py:114-116, torchvision
// ResNet conv1 on a grayscale image).  The generic tile kernel contracts the one real channel inside a 16-channel chunk (131 us at
// B=8 against a ~25 us floor of the 79 MB output).  Same construction as the head kernel above: K = filter row ky (7 rows + one
// zero row = four 16-wide K-steps, lane half h <-> row 2s + h) x 8 column slots (kx = 0..6 + one zero weight); a lane's B operand
// is 8 consecutive INPUT pixels of row 2y + ky - 3 starting at column 2x - 3 (the stride only spaces the lanes' start columns).
// Weights come from the generic bf16 pack ([tile][chunk][tap][8-channel block][cout][8], channel 0 of block 0).
constexpr int ST_W = 32, ST_H = 32, ST_IW = 72, ST_IH = 2 * ST_H + 6;

template <int MBS, bool SC>
__global__ __launch_bounds__(256) void conv_bf16_stem7_kernel(const ConvKArgs a, int tiles_x, int tiles_y) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
  __shared__ float tile[ST_IH * ST_IW];
  __shared__ __attribute__((aligned(16))) u32x4 wfrag[MBS * 8 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = logical % a.n_cout_tiles;
  const int sp = logical / a.n_cout_tiles;
  const int n_tiles = tiles_x * tiles_y;
  const int t = sp % n_tiles, n = sp / n_tiles;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int y0 = ty * ST_H, x0 = tx * ST_W;
  const unsigned HW = (unsigned)(a.Hout * a.Wout);
  constexpr int COT = MBS * 32;
  const unsigned short* wp = (const unsigned short*)a.wpk + (size_t)ct * (49 * 2 * COT * 8);  // (one 16-channel chunk = two 8-channel blocks)
  for (int i = tid; i < MBS * 8 * 32; i += 256) {
    const int m = i & 31, r = (i >> 5) & 7, mb = i >> 8;
    unsigned v[7];
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) v[kx] = r < 7 ? wp[(size_t)((r * 7 + kx) * 2 * COT + mb * 32 + m) * 8] : 0u;
    const u32x4 f = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6]};
    wfrag[i] = f;  // [mb][row r = 2 s + h][cout m]
  }
  {
    const ess_rsrc r_in = ess_make_rsrc(a.src0 + (size_t)n * a.Hin * a.Win, (size_t)a.Hin * a.Win * 4);
    constexpr int NLD = (ST_IH * ST_IW + 255) / 256;
    float ld[NLD];  // every load is in flight before the first LDS write
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 256;
      const int iy = i / ST_IW, ix = i - iy * ST_IW;
      const int gy = 2 * y0 - 3 + iy, gx = 2 * x0 - 3 + ix;
      const bool ok = (i < ST_IH * ST_IW) & (gy >= 0) & (gy < a.Hin) & (gx >= 0) & (gx < a.Win);
      ld[k] = ess_bload(r_in, ok ? (unsigned)(gy * a.Win + gx) * 4u : ESS_OOB, 0);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 256;
      if (i < ST_IH * ST_IW) tile[i] = ld[k];
    }
  }
  __syncthreads();
  u32x4 af[MBS][4];
  int roff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int r = 2 * s + half;
    roff[s] = r * ST_IW + 2 * p;
#pragma unroll
    for (int mb = 0; mb < MBS; ++mb) af[mb][s] = wfrag[(mb * 8 + r) * 32 + p];
  }
  const bool relu = a.act == ESS_ACT_RELU;
  const bool out8 = a.fmt_out == ESS_FMT_BF16_C8;
  const int nb_all = (a.Cout + 7) >> 3;
  void* dst8 = out8 ? (void*)a.out : a.out_bf;
  float* dst32 = out8 ? nullptr : a.out;
  const ess_rsrc r_8 = ess_make_rsrc(dst8 ? (const char*)dst8 + (size_t)n * nb_all * HW * 16 : (const char*)a.src0, dst8 ? (size_t)nb_all * HW * 16 : 0);
  const ess_rsrc r_32 = ess_make_rsrc(dst32 ? (const char*)(dst32 + (size_t)n * a.Cout * HW) : (const char*)a.src0, dst32 ? (size_t)a.Cout * HW * 4 : 0);
  const int x = x0 + p;
  for (int rr = 0; rr < ST_H / 4; ++rr) {
    const int ly = wave * (ST_H / 4) + rr;
    const int y = y0 + ly;
    if (y >= a.Hout) break;  // (wave-uniform)
    f32x16 acc[MBS];
#pragma unroll
    for (int mb = 0; mb < MBS; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float* src = tile + 2 * ly * ST_IW + roff[s];
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[j];
      const bf16x8 bfrag = __builtin_bit_cast(bf16x8, pack8(v));
#pragma unroll
      for (int mb = 0; mb < MBS; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mb][s]), bfrag, acc[mb], 0, 0, 0);
    }
    const bool inb = x < a.Wout;
    const unsigned pix = (unsigned)(y * a.Wout + x);
#pragma unroll
    for (int mb = 0; mb < MBS; ++mb) {
      const int cbase = ct * COT + mb * 32;
      float v[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = cbase + j * 8 + 4 * half;
        const float4 s4 = SC ? *(const float4*)(a.scale + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 h4 = a.shift ? *(const float4*)(a.shift + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[4 * j] = acc[mb][4 * j] * s4.x + h4.x; v[4 * j + 1] = acc[mb][4 * j + 1] * s4.y + h4.y;
        v[4 * j + 2] = acc[mb][4 * j + 2] * s4.z + h4.z; v[4 * j + 3] = acc[mb][4 * j + 3] * s4.w + h4.w;
      }
      if (relu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (dst32) {  // (uniform) accumulator register r = 4 j + i <-> channel cbase + 8 j + 4 half + i
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = cbase + 8 * (r >> 2) + 4 * half + (r & 3);
          ess_bstore(v[r], r_32, (inb && c < a.Cout) ? ((unsigned)c * HW + pix) * 4u : ESS_OOB, 0);
        }
      }
      if (dst8) {  // (uniform)
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          uint2 pk[2];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            float q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int c = cbase + 8 * (jp + jj) + 4 * half + i;
              q[i] = c < a.Cout ? v[4 * (jp + jj) + i] : 0.f;
            }
            if (a.out_f16) {  // (uniform) ESS_FMT_F16_C8: the stem's output is a pre-norm tensor (BatchNorm reads it)
              typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
              f16x4 h;
#pragma unroll
              for (int i = 0; i < 4; ++i) h[i] = ess_f16_sat(q[i]);
              pk[jj] = __builtin_bit_cast(uint2, h);
            } else {
              bf16x4 b;
#pragma unroll
              for (int i = 0; i < 4; ++i) b[i] = (__bf16)q[i];
              pk[jj] = __builtin_bit_cast(uint2, b);
            }
          }
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
          const u32x4e vec = {s0[0], s1[0], s0[1], s1[1]};
          const int myblk = (cbase >> 3) + jp + half;
          __builtin_amdgcn_raw_buffer_store_b128(vec, r_8, (int)((inb && myblk < nb_all) ? ((unsigned)myblk * HW + pix) * 16u : ESS_OOB), 0, 0);
        }
      }
    }
  }
}

}  // namespace

namespace essconv {

bool conv_bf16_head_applies(const EssConvDesc* d, const EssConvPlan& pl) {
  static const bool on = [] { const char* e = getenv("ESS_CONV_HEAD"); return !(e && e[0] == '0'); }();
  return on && d->compute == ESS_COMPUTE_BF16 && d->ksize == 5 && d->stride == 1 && d->pad == 2 && d->C1 == 0 && d->C0 <= 5 &&
         d->mode0 == ESS_SRC_DIRECT && d->fmt0 == ESS_FMT_F32_NCHW && d->epilogue == ESS_EPI_LINEAR && d->out_split == 0 &&
         (d->act == ESS_ACT_NONE || d->act == ESS_ACT_RELU) && pl.cout_tile == 32 && pl.n_chunks == 1 && pl.ck == 8;
}

bool conv_bf16_stem_applies(const EssConvDesc* d, const EssConvPlan& pl) {
  static const bool on = [] { const char* e = getenv("ESS_CONV_STEM"); return !(e && e[0] == '0'); }();
  return on && d->compute == ESS_COMPUTE_BF16 && d->ksize == 7 && d->stride == 2 && d->pad == 3 && d->C0 == 1 && d->C1 == 0 &&
         d->mode0 == ESS_SRC_DIRECT && d->fmt0 == ESS_FMT_F32_NCHW && d->epilogue == ESS_EPI_LINEAR && d->out_split == 0 &&
         (d->act == ESS_ACT_NONE || d->act == ESS_ACT_RELU) && (pl.cout_tile == 32 || pl.cout_tile == 64) && pl.n_chunks == 1 && pl.ck == 16;
}

void conv_bf16_launch_stem(const EssConvDesc* d, const EssConvPlan& pl, hipStream_t st, const ConvKArgs& a) {
  const int tiles_x = ceil_div(d->W_out, ST_W), tiles_y = ceil_div(d->H_out, ST_H);
  const dim3 grid((unsigned)(tiles_x * tiles_y * pl.n_cout_tiles * d->N));
  if (pl.cout_tile == 64) {
    if (a.scale) hipLaunchKernelGGL((conv_bf16_stem7_kernel<2, true>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv_bf16_stem7_kernel<2, false>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
  } else {
    if (a.scale) hipLaunchKernelGGL((conv_bf16_stem7_kernel<1, true>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv_bf16_stem7_kernel<1, false>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
  }
}

void conv_bf16_launch_head(const EssConvDesc* d, const EssConvPlan& pl, hipStream_t st, const ConvKArgs& a) {
  const int tiles_x = ceil_div(d->W_out, HT_W), tiles_y = ceil_div(d->H_out, HT_H);
  const dim3 grid((unsigned)(tiles_x * tiles_y * pl.n_cout_tiles * d->N));
  if (a.f16) {  // ESS_COMPUTE_F16
    if (d->C0 <= 2) {
      if (a.scale) hipLaunchKernelGGL((conv_bf16_head5_kernel<true, 2, true>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
      else hipLaunchKernelGGL((conv_bf16_head5_kernel<false, 2, true>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
    } else {
      if (a.scale) hipLaunchKernelGGL((conv_bf16_head5_kernel<true, 5, true>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
      else hipLaunchKernelGGL((conv_bf16_head5_kernel<false, 5, true>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
    }
    return;
  }
  if (d->C0 <= 2) {
    if (a.scale) hipLaunchKernelGGL((conv_bf16_head5_kernel<true, 2>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv_bf16_head5_kernel<false, 2>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
  } else {  // 3 .. 5 voxel-grid bins
    if (a.scale) hipLaunchKernelGGL((conv_bf16_head5_kernel<true, 5>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv_bf16_head5_kernel<false, 5>), grid, dim3(256), 0, st, a, tiles_x, tiles_y);
  }
}

}  // namespace essconv
