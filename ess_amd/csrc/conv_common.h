// Shared pieces of the direct-convolution kernels (fp32-MFMA and bf16-MFMA variants): kernel argument block,
// tile-geometry choice, descriptor validation, weight-pack row mapping.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace essconv {

constexpr int NBW = 2;  // pixel blocks per wave

struct ConvKArgs {
  const float* src0;
  const float* src1;
  const void* wpk;
  const float* scale;
  const float* shift;
  const float* residual;
  const float* aux0;
  const float* aux1;
  float* out;
  float* out2;
  int N, Hin, Win, C0, C1, mode0, mode1;
  int Cout, Hout, Wout, pad;
  int bwl, wxl, tiles_x, n_tiles, n_cout_tiles;
  int IH, IW, row_pitch, par_off, plane;
  int ck, n_chunks;
  int act, hid, out_split;
};


// ---- shared epilogue: accumulator block (MFMA 32x32 C/D layout: col = lane&31 = pixel, row = (r&3)+8(r>>2)+4(lane>>5)
// = output channel) -> fused affine / residual / activation / LSTM / GRU maths -> NCHW fp32 stores
template <int MB, int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvKArgs& a, f32x16 (&acc)[MB][NBW], int ct, int n, int half, int x,
                                              int y0, const int (&ly)[NBW]) {
  constexpr int COT = MB * 32;
  const size_t HW = (size_t)a.Hout * a.Wout;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int y = y0 + ly[nb];
    if (y >= a.Hout || x >= a.Wout) continue;
    const size_t pix = (size_t)y * a.Wout + x;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int rowbase = ct * COT + mb * 32;
      if constexpr (EPI == ESS_EPI_LINEAR || EPI == ESS_EPI_GRU_OUT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = rowbase + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (co >= a.Cout) continue;
          float v = acc[mb][nb][r];
          if (a.scale) v *= a.scale[co];
          if (a.shift) v += a.shift[co];
          if constexpr (EPI == ESS_EPI_LINEAR) {
            const size_t idx = ((size_t)n * a.Cout + co) * HW + pix;
            if (a.residual) v += a.residual[idx];
            if (a.act == ESS_ACT_RELU) v = fmaxf(v, 0.f);
            else if (a.act == ESS_ACT_SIGMOID) v = ess_sigmoid(v);
            else if (a.act == ESS_ACT_TANH) v = ess_tanh(v);
            if (a.out_split > 0) {
              if (co < a.out_split) a.out[((size_t)n * a.out_split + co) * HW + pix] = v;
              else a.out2[((size_t)n * (a.Cout - a.out_split) + (co - a.out_split)) * HW + pix] = v;
            } else {
              a.out[idx] = v;
            }
          } else {  // GRU candidate: h' = h (1-u) + tanh(.) u
            const size_t idx = ((size_t)n * a.hid + co) * HW + pix;
            const float o = ess_tanh(v), u = a.aux1[idx], h = a.aux0[idx];
            a.out[idx] = h * (1.f - u) + o * u;
          }
        }
      } else if constexpr (EPI == ESS_EPI_LSTM) {
        // packed row 8*g + j of this 32-row block = gate g (in, remember, out, cell) of hidden hb*8 + j
        const int hb = ct * MB + mb;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int hc = hb * 8 + 4 * half + jj;
          if (hc >= a.hid) continue;
          const int pr = rowbase + 4 * half + jj;
          const float gi = ess_sigmoid(acc[mb][nb][jj] + a.shift[pr]);
          const float gf = ess_sigmoid(acc[mb][nb][4 + jj] + a.shift[pr + 8]);
          const float go = ess_sigmoid(acc[mb][nb][8 + jj] + a.shift[pr + 16]);
          const float gc = ess_tanh(acc[mb][nb][12 + jj] + a.shift[pr + 24]);
          const size_t idx = ((size_t)n * a.hid + hc) * HW + pix;
          const float cprev = a.aux0 ? a.aux0[idx] : 0.f;
          const float cn = gf * cprev + gi * gc;
          a.out2[idx] = cn;
          a.out[idx] = go * ess_tanh(cn);
        }
      } else {  // ESS_EPI_GRU_UR
        // packed row 8*q + j: gate q&1 (0 update, 1 reset) of hidden hb*16 + (q>>1)*8 + j
        const int hb = ct * MB + mb;
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int hc = hb * 16 + q2 * 8 + 4 * half + jj;
            if (hc >= a.hid) continue;
            const int pr = rowbase + 16 * q2 + 4 * half + jj;
            const float u = ess_sigmoid(acc[mb][nb][8 * q2 + jj] + a.shift[pr]);
            const float rr = ess_sigmoid(acc[mb][nb][8 * q2 + 4 + jj] + a.shift[pr + 8]);
            const size_t idx = ((size_t)n * a.hid + hc) * HW + pix;
            const float h = a.aux0 ? a.aux0[idx] : 0.f;
            a.out[idx] = u;
            a.out2[idx] = rr * h;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
struct Geom {
  int bwl, wxl, TW, TH, tiles_x, tiles_y, IH, IW, row_pitch, par_off, plane;
};

inline bool is_bf16(const EssConvDesc* d) { return d->compute == ESS_COMPUTE_BF16; }

inline int pick_ck(const EssConvDesc* d) {
  const int cin = d->C0 + d->C1;
  if (is_bf16(d)) return (d->ksize == 1 && d->stride == 1 && cin >= 32) ? 32 : 16;  // one v_mfma_f32_32x32x16_bf16 K-step = 16 channels
  // fp32: K-step = 2 channels; chunk sized for LDS.  Only the 2-channel 5x5 head gets a narrower chunk.
  if (d->ksize == 5 && d->stride == 1 && cin <= 2) return 2;
  return d->ksize >= 7 ? 2 : (d->ksize == 5 ? 4 : 8);
}

inline int packed_rows(const EssConvDesc* d);
inline int pick_mb(const EssConvDesc* d) {
  // bf16 3x3/s1 (wave-specialised kernel): 128-channel tiles where there are enough output channels -- halves the
  // activation staging and the weight re-fetch per MFMA
  // (measured: pays only for the deepest layer -- 512 -> 1024 @ 60x80: 779 -> 859 TFLOP/s; one workgroup per CU hurts the rest)
  if (d->compute == ESS_COMPUTE_BF16 && d->ksize == 3 && d->stride == 1 && packed_rows(d) >= 256 && d->C0 + d->C1 >= 512) {
    static const bool mb4 = [] { const char* e = getenv("ESS_CONV_MB4"); return !(e && e[0] == '0'); }();
    if (mb4) return 4;
  }
  return d->C_out > 32 ? 2 : 1;
}

// tile positions (per channel / per 8-channel block) a thread can stage: bound of the kernels' register prefetch
constexpr int stage_kpc(int ks, int s) {
  return s == 1 ? (ks == 1 ? 1 : ks == 3 ? 2 : ks == 5 ? 2 : 3) : (ks == 1 ? 4 : ks == 3 ? 5 : ks == 5 ? 5 : 6);
}

inline Geom choose_geom(const EssConvDesc* d) {
  Geom best{};
  double best_cost = 1e300;
  const int KS = d->ksize, S = d->stride;
  for (int bwl = 5; bwl >= 3; --bwl) {
    for (int wxl = 0; wxl <= 2; ++wxl) {
      const int BW = 1 << bwl, RB = 32 >> bwl;
      const int TW = BW << wxl, TH = (4 >> wxl) * NBW * RB;
      const int tx = ceil_div(d->W_out, TW), ty = ceil_div(d->H_out, TH);
      const int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
      if (IH * IW > stage_kpc(KS, S) * 256) continue;  // would not fit the staging registers
      // padded MACs (dominant) + a small halo/staging term; prefer wide blocks on ties
      const double cost = (double)tx * ty * TW * TH * (1.0 + 0.02 * (double)(IH * IW) / (TH * TW * S * S)) +
                          1e-3 * (5 - bwl);
      if (cost < best_cost) {
        best_cost = cost;
        Geom g{};
        g.bwl = bwl; g.wxl = wxl; g.TW = TW; g.TH = TH; g.tiles_x = tx; g.tiles_y = ty; g.IH = IH; g.IW = IW;
        if (is_bf16(d)) {
          // units: one 16-byte pixel vector (8 bf16 channels), read with ds_read_b128 (16-lane groups): the rows
          // of one pixel block must start 0 (BW=16) / 8 (BW=8) vectors apart modulo 16 to stay conflict-free
          if (S == 1) {
            int rp = IW;
            if (BW == 16) while (rp & 15) ++rp;
            if (BW == 8) while ((rp & 15) != 8) ++rp;
            g.row_pitch = rp; g.par_off = 0;
          } else {
            int pw = (IW + 1) / 2;
            if (BW == 16) while (pw & 3) ++pw;
            if (BW == 8) while ((pw & 3) != 2) ++pw;
            g.row_pitch = 2 * pw; g.par_off = pw;
          }
          g.plane = IH * g.row_pitch;
        } else {
          if (S == 1) {
            int rp = IW;
            if (BW < 32) while ((rp & 31) != BW) ++rp;  // rows of one pixel block land on disjoint banks
            g.row_pitch = rp; g.par_off = 0;
          } else {
            int pw = (IW + 1) / 2;
            if (BW < 32) while (((4 * pw) & 31) != BW) ++pw;
            g.row_pitch = 2 * pw; g.par_off = pw;
          }
          g.plane = (IH * g.row_pitch + 3) & ~3;
        }
        best = g;
      }
    }
  }
  return best;
}

inline int validate(const EssConvDesc* d) {
  ESS_CHECK_ARG(d != nullptr, "conv: null descriptor");
  ESS_CHECK_ARG(d->N > 0 && d->H_in > 0 && d->W_in > 0 && d->C0 > 0 && d->C1 >= 0 && d->C_out > 0, "conv: bad extents");
  ESS_CHECK_ARG(d->ksize == 1 || d->ksize == 3 || d->ksize == 5 || d->ksize == 7, "conv: ksize %d unsupported", d->ksize);
  ESS_CHECK_ARG(d->stride == 1 || d->stride == 2, "conv: stride %d unsupported", d->stride);
  ESS_CHECK_ARG(d->H_out == (d->H_in + 2 * d->pad - d->ksize) / d->stride + 1 &&
                    d->W_out == (d->W_in + 2 * d->pad - d->ksize) / d->stride + 1,
                "conv: output extent %dx%d inconsistent with input %dx%d k%d s%d p%d", d->H_out, d->W_out, d->H_in,
                d->W_in, d->ksize, d->stride, d->pad);
  for (int s = 0; s < 2; ++s) {
    const int m = s ? d->mode1 : d->mode0;
    ESS_CHECK_ARG(m >= 0 && m <= 2, "conv: bad source mode");
    if (m != ESS_SRC_DIRECT) ESS_CHECK_ARG(!(d->H_in & 1) && !(d->W_in & 1), "conv: x2 source needs even extent");
  }
  ESS_CHECK_ARG(d->epilogue >= 0 && d->epilogue <= 3, "conv: bad epilogue");
  ESS_CHECK_ARG(d->compute == ESS_COMPUTE_FP32 || d->compute == ESS_COMPUTE_BF16, "conv: bad compute type");
  if (d->epilogue == ESS_EPI_LSTM) ESS_CHECK_ARG(d->C_out == 4 * d->hidden, "conv: LSTM needs C_out = 4*hidden");
  if (d->epilogue == ESS_EPI_GRU_UR) ESS_CHECK_ARG(d->C_out == 2 * d->hidden, "conv: GRU_UR needs C_out = 2*hidden");
  if (d->epilogue == ESS_EPI_GRU_OUT) ESS_CHECK_ARG(d->C_out == d->hidden, "conv: GRU_OUT needs C_out = hidden");
  if (d->epilogue != ESS_EPI_LINEAR)
    ESS_CHECK_ARG(d->ksize == 3 && d->stride == 1 && d->out_split == 0, "conv: recurrent epilogues are 3x3 s1");
  ESS_CHECK_ARG(d->out_split >= 0 && d->out_split < d->C_out, "conv: bad out_split");
  return ESS_OK;
}

// rows of the packed weight matrix that exist for this epilogue (before padding to the tile)
inline int packed_rows(const EssConvDesc* d) {
  switch (d->epilogue) {
    case ESS_EPI_LSTM: return ceil_div(d->hidden, 8) * 32;
    case ESS_EPI_GRU_UR: return ceil_div(d->hidden, 16) * 32;
    default: return d->C_out;
  }
}

inline void make_plan(const EssConvDesc* d, EssConvPlan* pl) {
  const int mb = pick_mb(d);
  pl->cout_tile = mb * 32;
  pl->ck = pick_ck(d);
  pl->n_chunks = ceil_div(d->C0 + d->C1, pl->ck);
  pl->n_cout_tiles = ceil_div(packed_rows(d), pl->cout_tile);
  pl->rows_padded = pl->n_cout_tiles * pl->cout_tile;
  pl->packed_elems = (int64_t)pl->rows_padded * pl->n_chunks * pl->ck * d->ksize * d->ksize;
  const Geom g = choose_geom(d);
  if (is_bf16(d)) {
    pl->packed_bytes = pl->packed_elems * 2;
    pl->lds_bytes = ((pl->ck / 8) * g.plane + d->ksize * d->ksize * (pl->ck / 8) * pl->cout_tile) * 16;
  } else {
    pl->packed_bytes = pl->packed_elems * 4;
    pl->lds_bytes = (pl->ck * g.plane + d->ksize * d->ksize * pl->ck * pl->cout_tile) * 4;
  }
}

// packed row -> (source tensor selector, source row) or -1
__device__ __forceinline__ int map_row(int prow, int epi, int hid, int cout, int* sel) {
  *sel = 0;
  if (epi == ESS_EPI_LSTM) {
    const int b = prow >> 5, q = (prow & 31) >> 3, j = prow & 7;
    const int hc = b * 8 + j;
    return hc < hid ? q * hid + hc : -1;
  }
  if (epi == ESS_EPI_GRU_UR) {
    const int b = prow >> 5, q = (prow & 31) >> 3, j = prow & 7;
    const int hc = b * 16 + (q >> 1) * 8 + j;
    *sel = q & 1;
    return hc < hid ? hc : -1;
  }
  return prow < cout ? prow : -1;
}

static __global__ void pack_weights_kernel(const float* w, const float* w2, float* out, int64_t total, int cot, int ck,
                                    int n_chunks, int ks, int cin, int cout, int epi, int hid, int w_kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int64_t t = i;
  const int col = t % cot; t /= cot;
  const int cb = t % ck; t /= ck;
  const int tap = t % (ks * ks); t /= ks * ks;
  const int ch = t % n_chunks;
  const int ct = t / n_chunks;
  const int c = ch * ck + cb;
  int sel;
  const int row = map_row(ct * cot + col, epi, hid, cout, &sel);
  float v = 0.f;
  if (row >= 0 && c < cin) {
    const int ky = tap / ks, kx = tap - ky * ks;
    const float* src = sel ? w2 : w;
    if (w_kind == ESS_W_CONV) {
      v = src[(((size_t)row * cin + c) * ks + ky) * ks + kx];
    } else {
      const int rows_src = (epi == ESS_EPI_GRU_UR) ? hid : cout;
      v = src[(((size_t)c * rows_src + row) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
    }
  }
  out[i] = v;
}

static __global__ void pack_rows_kernel(const float* v, const float* v2, float fill, float* out, int rows_padded, int epi,
                                 int hid, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_padded) return;
  int sel;
  const int row = map_row(i, epi, hid, cout, &sel);
  out[i] = row >= 0 ? (sel ? v2[row] : v[row]) : fill;
}


// bf16-MFMA variant (conv_bf16.hip)
int conv_bf16_launch(const EssConvDesc* d, const EssConvPlan& pl, const Geom& g, const ConvKArgs& a, hipStream_t st);
int conv_bf16_pack_weights(const EssConvDesc* d, const EssConvPlan& pl, int w_kind, const float* w, const float* w2, void* packed,
                           hipStream_t st);

}  // namespace essconv
