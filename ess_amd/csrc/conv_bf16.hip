// Direct convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// Same decomposition as conv_fwd.hip (A = weights, B = activations, 4 waves x 2 pixel blocks x MB channel blocks),
// with the MFMA K-step now 16 input channels: each lane feeds 8 consecutive channels of one pixel (B) / one output
// channel (A) as a 16-byte bf16x8 fragment.  Tensors stay NCHW fp32 in HBM; the conversion to bf16 and the
// channel-interleaving ("pixel vector" = 8 channels, 16 B) happen while the tile is staged into LDS, so one
// ds_read_b128 per fragment feeds the matrix core and consecutive lanes (= consecutive pixels) read consecutive
// 16-byte slots: conflict-free without a swizzle.  Weights are packed once to bf16 [tile][chunk][tap][c/8][cout][8].
// Staging is software-pipelined: the global loads of chunk i+1 are issued before the MFMA phase of chunk i and
// written to LDS after it, so HBM/L2 latency hides under the matrix work (one LDS buffer, two barriers per chunk).
#include "conv_common.h"

namespace {

using namespace essconv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
  return __builtin_bit_cast(u32x4, b);
}

// pixel vectors a thread stages per chunk (compile-time bound of the register prefetch), by filter geometry
constexpr int maxv(int ks, int s) {
  return s == 1 ? (ks == 1 ? 4 : ks == 3 ? 3 : ks == 5 ? 4 : 5) : (ks == 1 ? 8 : ks == 3 ? 9 : ks == 5 ? 10 : 12);
}

template <int KS, int S, int MB, int EPI>
__global__ __launch_bounds__(256) void conv_bf16_kernel(const ConvKArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  constexpr int COT = MB * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int BW = 1 << a.bwl, WX = 1 << a.wxl, RB = 32 >> a.bwl;
  const int TW = WX << a.bwl, TH = (4 >> a.wxl) * NBW * RB;
  const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int ct = blockIdx.y, n = blockIdx.z;
  const int CB8 = a.ck >> 3;
  u32x4* in_t = smem16;
  u32x4* w_t = smem16 + CB8 * a.plane;
  const int wsz = KS * KS * CB8 * COT;  // 16-byte units

  const int ox = p & (BW - 1), oy = p >> a.bwl;
  const int wx = wave & (WX - 1), wy = wave >> a.wxl;
  const int lx = wx * BW + ox;
  int ly[NBW], boff[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    ly[nb] = (wy * NBW + nb) * RB + oy;
    boff[nb] = half * a.plane + ly[nb] * S * a.row_pitch + lx;
  }

  f32x16 acc[MB][NBW];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  constexpr int MAXV = maxv(KS, S);
  const int iy0 = y0 * S - a.pad, ix0 = x0 * S - a.pad;
  // The input tile is a list of CB8*IH*IW pixel vectors; thread t stages vectors t, t+256, ... (lanes run along x, so
  // each of the 8 per-channel loads of a vector is a coalesced row segment).  Everything that does not depend on the
  // chunk index is resolved once: LDS slot, channel block, and the element offset inside a channel plane of either
  // source (-1: outside the image / a zero of the zero-insert mode).
  const int nvec = CB8 * a.IH * a.IW;
  const int sh0 = a.mode0 != ESS_SRC_DIRECT ? 1 : 0, sh1 = a.mode1 != ESS_SRC_DIRECT ? 1 : 0;
  const int Wp0 = a.Win >> sh0, Wp1 = a.Win >> sh1;
  const size_t pl0 = (size_t)(a.Hin >> sh0) * Wp0, pl1 = (size_t)(a.Hin >> sh1) * Wp1;
  const float* s0 = a.src0 + (size_t)n * a.C0 * pl0;
  const float* s1 = a.C1 ? a.src1 + (size_t)n * a.C1 * pl1 : nullptr;
  int v_lds[MAXV], v_cb[MAXV], v_o0[MAXV], v_o1[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = tid + i * 256;
    const int cb = vi / (a.IH * a.IW);
    const int r = vi - cb * a.IH * a.IW;
    const int iy = r / a.IW, ix = r - iy * a.IW;
    const int gy = iy0 + iy, gx = ix0 + ix;
    const bool in = vi < nvec && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
    const bool odd = ((gy | gx) & 1) != 0;
    v_cb[i] = cb;
    v_lds[i] = vi < nvec ? cb * a.plane + iy * a.row_pitch + (S == 2 ? (ix & 1) * a.par_off + (ix >> 1) : ix) : -1;
    v_o0[i] = (in && !(a.mode0 == ESS_SRC_ZERO_UP2 && odd)) ? (gy >> sh0) * Wp0 + (gx >> sh0) : -1;
    v_o1[i] = (in && !(a.mode1 == ESS_SRC_ZERO_UP2 && odd)) ? (gy >> sh1) * Wp1 + (gx >> sh1) : -1;
  }
  // 8 consecutive channels of one position -> bf16x8 (a vector never straddles the two sources: C0 % 8 == 0)
  auto load_vec = [&](int ch, int i) -> u32x4 {
    const int c0 = ch * a.ck + v_cb[i] * 8;
    const bool first = c0 < a.C0;
    const int off = first ? v_o0[i] : v_o1[i];
    const size_t pls = first ? pl0 : pl1;
    const int cc = first ? c0 : c0 - a.C0;
    const int lim = (first ? a.C0 : a.C1) - cc;  // channels left in this source
    const float* sp = (first ? s0 : s1) + (size_t)cc * pls + off;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (off >= 0 && j < lim) ? sp[(size_t)j * pls] : 0.f;
    return pack8(v);
  };

  u32x4 pre[MAXV];
  const u32x4* wbase = (const u32x4*)a.wpk + (size_t)ct * a.n_chunks * wsz;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (v_lds[i] >= 0) pre[i] = load_vec(0, i);

  for (int ch = 0; ch < a.n_chunks; ++ch) {
    __syncthreads();  // previous chunk's fragments have been read
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (v_lds[i] >= 0) in_t[v_lds[i]] = pre[i];
    {
      const u32x4* wsrc = wbase + (size_t)ch * wsz;
      for (int i = tid; i < wsz; i += 256) w_t[i] = wsrc[i];
    }
    __syncthreads();
    // prefetch the next chunk's activations; they land while the matrix cores work on this one
    if (ch + 1 < a.n_chunks) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (v_lds[i] >= 0) pre[i] = load_vec(ch + 1, i);
    }
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int tap = ky * KS + kx;
        const int toff = ky * a.row_pitch + (S == 2 ? (kx & 1) * a.par_off + (kx >> 1) : kx);
        const u32x4* wp = w_t + (tap * CB8 + half) * COT + p;
        const u32x4* ip = in_t + toff;
        for (int kk = 0; kk < CB8; kk += 2) {
          bf16x8 af[MB], bfr[NBW];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) af[mb] = __builtin_bit_cast(bf16x8, wp[kk * COT + mb * 32]);
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb) bfr[nb] = __builtin_bit_cast(bf16x8, ip[boff[nb] + kk * a.plane]);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);
        }
      }
    }
  }
  conv_epilogue<MB, EPI>(a, acc, ct, n, half, x0 + lx, y0, ly);
}

__global__ void pack_weights_bf16_kernel(const float* w, const float* w2, __bf16* out, int64_t total, int cot, int ck,
                                         int n_chunks, int ks, int cin, int cout, int epi, int hid, int w_kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int64_t t = i;
  const int kp = t & 7; t >>= 3;
  const int col = t % cot; t /= cot;
  const int cb8 = ck >> 3;
  const int cb = t % cb8; t /= cb8;
  const int tap = t % (ks * ks); t /= ks * ks;
  const int ch = t % n_chunks;
  const int ct = t / n_chunks;
  const int c = ch * ck + cb * 8 + kp;
  int sel;
  const int row = map_row(ct * cot + col, epi, hid, cout, &sel);
  float v = 0.f;
  if (row >= 0 && c < cin) {
    const int ky = tap / ks, kx = tap - ky * ks;
    const float* src = sel ? w2 : w;
    if (w_kind == ESS_W_CONV) v = src[(((size_t)row * cin + c) * ks + ky) * ks + kx];
    else v = src[(((size_t)c * cout + row) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
  }
  out[i] = (__bf16)v;
}

template <int KS, int S, int MB>
void launch_epi(int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if constexpr (KS == 3 && S == 1) {
    switch (epi) {
      case ESS_EPI_LSTM: { ess_allow_lds(conv_bf16_kernel<KS, S, MB, ESS_EPI_LSTM>, lds); hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MB, ESS_EPI_LSTM>), grid, dim3(256), lds, st, a); } return;
      case ESS_EPI_GRU_UR: { ess_allow_lds(conv_bf16_kernel<KS, S, MB, ESS_EPI_GRU_UR>, lds); hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MB, ESS_EPI_GRU_UR>), grid, dim3(256), lds, st, a); } return;
      case ESS_EPI_GRU_OUT: { ess_allow_lds(conv_bf16_kernel<KS, S, MB, ESS_EPI_GRU_OUT>, lds); hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MB, ESS_EPI_GRU_OUT>), grid, dim3(256), lds, st, a); } return;
      default: break;
    }
  }
  { ess_allow_lds(conv_bf16_kernel<KS, S, MB, ESS_EPI_LINEAR>, lds); hipLaunchKernelGGL((conv_bf16_kernel<KS, S, MB, ESS_EPI_LINEAR>), grid, dim3(256), lds, st, a); }
}

template <int KS, int S>
void launch_mb(int mb, int epi, dim3 grid, size_t lds, hipStream_t st, const ConvKArgs& a) {
  if (mb == 2) launch_epi<KS, S, 2>(epi, grid, lds, st, a);
  else launch_epi<KS, S, 1>(epi, grid, lds, st, a);
}

}  // namespace

namespace essconv {

int conv_bf16_pack_weights(const EssConvDesc* d, const EssConvPlan& pl, int w_kind, const float* w, const float* w2, void* packed,
                           hipStream_t st) {
  const int64_t total = pl.packed_elems;
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, w, w2, (__bf16*)packed,
                     total, pl.cout_tile, pl.ck, pl.n_chunks, d->ksize, d->C0 + d->C1, d->C_out, d->epilogue, d->hidden, w_kind);
  return ess_launch_status("pack_weights_bf16");
}

int conv_bf16_launch(const EssConvDesc* d, const EssConvPlan& pl, const Geom& g, const ConvKArgs& a, hipStream_t st) {
  const int nvec = (pl.ck / 8) * g.IH * g.IW;
  ESS_CHECK_ARG(nvec <= maxv(d->ksize, d->stride) * 256, "conv(bf16): input tile of %d pixel vectors exceeds the staging capacity",
                nvec);
  ESS_CHECK_ARG(d->C1 == 0 || (d->C0 % 8) == 0, "conv(bf16): the first source of a concat must have a multiple of 8 channels");
  const dim3 grid(g.tiles_x * g.tiles_y, pl.n_cout_tiles, d->N);
  const int mb = pl.cout_tile / 32;
  const int key = d->ksize * 10 + d->stride;
  switch (key) {
    case 11: launch_mb<1, 1>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 12: launch_mb<1, 2>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 31: launch_mb<3, 1>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 32: launch_mb<3, 2>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 51: launch_mb<5, 1>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 52: launch_mb<5, 2>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 71: launch_mb<7, 1>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    case 72: launch_mb<7, 2>(mb, d->epilogue, grid, pl.lds_bytes, st, a); break;
    default: ess_set_error("conv: no kernel for k%d s%d", d->ksize, d->stride); return ESS_ENOTSUP;
  }
  return ess_launch_status("conv2d_forward(bf16)");
}

}  // namespace essconv
