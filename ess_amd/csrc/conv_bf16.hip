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
#include "conv_bf16_common.h"
#include <atomic>

namespace {

using namespace essconv;

// split (ESS_COMPUTE_BF16X3): every (tile, chunk) block is followed by a second one holding lo = bf16(w - float(bf16(w)))
__global__ void pack_weights_bf16_kernel(const float* w, const float* w2, __bf16* out, int64_t total, int cot, int ck,
                                         int n_chunks, int ks, int cin, int cout, int epi, int hid, int w_kind, int split, int f16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int64_t t = i;
  const int kp = t & 7; t >>= 3;
  const int col = t % cot; t /= cot;
  const int cb8 = ck >> 3;
  const int cb = t % cb8; t /= cb8;
  const int tap = t % (ks * ks); t /= ks * ks;
  int lo = 0;
  if (split) { lo = (int)(t & 1); t >>= 1; }
  const int ch = t % n_chunks;
  const int ct = t / n_chunks;
  const int c = ch * ck + cb * 8 + kp;
  int sel;
  const int row = map_row(ct * cot + col, epi, hid, cout, &sel);
  float v = 0.f;
  if (w_kind == ESS_W_CONV5_S2D) {
    // the space-to-depth form of a 5x5 / stride-2 convolution (conv_bf16_wide.hip, ESS_SRC_S2D): virtual channel c = q * cin5 + cr
    // of parity class q = py + 2 px; `tap` is the SLOT of the chunk's slab -- a class keeps its used taps first
    // chunk position c / 16 -> (class, 16-channel group): conv_bf16_wide.hip s2d_class / s2d_group (column parities of a row parity adjacent)
    const int cin5 = cin >> 2, nq = cin5 >> 4, pos = c >> 4;
    const int q = pos < 2 * nq ? ((pos & 1) ? 2 : 0) : ((pos & 1) ? 3 : 1);
    const int cr = (((pos < 2 * nq ? pos : pos - 2 * nq) >> 1) << 4) + (c & 15), py = q & 1, px = q >> 1;
    int t3 = tap;  // slot -> tap (ty * 3 + tx)
    if (q == 2) { const int m[9] = {0, 1, 3, 4, 6, 7, 2, 5, 8}; t3 = m[tap]; }
    if (q == 3) { const int m[9] = {0, 1, 3, 4, 2, 5, 6, 7, 8}; t3 = m[tap]; }
    const int ky = 2 * (t3 / 3) + py, kx = 2 * (t3 % 3) + px;
    if (row >= 0 && ky < 5 && kx < 5) v = w[(((size_t)row * cin5 + cr) * 5 + ky) * 5 + kx];
  } else if (row >= 0 && c < cin) {
    const int ky = tap / ks, kx = tap - ky * ks;
    const float* src = sel ? w2 : w;
    if (w_kind == ESS_W_CONV) v = src[(((size_t)row * cin + c) * ks + ky) * ks + kx];
    else v = src[(((size_t)c * cout + row) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
  }
  if (f16) { ((_Float16*)out)[i] = ess_f16_sat(v); return; }  // (ESS_COMPUTE_F16: half weights, no split form)
  const __bf16 hi = (__bf16)v;
  out[i] = lo ? (__bf16)(v - (float)hi) : hi;
}

// weights for the tap-paired kernel: [tile][chunk of 8 channels][pair][half][cout][8]
__global__ void pack_weights_bf16_pair_kernel(const float* w, __bf16* out, int64_t total, int cot, int n_chunks, int ks, int cin,
                                              int cout, int w_kind, int split, int f16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int nt = ks * ks, np = (nt + 1) / 2;
  int64_t t = i;
  const int kp = t & 7; t >>= 3;
  const int col = t % cot; t /= cot;
  const int hf = t & 1; t >>= 1;
  const int pr = t % np; t /= np;
  int lo = 0;
  if (split) { lo = (int)(t & 1); t >>= 1; }  // [tile][chunk][hi | lo][pair][half][cout][8]
  const int ch = t % n_chunks;
  const int ct = t / n_chunks;
  const int tap = 2 * pr + hf, c = ch * 8 + kp, row = ct * cot + col;
  float v = 0.f;
  if (tap < nt && row < cout && c < cin) {
    const int ky = tap / ks, kx = tap - ky * ks;
    if (w_kind == ESS_W_CONV) v = w[(((size_t)row * cin + c) * ks + ky) * ks + kx];
    else v = w[(((size_t)c * cout + row) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
  }
  if (f16) { ((_Float16*)out)[i] = ess_f16_sat(v); return; }
  const __bf16 hi = (__bf16)v;
  out[i] = lo ? (__bf16)(v - (float)hi) : hi;
}
}  // namespace

namespace essconv {

int conv_bf16_pack_weights(const EssConvDesc* d, const EssConvPlan& pl, int w_kind, const float* w, const float* w2, void* packed,
                           hipStream_t st, bool split, bool f16) {
  const int64_t total = pl.packed_elems;
  ESS_CHECK_ARG(!(split && f16), "pack_weights: no split-operand form of half weights");
  if (is_paired(d)) {
    hipLaunchKernelGGL(pack_weights_bf16_pair_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, w, (__bf16*)packed, total,
                       pl.cout_tile, pl.n_chunks, d->ksize, d->C0 + d->C1, d->C_out, w_kind, split ? 1 : 0, f16 ? 1 : 0);
    return ess_launch_status("pack_weights_bf16(paired)");
  }
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, w, w2, (__bf16*)packed,
                     total, pl.cout_tile, pl.ck, pl.n_chunks, d->ksize, d->C0 + d->C1, d->C_out, d->epilogue, d->hidden, w_kind, split ? 1 : 0, f16 ? 1 : 0);
  return ess_launch_status("pack_weights_bf16");
}

// ---- many weight tensors in one launch (re-packing every trainable convolution after an optimiser step: ~65 tensors of a few
// hundred KB each, 5 us per single launch).  The job table travels in the kernel arguments; a block finds its job by a scan
// over the (wave-uniform) block ranges.  LINEAR layouts of pack_weights_bf16_kernel only.
struct PackJob {
  const float* w;
  __bf16* out;
  long long total;
  int blk_end;  // exclusive end of this job's block range
  int cot, ck, n_chunks, ks, cin, cout, w_kind;
  int f16;  // ESS_COMPUTE_F16: half elements
};
constexpr int PACK_JOBS = 48;
struct PackJobs { PackJob j[PACK_JOBS]; int count; };

// One 16-byte output vector (the 8 input channels of a (tap, block, row) slot) per thread: the index arithmetic of a slot -- five
// integer divisions -- is paid once per eight elements and the store is one coalesced 16-byte write; the eight source floats sit
// ks * ks floats apart ([row][c][ky][kx]), the taps of the same (row, c) are picked up by neighbouring slots out of the same lines.
// (One ELEMENT per thread, round 3's form: 61 us per launch of the decoder's ~40 tensors, 1.0 TB/s -- bound by its divisions.)
__global__ void pack_weights_bf16_multi_kernel(const PackJobs jobs) {
  int k = 0;
  while (k + 1 < jobs.count && (int)blockIdx.x >= jobs.j[k].blk_end) ++k;
  const PackJob& jb = jobs.j[k];
  const int blk0 = k ? jobs.j[k - 1].blk_end : 0;
  const long long v = (long long)(blockIdx.x - blk0) * blockDim.x + threadIdx.x;  // vector (rows jobs: element) index
  if (jb.w_kind == ESS_W_ROWS) {  // a bias vector, tile-padded with zeros (LINEAR: packed row = output channel)
    if (v < jb.total) ((float*)jb.out)[v] = v < jb.cout ? jb.w[v] : 0.f;
    return;
  }
  if (v * 8 >= jb.total) return;
  long long t = v;
  const int col = t % jb.cot; t /= jb.cot;
  const int cb8 = jb.ck >> 3;
  const int cb = t % cb8; t /= cb8;
  const int kk = jb.ks * jb.ks;
  const int tap = t % kk; t /= kk;
  const int ch = t % jb.n_chunks;
  const int ct = t / jb.n_chunks;
  const int c0 = ch * jb.ck + cb * 8;
  const int row = ct * jb.cot + col;  // LINEAR: packed row = output channel
  const int ky = tap / jb.ks, kx = tap - ky * jb.ks;
  float f[8];
#pragma unroll
  for (int kp = 0; kp < 8; ++kp) {
    const int c = c0 + kp;
    float x = 0.f;
    if (row < jb.cout && c < jb.cin) {
      if (jb.w_kind == ESS_W_CONV) x = jb.w[(((size_t)row * jb.cin + c) * jb.ks + ky) * jb.ks + kx];
      else x = jb.w[(((size_t)c * jb.cout + row) * jb.ks + (jb.ks - 1 - ky)) * jb.ks + (jb.ks - 1 - kx)];
    }
    f[kp] = x;
  }
  ((u32x4*)jb.out)[v] = jb.f16 ? pack8h(f) : pack8(f);
}

// returns ESS_EINVAL (nothing launched) when a descriptor is not a plain bf16 LINEAR layout
int conv_bf16_pack_weights_multi(const EssConvDesc* descs, const int32_t* kinds, const float* const* w, void* const* packed, int count,
                                 hipStream_t st) {
  for (int i = 0; i < count; ++i) {
    int rc = validate(&descs[i]);
    if (rc) return rc;
    const ResolvedDesc rd = resolve_compute(&descs[i]);  // (ESS_COMPUTE_F16 -> the bf16 layouts with half elements)
    ESS_CHECK_ARG(descs[i].compute != ESS_COMPUTE_BF16X3 && is_bf16(&rd.d) && descs[i].epilogue == ESS_EPI_LINEAR && w[i] && packed[i] &&
                      ((!is_paired(&rd.d) && (kinds[i] == ESS_W_CONV || kinds[i] == ESS_W_TRANSPOSED)) || kinds[i] == ESS_W_ROWS),
                  "pack_weights_multi: job %d is not a plain bf16 / f16 LINEAR layout", i);
  }
  for (int i0 = 0; i0 < count; i0 += PACK_JOBS) {
    PackJobs jobs{};
    int blocks = 0;
    jobs.count = count - i0 < PACK_JOBS ? count - i0 : PACK_JOBS;
    for (int k = 0; k < jobs.count; ++k) {
      const ResolvedDesc rd = resolve_compute(&descs[i0 + k]);
      const EssConvDesc* d = &rd.d;
      EssConvPlan pl;
      make_plan(d, &pl);
      PackJob& jb = jobs.j[k];
      jb.f16 = rd.f16 ? 1 : 0;
      jb.w = w[i0 + k]; jb.out = (__bf16*)packed[i0 + k];
      jb.total = kinds[i0 + k] == ESS_W_ROWS ? pl.rows_padded : pl.packed_elems;
      jb.cot = pl.cout_tile; jb.ck = pl.ck; jb.n_chunks = pl.n_chunks; jb.ks = d->ksize; jb.cin = d->C0 + d->C1; jb.cout = d->C_out;
      jb.w_kind = kinds[i0 + k];
      blocks += (int)ceil_div64(kinds[i0 + k] == ESS_W_ROWS ? jb.total : jb.total / 8, 256);  // (a weight job: one 16-byte vector per thread)
      jb.blk_end = blocks;
    }
    hipLaunchKernelGGL(pack_weights_bf16_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, jobs);
  }
  return ess_launch_status("pack_weights_multi");
}

// ---- wide-tile kernel (conv_bf16_wide.hip): chosen per launch by round count.
// Mode: ESS_CONV_WIDE = 0 off | 1 by the cost model (default) | 2 whenever a variant applies; `ess_tuning_set("conv_wide", v)`
// changes it at run time (tests compare the two kernels in one process: their results are bit-identical, both accumulate
// chunk by chunk, tap by tap in the same order).
static std::atomic<int> g_wide_mode{-1};
int wide_mode() {
  int m = g_wide_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("ESS_CONV_WIDE");
    m = e ? atoi(e) : 1;
    g_wide_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}
void set_wide_mode(int m) { g_wide_mode.store(m < 0 ? 0 : m, std::memory_order_relaxed); }

// ---- dispatcher constants, in ONE place, each with the shape it was measured at.  Read once (environment overrides at first use,
// then immutable); the compute-unit count comes from the device the calling thread is on (256 on MI355X), never from a literal.
//   constant      value  measured on (B = 8, bf16, MI355X)                                         meaning
//   tail_half     0.56   ws kernel, 256 -> 256 @ 60 x 80: a lone 64 x 256 tile 18.8 us vs a          cost of a ws-kernel tail round that fills at
//                        co-resident pair 33.8 us (round-3 wall-clock stamps)           most half of the resident slots
//   narrow_tile   1.35   ws kernel, 64^ -> 32 @ 480 x 640: 6.8 us per round of 32-channel tiles      32-channel ws tiles: a weight slab per 256
//                        vs 5.0 us for half a 64-channel round                                        pixels, one weight fragment per two MFMAs
//   kappa         1.12   wide vs ws at model parity on the multi-round decoder layers (120 x 160,     margin the wide kernel must win by (LINEAR):
//                        240 x 320): the wide kernel measures 4 - 18 % slower (tools/wide_probe.py)     one workgroup per CU exposes its epilogue
//   kappa_rec     0.9    lean ConvLSTM launches, levels 0 / 1 / 2 @ 480 x 640: + 8 / 5 / 11 % for     same for the recurrent epilogues (8 - 32
//                        the wide kernel                                                 chunks amortise the epilogue)
// The DDD17 shape (B = 2, 200 x 352: every launch is sub-round) is covered by an A/B of the switch in bench records
// (profiles/r5_bench_bf16_config2_ddd17*.json), not by constants of its own.
struct DispatchTuning { int cus; double tail_half, narrow_tile, kappa, kappa_rec; };
static const DispatchTuning& tuning() {
  static const DispatchTuning t = [] {
    DispatchTuning v{256, 0.56, 1.35, 1.12, 0.9};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      v.cus = cus;
    if (const char* e = getenv("ESS_WIDE_KAPPA")) v.kappa = atof(e);
    if (const char* e = getenv("ESS_WIDE_KAPPA_REC")) v.kappa_rec = atof(e);
    return v;
  }();
  return t;
}
int device_cus() { return tuning().cus; }

namespace {
// Relative launch time in units of (output channel x pixel) per CU: rounds of resident workgroups x tile work per CU; the ws
// kernel's tail round at tail_half when it is at most half full.  kappa scales the wide kernel's per-unit time against the ws kernel's.
double std_cost(int tiles, int cot, int per_cu) {
  const int slots = tuning().cus * per_cu;
  const int full = tiles / slots, rem = tiles % slots;
  const double tail = rem == 0 ? 0.0 : (rem * 2 <= slots ? tuning().tail_half : 1.0);
  return (full + tail) * per_cu * cot * 256.0;
}
struct WidePick { int mbw, cw, th, tiles_x, tiles_y, tiles; };
// the lean ConvLSTM step (BF16_C8 x / h, F32_C8 cell state in and out, bias in the accumulators: conv_epilogue_lstm_c8's form) on
// the 128 x 320 tile.  One workgroup per CU exposes its epilogue per TILE, so the wide kernel pays on long-K launches: measured on
// the lean launches themselves (trace_tmp-style A/B, B = 8, 480 x 640): level 0 386 -> 356 us, level 1 346 -> 328, level 2 (against the
// ws kernel's 128-row instance) 347 -> 311: + 8 / 5 / 11 % -- the long K loops (8 - 32 chunks) amortise the exposed epilogue, and on real
// data the wide tile's lower traffic per MFMA is worth more than its coarser rounds (power probe).  The round model
// decides with a margin of its own (ESS_WIDE_KAPPA_REC, 0.9: the three levels at B = 8 pass, a quarter-filled single round does not).
bool wide_pick_recurrent(const EssConvDesc* d, const EssConvPlan& pl, const Geom& g, const ConvKArgs& a, bool c8, WidePick* out) {
  const int mode = wide_mode();
  if (!mode || !c8 || !a.shift || a.scale || a.residual) return false;
  // the conditions under which the ws kernel takes the straight-line recurrent epilogues (conv_epilogue in conv_common.h), for a
  // wave that owns two 32-row blocks of a 128-row tile
  if (d->epilogue == ESS_EPI_LSTM) {
    if (a.fmt_out != ESS_FMT_F32_C8 || (a.aux0 && a.fmt_res != ESS_FMT_F32_C8)) return false;
  } else if (d->epilogue == ESS_EPI_GRU_UR) {
    if (!a.out || a.out2 || a.fmt_out != ESS_FMT_F32_C8 || (a.aux0 && a.fmt_res != ESS_FMT_F32_C8)) return false;
  } else if (d->epilogue == ESS_EPI_GRU_OUT) {
    if (a.fmt_res != ESS_FMT_F32_C8 || !a.aux1 || (a.out && a.fmt_out != ESS_FMT_F32_C8)) return false;
  } else {
    return false;
  }
  if (pl.cout_tile != 64 && pl.cout_tile != 128) return false;
  // ConvGRU: measured (B = 8, lean launches) the wide tile wins only where the ws kernel itself runs one workgroup per CU (its
  // 128-row instance, the deepest level: (update, reset) 197 -> 171 us, candidate 139 -> 96 us); on the large planes the pair moves
  // 14 + 12 bytes per hidden element and an exposed epilogue costs more than the rounds gain (level 0 / 1: 0.87 - 0.99 x)
  if (d->epilogue != ESS_EPI_LSTM && pl.cout_tile != 128 && mode < 2) return false;
  if (packed_rows(d) % 128) return false;  // every row of a workgroup's tile is a real gate row (hid % 32 / 64 / 128 by epilogue)
  const double kappa = tuning().kappa_rec;
  const int std_tiles = g.tiles_x * g.tiles_y * pl.n_cout_tiles * d->N;
  const double c_std = std_cost(std_tiles, pl.cout_tile, pl.cout_tile == 128 ? 1 : 2);
  int th, tw;
  conv_bf16_wide_tile(2, 2, &th, &tw);
  const int tx = ceil_div(d->W_out, tw), ty = ceil_div(d->H_out, th);
  if (pl.rows_padded % 128) return false;
  const int tiles = tx * ty * (pl.rows_padded / 128) * d->N;
  *out = WidePick{2, 2, th, tx, ty, tiles};
  return mode >= 2 || kappa * ceil_div(tiles, tuning().cus) * 128.0 * th * tw < c_std;
}

bool wide_pick(const EssConvDesc* d, const EssConvPlan& pl, const Geom& g, const ConvKArgs& a, bool c8, WidePick* out) {
  if (d->epilogue != ESS_EPI_LINEAR) return wide_pick_recurrent(d, pl, g, a, c8, out);
  const int mode = wide_mode();
  if (!mode || !c8 || a.fmt_out != ESS_FMT_BF16_C8 || d->epilogue != ESS_EPI_LINEAR || a.out_bf) return false;
  const bool relu = d->act == ESS_ACT_RELU, res = a.residual != nullptr;
  if (!(d->act == ESS_ACT_NONE || relu)) return false;
  if (a.out_f16 && (relu || res)) return false;
  if (a.f16 && res) return false;  // (the H instantiations of the wide-tile kernel carry no residual form)
  if (d->out_split > 0 && (relu || res || a.out_f16 || (d->out_split & 15) || a.scale)) return false;  // (conv_epilogue_c8_dgrad's form)
  // kappa: margin the wide kernel must win by.  Measured (tools/wide_probe.py, B = 8): per unit of work and round it runs exactly
  // as fast as the ws kernel (256 -> 256 @ 60 x 80: model 52.7 -> 42.2 us, measured 50.7 -> 42.4), but with one workgroup per CU
  // its epilogue is exposed in multi-round launches: at model parity (two rounds against 2.34) it measures 4-18 % slower.
  const double kappa = tuning().kappa;
  const int mb = pl.cout_tile / 32;
  const int std_tiles = g.tiles_x * g.tiles_y * pl.n_cout_tiles * d->N;
  // (32-channel tiles of the ws kernel stage a weight slab per 256 pixels and read one weight fragment per two MFMAs: measured
  // 6.8 us per round of 64^ -> 32 @ 480 x 640 against 5.0 for half a 64-channel round -- 1.35)
  const double c_std = std_cost(std_tiles, pl.cout_tile, 2) * (mb == 1 ? tuning().narrow_tile : 1.0);
  const int cands[4][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}};
  double best = 1e300;
  for (const auto& c : cands) {
    const int mbw = c[0], cw = c[1], cot = mbw * cw * 32;
    if ((mb == 2) != (cot >= 64) || mb > 2) continue;             // reads the plan's own weight pack (64- / 32-channel slabs)
    if (d->C_out % cot) continue;                                 // every channel of a workgroup's tile is real
    if (d->out_split > 0 && (d->out_split % (mbw * 32))) continue;  // (a wave's block pairs never straddle the split: % 16 above)
    int th, tw;
    conv_bf16_wide_tile(mbw, cw, &th, &tw);
    const int tx = ceil_div(d->W_out, tw), ty = ceil_div(d->H_out, th);
    const int tiles = tx * ty * (d->C_out / cot) * d->N;
    const double cost = kappa * ceil_div(tiles, tuning().cus) * (double)cot * th * tw;
    if (cost < best) { best = cost; *out = WidePick{mbw, cw, th, tx, ty, tiles}; }
  }
  if (best >= 1e300) return false;
  return mode >= 2 || best < c_std;
}
}  // namespace

// workgroup tile of the space-to-depth form (ESS_SRC_S2D): 128-channel tiles where the output channels allow and the round count is no
// worse, else 64-channel tiles; *cw = 0 when C_out is not a multiple of 64
void conv_bf16_s2d_pick(const EssConvDesc* d, int* cw_out, int* tiles_out, int* tx_out, int* ty_out) {
  const int cus = tuning().cus;
  double best = 1e300;
  *cw_out = 0; *tiles_out = 0; *tx_out = 0; *ty_out = 0;
  for (int cw = 2; cw >= 1; --cw) {
    const int cot = 64 * cw;
    if (d->C_out % cot) continue;
    int th, tw;
    conv_bf16_wide_tile(2, cw, &th, &tw);
    const int tx = ceil_div(d->W_out, tw), ty = ceil_div(d->H_out, th);
    const int tiles = tx * ty * (d->C_out / cot) * d->N;
    const double cost = (double)ceil_div(tiles, cus) * cot * th * tw;
    if (cost < best) { best = cost; *cw_out = cw; *tiles_out = tiles; *tx_out = tx; *ty_out = ty; }
  }
}
// Is the space-to-depth form the faster one for this launch?  One workgroup per CU on 64 x 640 / 128 x 320 tiles needs a launch that
// fills the chip: measured against the tap-paired kernel (32-channel x 256-pixel tiles, two workgroups per CU) on the encoder's three
// levels at B = 1 / 2 / 4 / 8 (tools/s2d_probe.py, profiles/r5_s2d_probe*.txt): 30 / 60 / 120 tiles 0.41 - 0.90 x, 240 tiles 1.09 - 1.22 x,
// 480 / 960 tiles 1.14 - 1.23 x -> from three quarters of the compute units' worth of tiles on.
bool conv_bf16_s2d_preferred(const EssConvDesc* d) {
  int cw, tiles, tx, ty;
  conv_bf16_s2d_pick(d, &cw, &tiles, &tx, &ty);
  return cw > 0 && 4 * tiles >= 3 * tuning().cus;
}

int conv_bf16_launch(const EssConvDesc* d, const EssConvPlan& pl, const Geom& g, const ConvKArgs& a, hipStream_t st) {
  ESS_CHECK_ARG(g.IH * g.IW <= kpc(d->ksize, d->stride) * 256, "conv(bf16): input tile of %d positions exceeds the staging capacity",
                g.IH * g.IW);
  ESS_CHECK_ARG((int64_t)(d->C0 > d->C1 ? d->C0 : d->C1) * d->H_in * d->W_in * 4 < (int64_t)1 << 31,
                "conv(bf16): one sample of a source must stay below 2 GiB (32-bit buffer offsets)");
  ESS_CHECK_ARG(d->C1 == 0 || (d->C0 % 8) == 0, "conv(bf16): the first source of a concat must have a multiple of 8 channels");
  const dim3 grid((unsigned)(g.tiles_x * g.tiles_y * pl.n_cout_tiles * d->N));
  const int mb = pl.cout_tile / 32;
  const bool c8 = a.fmt0 == ESS_FMT_BF16_C8;
  const bool split_generic = d->ksize == 3 && d->stride == 2 && d->epilogue == ESS_EPI_LINEAR;  // (resolve_compute's third class)
  ESS_CHECK_ARG(!a.split || (((ws_enabled() && d->ksize == 3 && d->stride == 1 && pl.ck == 16) || is_paired(d) || split_generic) && !c8 && a.fmt_out == ESS_FMT_F32_NCHW),
                "conv(bf16): split operands run on the 3x3 / stride-1 wave-specialised, the 5x5 tap-paired and the 3x3 / stride-2 generic kernels with fp32 tensors");
  if (c8) ESS_CHECK_ARG((((uintptr_t)a.src0 | (uintptr_t)a.src1) & 15) == 0, "conv(bf16): BF16_C8 sources must be 16-byte aligned");
  // ESS_COMPUTE_F16: F16_C8 sources through the kernels' H instantiations; fp32 NCHW sources only where a kernel rounds them to half itself
  if (a.f16)
    ESS_CHECK_ARG(c8 || (is_paired(d) && !a.residual && !a.split && conv_bf16_head_applies(d, pl)),
                  "conv(f16): sources must be F16_C8 tensors (fp32 NCHW only for the 2..5-channel 5x5 head)");
  if (a.f16) ESS_CHECK_ARG(!a.split && d->act != ESS_ACT_SUMPOOL2 && (a.fmt_out == ESS_FMT_F32_NCHW || d->out_split == 0 || d->epilogue != ESS_EPI_LINEAR),
                           "conv(f16): forward forms only");
  if (!c8 && !a.residual && conv_bf16_stem_applies(d, pl)) {  // 1-channel 7x7 / stride 2 stem: K = the filter rows
    conv_bf16_launch_stem(d, pl, st, a);
    return ess_launch_status("conv2d_forward(bf16, 7x7 stem)");
  }
  if (is_paired(d) && !c8 && !a.residual && !a.split && conv_bf16_head_applies(d, pl)) {  // 2-channel 5x5 head: K = the filter rows
    conv_bf16_launch_head(d, pl, st, a);
    return ess_launch_status("conv2d_forward(bf16, 5x5 head)");
  }
  if (is_paired(d)) {  // 5x5: tap-paired wave-specialised kernel (the plan and the weight pack are specific to it)
    const size_t lds2 = 2 * (size_t)pl.lds_bytes;
    ESS_CHECK_ARG(lds2 <= 160 * 1024, "conv(bf16, 5x5): two stages of %d B exceed the 160 KiB LDS", pl.lds_bytes);
    {  // persistent launch (see the 3x3 kernel below): one resident set of workgroups when the grid exceeds it
      static const int persist = [] { const char* e = getenv("ESS_WS_PERSIST"); return e ? atoi(e) : 1; }();
      const int by_regs = (d->stride == 2 && mb == 2) ? 1 : 2, by_lds = (int)((160 * 1024) / lds2);
      const int slots = tuning().cus * (by_regs < by_lds ? by_regs : by_lds);
      if (persist && (int)grid.x > slots) {
        ConvKArgs t = a;
        t.persist = 1;
        conv_bf16_launch_pair(d->stride, mb, c8, dim3((unsigned)slots), lds2, st, t);
        return ess_launch_status("conv2d_forward(bf16, tap-paired, persistent)");
      }
    }
    conv_bf16_launch_pair(d->stride, mb, c8, grid, lds2, st, a);
    return ess_launch_status("conv2d_forward(bf16, tap-paired)");
  }
  if (d->mode0 == ESS_SRC_S2D) {
    // the 5x5 / stride-2 convolution of a BF16_C8 tensor as a 3x3 over its space-to-depth view (validate() has checked the form):
    // the wide-tile kernel only; 128-channel workgroup tiles where the output channels allow and the round count is no worse
    ESS_CHECK_ARG(c8 && a.fmt_out == ESS_FMT_BF16_C8 && !a.out_bf && !a.residual && !a.out_f16 && pl.ck == 16 && pl.cout_tile == 64 && (!a.hilo || (d->C_out % 64) == 0) &&
                      (d->act == ESS_ACT_NONE || d->act == ESS_ACT_RELU),
                  "conv(bf16, S2D): BF16_C8 in and out, no residual / copy, act in {none, relu}");
    const int cus = tuning().cus;
    int best_cw = 0, best_tiles = 0, best_tx = 0, best_ty = 0;
    conv_bf16_s2d_pick(d, &best_cw, &best_tiles, &best_tx, &best_ty);
    ESS_CHECK_ARG(best_cw > 0, "conv(bf16, S2D): C_out %d is not a multiple of 64", d->C_out);
    ConvKArgs t = a;
    t.tiles_x = best_tx; t.n_tiles = best_tx * best_ty;
    t.persist = best_tiles > cus ? 1 : 0;
    t.slab = pl.cout_tile;
    conv_bf16_launch_wide(2, best_cw, d->epilogue, dim3((unsigned)(best_tiles > cus ? cus : best_tiles)), st, t);
    return ess_launch_status("conv2d_forward(bf16, wide tile, space-to-depth 5x5/s2)");
  }
  // one nearest-x2-upsampled BF16_C8 source, 3x3 / stride 1 / pad 1, BF16_C8 / F16_C8 output: the polyphase kernel (conv_bf16_poly.hip;
  // 16 instead of 36 tap products per source pixel).  ESS_CONV_POLY=0: the general kernels with the upsampling in the tile loader
  // (the rounding differs: four effective weights are rounded once each instead of nine weights)
  static const bool poly_on = [] { const char* e = getenv("ESS_CONV_POLY"); return !(e && e[0] == '0'); }();
  if (poly_on && ws_enabled() && c8 && a.fmt_out == ESS_FMT_BF16_C8 && d->ksize == 3 && d->stride == 1 && d->pad == 1 && pl.ck == 16 &&
      d->mode0 == ESS_SRC_NEAREST_UP2 && d->C1 == 0 && d->epilogue == ESS_EPI_LINEAR && d->out_split == 0 && !a.out_bf &&
      (d->act == ESS_ACT_NONE || d->act == ESS_ACT_RELU) && (d->C_out % 32) == 0 && (d->C0 % 16) == 0 && !(d->H_in & 1) && !(d->W_in & 1) &&
      (pl.cout_tile == 32 || pl.cout_tile == 64)) {
    int th, tw, max_cin;
    conv_bf16_poly_tile(&th, &tw, &max_cin);
    if (d->C0 <= max_cin) {  // (the kernel keeps the effective weights of every chunk in LDS)
    ConvKArgs t = a;
    t.Hin = d->H_in >> 1; t.Win = d->W_in >> 1;      // the stored (low-resolution) source
    // (t.Hout / t.Wout stay the full-resolution output extent; a workgroup's tile is th x tw LOW-resolution pixels = 2 th x 2 tw outputs)
    t.tiles_x = ceil_div(t.Win, tw);
    t.n_tiles = t.tiles_x * ceil_div(t.Hin, th);
    t.n_cout_tiles = d->C_out / 32;
    t.slab = pl.cout_tile;
    // one resident set of workgroups, each bound to ONE 32-channel output tile (its weights stay in LDS) and walking pixel tiles
    const int cus = tuning().cus, nct = t.n_cout_tiles, ptiles = t.n_tiles * d->N;
    int per_ct = cus / nct > 0 ? cus / nct : 1;
    if (per_ct > ptiles) per_ct = ptiles;
    t.persist = 1;
    conv_bf16_launch_poly(dim3((unsigned)(per_ct * nct)), st, t);
    return ess_launch_status("conv2d_forward(bf16, polyphase nearest-up2)");
    }
  }
  if (ws_enabled() && d->ksize == 3 && d->stride == 1 && pl.ck == 16) {
    const size_t lds2 = 2 * (size_t)pl.lds_bytes;  // double-buffered stages
    WidePick wp;
    if (wide_pick(d, pl, g, a, c8, &wp)) {
      ConvKArgs t = a;
      t.tiles_x = wp.tiles_x; t.n_tiles = wp.tiles_x * wp.tiles_y;
      const int cus = tuning().cus;  // one workgroup per CU
      t.persist = wp.tiles > cus ? 1 : 0;
      t.slab = pl.cout_tile;
      conv_bf16_launch_wide(wp.mbw, wp.cw, d->epilogue, dim3((unsigned)(wp.tiles > cus ? cus : wp.tiles)), st, t);
      return ess_launch_status("conv2d_forward(bf16, wide tile)");
    }
    if (lds2 <= 160 * 1024) {
      // persistent launch when the grid exceeds one resident set of workgroups (2 per CU by registers -- 1 for the 128-row
      // instance -- and by LDS): see the tile schedule in conv_bf16_ws.hip.  ESS_WS_PERSIST=0: one workgroup per tile (tuning).
      static const int persist = [] { const char* e = getenv("ESS_WS_PERSIST"); return e ? atoi(e) : 1; }();
      const int per_cu_lds = (int)((160 * 1024) / lds2), per_cu = (mb == 4 ? 1 : 2) < per_cu_lds ? (mb == 4 ? 1 : 2) : per_cu_lds;
      const int slots = tuning().cus * per_cu;
      if (persist && (int)grid.x > slots) {
        ConvKArgs t = a;
        t.persist = 1;
        conv_bf16_launch_ws(mb, d->epilogue, c8, dim3((unsigned)slots), lds2, st, t);
        return ess_launch_status("conv2d_forward(bf16, wave-specialised, persistent)");
      }
      conv_bf16_launch_ws(mb, d->epilogue, c8, grid, lds2, st, a);
      return ess_launch_status("conv2d_forward(bf16, wave-specialised)");
    }
  }
  const int key = d->ksize * 10 + d->stride;
  // (the generic tile kernel ignores a.split but would read a [tile][chunk][hi | lo] weight pack: refuse instead of computing garbage)
  ESS_CHECK_ARG(!a.split || split_generic, "conv(bf16): split operands reached the generic tile kernel (k%d s%d, %zu B of LDS for two stages)",
                d->ksize, d->stride, 2 * (size_t)pl.lds_bytes);
  if (c8)
    ESS_CHECK_ARG(d->epilogue == ESS_EPI_LINEAR && (key == 11 || key == 12 || key == 31 || key == 32),
                  "conv(bf16): the generic tile kernel stages BF16_C8 sources for 1x1 and 3x3 LINEAR convolutions only");
  ESS_CHECK_ARG(key == 11 || key == 12 || key == 31 || key == 32 || key == 51 || key == 52 || key == 71 || key == 72,
                "conv: no kernel for k%d s%d", d->ksize, d->stride);
  conv_bf16_launch_generic(key, mb, pl.ck / 8, d->epilogue, c8, grid, pl.lds_bytes, st, a);
  return ess_launch_status("conv2d_forward(bf16)");
}

}  // namespace essconv
