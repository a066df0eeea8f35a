// Image-branch augmentation on the device: the geometric + photometric core of the albumentations pipeline of
// datasets/cityscapes_loader.py:39-74 for a whole batch in ONE launch, instead of per-sample numpy / cv2 work in loader
// processes:  HorizontalFlip -> ShiftScaleRotate(rotate 0, constant border 0) -> PadIfNeeded (centred, constant 0) -> RandomCrop
// -> GaussNoise -> RandomBrightnessContrast -> uint8 quantisation -> ToTensor (/255), and for the label map the same geometry
// with nearest sampling followed by the id -> trainId table (utils/labels.py:123-127).  The random DECISIONS (which ops fire,
// their magnitudes, the crop offset) are drawn on the host per sample and arrive as a parameter row; the per-pixel noise is a
// counter-based hash of (seed, pixel) so that the oracle reproduces it.  albumentations and cv2 are third-party packages absent
// from /root/reference and from this image: their interpolation arithmetic is restated (bilinear with zero border, sample
// positions at pixel centres, round-half-up quantisation), i.e. PARITY UNPINNED against the libraries themselves.
// Round 4: the rest of the pipeline as a second stage (`ess_augment_perspective_filter`, below): Perspective(p=0.2) ->
// RandomBrightnessContrast -> OneOf(Sharpen, Blur(3), MotionBlur(3))(p=0.5) need the NEIGHBOURS of a pixel of the cropped, noisy
// image, so they cannot ride in the first stage's single pass: stage 1 then runs with alpha = 1 / beta = 0 and no id table, stage
// 2 applies them after the warp, in the reference's order.
#include "common.h"

namespace {

constexpr int NPARAM = 12;  // flip, scale, dx, dy, pad_top, pad_left, crop_y, crop_x, alpha, beta, sigma, seed

__device__ __forceinline__ unsigned hash32(unsigned x) {  // (lowbias32)
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ img, const int64_t* __restrict__ lab,
                                                      const float* __restrict__ params, const int64_t* __restrict__ lut,
                                                      float* __restrict__ out_img, int64_t* __restrict__ out_lab, int N, int Hs,
                                                      int Ws, int H, int W) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
    const float* p = params + (size_t)n * NPARAM;
    const bool flip = p[0] != 0.f;
    const float s = p[1], dx = p[2], dy = p[3];
    // output pixel -> padded image -> image after ShiftScaleRotate
    const int ys = y + (int)p[6] - (int)p[4], xs = x + (int)p[7] - (int)p[5];
    float v = 0.f;
    int64_t l = 0;
    if (ys >= 0 && ys < Hs && xs >= 0 && xs < Ws) {
      // inverse of  dst = s * (src - c) + c + d  about the image centre c = (size - 1) / 2
      const float cx = 0.5f * (Ws - 1), cy = 0.5f * (Hs - 1);
      float u = ((float)xs - cx - dx) / s + cx, w = ((float)ys - cy - dy) / s + cy;
      if (flip) u = (float)(Ws - 1) - u;
      const float uf = floorf(u), wf = floorf(w);
      const int x0 = (int)uf, y0 = (int)wf;
      const float fx = u - uf, fy = w - wf;
      auto at = [&](int yy, int xx) { return (yy >= 0 && yy < Hs && xx >= 0 && xx < Ws) ? img[((size_t)n * Hs + yy) * Ws + xx] : 0.f; };
      const float top = at(y0, x0) * (1.f - fx) + at(y0, x0 + 1) * fx, bot = at(y0 + 1, x0) * (1.f - fx) + at(y0 + 1, x0 + 1) * fx;
      v = floorf(top * (1.f - fy) + bot * fy + 0.5f);
      const int xn = (int)floorf(u + 0.5f), yn = (int)floorf(w + 0.5f);
      if (lab && xn >= 0 && xn < Ws && yn >= 0 && yn < Hs) l = lab[((size_t)n * Hs + yn) * Ws + xn];
    }
    const float sigma = p[10];
    if (sigma > 0.f) {  // GaussNoise: N(0, sigma^2) on the 0..255 scale, Box-Muller over two hashed uniforms
      const unsigned seed = (unsigned)p[11];
      const unsigned h1 = hash32(seed ^ hash32((unsigned)i * 2u + 1u)), h2 = hash32(seed + 0x9e3779b9U + hash32((unsigned)i * 2u + 2u));
      const float u1 = ((float)(h1 >> 8) + 1.f) * (1.f / 16777216.f), u2 = (float)(h2 >> 8) * (1.f / 16777216.f);
      v = floorf(fminf(fmaxf(v + sigma * sqrtf(-2.f * __logf(u1)) * __cosf(6.28318530718f * u2), 0.f), 255.f) + 0.5f);
    }
    v = floorf(fminf(fmaxf(p[8] * v + p[9], 0.f), 255.f) + 0.5f);  // RandomBrightnessContrast: alpha * v + beta (beta in levels)
    out_img[i] = v * (1.f / 255.f);
    if (out_lab) out_lab[i] = lut ? lut[l < 0 ? 0 : (l > 255 ? 255 : l)] : l;
  }
}

// ---- stage 2 ------------------------------------------------------------------------------------------------------------------
// Parameter row (NP2 floats): [0] perspective fired; [1..9] the INVERSE homography (row-major; warped pixel (x, y) of the
// max_w x max_h rectangle -> source position in the H x W image: what cv2.warpPerspective computes from the forward matrix);
// [10] max_w, [11] max_h (the rectangle the quadrilateral is mapped to, albumentations Perspective with keep_size: the warp
// is followed by a bilinear resize back to H x W); [12] alpha, [13] beta (brightness / contrast, beta in levels); [14] stencil
// fired; [15..23] the 3 x 3 correlation kernel (Sharpen / box blur / motion-blur line, normalised by the host).
constexpr int NP2 = 24;

__device__ __forceinline__ float lvl(const float* __restrict__ img, int H, int W, int y, int x) {  // 0..255 level of a stage-1 pixel, zero border
  return (y >= 0 && y < H && x >= 0 && x < W) ? floorf(img[(size_t)y * W + x] * 255.f + 0.5f) : 0.f;
}

// one pixel of cv2.warpPerspective(img, M, (max_w, max_h), INTER_LINEAR, BORDER_CONSTANT 0), restated in float arithmetic
__device__ __forceinline__ float warp_px(const float* __restrict__ img, const float* __restrict__ m, int H, int W, int xi, int yi) {
  const float X = (float)xi, Y = (float)yi;
  const float w = m[6] * X + m[7] * Y + m[8];
  const float iw = w != 0.f ? 1.f / w : 0.f;
  const float u = (m[0] * X + m[1] * Y + m[2]) * iw, v = (m[3] * X + m[4] * Y + m[5]) * iw;
  if (!(u > -1.f && u < (float)W && v > -1.f && v < (float)H)) return 0.f;
  const float uf = floorf(u), vf = floorf(v);
  const int x0 = (int)uf, y0 = (int)vf;
  const float fx = u - uf, fy = v - vf;
  const float top = lvl(img, H, W, y0, x0) * (1.f - fx) + lvl(img, H, W, y0, x0 + 1) * fx;
  const float bot = lvl(img, H, W, y0 + 1, x0) * (1.f - fx) + lvl(img, H, W, y0 + 1, x0 + 1) * fx;
  return floorf(top * (1.f - fy) + bot * fy + 0.5f);
}

// Perspective (warp to the max_w x max_h rectangle, bilinear resize back to H x W with cv2's pixel-centre convention and
// replicated edges; label: nearest warp, nearest resize) + brightness / contrast -> levels (fp32 0..255), labels
__global__ __launch_bounds__(256) void augment_warp_kernel(const float* __restrict__ img, const int64_t* __restrict__ lab,
                                                           const float* __restrict__ params, float* __restrict__ mid,
                                                           int64_t* __restrict__ out_lab, const int64_t* __restrict__ lut, int N, int H, int W) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
    const float* p = params + (size_t)n * NP2;
    const float* im = img + (size_t)n * H * W;
    float v;
    int64_t l = 0;
    if (p[0] != 0.f) {
      const int mw = (int)p[10], mh = (int)p[11];
      const float sx = ((float)x + 0.5f) * ((float)mw / (float)W) - 0.5f, sy = ((float)y + 0.5f) * ((float)mh / (float)H) - 0.5f;
      float xf = floorf(sx), yf = floorf(sy);
      float fx = sx - xf, fy = sy - yf;
      int x0 = (int)xf, y0 = (int)yf;
      if (x0 < 0) { x0 = 0; fx = 0.f; }
      if (y0 < 0) { y0 = 0; fy = 0.f; }
      if (x0 >= mw - 1) { x0 = mw - 1; fx = 0.f; }
      if (y0 >= mh - 1) { y0 = mh - 1; fy = 0.f; }
      const int x1 = x0 + 1 < mw ? x0 + 1 : mw - 1, y1 = y0 + 1 < mh ? y0 + 1 : mh - 1;
      const float top = warp_px(im, p + 1, H, W, x0, y0) * (1.f - fx) + warp_px(im, p + 1, H, W, x1, y0) * fx;
      const float bot = warp_px(im, p + 1, H, W, x0, y1) * (1.f - fx) + warp_px(im, p + 1, H, W, x1, y1) * fx;
      v = floorf(top * (1.f - fy) + bot * fy + 0.5f);
      if (lab) {
        int xn = (int)floorf((float)x * ((float)mw / (float)W)), yn = (int)floorf((float)y * ((float)mh / (float)H));
        xn = xn < mw - 1 ? xn : mw - 1; yn = yn < mh - 1 ? yn : mh - 1;
        const float* m = p + 1;
        const float w = m[6] * xn + m[7] * yn + m[8];
        const float iw = w != 0.f ? 1.f / w : 0.f;
        const int us = (int)floorf((m[0] * xn + m[1] * yn + m[2]) * iw + 0.5f), vs = (int)floorf((m[3] * xn + m[4] * yn + m[5]) * iw + 0.5f);
        if (us >= 0 && us < W && vs >= 0 && vs < H) l = lab[((size_t)n * H + vs) * W + us];
      }
    } else {
      v = lvl(im, H, W, y, x);
      if (lab) l = lab[i];
    }
    mid[i] = floorf(fminf(fmaxf(p[12] * v + p[13], 0.f), 255.f) + 0.5f);
    if (out_lab) out_lab[i] = lut ? lut[l < 0 ? 0 : (l > 255 ? 255 : l)] : l;
  }
}

// cv2.filter2D / cv2.blur with a 3 x 3 kernel on the uint8 image: correlation, BORDER_REFLECT_101, result rounded to the
// nearest level with ties to even (cvRound) and saturated; identity when the group did not fire; ToTensor (/255)
__global__ __launch_bounds__(256) void augment_stencil_kernel(const float* __restrict__ mid, const float* __restrict__ params,
                                                              float* __restrict__ out, int N, int H, int W) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((int64_t)W * H));
    const float* p = params + (size_t)n * NP2;
    const float* im = mid + (size_t)n * H * W;
    float v = im[(size_t)y * W + x];
    if (p[14] != 0.f) {
      float acc = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        int yy = y + ky - 1;
        yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
        yy = yy < 0 ? 0 : yy;  // (H == 1)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          int xx = x + kx - 1;
          xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
          xx = xx < 0 ? 0 : xx;
          acc += p[15 + ky * 3 + kx] * im[(size_t)yy * W + xx];
        }
      }
      v = rintf(fminf(fmaxf(acc, 0.f), 255.f));
    }
    out[i] = v * (1.f / 255.f);
  }
}

}  // namespace

extern "C" int ess_augment_perspective_filter(const float* img, const int64_t* label, const float* params, const int64_t* id_lut,
                                              float* scratch, float* out_img, int64_t* out_label, int32_t N, int32_t H, int32_t W,
                                              ess_stream_t stream) {
  ESS_CHECK_ARG(img && params && scratch && out_img && N > 0 && H > 0 && W > 0, "augment_perspective_filter: bad arguments");
  ESS_CHECK_ARG((label == nullptr) == (out_label == nullptr), "augment_perspective_filter: label and out_label come together");
  ESS_CHECK_ARG(scratch != img && scratch != out_img && img != out_img, "augment_perspective_filter: img, scratch and out_img must be distinct buffers");
  const int64_t total = (int64_t)N * H * W;
  int64_t blocks = ceil_div64(total, 256);
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  hipLaunchKernelGGL(augment_warp_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, label, params, scratch,
                     out_label, id_lut, N, H, W);
  hipLaunchKernelGGL(augment_stencil_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, params,
                     out_img, N, H, W);
  return ess_launch_status("augment_perspective_filter");
}

extern "C" int ess_augment_image_label(const float* img, const int64_t* label, const float* params, const int64_t* id_lut,
                                       float* out_img, int64_t* out_label, int32_t N, int32_t H_src, int32_t W_src, int32_t H,
                                       int32_t W, ess_stream_t stream) {
  ESS_CHECK_ARG(img && params && out_img && N > 0 && H_src > 0 && W_src > 0 && H > 0 && W > 0, "augment_image_label: bad arguments");
  ESS_CHECK_ARG((label == nullptr) == (out_label == nullptr), "augment_image_label: label and out_label come together");
  const int64_t total = (int64_t)N * H * W;
  int64_t blocks = ceil_div64(total, 256);
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  hipLaunchKernelGGL(augment_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, label, params, id_lut, out_img,
                     out_label, N, H_src, W_src, H, W);
  return ess_launch_status("augment_image_label");
}
