// Image-branch augmentation on the device: the geometric + photometric core of the albumentations pipeline of
// datasets/cityscapes_loader.py:39-74 for a whole batch in ONE launch, instead of per-sample numpy / cv2 work in loader
// processes:  HorizontalFlip -> ShiftScaleRotate(rotate 0, constant border 0) -> PadIfNeeded (centred, constant 0) -> RandomCrop
// -> GaussNoise -> RandomBrightnessContrast -> uint8 quantisation -> ToTensor (/255), and for the label map the same geometry
// with nearest sampling followed by the id -> trainId table (utils/labels.py:123-127).  The random DECISIONS (which ops fire,
// their magnitudes, the crop offset) are drawn on the host per sample and arrive as a parameter row; the per-pixel noise is a
// counter-based hash of (seed, pixel) so that the oracle reproduces it.  albumentations and cv2 are third-party packages absent
// from /root/reference and from this image: their interpolation arithmetic is restated (bilinear with zero border, sample
// positions at pixel centres, round-half-up quantisation), i.e. PARITY UNPINNED against the libraries themselves; Perspective
// and the Sharpen / Blur / MotionBlur group are not provided.
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

}  // namespace

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
