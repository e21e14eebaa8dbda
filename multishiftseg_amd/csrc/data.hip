// On-device data path of the DeepLab trainer: what DiverseCityscapes.__getitem__ and its transforms do on
// CPU workers per sample (lib/dataset/cityscapes.py:153-171, lib/utils/img_utils.py:110-153,246-259,367-435), as ONE kernel
// over a batch of pre-decoded uint8 images resident in HBM:
//   mixup of the generated image with the original (cityscapes.py:161-164, float64 arithmetic, truncation to uint8),
//   ToTensor (/255), the SAME random crop for the four maps (RandCrop, img_utils.py:246-259), optional horizontal flip,
//   Normalize ((x - mean) / std, float32), COCO-object paste into the original image and its label map (mix_func,
//   img_utils.py:398-435: object pixels normalised in float64, label = the object mask's value),
// and the batch layout the trainer builds afterwards: images [orig...; aug...] NCHW float32, targets int64
// (train_deeplab.py:190-195). The random decisions (mixing weight, crop corner, flip, object, scale, paste corner) are
// drawn on the host exactly where the reference draws them (multishiftseg_amd/datapath.py) and arrive as small arrays.
// HBM traffic: 8 B read + 40 B written per output pixel.
#include "mss_common.h"
#include "../../include/mss_hip.h"

namespace {

__device__ __forceinline__ double pin(double v) { asm volatile("" : "+v"(v)); return v; }   // forbid fma contraction
__device__ __forceinline__ float pinf(float v) { asm volatile("" : "+v"(v)); return v; }

struct DataArgs {
  const uint8_t* img; const uint8_t* gen; const uint8_t* tgt; const uint8_t* gen_tgt;
  int B, H, W, h, w;
  const double* mix_p; const int* crop; const int* flip;
  float mean[3], stdv[3];          // float32(mean), float32(std): what torchvision's Normalize subtracts / divides by
  double mean_d[3], std_d[3];      // the Python floats themselves: what img_utils.normalize() uses on the pasted object
  const float* obj_img; const uint8_t* obj_mask; const int* obj_geom; int OHmax, OWmax;
  float* out_img; int64_t* out_tgt;
};

// grid (ceil(w / 256), h, B)
__global__ __launch_bounds__(256) void data_pair_kernel(DataArgs a) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= a.w) return;
  const int top = a.crop[2 * b], left = a.crop[2 * b + 1];
  const int xs = (a.flip && a.flip[b]) ? a.w - 1 - x : x;            // F.hflip of the cropped window
  const long long src = ((long long)b * a.H + top + y) * a.W + left + xs;
  const uint8_t* pi = a.img + src * 3;
  const uint8_t* pg = a.gen + src * 3;
  const long long plane = (long long)a.h * a.w, opix = (long long)y * a.w + x;
  float* o_orig = a.out_img + (long long)b * 3 * plane + opix;
  float* o_aug = a.out_img + (long long)(a.B + b) * 3 * plane + opix;
  const double p = a.mix_p ? a.mix_p[b] : 0.0;
  // COCO paste: object region rows [y1, y1+bh) x cols [x1, x1+bw) of this sample's object, placed at (h0, w0) of the crop
  // (the reference pastes after the transforms, i.e. in crop coordinates, before any flip exists in its pipeline; with a
  // flip requested here the paste still refers to the final, flipped window)
  bool pasted = false;
  float pv[3] = {0.f, 0.f, 0.f};
  uint8_t pm = 0;
  if (a.obj_geom) {
    const int* g = a.obj_geom + 6 * b;
    const int y1 = g[0], x1 = g[1], bh = g[2], bw = g[3], h0 = g[4], w0 = g[5];
    const int oy = y - h0, ox = x - w0;
    if (bh > 0 && oy >= 0 && oy < bh && ox >= 0 && ox < bw) {
      const long long oi = ((long long)b * a.OHmax + y1 + oy) * a.OWmax + x1 + ox;
      pm = a.obj_mask[oi];
      if (pm != 0 && pm != 255) {
        pasted = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          // normalize() of img_utils.py:355-361: float32 / 255.0 stays float32, "- mean" and "/ std" promote to float64
          const float v = pinf(a.obj_img[oi * 3 + c] / 255.0f);
          pv[c] = (float)(pin((double)v - a.mean_d[c]) / a.std_d[c]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float vi = (float)pi[c];
    // mixup, cityscapes.py:161-164: (p * image + (1 - p) * gen_image).astype(np.uint8), all in float64
    uint8_t gq = pg[c];
    if (a.mix_p) {
      const double t1 = pin(p * (double)pi[c]), t2 = pin((1.0 - p) * (double)pg[c]);
      gq = (uint8_t)(int)(t1 + t2);
    }
    // ToTensor: uint8 -> float32 / 255; Normalize: sub_(mean).div_(std), three separately rounded float32 operations
    const float n_orig = pinf(pinf(vi / 255.0f) - a.mean[c]) / a.stdv[c];
    const float n_aug = pinf(pinf((float)gq / 255.0f) - a.mean[c]) / a.stdv[c];
    o_orig[(long long)c * plane] = pasted ? pv[c] : n_orig;
    o_aug[(long long)c * plane] = n_aug;
  }
  a.out_tgt[(long long)b * plane + opix] = pasted ? (int64_t)pm : (int64_t)a.tgt[src];
  a.out_tgt[(long long)(a.B + b) * plane + opix] = (int64_t)a.gen_tgt[src];
}

}  // namespace

extern "C" {

int mss_data_pair_f32(const uint8_t* img, const uint8_t* gen, const uint8_t* tgt, const uint8_t* gen_tgt, int B, int H, int W,
                      int h, int w, const double* mix_p, const int* crop, const int* flip, const double* mean3,
                      const double* std3, const float* obj_img, const uint8_t* obj_mask, const int* obj_geom, int OHmax,
                      int OWmax, float* out_img, int64_t* out_tgt, void* stream) {
  if (!img || !gen || !tgt || !gen_tgt || !crop || !mean3 || !std3 || !out_img || !out_tgt) return MSS_ERR_BAD_ARG;
  if (B <= 0 || h <= 0 || w <= 0 || h > H || w > W || h > 65535 || B > 65535) return MSS_ERR_BAD_ARG;
  if (obj_geom && (!obj_img || !obj_mask || OHmax <= 0 || OWmax <= 0)) return MSS_ERR_BAD_ARG;
  DataArgs a;
  a.img = img; a.gen = gen; a.tgt = tgt; a.gen_tgt = gen_tgt;
  a.B = B; a.H = H; a.W = W; a.h = h; a.w = w;
  a.mix_p = mix_p; a.crop = crop; a.flip = flip;
  for (int c = 0; c < 3; ++c) {                                                // HOST pointers: three doubles each
    a.mean_d[c] = mean3[c]; a.std_d[c] = std3[c];
    a.mean[c] = (float)mean3[c]; a.stdv[c] = (float)std3[c];
  }
  a.obj_img = obj_img; a.obj_mask = obj_mask; a.obj_geom = obj_geom; a.OHmax = OHmax; a.OWmax = OWmax;
  a.out_img = out_img; a.out_tgt = out_tgt;
  hipLaunchKernelGGL(data_pair_kernel, dim3((w + 255) / 256, h, B), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return mss_launch_status();
}

}  // extern "C"
