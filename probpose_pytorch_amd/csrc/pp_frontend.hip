// Crop + LANCZOS resize + [0,1] scaling of person boxes: the step in front of the forward pass.
// Replaces, per box (reference file:line):
//   dataset.py:71-90   scale_box: image.crop(box) then .resize(image_size, PIL.Image.LANCZOS)
//   inference.py:74-82 image.resize(input_size, LANCZOS); v2.ToImage(); v2.ToDtype(float32, scale=True)
// The arithmetic of those calls lives in Pillow (third party; 12.2.0 in this image): Image.crop rounds the
// box to integers and pads with zeros outside the image; ImagingResample (8 bits per channel) runs a
// horizontal then a vertical pass, each output = clip8((2^21 + sum_k u8 * coeff_k) >> 22) with the
// coefficients = normalised Lanczos-3 weights (double, libm sin) rounded to 22-bit fixed point, and the
// intermediate image is uint8.  The result is integer work, so parity is bit-exact:
//   * pp_frontend_plan_build (HOST code, this file) computes the per-box bounds / fixed-point tables with
//     the same double-precision expressions and the same libm as Pillow, so they are identical;
//   * crop_resize_kernel applies them: one workgroup per (box, block of output rows) runs the horizontal
//     pass for the source rows that block needs into LDS (uint8, like Pillow's temporary image), then the
//     vertical pass out of LDS, and stores f32(u8) * f32(1/255) as NCHW f32: torchvision's v2 ToDtype(scale=True)
//     is image.to(float32).mul_(1.0 / 255) (torchvision is not installed here: that step follows its
//     published source and is pinned only by this restatement).
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "pp_common.h"

namespace pp {

constexpr int FE_PRECISION_BITS = 32 - 8 - 2;   // Pillow Resample.c PRECISION_BITS
constexpr int FE_HDR = 16;                      // int32 words per box header
constexpr int FE_THREADS = 256;
#ifndef FE_UNROLL
#define FE_UNROLL 4     // taps in flight per thread in the horizontal pass (measured: 1 -> 4: 188 -> 170 us per 64 crops; 8: same)
#endif
constexpr size_t FE_LDS_TARGET = 64 * 1024, FE_LDS_MAX = 152 * 1024;

// header words
enum { H_X0 = 0, H_Y0, H_CW, H_CH, H_KSH, H_KSV, H_COEF_OFF, H_SRC_OFF, H_OFF_BH, H_OFF_KH, H_OFF_BV, H_OFF_KV,
       H_NEED_H, H_NEED_V, H_RB, H_STAGED };
constexpr int FE_SRC_ROWS = 8;              // source rows staged in LDS per trip of the horizontal pass

static inline double fe_sinc(double x) {
  if (x == 0.0) return 1.0;
  x = x * M_PI;
  return sin(x) / x;
}
static inline double fe_lanczos(double x) {   // Pillow Resample.c lanczos_filter, support 3
  if (-3.0 <= x && x < 3.0) return fe_sinc(x) * fe_sinc(x / 3);
  return 0.0;
}

// Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for in0 = 0, in1 = inSize (a whole crop)
// (want_kk = false: bounds and ksize only -- no libm calls; the plan's size depends on nothing else)
static int fe_coeffs(int inSize, int outSize, std::vector<int> &bounds, std::vector<int> &kk, bool want_kk) {
  const float in0 = 0.0f, in1 = (float)inSize;
  double filterscale, scale;
  filterscale = scale = (double)(in1 - in0) / outSize;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 3.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  bounds.assign((size_t)outSize * 2, 0);
  if (want_kk) kk.assign((size_t)outSize * ksize, 0);
  std::vector<double> k((size_t)ksize);
  for (int xx = 0; xx < outSize; xx++) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    bounds[xx * 2 + 0] = xmin;
    bounds[xx * 2 + 1] = xmax;
    if (!want_kk) continue;
    for (int x = 0; x < xmax; x++) {
      const double w = fe_lanczos((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; x++) {
      if (ww != 0.0) k[x] /= ww;
    }
    for (int x = 0; x < xmax; x++) {
      const double v = k[x];
      kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << FE_PRECISION_BITS))
                                         : (int)(0.5 + v * (1 << FE_PRECISION_BITS));
    }
  }
  return ksize;
}

struct FeBox {
  std::vector<int> bh, kh, bv, kv;
  int ksh, ksv, rb, nblocks;
  size_t lds;                    // total LDS of a workgroup of this box
  size_t coef_off, src_off;      // staged horizontal pass: byte offsets of the coefficient table / source-row buffer
  bool staged;
};

// Output rows per workgroup: as many as keep >= 1024 workgroups in the launch (4 per CU; fewer, longer
// workgroups measured slower: 512 -> +17 %), at most 32; fe_prepare lowers it further to fit LDS.
static int fe_rb_max(int n_boxes, int out_h) {
  int rb = 32;
  while (rb > 1 && (long long)n_boxes * ((out_h + rb - 1) / rb) < 1024) rb >>= 1;
  return rb;
}

// rows of the temporary image one block of output rows [r0, r0 + rb) reads
static int fe_span(const std::vector<int> &bv, int out_h, int r0, int rb) {
  const int r1 = std::min(out_h, r0 + rb) - 1;
  return bv[r1 * 2] + bv[r1 * 2 + 1] - bv[r0 * 2];
}

static int fe_prepare(const int *box, int out_w, int out_h, int rb_max, FeBox &b, bool want_kk) {
  const int cw = box[2] - box[0], ch = box[3] - box[1];
  if (cw <= 0 || ch <= 0) return fail("pp_frontend: empty box (%d,%d,%d,%d)", box[0], box[1], box[2], box[3]);
  b.ksh = fe_coeffs(cw, out_w, b.bh, b.kh, want_kk);
  b.ksv = fe_coeffs(ch, out_h, b.bv, b.kv, want_kk);
  // largest block of output rows whose source-row span fits the LDS target (span * out_w RGB bytes)
  b.rb = 0;
  for (int rb = rb_max; rb >= 1; rb >>= 1) {
    int span = 0;
    for (int r0 = 0; r0 < out_h; r0 += rb) span = std::max(span, fe_span(b.bv, out_h, r0, rb));
    const size_t lds = ((size_t)span * out_w * 3 + 15) & ~(size_t)15;
    if (lds <= FE_LDS_TARGET || (rb == 1 && lds <= FE_LDS_MAX)) {
      b.rb = rb;
      b.lds = lds;
      // staged horizontal pass when its tables fit beside the temporary rows
      const size_t coef = (((size_t)out_w * (b.ksh + 2) * sizeof(int)) + 15) & ~(size_t)15;
      const size_t src = (size_t)FE_SRC_ROWS * ((((size_t)3 * cw + 3) & ~(size_t)3) + 4);
      b.staged = lds + coef + src <= 2 * FE_LDS_TARGET && lds + coef + src <= FE_LDS_MAX;
      b.coef_off = lds;
      b.src_off = lds + coef;
      if (b.staged) b.lds = lds + coef + ((src + 15) & ~(size_t)15);
      break;
    }
  }
  if (b.rb == 0)
    return fail("pp_frontend: box %dx%d -> %dx%d needs more than %zu bytes of LDS per output row", cw, ch, out_w,
                out_h, FE_LDS_MAX);
  b.nblocks = (out_h + b.rb - 1) / b.rb;
  return 0;
}

// ---- device side ----------------------------------------------------------------------------------
__device__ __forceinline__ int fe_clip8(int v) {   // Pillow clip8: lookup of v >> 22 clamped to [0, 255]
  v >>= FE_PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(FE_THREADS) void crop_resize_kernel(const unsigned char *__restrict__ image, int img_w,
                                                                 int img_h, long long img_stride,
                                                                 const int *__restrict__ plan, int n_boxes,
                                                                 int out_w, int out_h, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tmp[];   // [span][out_w][3] u8
  // block table behind the headers: (box, first output row)
  const int *bt = plan + (size_t)n_boxes * FE_HDR + 2 * blockIdx.x;
  const int c = bt[0], r0 = bt[1];
  const int *h = plan + (size_t)c * FE_HDR;
  const int x0 = h[H_X0], y0 = h[H_Y0], ksh = h[H_KSH], ksv = h[H_KSV];
  const int *bh = plan + h[H_OFF_BH], *kh = plan + h[H_OFF_KH], *bv = plan + h[H_OFF_BV], *kv = plan + h[H_OFF_KV];
  const int need_h = h[H_NEED_H], need_v = h[H_NEED_V];
  const int r1 = min(out_h, r0 + h[H_RB]);
  // source rows (crop coordinates) of this block: Pillow's temporary image rows, shifted by ybox_first there
  const int s0 = bv[r0 * 2], s1 = bv[(r1 - 1) * 2] + bv[(r1 - 1) * 2 + 1];
  const int span = s1 - s0;

  if (h[H_STAGED]) {
    // ---- staged horizontal pass: the block's source rows go through LDS FE_SRC_ROWS at a time with coalesced dword
    // loads (a wave reads 256 contiguous bytes; unaligned dword addresses are fine on gfx9), the box's bounds +
    // coefficient rows sit in LDS too ([out_w][2 + ksh] ints, odd stride: conflict-free), so a tap costs three LDS
    // byte reads + one LDS coefficient read instead of a scattered global dword load + a scattered coefficient load.
    const int cw = h[H_CW];
    int *coef = reinterpret_cast<int *>(tmp + h[H_COEF_OFF]);
    unsigned char *src = tmp + h[H_SRC_OFF];
    const int cstride = ksh + 2;
    const int srcrow = ((3 * cw + 3) & ~3) + 4, ndw = (3 * cw + 3) >> 2;
    if (need_h)
      for (int i = threadIdx.x; i < out_w * cstride; i += FE_THREADS) {
        const int xx = i / cstride, j = i - xx * cstride;
        coef[i] = j < 2 ? bh[xx * 2 + j] : kh[(size_t)xx * ksh + (j - 2)];
      }
    const int rowbytes = 3 * img_w;
    for (int c0 = 0; c0 < span; c0 += FE_SRC_ROWS) {
      const int nrows = min(FE_SRC_ROWS, span - c0);
      for (int i = threadIdx.x; i < nrows * ndw; i += FE_THREADS) {
        const int rr = i / ndw, d = i - rr * ndw;
        const int iy = y0 + s0 + c0 + rr;
        const int o = 3 * x0 + 4 * d;                  // byte offset inside the frame row (may lie outside it)
        unsigned v = 0;
        if (iy >= 0 && iy < img_h) {
          const unsigned char *row = image + (long long)iy * img_stride;
          if (o >= 0 && o + 4 <= rowbytes) {
            __builtin_memcpy(&v, row + o, 4);
          } else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
              if (o + b >= 0 && o + b < rowbytes) v |= (unsigned)row[o + b] << (8 * b);
          }
        }
        *reinterpret_cast<unsigned *>(src + rr * srcrow + 4 * d) = v;
      }
      __syncthreads();
      for (int it = threadIdx.x; it < nrows * out_w; it += FE_THREADS) {
        const int rr = it / out_w, xx = it - rr * out_w;
        const unsigned char *srow = src + rr * srcrow;
        int v0, v1, v2;
        if (need_h) {
          const int *cx = coef + xx * cstride;
          const int xmin = cx[0], xmax = cx[1];
          const unsigned char *px = srow + 3 * xmin;
          int a0 = 1 << (FE_PRECISION_BITS - 1), a1 = a0, a2 = a0;
          for (int x = 0; x < xmax; ++x) {   // (a 4-tap unrolled form of this LDS loop measured 2x slower)
            const int kx = cx[2 + x];
            a0 += (int)px[3 * x] * kx;
            a1 += (int)px[3 * x + 1] * kx;
            a2 += (int)px[3 * x + 2] * kx;
          }
          v0 = fe_clip8(a0);
          v1 = fe_clip8(a1);
          v2 = fe_clip8(a2);
        } else {
          v0 = srow[3 * xx];
          v1 = srow[3 * xx + 1];
          v2 = srow[3 * xx + 2];
        }
        unsigned char *t = tmp + ((size_t)(c0 + rr) * out_w + xx) * 3;
        t[0] = (unsigned char)v0;
        t[1] = (unsigned char)v1;
        t[2] = (unsigned char)v2;
      }
      __syncthreads();
    }
  } else
  // ---- direct horizontal pass (or copy) of crop rows [s0, s1) into LDS; pixels outside the image are 0 (Image.crop)
  for (int it = threadIdx.x; it < span * out_w; it += FE_THREADS) {
    const int tr = it / out_w, xx = it - tr * out_w;
    const int iy = y0 + s0 + tr;
    int v0 = 0, v1 = 0, v2 = 0;
    if (iy >= 0 && iy < img_h) {
      const unsigned char *row = image + (long long)iy * img_stride;
      if (need_h) {
        const int xmin = bh[xx * 2], xmax = bh[xx * 2 + 1];
        const int *k = kh + (size_t)xx * ksh;
        int a0 = 1 << (FE_PRECISION_BITS - 1), a1 = a0, a2 = a0;
        // one (unaligned) dword load per pixel instead of three byte loads; the 4th byte is ignored.  The
        // frame's very last pixel is read bytewise so that nothing past the buffer is touched.
        const bool last_row = iy == img_h - 1;
        // four taps per trip: the four pixel loads (and coefficient loads) are independent, so they are in flight
        // together; taps past xmax or outside the frame contribute 0 (a zero pixel, as Image.crop pads)
        for (int x = 0; x < xmax; x += FE_UNROLL) {
          unsigned rgb[FE_UNROLL];
          int kx[FE_UNROLL];
#pragma unroll
          for (int u = 0; u < FE_UNROLL; ++u) {
            const int ix = x0 + xmin + x + u;
            rgb[u] = 0;
            kx[u] = 0;
            if (x + u < xmax && ix >= 0 && ix < img_w) {
              kx[u] = k[x + u];
              const unsigned char *px = row + 3 * ix;
              if (last_row && ix == img_w - 1)
                rgb[u] = (unsigned)px[0] | ((unsigned)px[1] << 8) | ((unsigned)px[2] << 16);
              else
                __builtin_memcpy(&rgb[u], px, 4);
            }
          }
#pragma unroll
          for (int u = 0; u < FE_UNROLL; ++u) {
            a0 += (int)(rgb[u] & 255u) * kx[u];
            a1 += (int)((rgb[u] >> 8) & 255u) * kx[u];
            a2 += (int)((rgb[u] >> 16) & 255u) * kx[u];
          }
        }
        v0 = fe_clip8(a0);
        v1 = fe_clip8(a1);
        v2 = fe_clip8(a2);
      } else {
        const int ix = x0 + xx;
        if (ix >= 0 && ix < img_w) {
          v0 = row[3 * ix];
          v1 = row[3 * ix + 1];
          v2 = row[3 * ix + 2];
        }
      }
    } else if (need_h) {
      // a zero row still passes through the filter: clip8(2^21 >> 22) = 0
      v0 = v1 = v2 = 0;
    }
    unsigned char *t = tmp + ((size_t)tr * out_w + xx) * 3;
    t[0] = (unsigned char)v0;
    t[1] = (unsigned char)v1;
    t[2] = (unsigned char)v2;
  }
  __syncthreads();

  // ---- vertical pass (or copy) out of LDS, then ToDtype(float32, scale=True)
  const size_t plane = (size_t)out_h * out_w;
  float *o = out + (size_t)c * 3 * plane;
  for (int it = threadIdx.x; it < (r1 - r0) * out_w; it += FE_THREADS) {
    const int yy = r0 + it / out_w, xx = it % out_w;
    int v0, v1, v2;
    if (need_v) {
      const int ymin = bv[yy * 2] - s0, ymax = bv[yy * 2 + 1];
      const int *k = kv + (size_t)yy * ksv;
      int a0 = 1 << (FE_PRECISION_BITS - 1), a1 = a0, a2 = a0;
      for (int y = 0; y < ymax; ++y) {
        const unsigned char *t = tmp + ((size_t)(ymin + y) * out_w + xx) * 3;
        const int ky = k[y];
        a0 += (int)t[0] * ky;
        a1 += (int)t[1] * ky;
        a2 += (int)t[2] * ky;
      }
      v0 = fe_clip8(a0);
      v1 = fe_clip8(a1);
      v2 = fe_clip8(a2);
    } else {
      const unsigned char *t = tmp + ((size_t)(yy - s0) * out_w + xx) * 3;
      v0 = t[0];
      v1 = t[1];
      v2 = t[2];
    }
    const size_t idx = (size_t)yy * out_w + xx;
    // torchvision v2 to_dtype_image (int -> float, scale=True): image.to(float32).mul_(1.0 / 255)
    constexpr float inv255 = (float)(1.0 / 255.0);
    o[idx] = (float)v0 * inv255;
    o[plane + idx] = (float)v1 * inv255;
    o[2 * plane + idx] = (float)v2 * inv255;
  }
}

}  // namespace pp

// ---- C ABI ------------------------------------------------------------------------------------------
using namespace pp;

extern "C" long long pp_frontend_plan_bytes(int n_boxes, const int *boxes_xyxy, int out_w, int out_h) {
  if (n_boxes < 0 || out_w <= 0 || out_h <= 0 || (n_boxes > 0 && !boxes_xyxy)) {
    fail("pp_frontend_plan_bytes: bad arguments");
    return -1;
  }
  size_t words = (size_t)n_boxes * FE_HDR;
  for (int c = 0; c < n_boxes; ++c) {
    FeBox b;
    if (fe_prepare(boxes_xyxy + 4 * c, out_w, out_h, fe_rb_max(n_boxes, out_h), b, false) != 0) return -1;
    words += 2 * (size_t)b.nblocks + b.bh.size() + b.bv.size() + (size_t)out_w * b.ksh + (size_t)out_h * b.ksv;
  }
  return (long long)(words * sizeof(int));
}

// Fills `plan` (host memory, pp_frontend_plan_bytes bytes) and returns the launch geometry.
extern "C" int pp_frontend_plan_build(int n_boxes, const int *boxes_xyxy, int out_w, int out_h, void *plan,
                                      int *n_blocks_out, long long *lds_bytes_out) {
  PP_REQUIRE(n_boxes >= 0 && out_w > 0 && out_h > 0 && plan && n_blocks_out && lds_bytes_out &&
                 (n_boxes == 0 || boxes_xyxy),
             "pp_frontend_plan_build: bad arguments");
  int *w = static_cast<int *>(plan);
  std::vector<FeBox> bx((size_t)n_boxes);
  int nblocks = 0;
  size_t lds = 0;
  {
    // the coefficient tables cost ~150 us of libm per box: spread the boxes over host threads
    const int nt = std::max(1, std::min({n_boxes / 2, (int)std::thread::hardware_concurrency(), 16}));
    const int rb_max = fe_rb_max(n_boxes, out_h);
    std::vector<int> rcs((size_t)n_boxes, 0);
    std::vector<std::string> msgs((size_t)nt);
    auto work = [&](int t) {
      for (int c = t; c < n_boxes; c += nt) {
        rcs[c] = fe_prepare(boxes_xyxy + 4 * c, out_w, out_h, rb_max, bx[c], true);
        if (rcs[c] != 0) msgs[t] = err_buf();     // the error text is thread-local: carry it back
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    for (int c = 0; c < n_boxes; ++c)
      if (rcs[c] != 0) return fail("%s", msgs[c % nt].c_str());
  }
  for (int c = 0; c < n_boxes; ++c) {
    nblocks += bx[c].nblocks;
    lds = std::max(lds, bx[c].lds);
  }
  size_t off = (size_t)n_boxes * FE_HDR;
  int blk = 0;
  for (int c = 0; c < n_boxes; ++c)   // block table: (box, first output row)
    for (int r0 = 0; r0 < out_h; r0 += bx[c].rb) {
      w[off + 2 * blk] = c;
      w[off + 2 * blk + 1] = r0;
      ++blk;
    }
  off += 2 * (size_t)nblocks;
  for (int c = 0; c < n_boxes; ++c) {
    const int *box = boxes_xyxy + 4 * c;
    FeBox &b = bx[c];
    const int cw = box[2] - box[0], ch = box[3] - box[1];
    int *h = w + (size_t)c * FE_HDR;
    h[H_X0] = box[0];
    h[H_Y0] = box[1];
    h[H_CW] = cw;
    h[H_CH] = ch;
    h[H_KSH] = b.ksh;
    h[H_KSV] = b.ksv;
    h[H_COEF_OFF] = (int)b.coef_off;
    h[H_SRC_OFF] = (int)b.src_off;
    h[H_NEED_H] = cw != out_w;   // Resample.c: need_horizontal = xsize != imIn->xsize || box[0] || box[2] != xsize
    h[H_NEED_V] = ch != out_h;
    h[H_RB] = b.rb;
    h[H_STAGED] = b.staged ? 1 : 0;
    auto put = [&](int slot, const std::vector<int> &v) {
      h[slot] = (int)off;
      memcpy(w + off, v.data(), v.size() * sizeof(int));
      off += v.size();
    };
    put(H_OFF_BH, b.bh);
    put(H_OFF_KH, b.kh);
    put(H_OFF_BV, b.bv);
    put(H_OFF_KV, b.kv);
    PP_REQUIRE(off < (1ull << 31), "pp_frontend_plan_build: plan exceeds 2^31 words");
  }
  *n_blocks_out = nblocks;
  *lds_bytes_out = (long long)lds;
  return 0;
}

extern "C" int pp_frontend_crop_resize(const unsigned char *image, int img_w, int img_h, long long img_stride,
                                       const void *plan_dev, int n_boxes, int n_blocks, long long lds_bytes,
                                       int out_w, int out_h, float *out, void *stream) {
  PP_REQUIRE(n_boxes >= 0 && n_blocks >= 0 && out_w > 0 && out_h > 0 && img_w > 0 && img_h > 0 &&
                 img_stride >= 3ll * img_w,
             "pp_frontend_crop_resize: bad shape");
  if (n_boxes == 0 || n_blocks == 0) return 0;
  PP_REQUIRE(image && plan_dev && out, "pp_frontend_crop_resize: null pointer");
  PP_REQUIRE(lds_bytes > 0 && (size_t)lds_bytes <= FE_LDS_MAX, "pp_frontend_crop_resize: bad LDS size %lld", lds_bytes);
  if (lds_bytes > 64 * 1024) {
    static thread_local unsigned long long attr_mask = 0;
    int dev_ = 0;
    if (attr_needed(attr_mask, dev_))
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(crop_resize_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)FE_LDS_MAX));
  }
  hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)n_blocks), dim3(FE_THREADS), (size_t)lds_bytes,
                     (hipStream_t)stream, image, img_w, img_h, img_stride, static_cast<const int *>(plan_dev), n_boxes,
                     out_w, out_h, out);
  PP_CHECK_LAUNCH("crop_resize_kernel");
  return 0;
}
