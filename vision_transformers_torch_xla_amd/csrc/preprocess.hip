// Device tail of the input pipeline: a batch of decoded, cropped images in host layout (NHWC, uint8 or fp32) becomes the
// normalised NCHW fp32 batch the model consumes.  It restates, per element and in the reference's order,
//   flip_lr                      (big_vision/pp/ops_image.py:69-82)   x -> W-1-x for the samples whose flag is set
//   value_range(vmin, vmax)      (big_vision/pp/ops_general.py:26-43) integers: float(u) / 255, then  v * (vmax-vmin) + vmin
//   HWC -> CHW, float32          (datasets.py:301-307)
//   (v - mean[c]) / std[c]       (datasets.py:249-252, image.sub_(mean).div_(std))
// with every operation rounded on its own (no fma contraction, IEEE division), so the result is bit-identical to the host code.
// The optional window (top, left) reads an H x W crop out of larger [Hs, Ws] source images (central_crop of equally sized
// images, ops_image.py:121-134).  HBM-bound: 3 or 12 bytes read and 12 bytes written per pixel.
#include "common.h"

// One item = 4 consecutive output pixels of one image row: 12 contiguous source elements (reversed when the sample is
// flipped), three 16-byte stores - one per channel plane - so a wave writes 1 KB contiguous per plane.
// A byte has 256 values, so for uint8 sources each block first evaluates the whole chain once per (channel, value) into an
// LDS table - with the very operations the float path applies per element - and the pixel loop is a table lookup: two
// IEEE divisions per element would otherwise make the kernel VALU-bound at half the HBM rate.
template <bool U8>
__global__ __launch_bounds__(256) void images_to_nchw_kernel(const void* __restrict__ src_, const unsigned char* __restrict__ flip,
                                                              float* __restrict__ dst, int B, int H, int W, int Hs, int Ws, int top,
                                                              int left, float scale, float vmin, int normalise, float m0, float m1,
                                                              float m2, float s0, float s1, float s2) {
#pragma clang fp contract(off)
    __shared__ float tab[U8 ? 3 * 256 : 1];
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    if (U8) {
        for (int k = threadIdx.x; k < 3 * 256; k += blockDim.x) {
            const int c = k >> 8;
            float t = (float)(k & 255) / 255.0f;
            t = t * scale + vmin;
            if (normalise) t = (t - mean[c]) / stdv[c];
            tab[k] = t;
        }
        __syncthreads();
    }
    const int wg = W / 4;
    const size_t total = (size_t)B * H * wg;
    const size_t plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xg = (int)(i % wg);
        const size_t r = i / wg;
        const int y = (int)(r % H), b = (int)(r / H);
        const bool fl = flip != nullptr && flip[b] != 0;
        const int x = xg * 4;
        const int sx = fl ? W - 4 - x : x;                              // first source pixel of the group
        const size_t e0 = (((size_t)b * Hs + top + y) * Ws + left + sx) * 3;   // first source element
        float v[12];            // uint8 sources: the finished values; float sources: the raw elements
        if (U8) {
            const unsigned char* s = static_cast<const unsigned char*>(src_) + e0;
            unsigned u[12];
            if ((e0 & 3) == 0) {
                const uint3 w = *reinterpret_cast<const uint3*>(s);
                const unsigned ww[3] = {w.x, w.y, w.z};
#pragma unroll
                for (int k = 0; k < 12; ++k) u[k] = (ww[k >> 2] >> (8 * (k & 3))) & 0xffu;
            } else {
#pragma unroll
                for (int k = 0; k < 12; ++k) u[k] = s[k];
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = tab[(k % 3) * 256 + u[k]];
        } else {
            const float* s = static_cast<const float*>(src_) + e0;
            if ((e0 & 3) == 0) {
                const float4 a = *reinterpret_cast<const float4*>(s), c = *reinterpret_cast<const float4*>(s + 4),
                             d = *reinterpret_cast<const float4*>(s + 8);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
                v[8] = d.x; v[9] = d.y; v[10] = d.z; v[11] = d.w;
            } else {
#pragma unroll
                for (int k = 0; k < 12; ++k) v[k] = s[k];
            }
        }
        float* o = dst + (size_t)b * 3 * plane + (size_t)y * W + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float q[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float t = v[(fl ? 3 - p : p) * 3 + c];
                if (!U8) {
                    t = t * scale + vmin;
                    if (normalise) t = (t - mean[c]) / stdv[c];
                }
                q[p] = t;
            }
            *reinterpret_cast<float4*>(o + c * plane) = make_float4(q[0], q[1], q[2], q[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Crop + antialiased resize of decoded images of different sizes (tf.image.resize(..., antialias=True) = ScaleAndTranslate).
// One workgroup per (image, output row).  The row's vertical span and weights are computed once (LDS); every thread then forms
// the vertically filtered source row for its columns of the box - the intermediate row of the two-pass algorithm, kept in LDS -
// and after a barrier every thread filters it horizontally for its output pixels.  Sums start at zero and add weight * value in
// ascending source order, multiply and add rounded separately, so the result is bit-identical to the two-pass float32 host code.
static __device__ __forceinline__ float resample_kernel(int kind, float x) {
#pragma clang fp contract(off)
    x = fabsf(x);
    if (kind == 0) return x < 1.0f ? 1.0f - x : 0.0f;
    if (x >= 2.0f) return 0.0f;
    if (x >= 1.0f) return ((-0.5f * x + 2.5f) * x - 4.0f) * x + 2.0f;
    return ((1.5f * x - 2.5f) * x) * x + 1.0f;
}
// span of output index o: first source index, count, sampling position; in / out sizes as ints
struct ResampleAxis { float inv_scale, kernel_scale, one_over, radius; int in_size; };
static __device__ __forceinline__ ResampleAxis resample_axis(int in_size, int out_size, int kind) {
#pragma clang fp contract(off)
    ResampleAxis a;
    const float scale = (float)out_size / (float)in_size;
    a.inv_scale = (float)(1.0 / (double)scale);
    a.kernel_scale = fmaxf(a.inv_scale, 1.0f);
    a.one_over = 1.0f / a.kernel_scale;
    a.radius = kind == 0 ? 1.0f : 2.0f;
    a.in_size = in_size;
    return a;
}
static __device__ __forceinline__ void resample_span(const ResampleAxis& a, int o, int& start, int& count, float& sample) {
#pragma clang fp contract(off)
    sample = ((float)o + 0.5f) * a.inv_scale;
    const float reach = a.radius * a.kernel_scale;
    int s = (int)ceilf((sample - reach) - 0.5f);
    int e = (int)floorf((sample + reach) - 0.5f);
    s = min(max(s, 0), a.in_size - 1);
    e = min(max(e, 0), a.in_size - 1) + 1;
    start = s;
    count = e - s;
}
static __device__ __forceinline__ float resample_weight(const ResampleAxis& a, int kind, int src, float sample) {
#pragma clang fp contract(off)
    const float pos = ((float)src + 0.5f) - sample;
    return resample_kernel(kind, pos * a.one_over);
}

#define RESAMPLE_MAX_TAPS 512      // vertical taps kept in LDS: covers a 127-fold reduction with the cubic kernel (2 * 2 * 127 + 3 taps)

// dst: fp32 NCHW [B, 3, S, S] - or, with patches != null, the model's bf16 patch matrix [B * (S/patch)^2, 3 * patch^2] (row =
// patch gy * (S/patch) + gx of image b, column c * patch^2 + ky * patch + kx: the operand order of the patch-embedding GEMM),
// each value rounded to bf16 once, as the im2col kernel rounds the fp32 image.
// The descriptors the kernel reads are the caller's DEVICE copy; the host copy was validated.  So that a stale or mismatched
// device copy cannot turn into an access outside the buffers, every source byte offset is clamped to the source buffer and the
// box width to the width the LDS row was sized for.
__global__ __launch_bounds__(256) void resample_kernel_rows(const unsigned char* __restrict__ src, const vitk_resample_desc* __restrict__ desc,
                                                            float* __restrict__ dst, int S, int kind, float scale, float vmin, int normalise,
                                                            float m0, float m1, float m2, float s0, float s1, float s2,
                                                            bf16_t* __restrict__ patches, int patch, long long src_bytes, int max_box_w) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float rs_smem[];
    float* wy = rs_smem;                               // [RESAMPLE_MAX_TAPS] normalised vertical weights of this output row
    float* inter = rs_smem + RESAMPLE_MAX_TAPS;        // [box_w][3] vertically filtered row
    const int b = blockIdx.x / S, y = blockIdx.x % S;
    vitk_resample_desc d = desc[b];
    d.box_w = min(max(d.box_w, 1), max_box_w);
    d.box_h = max(d.box_h, 1);
    d.out_w = max(d.out_w, 1);
    d.out_h = max(d.out_h, 1);
    const size_t plane = (size_t)S * S;
    float* out_row = patches ? nullptr : dst + (size_t)b * 3 * plane + (size_t)y * S;
    // patch-matrix output: this output row is row y % patch of the patches gy = y / patch
    const int gw = patches ? S / patch : 1;
    bf16_t* prow = patches ? patches + ((size_t)b * gw * gw + (size_t)(y / patch) * gw) * 3 * patch * patch + (size_t)(y % patch) * patch : nullptr;
    auto put = [&](int x, int c, float v) {
        if (patches) prow[(size_t)(x / patch) * 3 * patch * patch + (size_t)c * patch * patch + (x % patch)] = f2bf(v);
        else out_row[c * plane + x] = v;
    };
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    auto finish = [&](float t, int c) {                // value_range (float input) and normalisation, as vitk_preprocess_images
        t = t * scale + vmin;
        if (normalise) t = (t - mean[c]) / stdv[c];
        return t;
    };
    const int oy = y + d.win_y;
    if (oy < 0 || oy >= d.out_h) {                      // a padded row (the resized image is shorter than the window)
        for (int x = threadIdx.x; x < S; x += blockDim.x)
#pragma unroll
            for (int c = 0; c < 3; ++c) put(x, c, finish(0.0f, c));
        return;
    }
    const ResampleAxis ay = resample_axis(d.box_h, d.out_h, kind);
    int ys, yn;
    float ysample;
    resample_span(ay, oy, ys, yn, ysample);
    if (yn > RESAMPLE_MAX_TAPS) yn = RESAMPLE_MAX_TAPS;              // refused on the host; never taken
    for (int i = threadIdx.x; i < yn; i += blockDim.x) wy[i] = resample_weight(ay, kind, ys + i, ysample);
    __syncthreads();
    float ytotal = 0.0f;
    for (int i = 0; i < yn; ++i) ytotal += wy[i];
    const bool ynorm = fabsf(ytotal) >= 1000.0f * 1.17549435e-38f;
    const float yinv = 1.0f / ytotal;
    // vertical pass: inter[col][c] = sum_i (wy[i] * yinv) * (u8 * (1/255))
    const long long box = (long long)d.src_offset + ((long long)(d.box_y + ys) * d.src_w + d.box_x) * 3;
    const long long row_bytes = (long long)d.src_w * 3;
    const float k255 = (float)(1.0 / 255.0);
    for (int e = threadIdx.x; e < d.box_w * 3; e += blockDim.x) {
        float acc = 0.0f;
        for (int i = 0; i < yn; ++i) {
            const float w = ynorm ? wy[i] * yinv : 0.0f;
            long long at = box + e + (long long)i * row_bytes;
            at = at < 0 ? 0 : (at < src_bytes ? at : src_bytes - 1);
            const float v = (float)src[at] * k255;
            acc = acc + w * v;
        }
        inter[e] = acc;
    }
    __syncthreads();
    // horizontal pass
    const ResampleAxis ax = resample_axis(d.box_w, d.out_w, kind);
    for (int x = threadIdx.x; x < S; x += blockDim.x) {
        const int xo = (d.flip ? S - 1 - x : x) + d.win_x;      // flip_lr acts on the finished S x S image
        float r[3] = {0.0f, 0.0f, 0.0f};
        if (xo >= 0 && xo < d.out_w) {
            int xs, xn;
            float xsample;
            resample_span(ax, xo, xs, xn, xsample);
            float total = 0.0f;
            for (int i = 0; i < xn; ++i) total += resample_weight(ax, kind, xs + i, xsample);
            const bool norm = fabsf(total) >= 1000.0f * 1.17549435e-38f;
            const float inv = 1.0f / total;
            for (int i = 0; i < xn; ++i) {
                const float w = norm ? resample_weight(ax, kind, xs + i, xsample) * inv : 0.0f;
                const float* q = inter + (size_t)min(xs + i, d.box_w - 1) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) r[c] = r[c] + w * q[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) put(x, c, finish(r[c], c));
    }
}

// NHWC window -> bf16 patch matrix, the device tail of the input pipeline for a consumer that is the model itself: one item = 8
// consecutive pixels of one patch row (24 contiguous source elements, reversed when the sample is flipped) -> three 16-byte
// stores, one per channel, at column c * p^2 + ky * p + kx0 of the patch's row; consecutive items are consecutive 16-byte chunks
// of a patch's channel plane (p^2 * 2 contiguous bytes), so a wave's stores cover whole lines.  Same arithmetic and LDS table as
// images_to_nchw_kernel, then one rounding to bf16 - what vitk_im2col_patches makes of that kernel's fp32 output, bit for bit.
template <bool U8>
__global__ __launch_bounds__(256) void images_to_patches_kernel(const void* __restrict__ src_, const unsigned char* __restrict__ flip,
                                                                 bf16_t* __restrict__ dst, int B, int H, int W, int Hs, int Ws, int top,
                                                                 int left, int p, float scale, float vmin, int normalise, float m0, float m1,
                                                                 float m2, float s0, float s1, float s2) {
#pragma clang fp contract(off)
    __shared__ float tab[U8 ? 3 * 256 : 1];
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    if (U8) {
        for (int k = threadIdx.x; k < 3 * 256; k += blockDim.x) {
            const int c = k >> 8;
            float t = (float)(k & 255) / 255.0f;
            t = t * scale + vmin;
            if (normalise) t = (t - mean[c]) / stdv[c];
            tab[k] = t;
        }
        __syncthreads();
    }
    const int gw = W / p, gh = H / p, kg = p / 8;
    const size_t per_patch = (size_t)p * kg;
    const size_t total = (size_t)B * gh * gw * per_patch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t patch_id = i / per_patch;
        const int in_patch = (int)(i - patch_id * per_patch);
        const int ky = in_patch / kg, kx0 = (in_patch - ky * kg) * 8;
        const int gx = (int)(patch_id % gw);
        const size_t t2 = patch_id / gw;
        const int gy = (int)(t2 % gh), b = (int)(t2 / gh);
        const bool fl = flip != nullptr && flip[b] != 0;
        const int y = gy * p + ky, x = gx * p + kx0;
        const int sx = fl ? W - 8 - x : x;                              // first source pixel of the group
        const size_t e0 = (((size_t)b * Hs + top + y) * Ws + left + sx) * 3;
        float v[24];
        if (U8) {
            const unsigned char* s = static_cast<const unsigned char*>(src_) + e0;
            unsigned u[24];
            if ((reinterpret_cast<uintptr_t>(s) & 7) == 0) {
                const uint2 w0 = *reinterpret_cast<const uint2*>(s), w1 = *reinterpret_cast<const uint2*>(s + 8), w2 = *reinterpret_cast<const uint2*>(s + 16);
                const unsigned ww[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
#pragma unroll
                for (int k = 0; k < 24; ++k) u[k] = (ww[k >> 2] >> (8 * (k & 3))) & 0xffu;
            } else {
#pragma unroll
                for (int k = 0; k < 24; ++k) u[k] = s[k];
            }
#pragma unroll
            for (int k = 0; k < 24; ++k) v[k] = tab[(k % 3) * 256 + u[k]];
        } else {
            const float* s = static_cast<const float*>(src_) + e0;
#pragma unroll
            for (int k = 0; k < 24; ++k) v[k] = s[k];
        }
        bf16_t* o = dst + patch_id * 3 * p * p + (size_t)ky * p + kx0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float q[8];
#pragma unroll
            for (int px = 0; px < 8; ++px) {
                float t = v[(fl ? 7 - px : px) * 3 + c];
                if (!U8) {
                    t = t * scale + vmin;
                    if (normalise) t = (t - mean[c]) / stdv[c];
                }
                q[px] = t;
            }
            store8(o + (size_t)c * p * p, q);
        }
    }
}

static int resample_impl(const unsigned char* src, int64_t src_bytes, const vitk_resample_desc* desc_host,
                         const vitk_resample_desc* desc_dev, float* dst, bf16_t* patches, int patch, int B, int S, int kernel, float vmin,
                         float vmax, const float* mean3, const float* std3, void* stream);

extern "C" {

int vitk_resample_images(const unsigned char* src, int64_t src_bytes, const vitk_resample_desc* desc_host,
                         const vitk_resample_desc* desc_dev, float* dst, int B, int S, int kernel, float vmin, float vmax,
                         const float* mean3, const float* std3, void* stream) {
    VITK_REQUIRE(dst != nullptr, "resample_images: bad arguments");
    return resample_impl(src, src_bytes, desc_host, desc_dev, dst, nullptr, 0, B, S, kernel, vmin, vmax, mean3, std3, stream);
}

int vitk_resample_patches(const unsigned char* src, int64_t src_bytes, const vitk_resample_desc* desc_host,
                          const vitk_resample_desc* desc_dev, void* patches, int patch, int B, int S, int kernel, float vmin, float vmax,
                          const float* mean3, const float* std3, void* stream) {
    VITK_REQUIRE(patches != nullptr && patch > 0 && patch % 8 == 0 && S % patch == 0,
                 "resample_patches: the patch size %d must be a multiple of 8 that divides the image size %d", patch, S);
    return resample_impl(src, src_bytes, desc_host, desc_dev, nullptr, (bf16_t*)patches, patch, B, S, kernel, vmin, vmax, mean3, std3, stream);
}

}  // extern "C"

static int resample_impl(const unsigned char* src, int64_t src_bytes, const vitk_resample_desc* desc_host,
                         const vitk_resample_desc* desc_dev, float* dst, bf16_t* patches, int patch, int B, int S, int kernel, float vmin,
                         float vmax, const float* mean3, const float* std3, void* stream) {
    VITK_REQUIRE(src && desc_host && desc_dev && (dst || patches) && B > 0 && src_bytes > 0, "resample_images: bad arguments");
    VITK_REQUIRE(S > 0 && S % 4 == 0 && S <= 1024, "resample_images: output size %d must be a multiple of 4, at most 1024", S);
    VITK_REQUIRE(kernel == 0 || kernel == 1, "resample_images: kernel must be 0 (triangle / BILINEAR) or 1 (keyscubic / BICUBIC)");
    VITK_REQUIRE((mean3 == nullptr) == (std3 == nullptr), "resample_images: mean and std go together");
    VITK_REQUIRE(vitk_aligned(dst, 16) && vitk_aligned(patches, 16) && vitk_aligned(desc_dev, 8), "resample_images: dst / desc alignment");
    int max_box_w = 1;
    for (int b = 0; b < B; ++b) {
        const vitk_resample_desc& d = desc_host[b];
        VITK_REQUIRE(d.src_h > 0 && d.src_w > 0 && d.src_offset >= 0 && d.src_offset + (int64_t)d.src_h * d.src_w * 3 <= src_bytes,
                     "resample_images: image %d (%d x %d at byte %lld) does not fit the %lld source bytes", b, d.src_h, d.src_w,
                     (long long)d.src_offset, (long long)src_bytes);
        VITK_REQUIRE(d.box_h > 0 && d.box_w > 0 && d.box_y >= 0 && d.box_x >= 0 && (int64_t)d.box_y + d.box_h <= d.src_h &&
                         (int64_t)d.box_x + d.box_w <= d.src_w,
                     "resample_images: image %d: box %d x %d at (%d, %d) outside the %d x %d image", b, d.box_h, d.box_w, d.box_y, d.box_x,
                     d.src_h, d.src_w);
        VITK_REQUIRE(d.out_h > 0 && d.out_w > 0 && d.box_w <= 12288, "resample_images: image %d: bad output size or a box wider than 12288", b);
        // vertical taps of one output row: 2 * ceil(radius * max(in / out, 1)) + 1
        const double ks = d.box_h > d.out_h ? (double)d.box_h / d.out_h : 1.0;
        VITK_REQUIRE(2.0 * (kernel == 0 ? 1.0 : 2.0) * ks + 3.0 <= RESAMPLE_MAX_TAPS,
                     "resample_images: image %d: a %d -> %d row reduction needs more than %d filter taps", b, d.box_h, d.out_h, RESAMPLE_MAX_TAPS);
        if (d.box_w > max_box_w) max_box_w = d.box_w;
    }
    const int normalise = mean3 != nullptr;
    const float m0 = normalise ? mean3[0] : 0.f, m1 = normalise ? mean3[1] : 0.f, m2 = normalise ? mean3[2] : 0.f;
    const float s0 = normalise ? std3[0] : 1.f, s1 = normalise ? std3[1] : 1.f, s2 = normalise ? std3[2] : 1.f;
    VITK_REQUIRE(s0 != 0.f && s1 != 0.f && s2 != 0.f, "resample_images: std must be non-zero");
    const size_t smem = ((size_t)RESAMPLE_MAX_TAPS + (size_t)max_box_w * 3) * sizeof(float);
    static VitkLdsOnce once = {};
    if (int rc_ = vitk_lds_once(once, (const void*)resample_kernel_rows, ((size_t)RESAMPLE_MAX_TAPS + 12288 * 3) * sizeof(float), "resample_images"))
        return rc_;
    resample_kernel_rows<<<B * S, 256, smem, (hipStream_t)stream>>>(src, desc_dev, dst, S, kernel, vmax - vmin, vmin, normalise, m0, m1, m2, s0,
                                                                    s1, s2, patches, patch, (long long)src_bytes, max_box_w);
    return vitk_check_launch("resample_images");
}

extern "C" {

int vitk_preprocess_images(const void* src, int src_is_u8, const unsigned char* flip, float* dst, int B, int H, int W, int src_h,
                           int src_w, int top, int left, float vmin, float vmax, const float* mean3, const float* std3,
                           void* stream) {
    VITK_REQUIRE(src && dst && B > 0 && H > 0 && W > 0, "preprocess_images: bad arguments");
    VITK_REQUIRE(W % 4 == 0, "preprocess_images: output width %d must be a multiple of 4", W);
    VITK_REQUIRE(top >= 0 && left >= 0 && top + H <= src_h && left + W <= src_w,
                 "preprocess_images: window %dx%d at (%d,%d) outside the %dx%d source", H, W, top, left, src_h, src_w);
    VITK_REQUIRE((mean3 == nullptr) == (std3 == nullptr), "preprocess_images: mean and std go together");
    VITK_REQUIRE(vitk_aligned(dst, 16) && vitk_aligned(src, src_is_u8 ? 4 : 16), "preprocess_images: src / dst alignment");
    const int normalise = mean3 != nullptr;
    const float m0 = normalise ? mean3[0] : 0.f, m1 = normalise ? mean3[1] : 0.f, m2 = normalise ? mean3[2] : 0.f;
    const float s0 = normalise ? std3[0] : 1.f, s1 = normalise ? std3[1] : 1.f, s2 = normalise ? std3[2] : 1.f;
    VITK_REQUIRE(s0 != 0.f && s1 != 0.f && s2 != 0.f, "preprocess_images: std must be non-zero");
    const float scale = vmax - vmin;
    const size_t items = (size_t)B * H * (W / 4);
    size_t g = (items + 255) / 256;
    if (g > 4096) g = 4096;
    if (src_is_u8)
        images_to_nchw_kernel<true><<<(int)g, 256, 0, (hipStream_t)stream>>>(src, flip, dst, B, H, W, src_h, src_w, top, left, scale,
                                                                           vmin, normalise, m0, m1, m2, s0, s1, s2);
    else
        images_to_nchw_kernel<false><<<(int)g, 256, 0, (hipStream_t)stream>>>(src, flip, dst, B, H, W, src_h, src_w, top, left, scale,
                                                                            vmin, normalise, m0, m1, m2, s0, s1, s2);
    return vitk_check_launch("preprocess_images");
}

int vitk_preprocess_patches(const void* src, int src_is_u8, const unsigned char* flip, void* dst, int patch, int B, int H, int W, int src_h,
                            int src_w, int top, int left, float vmin, float vmax, const float* mean3, const float* std3, void* stream) {
    VITK_REQUIRE(src && dst && B > 0 && H > 0 && W > 0, "preprocess_patches: bad arguments");
    VITK_REQUIRE(patch > 0 && patch % 8 == 0 && H % patch == 0 && W % patch == 0,
                 "preprocess_patches: the patch size %d must be a multiple of 8 that divides the %d x %d window", patch, H, W);
    VITK_REQUIRE(top >= 0 && left >= 0 && (int64_t)top + H <= src_h && (int64_t)left + W <= src_w,
                 "preprocess_patches: window %dx%d at (%d,%d) outside the %dx%d source", H, W, top, left, src_h, src_w);
    VITK_REQUIRE((mean3 == nullptr) == (std3 == nullptr), "preprocess_patches: mean and std go together");
    VITK_REQUIRE(vitk_aligned(dst, 16) && vitk_aligned(src, src_is_u8 ? 4 : 16), "preprocess_patches: src / dst alignment");
    const int normalise = mean3 != nullptr;
    const float m0 = normalise ? mean3[0] : 0.f, m1 = normalise ? mean3[1] : 0.f, m2 = normalise ? mean3[2] : 0.f;
    const float s0 = normalise ? std3[0] : 1.f, s1 = normalise ? std3[1] : 1.f, s2 = normalise ? std3[2] : 1.f;
    VITK_REQUIRE(s0 != 0.f && s1 != 0.f && s2 != 0.f, "preprocess_patches: std must be non-zero");
    const float scale = vmax - vmin;
    const size_t items = (size_t)B * H * (W / 8);
    size_t g = (items + 255) / 256;
    if (g > 4096) g = 4096;
    if (src_is_u8)
        images_to_patches_kernel<true><<<(int)g, 256, 0, (hipStream_t)stream>>>(src, flip, (bf16_t*)dst, B, H, W, src_h, src_w, top, left, patch,
                                                                              scale, vmin, normalise, m0, m1, m2, s0, s1, s2);
    else
        images_to_patches_kernel<false><<<(int)g, 256, 0, (hipStream_t)stream>>>(src, flip, (bf16_t*)dst, B, H, W, src_h, src_w, top, left, patch,
                                                                               scale, vmin, normalise, m0, m1, m2, s0, s1, s2);
    return vitk_check_launch("preprocess_patches");
}

}  // extern "C"
