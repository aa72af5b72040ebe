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

}  // extern "C"
