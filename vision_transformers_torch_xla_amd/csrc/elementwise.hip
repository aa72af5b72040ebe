// HBM-bound helpers around the GEMM/attention core: patch gather (im2col), prefix tokens + position table,
// token pooling, per-sample row scaling (stochastic depth), casts and the bf16 weight transpose.
#include "common.h"

#define EW_BLOCK 256
static inline int ew_grid(size_t work_items) {
    size_t g = (work_items + EW_BLOCK - 1) / EW_BLOCK;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// img NCHW fp32 -> patches bf16 [B*gh*gw, C*p*p]; one thread = 8 consecutive kx of one (patch, c, ky) row:
// 32-byte contiguous fp32 read, 16-byte bf16 write.  p % 8 == 0.
__global__ void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int C, int H, int W, int p) {
    const int gh = H / p, gw = W / p;
    const int kvec = C * p * p / 8;                       // 8-element vectors per patch row
    const size_t total = (size_t)B * gh * gw * kvec;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kv = (int)(i % kvec);
        const size_t prow = i / kvec;
        const int k = kv * 8;
        const int c = k / (p * p), ky = (k / p) % p, kx = k % p;
        const int gx = (int)(prow % gw), gy = (int)((prow / gw) % gh), b = (int)(prow / ((size_t)gw * gh));
        const float* src = img + (((size_t)b * C + c) * H + (gy * p + ky)) * W + gx * p + kx;
        const float4 a = *reinterpret_cast<const float4*>(src);
        const float4 d = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, d.x, d.y, d.z, d.w};
        store8(out + prow * (size_t)(kvec * 8) + k, v);
    }
}

__global__ void prefix_tokens_kernel(const float* __restrict__ tokens, const float* __restrict__ pos, bf16_t* __restrict__ x,
                                     int B, int N, int D, int prefix) {
    const size_t total = (size_t)B * prefix * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int t = (int)((i / D) % prefix);
        const int b = (int)(i / ((size_t)D * prefix));
        x[((size_t)b * N + t) * D + d] = f2bf(tokens[(size_t)t * D + d] + pos[(size_t)t * D + d]);
    }
}

// dpos[t][:] = sum_b dx[b][t][:].  One block per token t: thread = (8-column vector cv, batch group bg); every group sums
// its share of the batch with 16-byte loads, the groups are folded through LDS in a fixed order (deterministic).
__global__ void pos_tokens_bwd_kernel(const bf16_t* __restrict__ dx, float* __restrict__ dpos, float* __restrict__ dtok,
                                      int accumulate, int B, int N, int D, int prefix, int groups) {
    extern __shared__ float ptb_smem[];     // [groups][D]
    const int nv = D / 8;
    const int t = blockIdx.x;
    const int cv = threadIdx.x % nv, bg = threadIdx.x / nv;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (bg < groups) {
        for (int b = bg; b < B; b += groups) {
            float v[8];
            load8(dx + ((size_t)b * N + t) * D + cv * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += v[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ptb_smem[bg * D + cv * 8 + j] = s[j];
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float r = 0.f;
        for (int k = 0; k < groups; ++k) r += ptb_smem[k * D + d];
        float* pp = dpos + (size_t)t * D + d;
        *pp = accumulate ? *pp + r : r;
        if (t < prefix && dtok != nullptr) {
            float* tp = dtok + (size_t)t * D + d;
            *tp = accumulate ? *tp + r : r;
        }
    }
}

// y[b][:] = mean_{t>=first} x[b][t][:].  Block = (sample b, 32 column vectors); thread = (vector, one of 8 token groups);
// the token groups are folded through LDS in a fixed order.
__global__ __launch_bounds__(256) void pool_avg_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int N, int D,
                                                            int first) {
    __shared__ float part[8][32 * 8 + 4];
    const int nv = D / 8;
    const int cgroups = (nv + 31) / 32;
    const int b = blockIdx.x / cgroups, cg = blockIdx.x % cgroups;
    const int cv = cg * 32 + (threadIdx.x & 31), tg = threadIdx.x >> 5;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (cv < nv) {
        for (int t = first + tg; t < N; t += 8) {
            float v[8];
            load8(x + ((size_t)b * N + t) * D + cv * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[tg][(threadIdx.x & 31) * 8 + j] = s[j];
    __syncthreads();
    if (tg == 0 && cv < nv) {
        const float inv = 1.0f / (float)(N - first);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) r += part[k][(threadIdx.x & 31) * 8 + j];
            o[j] = r * inv;
        }
        store8(y + (size_t)b * D + cv * 8, o);
    }
}

__global__ void pool_avg_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int B, int N, int D, int first) {
    const int nv = D / 8;
    const size_t total = (size_t)B * N * nv;
    const float inv = 1.0f / (float)(N - first);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nv) * 8;
        const int t = (int)((i / nv) % N);
        const int b = (int)(i / ((size_t)nv * N));
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t >= first) {
            load8(dy + (size_t)b * D + c, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= inv;
        }
        store8(dx + ((size_t)b * N + t) * D + c, v);
    }
}

__global__ void gather_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int N, int D, int token) {
    const int nv = D / 8;
    const size_t total = (size_t)B * nv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nv) * 8;
        const int b = (int)(i / nv);
        *reinterpret_cast<uint4*>(y + (size_t)b * D + c) = *reinterpret_cast<const uint4*>(x + ((size_t)b * N + token) * D + c);
    }
}

// dx[b][token] = dy[b]; with zero_others every other row of dx is cleared
__global__ void scatter_rows_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int B, int N, int D, int token,
                                    int zero_others) {
    const int nv = D / 8;
    const size_t total = zero_others ? (size_t)B * N * nv : (size_t)B * nv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nv) * 8;
        int t, b;
        if (zero_others) { t = (int)((i / nv) % N); b = (int)(i / ((size_t)nv * N)); }
        else { t = token; b = (int)(i / nv); }
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t == token) v = *reinterpret_cast<const uint4*>(dy + (size_t)b * D + c);
        *reinterpret_cast<uint4*>(dx + ((size_t)b * N + t) * D + c) = v;
    }
}

// y[b][t-first][:] = x[b][t][:] for t >= first (drops the prefix tokens; compact patch-token rows)
__global__ void slice_tokens_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int N, int D, int first) {
    const int nv = D / 8;
    const int T = N - first;
    const size_t total = (size_t)B * T * nv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nv) * 8;
        const int t = (int)((i / nv) % T);
        const int b = (int)(i / ((size_t)nv * T));
        *reinterpret_cast<uint4*>(y + ((size_t)b * T + t) * D + c) =
            *reinterpret_cast<const uint4*>(x + ((size_t)b * N + first + t) * D + c);
    }
}

__global__ void rowscale_kernel(const bf16_t* __restrict__ x, const float* __restrict__ rowscale, int rows_per_group,
                                bf16_t* __restrict__ y, int M, int D) {
    const int nv = D / 8;
    const size_t total = (size_t)M * nv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / nv);
        const float sc = rowscale[m / rows_per_group];
        float v[8];
        load8(x + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= sc;
        store8(y + i * 8, v);
    }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    const int64_t n8 = n / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = *reinterpret_cast<const float4*>(src + i * 8);
        const float4 b = *reinterpret_cast<const float4*>(src + i * 8 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        store8(dst + i * 8, v);
    }
    if (blockIdx.x == 0) {
        for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) dst[i] = f2bf(src[i]);
    }
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = bf2f(src[i]);
}
__global__ void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ y, int64_t n) {
    const int64_t n8 = n / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float u[8], v[8];
        load8(a + i * 8, u);
        load8(b + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] += v[j];
        store8(y + i * 8, u);
    }
    if (blockIdx.x == 0) {
        for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) y[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
    }
}

// dst[c][r] = src[r][c], 64x64 tiles through LDS (padded), 256 threads
__global__ void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int gr = r0 + r, gc = c0 + tx;
        tile[r][tx] = (gr < rows && gc < cols) ? src[(size_t)gr * cols + gc] : (bf16_t)0;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        const int gc = c0 + c, gr = r0 + tx;
        if (gc < cols && gr < rows) dst[(size_t)gc * rows + gr] = tile[tx][c];
    }
}

// The same for a list of matrices in one launch.  desc[i] = {src address, dst address, rows, cols, first tile}: five 64-bit
// words per matrix, first tile = number of 64x64 tiles of the matrices before it; desc[n].first tile = total tiles.
__global__ void transpose_bf16_batched_kernel(const long long* __restrict__ desc, int n) {
    __shared__ bf16_t tile[64][66];
    int lo = 0, hi = n - 1;                 // last matrix whose first tile <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid * 5 + 4] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const bf16_t* src = reinterpret_cast<const bf16_t*>(desc[lo * 5]);
    bf16_t* dst = reinterpret_cast<bf16_t*>(desc[lo * 5 + 1]);
    const int rows = (int)desc[lo * 5 + 2], cols = (int)desc[lo * 5 + 3];
    const int t = (int)(blockIdx.x - desc[lo * 5 + 4]);
    const int tiles_c = (cols + 63) / 64;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int gr = r0 + r, gc = c0 + tx;
        tile[r][tx] = (gr < rows && gc < cols) ? src[(size_t)gr * cols + gc] : (bf16_t)0;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        const int gc = c0 + c, gr = r0 + tx;
        if (gc < cols && gr < rows) dst[(size_t)gc * rows + gr] = tile[tx][c];
    }
}

// ---------------------------------------------------------------------------------------- mixup / cutmix (batch mode)
// timm.data.Mixup in 'batch' mode, applied by the reference before every training step (main.py:622-629,
// engine.py:132-133, 259-260).  Sample b is paired with sample B-1-b (x.flip(0)).
//   mixup : x = x.mul_(lam).add_(x.flip(0).mul_(1-lam))  -> both products are rounded before the sum: floating-point
//           contraction is switched off in these kernels (hipcc would otherwise fuse product and sum into one fma, and HIP's
//           __fmul_rn / __fadd_rn are plain operators that do not prevent it), which makes them bit-exact with that order.
//   cutmix: x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]  -> the two samples of a pair swap the box.
__global__ void mixup_pairs_kernel(float* __restrict__ x, int half, int B, size_t per4, float lam, float oml) {
#pragma clang fp contract(off)
    const size_t total = (size_t)half * per4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per4, e = i - b * per4;
        float4* pa = reinterpret_cast<float4*>(x) + b * per4 + e;
        float4* pc = reinterpret_cast<float4*>(x) + (size_t)(B - 1 - b) * per4 + e;
        const float4 a = *pa, c = *pc;
        float4 na, nc;
        na.x = a.x * lam + c.x * oml; nc.x = c.x * lam + a.x * oml;
        na.y = a.y * lam + c.y * oml; nc.y = c.y * lam + a.y * oml;
        na.z = a.z * lam + c.z * oml; nc.z = c.z * lam + a.z * oml;
        na.w = a.w * lam + c.w * oml; nc.w = c.w * lam + a.w * oml;
        *pa = na;
        *pc = nc;
    }
}

// one block per (pair, channel, box row); threads walk the box columns
__global__ void cutmix_swap_kernel(float* __restrict__ x, int B, int C, int H, int W, int yl, int yh, int xl, int xh) {
    const int bh = yh - yl;
    int r = blockIdx.x;
    const int y = yl + r % bh; r /= bh;
    const int c = r % C;
    const int b = r / C;
    float* pa = x + (((size_t)b * C + c) * H + y) * W;
    float* pc = x + (((size_t)(B - 1 - b) * C + c) * H + y) * W;
    for (int xx = xl + threadIdx.x; xx < xh; xx += blockDim.x) {
        const float a = pa[xx];
        pa[xx] = pc[xx];
        pc[xx] = a;
    }
}

// timm mixup_target: one_hot(target, on, off) * lam + one_hot(target.flip(0), on, off) * (1 - lam), products rounded first
__global__ void mixup_targets_kernel(const int64_t* __restrict__ target, float* __restrict__ out, int ldo, int B, int C, float lam,
                                     float oml, float on, float off) {
#pragma clang fp contract(off)
    const int b = blockIdx.x;
    const int64_t t1 = target[b], t2 = target[B - 1 - b];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float v1 = (c == t1) ? on : off, v2 = (c == t2) ? on : off;
        out[(size_t)b * ldo + c] = v1 * lam + v2 * oml;
    }
}

// timm.data.Mixup in 'elem' and 'pair' mode (Mixup._mix_elem / _mix_pair; the reference forwards --mixup_mode, main.py:229, 628):
// every sample (elem) or every pair (pair) has its own lambda, kind and box.  timm mixes sample i with the ORIGINAL of sample
// j = B-1-i (it keeps a clone of the batch); here one thread holds the originals of both samples of a pair in registers and
// writes both, so the batch is mixed in place without a copy.  Per sample: kind 0 = untouched (lambda = 1), 1 = mixup
// (x_i*lam + x_j*(1-lam), products rounded first), 2 = cutmix (the box comes from x_j).
__global__ void mixup_elem_kernel(float* __restrict__ x, int half, int B, int H, int W, size_t per4,
                                  const vitk_mixup_sample* __restrict__ prm) {
#pragma clang fp contract(off)
    const size_t total = (size_t)half * per4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per4, e = i - b * per4;
        const vitk_mixup_sample pa_ = prm[b], pc_ = prm[B - 1 - b];
        if (pa_.kind == 0 && pc_.kind == 0) continue;
        float4* pa = reinterpret_cast<float4*>(x) + b * per4 + e;
        float4* pc = reinterpret_cast<float4*>(x) + (size_t)(B - 1 - b) * per4 + e;
        const float4 a4 = *pa, c4 = *pc;
        const float a[4] = {a4.x, a4.y, a4.z, a4.w}, c[4] = {c4.x, c4.y, c4.z, c4.w};
        float na[4], nc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t idx = e * 4 + k;                 // element index inside the sample [C, H, W]
            const int xx = (int)(idx % W), yy = (int)((idx / W) % H);
            auto mixed = [&](const vitk_mixup_sample& p, float own, float other) {
                if (p.kind == 1) return own * p.lam + other * p.one_minus_lam;
                if (p.kind == 2 && yy >= p.yl && yy < p.yh && xx >= p.xl && xx < p.xh) return other;
                return own;
            };
            na[k] = mixed(pa_, a[k], c[k]);
            nc[k] = mixed(pc_, c[k], a[k]);
        }
        *pa = make_float4(na[0], na[1], na[2], na[3]);
        *pc = make_float4(nc[0], nc[1], nc[2], nc[3]);
    }
}

// timm mixup_target with one lambda per sample (a [B,1] tensor): y1 * lam_b + y2 * (1 - lam_b), the difference formed in fp32
__global__ void mixup_targets_elem_kernel(const int64_t* __restrict__ target, float* __restrict__ out, int ldo, int B, int C,
                                          const float* __restrict__ lam_rows, float on, float off) {
#pragma clang fp contract(off)
    const int b = blockIdx.x;
    const int64_t t1 = target[b], t2 = target[B - 1 - b];
    const float lam = lam_rows[b], oml = 1.0f - lam;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float v1 = (c == t1) ? on : off, v2 = (c == t2) ? on : off;
        out[(size_t)b * ldo + c] = v1 * lam + v2 * oml;
    }
}

// dst[list[i]] = src[list[i]] for whole samples of `per8` 16-byte pieces each: the samples that skip a residual branch under
// stochastic depth keep their activations (x + 0 * branch(x), timm DropPath as used at models/vision_transformer.py:176-177)
__global__ void copy_samples_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const int* __restrict__ list, int count,
                                    size_t per8) {
    const size_t total = (size_t)count * per8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t k = i / per8, e = i - k * per8;
        const size_t o = (size_t)list[k] * per8 + e;
        dst[o] = src[o];
    }
}

// Plain 16-byte copy.  The source may be pinned host memory (mapped into the device's address space): the per-step index tables
// of stochastic depth are fetched this way, as an ordinary launch on the compute stream, instead of through hipMemcpyAsync -
// whose host call was seen to block for 5-8 ms every few steps once the host runs several steps ahead, with the launch in
// flight at that moment stretched by the same amount.
__global__ void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// y[c * rpg + t][:] = rowscale[smap[c]] * x[smap[c] * rpg + t][:] for the nk samples of smap: the scaled gradient that enters a
// residual branch, compact over the samples that took the branch
__global__ void rowscale_gather_kernel(const bf16_t* __restrict__ x, const float* __restrict__ rowscale, int rows_per_group,
                                       const int* __restrict__ smap, int nk, bf16_t* __restrict__ y, int D) {
    const int nv = D / 8;
    const size_t total = (size_t)nk * rows_per_group * nv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / nv;
        const int c8 = (int)(i - row * nv) * 8;
        const int c = (int)(row / rows_per_group), t = (int)(row - (size_t)c * rows_per_group);
        const int smp = smap[c];
        const float sc = rowscale[smp];
        float v[8];
        load8(x + ((size_t)smp * rows_per_group + t) * D + c8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= sc;
        store8(y + row * D + c8, v);
    }
}

extern "C" {

int vitk_im2col_patches(const float* img, void* patches, int B, int C, int H, int W, int p, void* stream) {
    VITK_REQUIRE(img && patches, "im2col: null pointer");
    VITK_REQUIRE(B > 0 && C > 0 && p > 0 && p % 8 == 0 && H % p == 0 && W % p == 0,
                 "im2col: need p%%8==0 and H,W multiples of p (B=%d C=%d H=%d W=%d p=%d)", B, C, H, W, p);
    VITK_REQUIRE(vitk_aligned(img, 16) && vitk_aligned(patches, 16), "im2col: pointers must be 16-byte aligned");
    const size_t total = (size_t)B * (H / p) * (W / p) * (C * p * p / 8);
    im2col_kernel<<<ew_grid(total), EW_BLOCK, 0, (hipStream_t)stream>>>(img, (bf16_t*)patches, B, C, H, W, p);
    return vitk_check_launch("im2col");
}

int vitk_prefix_tokens_fwd(const float* tokens, const float* pos, void* x, int B, int N, int D, int prefix, void* stream) {
    VITK_REQUIRE(tokens && pos && x, "prefix_tokens_fwd: null pointer");
    VITK_REQUIRE(B > 0 && N > 0 && D > 0 && prefix > 0 && prefix <= N, "prefix_tokens_fwd: bad shape");
    prefix_tokens_kernel<<<ew_grid((size_t)B * prefix * D), EW_BLOCK, 0, (hipStream_t)stream>>>(tokens, pos, (bf16_t*)x, B, N, D, prefix);
    return vitk_check_launch("prefix_tokens_fwd");
}

int vitk_pos_tokens_bwd(const void* dx, float* dpos, float* dtokens, int accumulate, int B, int N, int D, int prefix, void* stream) {
    VITK_REQUIRE(dx && dpos, "pos_tokens_bwd: null pointer");
    VITK_REQUIRE(B > 0 && N > 0 && D > 0 && D % 8 == 0 && D <= 8192 && prefix >= 0 && prefix <= N, "pos_tokens_bwd: bad shape (D %% 8 == 0, D <= 8192)");
    VITK_REQUIRE(vitk_aligned(dx, 16), "pos_tokens_bwd: dx must be 16-byte aligned");
    const int nv = D / 8;
    int groups = 1024 / nv;
    if (groups > 12) groups = 12;          // groups * D * 4 bytes of LDS <= 48 KB at D = 1024
    if (groups > B) groups = B;
    if (groups < 1) groups = 1;
    pos_tokens_bwd_kernel<<<N, nv * groups, (size_t)groups * D * sizeof(float), (hipStream_t)stream>>>((const bf16_t*)dx, dpos, dtokens, accumulate, B, N, D,
                                                                                                   prefix, groups);
    return vitk_check_launch("pos_tokens_bwd");
}

int vitk_pool_avg_fwd(const void* x, void* y, int B, int N, int D, int first, void* stream) {
    VITK_REQUIRE(x && y, "pool_avg_fwd: null pointer");
    VITK_REQUIRE(B > 0 && N > first && first >= 0 && D % 8 == 0, "pool_avg_fwd: bad shape");
    VITK_REQUIRE(vitk_aligned(x, 16) && vitk_aligned(y, 16), "pool_avg_fwd: pointers must be 16-byte aligned");
    pool_avg_fwd_kernel<<<B * ((D / 8 + 31) / 32), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)y, B, N, D, first);
    return vitk_check_launch("pool_avg_fwd");
}

int vitk_pool_avg_bwd(const void* dy, void* dx, int B, int N, int D, int first, void* stream) {
    VITK_REQUIRE(dy && dx, "pool_avg_bwd: null pointer");
    VITK_REQUIRE(B > 0 && N > first && first >= 0 && D % 8 == 0, "pool_avg_bwd: bad shape");
    VITK_REQUIRE(vitk_aligned(dy, 16) && vitk_aligned(dx, 16), "pool_avg_bwd: pointers must be 16-byte aligned");
    pool_avg_bwd_kernel<<<ew_grid((size_t)B * N * D / 8), EW_BLOCK, 0, (hipStream_t)stream>>>((const bf16_t*)dy, (bf16_t*)dx, B, N, D, first);
    return vitk_check_launch("pool_avg_bwd");
}

int vitk_gather_rows(const void* x, void* y, int B, int N, int D, int token, void* stream) {
    VITK_REQUIRE(x && y, "gather_rows: null pointer");
    VITK_REQUIRE(B > 0 && N > 0 && token >= 0 && token < N && D % 8 == 0, "gather_rows: bad shape");
    VITK_REQUIRE(vitk_aligned(x, 16) && vitk_aligned(y, 16), "gather_rows: pointers must be 16-byte aligned");
    gather_rows_kernel<<<ew_grid((size_t)B * D / 8), EW_BLOCK, 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)y, B, N, D, token);
    return vitk_check_launch("gather_rows");
}

int vitk_scatter_rows(const void* dy, void* dx, int B, int N, int D, int token, int zero_others, void* stream) {
    VITK_REQUIRE(dy && dx, "scatter_rows: null pointer");
    VITK_REQUIRE(B > 0 && N > 0 && token >= 0 && token < N && D % 8 == 0, "scatter_rows: bad shape");
    VITK_REQUIRE(vitk_aligned(dy, 16) && vitk_aligned(dx, 16), "scatter_rows: pointers must be 16-byte aligned");
    const size_t total = zero_others ? (size_t)B * N * D / 8 : (size_t)B * D / 8;
    scatter_rows_kernel<<<ew_grid(total), EW_BLOCK, 0, (hipStream_t)stream>>>((const bf16_t*)dy, (bf16_t*)dx, B, N, D, token, zero_others);
    return vitk_check_launch("scatter_rows");
}

int vitk_slice_tokens(const void* x, void* y, int B, int N, int D, int first, void* stream) {
    VITK_REQUIRE(x && y, "slice_tokens: null pointer");
    VITK_REQUIRE(B > 0 && N > first && first >= 0 && D % 8 == 0, "slice_tokens: bad shape");
    VITK_REQUIRE(vitk_aligned(x, 16) && vitk_aligned(y, 16), "slice_tokens: pointers must be 16-byte aligned");
    slice_tokens_kernel<<<ew_grid((size_t)B * (N - first) * D / 8), EW_BLOCK, 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)y, B, N, D, first);
    return vitk_check_launch("slice_tokens");
}

int vitk_rowscale(const void* x, const float* rowscale, int rows_per_group, void* y, int M, int D, void* stream) {
    VITK_REQUIRE(x && rowscale && y, "rowscale: null pointer");
    VITK_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && rows_per_group > 0, "rowscale: bad shape");
    VITK_REQUIRE(vitk_aligned(x, 16) && vitk_aligned(y, 16), "rowscale: pointers must be 16-byte aligned");
    rowscale_kernel<<<ew_grid((size_t)M * D / 8), EW_BLOCK, 0, (hipStream_t)stream>>>((const bf16_t*)x, rowscale, rows_per_group, (bf16_t*)y, M, D);
    return vitk_check_launch("rowscale");
}

int vitk_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    VITK_REQUIRE(src && dst && n > 0, "cast_f32_to_bf16: bad arguments");
    VITK_REQUIRE(vitk_aligned(src, 16) && vitk_aligned(dst, 16), "cast_f32_to_bf16: pointers must be 16-byte aligned");
    cast_f32_bf16_kernel<<<ew_grid((size_t)(n / 8 + 1)), EW_BLOCK, 0, (hipStream_t)stream>>>(src, (bf16_t*)dst, n);
    return vitk_check_launch("cast_f32_to_bf16");
}

int vitk_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
    VITK_REQUIRE(src && dst && n > 0, "cast_bf16_to_f32: bad arguments");
    cast_bf16_f32_kernel<<<ew_grid((size_t)n), EW_BLOCK, 0, (hipStream_t)stream>>>((const bf16_t*)src, dst, n);
    return vitk_check_launch("cast_bf16_to_f32");
}

int vitk_transpose_bf16(const void* src, void* dst, int rows, int cols, void* stream) {
    VITK_REQUIRE(src && dst && rows > 0 && cols > 0, "transpose_bf16: bad arguments");
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    transpose_bf16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)src, (bf16_t*)dst, rows, cols);
    return vitk_check_launch("transpose_bf16");
}

int vitk_transpose_bf16_batched(const void* desc, int n, int total_tiles, void* stream) {
    VITK_REQUIRE(desc && n > 0 && total_tiles > 0, "transpose_bf16_batched: bad arguments");
    VITK_REQUIRE(vitk_aligned(desc, 8), "transpose_bf16_batched: descriptor table must be 8-byte aligned");
    transpose_bf16_batched_kernel<<<total_tiles, 256, 0, (hipStream_t)stream>>>((const long long*)desc, n);
    return vitk_check_launch("transpose_bf16_batched");
}

int vitk_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream) {
    VITK_REQUIRE(a && b && y && n > 0, "add_bf16: bad arguments");
    VITK_REQUIRE(vitk_aligned(a, 16) && vitk_aligned(b, 16) && vitk_aligned(y, 16), "add_bf16: pointers must be 16-byte aligned");
    add_bf16_kernel<<<ew_grid((size_t)(n / 8 + 1)), EW_BLOCK, 0, (hipStream_t)stream>>>((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n);
    return vitk_check_launch("add_bf16");
}

int vitk_mixup_batch(float* x, int B, int64_t per_sample, float lam, float one_minus_lam, void* stream) {
    VITK_REQUIRE(x && B > 0 && per_sample > 0, "mixup_batch: bad arguments");
    VITK_REQUIRE(B % 2 == 0, "mixup_batch: batch size should be even when using this (B=%d)", B);
    VITK_REQUIRE(per_sample % 4 == 0 && vitk_aligned(x, 16), "mixup_batch: elements per sample must be a multiple of 4, x 16-byte aligned");
    const size_t per4 = (size_t)per_sample / 4;
    mixup_pairs_kernel<<<ew_grid((size_t)(B / 2) * per4), EW_BLOCK, 0, (hipStream_t)stream>>>(x, B / 2, B, per4, lam, one_minus_lam);
    return vitk_check_launch("mixup_batch");
}

int vitk_cutmix_batch(float* x, int B, int C, int H, int W, int yl, int yh, int xl, int xh, void* stream) {
    VITK_REQUIRE(x && B > 0 && C > 0 && H > 0 && W > 0, "cutmix_batch: bad arguments");
    VITK_REQUIRE(B % 2 == 0, "cutmix_batch: batch size should be even when using this (B=%d)", B);
    VITK_REQUIRE(0 <= yl && yl <= yh && yh <= H && 0 <= xl && xl <= xh && xh <= W, "cutmix_batch: box [%d,%d)x[%d,%d) outside %dx%d", yl, yh,
                 xl, xh, H, W);
    if (yh == yl || xh == xl) return VITK_OK;       // empty box: nothing moves
    const long long blocks = (long long)(B / 2) * C * (yh - yl);
    VITK_REQUIRE(blocks <= 0x7fffffffLL, "cutmix_batch: too many rows");
    cutmix_swap_kernel<<<(unsigned)blocks, 64, 0, (hipStream_t)stream>>>(x, B, C, H, W, yl, yh, xl, xh);
    return vitk_check_launch("cutmix_batch");
}

int vitk_mixup_targets(const int64_t* target, float* out, int ldo, int B, int C, float lam, float one_minus_lam, float on_value,
                       float off_value, void* stream) {
    VITK_REQUIRE(target && out && B > 0 && C > 0 && ldo >= C, "mixup_targets: bad arguments");
    mixup_targets_kernel<<<B, 256, 0, (hipStream_t)stream>>>(target, out, ldo, B, C, lam, one_minus_lam, on_value, off_value);
    return vitk_check_launch("mixup_targets");
}

int vitk_mixup_elem(float* x, int B, int C, int H, int W, const vitk_mixup_sample* params_host, const vitk_mixup_sample* params_dev,
                    void* stream) {
    VITK_REQUIRE(x && params_host && params_dev && B > 0 && C > 0 && H > 0 && W > 0, "mixup_elem: bad arguments");
    VITK_REQUIRE(B % 2 == 0, "mixup_elem: batch size should be even when using this (B=%d)", B);
    const int64_t per_sample = (int64_t)C * H * W;
    VITK_REQUIRE(per_sample % 4 == 0 && vitk_aligned(x, 16) && vitk_aligned(params_dev, 4),
                 "mixup_elem: elements per sample must be a multiple of 4, x 16-byte aligned");
    for (int b = 0; b < B; ++b) {
        const vitk_mixup_sample& p = params_host[b];
        VITK_REQUIRE(p.kind >= 0 && p.kind <= 2, "mixup_elem: sample %d: kind %d is not 0 (none), 1 (mixup) or 2 (cutmix)", b, p.kind);
        VITK_REQUIRE(p.kind != 2 || (0 <= p.yl && p.yl <= p.yh && p.yh <= H && 0 <= p.xl && p.xl <= p.xh && p.xh <= W),
                     "mixup_elem: sample %d: box [%d,%d)x[%d,%d) outside %dx%d", b, p.yl, p.yh, p.xl, p.xh, H, W);
    }
    const size_t per4 = (size_t)per_sample / 4;
    mixup_elem_kernel<<<ew_grid((size_t)(B / 2) * per4), EW_BLOCK, 0, (hipStream_t)stream>>>(x, B / 2, B, H, W, per4, params_dev);
    return vitk_check_launch("mixup_elem");
}

int vitk_mixup_targets_elem(const int64_t* target, float* out, int ldo, int B, int C, const float* lam_rows, float on_value,
                            float off_value, void* stream) {
    VITK_REQUIRE(target && out && lam_rows && B > 0 && C > 0 && ldo >= C, "mixup_targets_elem: bad arguments");
    mixup_targets_elem_kernel<<<B, 256, 0, (hipStream_t)stream>>>(target, out, ldo, B, C, lam_rows, on_value, off_value);
    return vitk_check_launch("mixup_targets_elem");
}

int vitk_copy_samples(const void* src, void* dst, const int* sample_list, int count, int64_t elems_per_sample, void* stream) {
    VITK_REQUIRE(src && dst && sample_list && count > 0 && elems_per_sample > 0, "copy_samples: bad arguments");
    VITK_REQUIRE(elems_per_sample % 8 == 0 && vitk_aligned(src, 16) && vitk_aligned(dst, 16),
                 "copy_samples: bf16 elements per sample must be a multiple of 8, pointers 16-byte aligned");
    const size_t per8 = (size_t)elems_per_sample / 8;
    copy_samples_kernel<<<ew_grid((size_t)count * per8), EW_BLOCK, 0, (hipStream_t)stream>>>((const uint4*)src, (uint4*)dst, sample_list, count, per8);
    return vitk_check_launch("copy_samples");
}

int vitk_copy_from_host(const void* pinned_src, void* dst, int64_t bytes, void* stream) {
    VITK_REQUIRE(pinned_src && dst && bytes > 0, "copy_from_host: bad arguments");
    VITK_REQUIRE(bytes % 16 == 0 && vitk_aligned(pinned_src, 16) && vitk_aligned(dst, 16), "copy_from_host: 16-byte multiples and alignment");
    const size_t n16 = (size_t)bytes / 16;
    copy16_kernel<<<ew_grid(n16), EW_BLOCK, 0, (hipStream_t)stream>>>((const uint4*)pinned_src, (uint4*)dst, n16);
    return vitk_check_launch("copy_from_host");
}

int vitk_rowscale_gather(const void* x, const float* rowscale, int rows_per_group, const int* sample_map, int nk, void* y, int D,
                         void* stream) {
    VITK_REQUIRE(x && rowscale && sample_map && y, "rowscale_gather: null pointer");
    VITK_REQUIRE(nk > 0 && D > 0 && D % 8 == 0 && rows_per_group > 0, "rowscale_gather: bad shape");
    VITK_REQUIRE(vitk_aligned(x, 16) && vitk_aligned(y, 16), "rowscale_gather: pointers must be 16-byte aligned");
    rowscale_gather_kernel<<<ew_grid((size_t)nk * rows_per_group * D / 8), EW_BLOCK, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, rowscale, rows_per_group, sample_map, nk, (bf16_t*)y, D);
    return vitk_check_launch("rowscale_gather");
}

}  // extern "C"
