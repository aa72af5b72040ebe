// Fused log-softmax losses on fp32 logits (forward value and gradient in one pass), one 256-thread block per row.
#include "common.h"

static __device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
static __device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// hard labels + label smoothing
__global__ __launch_bounds__(256) void xent_label_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ target,
                                                         float smoothing, float* __restrict__ loss_rows, float* __restrict__ dlogits,
                                                         int lddl, float grad_scale, int C) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    const float* x = logits + (size_t)row * ld;
    float mx = -INFINITY, sx = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { const float v = x[c]; mx = fmaxf(mx, v); sx += v; }
    mx = block_max(mx, sh);
    sx = block_sum(sx, sh);
    float se = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) se += __expf(x[c] - mx);
    se = block_sum(se, sh);
    const float lse = mx + __logf(se);
    // A label outside [0, C) (a foreign ignore_index, a class count that does not match the head) must never become an
    // address: the row's loss and gradient are NaN, so the step fails loudly where torch would raise.
    const int64_t t64 = target[row];
    const bool valid = t64 >= 0 && t64 < (int64_t)C;
    const int t = valid ? (int)t64 : 0;
    if (threadIdx.x == 0) {
        const float nll = lse - x[t];
        const float smooth = lse - sx / (float)C;
        loss_rows[row] = valid ? (1.0f - smoothing) * nll + smoothing * smooth : __builtin_nanf("");
    }
    if (dlogits != nullptr) {
        float* d = dlogits + (size_t)row * lddl;
        const float inv = 1.0f / se, u = smoothing / (float)C;
        for (int c = threadIdx.x; c < C; c += 256) {
            float g = __expf(x[c] - mx) * inv - u;
            if (c == t) g -= (1.0f - smoothing);
            d[c] = valid ? g * grad_scale : __builtin_nanf("");
        }
    }
}

// soft targets: loss = sum_c -t_c * logp_c
__global__ __launch_bounds__(256) void xent_soft_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ target,
                                                        int ldt, float* __restrict__ loss_rows, float* __restrict__ dlogits, int lddl,
                                                        float grad_scale, int C) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    const float* x = logits + (size_t)row * ld;
    const float* tg = target + (size_t)row * ldt;
    float mx = -INFINITY, st = 0.f, stx = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { const float v = x[c], t = tg[c]; mx = fmaxf(mx, v); st += t; stx += t * v; }
    mx = block_max(mx, sh);
    st = block_sum(st, sh);
    stx = block_sum(stx, sh);
    float se = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) se += __expf(x[c] - mx);
    se = block_sum(se, sh);
    const float lse = mx + __logf(se);
    if (threadIdx.x == 0) loss_rows[row] = lse * st - stx;
    if (dlogits != nullptr) {
        float* d = dlogits + (size_t)row * lddl;
        const float inv = 1.0f / se;
        for (int c = threadIdx.x; c < C; c += 256) d[c] = (__expf(x[c] - mx) * inv * st - tg[c]) * grad_scale;
    }
}

// T^2 * KL(softmax(t/T) || softmax(s/T)) per row; gradient w.r.t. the student logits
__global__ __launch_bounds__(256) void kd_kl_kernel(const float* __restrict__ student, int lds_, const float* __restrict__ teacher,
                                                    int ldt, float T, float* __restrict__ loss_rows, float* __restrict__ dstudent,
                                                    int ldd, float grad_scale, int C) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    const float* s = student + (size_t)row * lds_;
    const float* t = teacher + (size_t)row * ldt;
    const float invT = 1.0f / T;
    float ms = -INFINITY, mt = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) { ms = fmaxf(ms, s[c] * invT); mt = fmaxf(mt, t[c] * invT); }
    ms = block_max(ms, sh);
    mt = block_max(mt, sh);
    float es = 0.f, et = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { es += __expf(s[c] * invT - ms); et += __expf(t[c] * invT - mt); }
    es = block_sum(es, sh);
    et = block_sum(et, sh);
    const float lses = ms + __logf(es), lset = mt + __logf(et);
    float kl = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float lpt = t[c] * invT - lset, lps = s[c] * invT - lses;
        const float pt = __expf(lpt);
        kl += pt > 0.f ? pt * (lpt - lps) : 0.f;
    }
    kl = block_sum(kl, sh);
    if (threadIdx.x == 0) loss_rows[row] = kl * T * T;
    if (dstudent != nullptr) {
        float* d = dstudent + (size_t)row * ldd;
        for (int c = threadIdx.x; c < C; c += 256) {
            const float ps = __expf(s[c] * invT - lses), pt = __expf(t[c] * invT - lset);
            d[c] = (ps - pt) * T * grad_scale;
        }
    }
}

// first maximal index per row (torch.max / argmax tie rule)
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int ld, int64_t* __restrict__ idx, int C) {
    __shared__ float shv[4];
    __shared__ int shi[4];
    const int row = blockIdx.x;
    const float* x = logits + (size_t)row * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float v = x[c];
        if (v > bv || (v == bv && c < bi) || bi == 0x7fffffff) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { shv[wave] = bv; shi[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (shv[w] > bv || (shv[w] == bv && shi[w] < bi)) { bv = shv[w]; bi = shi[w]; }
        idx[row] = bi;
    }
}

extern "C" {

int vitk_xent_label_fwd_bwd(const float* logits, int ld, const int64_t* target, float smoothing, float* loss_rows,
                            float* dlogits, int lddl, float grad_scale, int B, int C, void* stream) {
    VITK_REQUIRE(logits && target && loss_rows, "xent_label: null pointer");
    VITK_REQUIRE(B > 0 && C > 0 && ld >= C && (!dlogits || lddl >= C), "xent_label: bad shape B=%d C=%d ld=%d", B, C, ld);
    xent_label_kernel<<<B, 256, 0, (hipStream_t)stream>>>(logits, ld, target, smoothing, loss_rows, dlogits, lddl, grad_scale, C);
    return vitk_check_launch("xent_label");
}

int vitk_xent_soft_fwd_bwd(const float* logits, int ld, const float* target, int ldt, float* loss_rows, float* dlogits,
                           int lddl, float grad_scale, int B, int C, void* stream) {
    VITK_REQUIRE(logits && target && loss_rows, "xent_soft: null pointer");
    VITK_REQUIRE(B > 0 && C > 0 && ld >= C && ldt >= C && (!dlogits || lddl >= C), "xent_soft: bad shape B=%d C=%d", B, C);
    xent_soft_kernel<<<B, 256, 0, (hipStream_t)stream>>>(logits, ld, target, ldt, loss_rows, dlogits, lddl, grad_scale, C);
    return vitk_check_launch("xent_soft");
}

int vitk_kd_kl_fwd_bwd(const float* student, int lds_, const float* teacher, int ldt, float temperature, float* loss_rows,
                       float* dstudent, int ldd, float grad_scale, int B, int C, void* stream) {
    VITK_REQUIRE(student && teacher && loss_rows, "kd_kl: null pointer");
    VITK_REQUIRE(B > 0 && C > 0 && lds_ >= C && ldt >= C && temperature > 0.f && (!dstudent || ldd >= C), "kd_kl: bad shape");
    kd_kl_kernel<<<B, 256, 0, (hipStream_t)stream>>>(student, lds_, teacher, ldt, temperature, loss_rows, dstudent, ldd, grad_scale, C);
    return vitk_check_launch("kd_kl");
}

int vitk_argmax_rows(const float* logits, int ld, int64_t* idx, int B, int C, void* stream) {
    VITK_REQUIRE(logits && idx && B > 0 && C > 0 && ld >= C, "argmax_rows: bad arguments");
    argmax_kernel<<<B, 256, 0, (hipStream_t)stream>>>(logits, ld, idx, C);
    return vitk_check_launch("argmax_rows");
}

}  // extern "C"
