// Fused AdamW over a flat fp32 range: reads p,g,m,v (16 B/param), writes p,m,v (12 B) + bf16 copy (2 B)
// [+ EMA read/write 8 B].  HBM-bound.  Arithmetic follows torch.optim.AdamW (decoupled decay, bias-corrected).
#include "common.h"

bool vitk_prof_begin(void* stream, double flop, int kind);
void vitk_prof_end(void* stream);

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ pb, float* __restrict__ ema,
                                                    float ema_decay, float omd, int64_t n, float beta2, float omb1, float omb2, float eps,
                                                    float decay_mul, float step_size, float inv_bc2_sqrt, float grad_scale) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 p4 = reinterpret_cast<float4*>(p)[i];
        const float4 g4 = reinterpret_cast<const float4*>(g)[i];
        float4 m4 = reinterpret_cast<float4*>(m)[i];
        float4 v4 = reinterpret_cast<float4*>(v)[i];
        float pp[4] = {p4.x, p4.y, p4.z, p4.w};
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
        float mm[4] = {m4.x, m4.y, m4.z, m4.w};
        float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] * grad_scale;
            pp[j] *= decay_mul;
            mm[j] = mm[j] + (gr - mm[j]) * omb1;                      // lerp_(grad, 1-beta1)
            vv[j] = vv[j] * beta2 + omb2 * gr * gr;
            const float denom = sqrtf(vv[j]) * inv_bc2_sqrt + eps;
            pp[j] -= step_size * (mm[j] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (pb != nullptr) {
            uint2 w;
            w.x = pack2bf(pp[0], pp[1]);
            w.y = pack2bf(pp[2], pp[3]);
            reinterpret_cast<uint2*>(pb)[i] = w;
        }
        if (ema != nullptr) {
            float4 e4 = reinterpret_cast<float4*>(ema)[i];
            e4.x = e4.x * ema_decay + omd * pp[0];
            e4.y = e4.y * ema_decay + omd * pp[1];
            e4.z = e4.z * ema_decay + omd * pp[2];
            e4.w = e4.w * ema_decay + omd * pp[3];
            reinterpret_cast<float4*>(ema)[i] = e4;
        }
    }
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            const float gr = g[i] * grad_scale;
            float pp = p[i] * decay_mul;
            const float mm = m[i] + (gr - m[i]) * omb1;
            const float vv = v[i] * beta2 + omb2 * gr * gr;
            pp -= step_size * (mm / (sqrtf(vv) * inv_bc2_sqrt + eps));
            p[i] = pp; m[i] = mm; v[i] = vv;
            if (pb != nullptr) pb[i] = f2bf(pp);
            if (ema != nullptr) ema[i] = ema[i] * ema_decay + omd * pp;
        }
    }
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = g[i];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ out, int accumulate) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = sh[0] + sh[1] + sh[2] + sh[3];
        out[0] = accumulate ? out[0] + t : t;
    }
}

#define SUMSQ_BLOCKS 1024

// ema = decay * ema + (1 - decay) * p over a flat fp32 range
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, float decay, float omd, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 e = reinterpret_cast<float4*>(ema)[i];
        const float4 q = reinterpret_cast<const float4*>(p)[i];
        e.x = e.x * decay + omd * q.x; e.y = e.y * decay + omd * q.y;
        e.z = e.z * decay + omd * q.z; e.w = e.w * decay + omd * q.w;
        reinterpret_cast<float4*>(ema)[i] = e;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) ema[i] = ema[i] * decay + omd * p[i];
}

// g *= scale[0]-derived clip coefficient: coef = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))  (clip_grad_norm_)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ sumsq, float max_norm) {
    const float total = sqrtf(sumsq[0]);
    float coef = max_norm / (total + 1e-6f);
    if (coef >= 1.0f) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] *= coef;
}

extern "C" {

int vitk_ema_update(float* ema, const float* p, double decay, int64_t n, void* stream) {
    VITK_REQUIRE(ema && p && n > 0, "ema_update: bad arguments");
    VITK_REQUIRE(vitk_aligned(ema, 16) && vitk_aligned(p, 16), "ema_update: pointers must be 16-byte aligned");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    ema_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(ema, p, (float)decay, (float)(1.0 - decay), n);
    return vitk_check_launch("ema_update");
}

int vitk_clip_grad_scale(float* g, int64_t n, const float* sumsq, float max_norm, void* stream) {
    VITK_REQUIRE(g && sumsq && n > 0 && max_norm > 0.f, "clip_grad_scale: bad arguments");
    clip_scale_kernel<<<2048, 256, 0, (hipStream_t)stream>>>(g, n, sumsq, max_norm);
    return vitk_check_launch("clip_grad_scale");
}


int vitk_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, float* ema, double ema_decay, int64_t n,
                    double lr, double beta1, double beta2, double eps, double weight_decay, int step, float grad_scale,
                    void* stream) {
    VITK_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw_step: bad arguments (n=%lld step=%d)", (long long)n, step);
    VITK_REQUIRE(vitk_aligned(p, 16) && vitk_aligned(g, 16) && vitk_aligned(m, 16) && vitk_aligned(v, 16) &&
                     (!p_bf16 || vitk_aligned(p_bf16, 8)) && (!ema || vitk_aligned(ema, 16)),
                 "adamw_step: pointers must be 16-byte aligned (bf16 copy: 8)");
    // host-side scalars in double, as torch's single-tensor path computes them in Python floats
    // (hyper-parameters arrive as doubles for that reason: 1 - 0.999f is off by 1.3e-5 relative from 1 - 0.999)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const float decay_mul = (float)(1.0 - lr * weight_decay);
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    const float omd = (float)(1.0 - ema_decay);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const bool pr = vitk_prof_begin(stream, (28.0 + (p_bf16 ? 2.0 : 0.0) + (ema ? 8.0 : 0.0)) * (double)n, VITK_PROF_ADAMW);
    adamw_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, (bf16_t*)p_bf16, ema, (float)ema_decay, omd, n, (float)beta2, omb1, omb2,
                                                              (float)eps, decay_mul, step_size, inv_bc2_sqrt, grad_scale);
    if (pr) vitk_prof_end(stream);
    return vitk_check_launch("adamw_step");
}

size_t vitk_sumsq_workspace(int64_t n) { return n > 0 ? SUMSQ_BLOCKS * sizeof(float) : 0; }

int vitk_sumsq(const float* g, int64_t n, float* out, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    VITK_REQUIRE(g && out && workspace && n > 0, "sumsq: bad arguments");
    if (workspace_bytes < vitk_sumsq_workspace(n)) {
        vitk_set_error("sumsq: workspace %zu < %zu", workspace_bytes, vitk_sumsq_workspace(n));
        return VITK_ENOSPC;
    }
    sumsq_partial_kernel<<<SUMSQ_BLOCKS, 256, 0, (hipStream_t)stream>>>(g, n, (float*)workspace);
    int rc = vitk_check_launch("sumsq_partial");
    if (rc) return rc;
    sumsq_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>((const float*)workspace, SUMSQ_BLOCKS, out, accumulate);
    return vitk_check_launch("sumsq_final");
}

}  // extern "C"
