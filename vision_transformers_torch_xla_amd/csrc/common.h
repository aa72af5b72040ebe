// Shared device helpers and host-side error plumbing for libvitk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/vitk.h"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define VITK_WAVE 64

static __device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN)
static __device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// two floats -> one dword of two bf16 with a single v_cvt_pk_bf16_f32 (the vector cast; two scalar casts cost four ops)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
static __device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// 16 bytes per lane, global -> LDS: lane i's data (read at base + voff) lands at LDS byte m0 + 16*i.  Issued as inline
// assembly on purpose: when hipcc sees the LDS-DMA builtin in a loop it stops counting LDS reads and waits for ALL of them
// (lgkmcnt(0)) in front of every MFMA group, which serialises fragment reads and MFMAs; hidden from it, the reads get
// counted waits.  Completion is tracked by hand (s_waitcnt vmcnt before the publishing barrier).
static __device__ __forceinline__ void glds16(const bf16_t* base, unsigned voff, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_byte) : "memory", "m0");
}
static __device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) void*)p;
}
// the same with a full 64-bit address per lane
static __device__ __forceinline__ void glds16_v(const void* src, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte) : "memory", "m0");
}

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf GELU (nn.GELU() default) and its derivative, without erff's branches.  For u = min(|x|, 8):  0.5 * erfc(u / sqrt 2) = 2^(u * r(u) - 1) with r a degree-6
// polynomial fitted to log2(erfc(u / sqrt 2)) / u on [0, 8] (smooth: r(0) = -sqrt(2/pi) * log2 e).  Then
//   Phi(x) = x > 0 ? 1 - h : h          with h = 0.5 * erfc(u / sqrt 2),
//   gelu(x) = x * Phi(x) = max(x, 0) - u * h   (no cancellation in the negative tail, unlike 0.5 x (1 + erf)),
// relative error <= 4e-6 for x > 0 and <= 8e-5 in the far negative tail (outputs below 1e-6), i.e. far inside one bf16
// rounding step; |x| > 8 gives x or -0 up to 1e-14.  12 instructions instead of ~40.
static __device__ __forceinline__ float half_erfc_poly(float u) {
    float r = -1.3601819546238403e-06f;
    r = __builtin_fmaf(r, u, 5.1214541599620134e-05f);
    r = __builtin_fmaf(r, u, -0.0008435604977421463f);
    r = __builtin_fmaf(r, u, 0.008160343393683434f);
    r = __builtin_fmaf(r, u, -0.05328623577952385f);
    r = __builtin_fmaf(r, u, -0.4590398073196411f);
    r = __builtin_fmaf(r, u, -1.1510837078094482f);
    return __builtin_amdgcn_exp2f(__builtin_fmaf(r, u, -1.0f));
}
static __device__ __forceinline__ float gelu_erf(float x) {
    const float u = fminf(fabsf(x), 8.0f);
    return __builtin_fmaf(-u, half_erfc_poly(u), fmaxf(x, 0.0f));
}
// d/dx gelu(x) = Phi(x) + x * pdf(x)
static __device__ __forceinline__ float dgelu_erf(float x) {
    const float u = fminf(fabsf(x), 8.0f);
    const float h = half_erfc_poly(u);
    const float cdf = 0.5f + copysignf(0.5f - h, x);
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * u * u);
    return __builtin_fmaf(x, pdf, cdf);
}
// Two elements at a time: the polynomial runs on v_pk_fma_f32 (half the instructions of the VALU-bound GEMM epilogues).
static __device__ __forceinline__ f32x2_t half_erfc_poly2(f32x2_t u) {
    const f32x2_t c6 = {-1.3601819546238403e-06f, -1.3601819546238403e-06f}, c5 = {5.1214541599620134e-05f, 5.1214541599620134e-05f};
    const f32x2_t c4 = {-0.0008435604977421463f, -0.0008435604977421463f}, c3 = {0.008160343393683434f, 0.008160343393683434f};
    const f32x2_t c2 = {-0.05328623577952385f, -0.05328623577952385f}, c1 = {-0.4590398073196411f, -0.4590398073196411f};
    const f32x2_t c0 = {-1.1510837078094482f, -1.1510837078094482f}, m1 = {-1.0f, -1.0f};
    f32x2_t r = __builtin_elementwise_fma(c6, u, c5);
    r = __builtin_elementwise_fma(r, u, c4);
    r = __builtin_elementwise_fma(r, u, c3);
    r = __builtin_elementwise_fma(r, u, c2);
    r = __builtin_elementwise_fma(r, u, c1);
    r = __builtin_elementwise_fma(r, u, c0);
    r = __builtin_elementwise_fma(r, u, m1);
    return (f32x2_t){__builtin_amdgcn_exp2f(r[0]), __builtin_amdgcn_exp2f(r[1])};
}
static __device__ __forceinline__ void gelu_erf2(float x0, float x1, float& y0, float& y1) {
    const f32x2_t u = {fminf(fabsf(x0), 8.0f), fminf(fabsf(x1), 8.0f)};
    const f32x2_t h = half_erfc_poly2(u);
    const f32x2_t mx = {fmaxf(x0, 0.0f), fmaxf(x1, 0.0f)};
    const f32x2_t y = __builtin_elementwise_fma(-u, h, mx);
    y0 = y[0]; y1 = y[1];
}
static __device__ __forceinline__ void dgelu_erf2(float x0, float x1, float& y0, float& y1) {
    const f32x2_t u = {fminf(fabsf(x0), 8.0f), fminf(fabsf(x1), 8.0f)};
    const f32x2_t h = half_erfc_poly2(u);
    const f32x2_t k = {-0.72134752044448170368f, -0.72134752044448170368f};
    const f32x2_t e = (k * u) * u;
    const f32x2_t x = {x0, x1};
    const f32x2_t xs = x * (f32x2_t){0.39894228040143267794f, 0.39894228040143267794f};
    const f32x2_t pdf = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
    const f32x2_t cdf = {0.5f + copysignf(0.5f - h[0], x0), 0.5f + copysignf(0.5f - h[1], x1)};
    const f32x2_t y = __builtin_elementwise_fma(xs, pdf, cdf);
    y0 = y[0]; y1 = y[1];
}

// gelu(x) and gelu'(x) of two elements at once: they share u, the erfc polynomial and its exponential; the derivative adds the
// Gaussian's exponential
static __device__ __forceinline__ void gelu_dgelu_erf2(float x0, float x1, float& y0, float& y1, float& g0, float& g1) {
    const f32x2_t u = {fminf(fabsf(x0), 8.0f), fminf(fabsf(x1), 8.0f)};
    const f32x2_t h = half_erfc_poly2(u);
    const f32x2_t mx = {fmaxf(x0, 0.0f), fmaxf(x1, 0.0f)};
    const f32x2_t y = __builtin_elementwise_fma(-u, h, mx);
    const f32x2_t k = {-0.72134752044448170368f, -0.72134752044448170368f};
    const f32x2_t e = (k * u) * u;
    const f32x2_t x = {x0, x1};
    const f32x2_t xs = x * (f32x2_t){0.39894228040143267794f, 0.39894228040143267794f};
    const f32x2_t pdf = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
    const f32x2_t cdf = {0.5f + copysignf(0.5f - h[0], x0), 0.5f + copysignf(0.5f - h[1], x1)};
    const f32x2_t g = __builtin_elementwise_fma(xs, pdf, cdf);
    y0 = y[0]; y1 = y[1]; g0 = g[0]; g1 = g[1];
}

// load 8 bf16 (16 B) as floats
static __device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
static __device__ __forceinline__ uint4 pack8_bf16(const float (&v)[8]) {
    uint4 r;
    r.x = pack2bf(v[0], v[1]); r.y = pack2bf(v[2], v[3]); r.z = pack2bf(v[4], v[5]); r.w = pack2bf(v[6], v[7]);
    return r;
}
static __device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) { *reinterpret_cast<uint4*>(p) = pack8_bf16(v); }

// ---- host side
void vitk_set_error(const char* fmt, ...);
int vitk_check_launch(const char* what);

#define VITK_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            vitk_set_error(__VA_ARGS__);        \
            return VITK_EINVAL;                 \
        }                                       \
    } while (0)

static inline bool vitk_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Kernels that declare more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per kernel
// AND per device (a process that drives several GPUs sets it on each).  Returns 0 or VITK_ELAUNCH with the error text set.
struct VitkLdsOnce { bool done[64]; };
static inline int vitk_lds_once(VitkLdsOnce& st, const void* fn, size_t bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        vitk_set_error("%s: cannot query the current device", what);
        return VITK_ELAUNCH;
    }
    if (st.done[dev]) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        vitk_set_error("%s: cannot set LDS size %zu: %s", what, bytes, hipGetErrorString(e));
        return VITK_ELAUNCH;
    }
    st.done[dev] = true;
    return 0;
}
