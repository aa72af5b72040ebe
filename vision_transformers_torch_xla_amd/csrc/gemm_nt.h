// What the NT GEMM kernels of libvitk share: the launch parameters, the tile-counter protocol of the persistent kernels, the
// XCD-contiguous tile order and the fused-epilogue arithmetic (gemm.hip: gemm_nt_kernel, gemm_ntp_kernel; gemm_ntq.hip: gemm_ntq_kernel).
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int V> struct itag { static constexpr int value = V; };

bool vitk_prof_begin(void* stream, double flop, int kind);
void vitk_prof_end(void* stream);

#define GBM 128
#define GBN 128
#define GBK 64

struct GemmNTParams {
    const bf16_t* A; int lda;
    const bf16_t* B; int ldb;
    void* C; int ldc;
    int M, N, K;
    const float* bias;
    const bf16_t* res; int ldres;
    const float* rowscale; int rows_per_group;
    bf16_t* out2; int ldout2;
    const bf16_t* aux; int ldaux;
    const float* pos; int patches; int prefix;
    int tiles_n; int num_tiles;
    int tiles_m; int cb;          // persistent kernel: row tiles; column tiles per block of the tile order (cb = tiles_n: row-major)
    const int* smap; int rps;     // BIAS_RES only: output / residual row of GEMM row m is smap[m / rps] * rps + m % rps
    unsigned* sched;              // persistent kernel: this launch's tile counters, [8] per XCD + [8] = workgroups that have left
    unsigned* sched_err;          // persistent kernel: sticky error word (host memory mapped into the device), null: no reporting
    void* sink;                   // four-wave kernel: 1 KB of device memory that lanes past the matrix edge store into (never read)
};

// Tile counters of the persistent kernels.  Word x (0..7) of a counter set belongs to the workgroups with blockIdx % 8 == x: its
// low 24 bits count the claims made so far, its top byte the workgroups that have entered the kernel (the entry atomic adds
// NTP_ARRIVE + 1: arrival ticket and first claim in one operation).  A set must be all zero when a launch starts; the kernel
// cannot see that directly (other workgroups of the same launch may have been there first), but two things it can see: a
// workgroup whose ticket is 0 must find no claims, and no ticket may reach the number of workgroups the launch has on that
// counter - a set left dirty by a launch that never finished, or shared with a launch that runs at the same time (a captured
// graph replayed on another stream than the one it was captured on), breaks one of them in some workgroup.  That workgroup
// writes the sticky error word; the next GEMM call on the host refuses to launch and reports it.
#define NTP_ARRIVE 0x01000000u
#define NTP_CLAIM_MASK 0x00ffffffu
static __device__ __forceinline__ void ntp_check_entry(const GemmNTParams& p, unsigned first, int xcd, int G) {
    const unsigned arrivals = first >> 24, claims = first & NTP_CLAIM_MASK;
    const unsigned wgs = (unsigned)((G - xcd + 7) >> 3);
    const int later = p.num_tiles - G - xcd + 7;
    const unsigned limit = wgs + (later > 0 ? (unsigned)(later >> 3) : 0u);
    if ((arrivals >= wgs || (arrivals == 0u && claims != 0u) || claims >= limit) && p.sched_err != nullptr)
        __hip_atomic_store(p.sched_err, 0x80000000u | ((unsigned)xcd << 24) | claims, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Blocks are dealt round-robin over the 8 XCDs (each with a private L2).  Remap so that every XCD works on a
// contiguous range of logical tiles (neighbouring tiles share the A row panel); bijective for any grid size.
static __device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// W (4 or 8) consecutive bf16 <-> fp32
template <int W>
static __device__ __forceinline__ void store_bf16(bf16_t* p, const float (&v)[W]) {
    if (W == 4) {
        uint2 r;
        r.x = pack2bf(v[0], v[1]);
        r.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = r;
    } else {
        // streaming store: a tile round writes as much as an XCD's whole L2 holds; kept out of it, the weight panel and
        // the activation panels in use stay resident
        u32x4 r = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[W - 4], v[W - 3]), pack2bf(v[W - 2], v[W - 1])};
        __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(p));
    }
}
template <int W>
static __device__ __forceinline__ void load_bf16(const bf16_t* p, float (&v)[W]) {
    if (W == 4) {
        const uint2 r = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    } else {
        const u32x4 r = *reinterpret_cast<const u32x4*>(p);
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
        v[W - 4] = __uint_as_float(r.z << 16); v[W - 3] = __uint_as_float(r.z & 0xffff0000u);
        v[W - 2] = __uint_as_float(r.w << 16); v[W - 1] = __uint_as_float(r.w & 0xffff0000u);
    }
}
template <int W>
static __device__ __forceinline__ void add_f32(const float* q, float (&v)[W]) {
#pragma unroll
    for (int i = 0; i < W; i += 4) {
        const float4 b4 = *reinterpret_cast<const float4*>(q + i);
        v[i] += b4.x; v[i + 1] += b4.y; v[i + 2] += b4.z; v[i + 3] += b4.w;
    }
}

// Fused epilogue on W (4 or 8) consecutive output columns n..n+W-1 of row m held by one lane (fp32 accumulators in v).
// side operand of an epilogue (residual for BIAS_RES, pre-activation for DGELU): W bf16 of row m starting at column n
template <int EPI>
static __device__ __forceinline__ const bf16_t* nt_side_ptr(const GemmNTParams& p, int m, int n) {
    return EPI == VITK_EPI_BIAS_RES ? p.res + (size_t)m * p.ldres + n : p.aux + (size_t)m * p.ldaux + n;      // DGELU, MUL: aux
}
static __device__ __forceinline__ void unpack_bf16x8(const u32x4 r, float (&v)[8]) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}

template <int EPI, int W>
static __device__ __forceinline__ void nt_epilogue(const GemmNTParams& p, float (&v)[W], int m, int n, int orow, float rs,
                                                   const float* posrow, const float* side = nullptr,
                                                   const float* bias_regs = nullptr) {
    if (EPI != VITK_EPI_PLAIN && EPI != VITK_EPI_DGELU && EPI != VITK_EPI_MUL) {
        if (bias_regs) {
#pragma unroll
            for (int j = 0; j < W; ++j) v[j] += bias_regs[j];
        } else {
            add_f32<W>(p.bias + n, v);
        }
    }
    if (EPI == VITK_EPI_PLAIN || EPI == VITK_EPI_BIAS) {
        store_bf16<W>((bf16_t*)p.C + (size_t)m * p.ldc + n, v);
    } else if (EPI == VITK_EPI_BIAS_F32) {
#pragma unroll
        for (int i = 0; i < W; i += 4)
            *reinterpret_cast<float4*>((float*)p.C + (size_t)m * p.ldc + n + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
    } else if (EPI == VITK_EPI_BIAS_GELU) {
        // round the pre-activation to bf16 once (the stored value), apply GELU to exactly that value
        unsigned hp[W / 2];
        float a[W];
#pragma unroll
        for (int j = 0; j < W; j += 2) {
            hp[j / 2] = pack2bf(v[j], v[j + 1]);
            gelu_erf2(__uint_as_float(hp[j / 2] << 16), __uint_as_float(hp[j / 2] & 0xffff0000u), a[j], a[j + 1]);
        }
        bf16_t* cp = (bf16_t*)p.C + (size_t)m * p.ldc + n;
        if (W == 4) {
            uint2 r; r.x = hp[0]; r.y = hp[1];
            *reinterpret_cast<uint2*>(cp) = r;
        } else {
            const u32x4 r = {hp[0], hp[1], hp[W / 2 - 2], hp[W / 2 - 1]};
            __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(cp));
        }
        store_bf16<W>(p.out2 + (size_t)m * p.ldout2 + n, a);
    } else if (EPI == VITK_EPI_BIAS_GELU_DG) {
        // what is kept for the backward pass is gelu'(h), rounded to bf16, in place of h; h itself is never stored, so both
        // functions are evaluated on the fp32 pre-activation
        unsigned gp[W / 2];
        float a[W];
#pragma unroll
        for (int j = 0; j < W; j += 2) {
            float g0, g1;
            gelu_dgelu_erf2(v[j], v[j + 1], a[j], a[j + 1], g0, g1);
            gp[j / 2] = pack2bf(g0, g1);
        }
        bf16_t* cp = (bf16_t*)p.C + (size_t)m * p.ldc + n;
        if (W == 4) {
            uint2 r; r.x = gp[0]; r.y = gp[1];
            *reinterpret_cast<uint2*>(cp) = r;
        } else {
            const u32x4 r = {gp[0], gp[1], gp[W / 2 - 2], gp[W / 2 - 1]};
            __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(cp));
        }
        store_bf16<W>(p.out2 + (size_t)m * p.ldout2 + n, a);
    } else if (EPI == VITK_EPI_MUL) {
        float h[W];
        if (side == nullptr) load_bf16<W>(p.aux + (size_t)m * p.ldaux + n, h);
#pragma unroll
        for (int j = 0; j < W; ++j) v[j] *= side ? side[j] : h[j];
        store_bf16<W>((bf16_t*)p.C + (size_t)m * p.ldc + n, v);
    } else if (EPI == VITK_EPI_BIAS_RES) {
        float r[W];
        if (side == nullptr) load_bf16<W>(p.res + (size_t)orow * p.ldres + n, r);
#pragma unroll
        for (int j = 0; j < W; ++j) v[j] = (side ? side[j] : r[j]) + rs * v[j];
        store_bf16<W>((bf16_t*)p.C + (size_t)orow * p.ldc + n, v);
    } else if (EPI == VITK_EPI_PATCH) {
        add_f32<W>(posrow + n, v);
        store_bf16<W>((bf16_t*)p.C + (size_t)orow * p.ldc + n, v);
    } else if (EPI == VITK_EPI_DGELU) {
        float h[W];
        if (side == nullptr) load_bf16<W>(p.aux + (size_t)m * p.ldaux + n, h);
#pragma unroll
        for (int j = 0; j < W; j += 2) {
            float g0, g1;
            dgelu_erf2(side ? side[j] : h[j], side ? side[j + 1] : h[j + 1], g0, g1);
            v[j] *= g0;
            v[j + 1] *= g1;
        }
        store_bf16<W>((bf16_t*)p.C + (size_t)m * p.ldc + n, v);
    }
}

#define LBM 256
#define LBN 256
#define LTILE (256 * GBK)            // elements per operand tile of the 256 x 256 kernels

// gemm_ntq.hip: the four-wave persistent kernel with the deferred epilogue.  vitk_gemm_ntq_takes: whether it takes this problem
// (host side, no launch); vitk_gemm_ntq_launch: 0 or VITK_ELAUNCH with the error text set.
bool vitk_gemm_ntq_takes(const GemmNTParams& p, int epilogue);
int vitk_gemm_ntq_launch(const GemmNTParams& p, int epilogue, int grid, hipStream_t stream);
