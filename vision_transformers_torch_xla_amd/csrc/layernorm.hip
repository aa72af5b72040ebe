// LayerNorm forward/backward for bf16 rows, one 64-lane wave per row, 16-byte loads, shuffle reductions.
// HBM-bound: forward moves 4 B/element (read x, write y), backward 6 B/element (read dy, x; write dx)
// plus 2 B/element when the residual gradient is fused in.
#include "common.h"
#include <mutex>

bool vitk_prof_begin(void* stream, double flop, int kind);
void vitk_prof_end(void* stream);

#define LN_ROWS_PER_BLOCK 4   // 4 waves of 64 lanes
#define LN_MAX_BLOCKS 2048

typedef __attribute__((ext_vector_type(4))) unsigned ln_u4;
static __device__ __forceinline__ void ln_unpack(const ln_u4 r, float (&v)[8]) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}

// The row loop is software-pipelined: the loads of the wave's next row are issued before the current row is reduced, so a
// wave always has a row in flight (one row per wave at a time leaves too few bytes in flight to cover the HBM latency).
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                      float eps, const int* __restrict__ smap, int rps) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row indices and map lookups stay scalar
    const int nvec = D >> 3;
    const float invD = 1.0f / (float)D;
    const int stride = gridDim.x * LN_ROWS_PER_BLOCK;
    int row = blockIdx.x * LN_ROWS_PER_BLOCK + wave;
    if (row >= M) return;
    // with a sample map, output row r (compact) normalises input row smap[r / rps] * rps + r % rps
    auto src_row = [&](int r) -> size_t {
        if (smap == nullptr) return (size_t)r;
        const int c = r / rps;
        return (size_t)smap[c] * rps + (r - c * rps);
    };
    int cc[NV];
    bool ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        ok[i] = c < nvec;
        cc[i] = ok[i] ? c : 0;            // lanes past the row re-read chunk 0 and ignore it
    }
    // gamma and beta of the lane's columns stay in registers for all rows of the wave
    float g[NV][8], bt[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + 8 * cc[i]);
        const float4 g1 = *reinterpret_cast<const float4*>(gamma + 8 * cc[i] + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + 8 * cc[i]);
        const float4 b1 = *reinterpret_cast<const float4*>(beta + 8 * cc[i] + 4);
        g[i][0] = g0.x; g[i][1] = g0.y; g[i][2] = g0.z; g[i][3] = g0.w; g[i][4] = g1.x; g[i][5] = g1.y; g[i][6] = g1.z; g[i][7] = g1.w;
        bt[i][0] = b0.x; bt[i][1] = b0.y; bt[i][2] = b0.z; bt[i][3] = b0.w; bt[i][4] = b1.x; bt[i][5] = b1.y; bt[i][6] = b1.z; bt[i][7] = b1.w;
    }
    ln_u4 cur[NV], nxt[NV];
    {
        const size_t sr = src_row(row);
#pragma unroll
        for (int i = 0; i < NV; ++i) cur[i] = *reinterpret_cast<const ln_u4*>(x + sr * D + 8 * cc[i]);
    }
    for (; row < M; row += stride) {
        const size_t nsr = src_row(row + stride < M ? row + stride : row);
#pragma unroll
        for (int i = 0; i < NV; ++i) nxt[i] = *reinterpret_cast<const ln_u4*>(x + nsr * D + 8 * cc[i]);
        float v[NV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            ln_unpack(cur[i], v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = ok[i] ? v[i][j] : 0.f; s += v[i][j]; }
        }
        const float mu = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (ok[i]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mu; q += d * d; }
            }
        }
        const float var = wave_sum(q) * invD;
        const float rs = 1.0f / sqrtf(var + eps);
        bf16_t* yr = y + (size_t)row * D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (ok[i]) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mu) * rs * g[i][j] + bt[i][j];
                store8(yr + 8 * c, o);
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
        for (int i = 0; i < NV; ++i) cur[i] = nxt[i];
    }
}

// Backward.  Each wave keeps per-lane column partials of dgamma/dbeta over all the rows it visits; the block
// folds its 4 waves through LDS and writes one partial row; ln_bwd_reduce_kernel sums the partial rows in a
// fixed order (deterministic).  Every global load of a row (x, dy, the residual gradient) is issued before the
// first is used.  With rowscale the kernel also writes dxs = rowscale[row / rows_per_group] * dx, the gradient
// that enters the next residual branch under stochastic depth (one extra store instead of a separate pass).

template <int NV, bool RES, bool SCALED>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const bf16_t* __restrict__ dres,
                                                      bf16_t* __restrict__ dx, const float* __restrict__ rowscale,
                                                      int rows_per_group, bf16_t* __restrict__ dxs,
                                                      float* __restrict__ partial /*[grid][2][D]*/, int M, int D,
                                                      const int* __restrict__ in_inv, const int* __restrict__ out_inv) {
    extern __shared__ __attribute__((aligned(16))) float ln_smem[];   // [4][2][D]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row indices and map lookups stay scalar
    const int nvec = D >> 3;
    const float invD = 1.0f / (float)D;
    float g[NV][8], dg[NV][8], db[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; g[i][j] = (c < nvec) ? gamma[8 * c + j] : 0.f; }
    }
    const int stride = gridDim.x * LN_ROWS_PER_BLOCK;
    int row = blockIdx.x * LN_ROWS_PER_BLOCK + wave;
    int cc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        cc[i] = c < nvec ? c : 0;        // lanes past the row re-read chunk 0 and ignore it
    }
    // Sample maps (stochastic depth, rows_per_group rows per sample): dy, mean and rstd may be COMPACT over the samples that
    // took the branch - in_inv[sample] is the sample's compact index, or -1 when it skipped the branch (then dx = dres) - and
    // the scaled copy may be compact over the samples of the next branch (out_inv likewise).  x, dres and dx are full.
    auto compact_row = [&](int r) -> int {          // row of dy / mean / rstd for full row r, -1: the sample skipped the branch
        if (in_inv == nullptr) return r;
        const int smp = r / rows_per_group;
        const int c = in_inv[smp];
        return c < 0 ? -1 : c * rows_per_group + (r - smp * rows_per_group);
    };
    // Per-row scalars (statistics, branch scale, row of the scaled copy) are fetched one row ahead and the compact row
    // index two rows ahead, so that no map or statistics load sits between a row's vector loads and its arithmetic.
    struct RowMeta { float mu, rs, sc; long long orow; };
    auto row_meta = [&](int r, int cr) -> RowMeta {
        RowMeta m;
        const int srow = cr < 0 ? 0 : cr;            // row of the statistics
        m.mu = mean[srow]; m.rs = rstd[srow];
        m.sc = 1.0f;
        m.orow = r;                                  // row of the scaled copy, -1: the sample skips the next branch
        if (SCALED) {
            const int smp = r / rows_per_group;
            m.sc = rowscale[smp];
            if (out_inv != nullptr) {
                const int c = out_inv[smp];
                m.orow = c < 0 ? -1 : (long long)c * rows_per_group + (r - smp * rows_per_group);
            }
        }
        return m;
    };
    auto clamp_row = [&](int r) -> int { return r < M ? r : (M - 1); };
    ln_u4 rx[NV], rd[NV], rr[NV], nx[NV], nd[NV], nr[NV];
    int crow = -1, ncrow = -1, nncrow = -1;
    RowMeta meta = {0.f, 0.f, 1.f, -1}, nmeta = meta;
    if (row < M) {
        crow = compact_row(row);
        ncrow = compact_row(clamp_row(row + stride));
        meta = row_meta(row, crow);
        const int cr = crow < 0 ? 0 : crow;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const size_t o = (size_t)row * D + 8 * cc[i];
            rx[i] = __builtin_nontemporal_load(reinterpret_cast<const ln_u4*>(x + o));
            rd[i] = __builtin_nontemporal_load(reinterpret_cast<const ln_u4*>(dy + (size_t)cr * D + 8 * cc[i]));
            if (RES) rr[i] = __builtin_nontemporal_load(reinterpret_cast<const ln_u4*>(dres + o));
        }
    }
    for (; row < M; row += stride) {
        const size_t off = (size_t)row * D;
        {   // the next row's loads travel under this row's arithmetic (the last trip re-reads its own row)
            const int nrow = row + stride < M ? row + stride : row;
            if (row + stride >= M) ncrow = crow;
            const int cr = ncrow < 0 ? 0 : ncrow;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const size_t o = (size_t)nrow * D + 8 * cc[i];
                nx[i] = __builtin_nontemporal_load(reinterpret_cast<const ln_u4*>(x + o));
                nd[i] = __builtin_nontemporal_load(reinterpret_cast<const ln_u4*>(dy + (size_t)cr * D + 8 * cc[i]));
                if (RES) nr[i] = __builtin_nontemporal_load(reinterpret_cast<const ln_u4*>(dres + o));
            }
            nmeta = row_meta(nrow, ncrow);
            nncrow = compact_row(clamp_row(row + 2 * stride));
        }
        const bool took = crow >= 0;                 // wave-uniform
        const float mu = meta.mu, rs = meta.rs, sc = meta.sc;
        const long long orow = meta.orow;
        float xh[NV][8], gd[NV][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            float xv[8], dv[8];
            ln_unpack(rx[i], xv);
            ln_unpack(rd[i], dv);
            const bool ok = c < nvec && took;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float h = ok ? (xv[j] - mu) * rs : 0.f;
                const float d = ok ? dv[j] : 0.f;
                xh[i][j] = h;
                dg[i][j] += d * h;
                db[i][j] += d;
                const float t = d * g[i][j];
                gd[i][j] = t;
                s1 += t;
                s2 += t * h;
            }
        }
        const float c1 = wave_sum(s1) * invD;
        const float c2 = wave_sum(s2) * invD;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rs * (gd[i][j] - c1 - xh[i][j] * c2);
                if (RES) {
                    float r[8];
                    ln_unpack(rr[i], r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += r[j];
                }
                store8(dx + off + 8 * c, o);
                if (SCALED && orow >= 0) {
                    // scale the bf16 value that was just stored, as a separate pass over dx would
                    const uint4 w = pack8_bf16(o);
                    float q[8];
                    ln_unpack((ln_u4){w.x, w.y, w.z, w.w}, q);
#pragma unroll
                    for (int j = 0; j < 8; ++j) q[j] *= sc;
                    store8(dxs + (size_t)orow * D + 8 * c, q);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) { rx[i] = nx[i]; rd[i] = nd[i]; if (RES) rr[i] = nr[i]; }
        crow = ncrow; ncrow = nncrow; meta = nmeta;
    }
    // fold the 4 waves
    float* sg = ln_smem + (size_t)wave * 2 * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { sg[8 * c + j] = dg[i][j]; sg[D + 8 * c + j] = db[i][j]; }
        }
    }
    __syncthreads();
    float* prow = partial + (size_t)blockIdx.x * 2 * D;
    for (int c = threadIdx.x; c < 2 * D; c += blockDim.x) {
        prow[c] = ln_smem[c] + ln_smem[2 * D + c] + ln_smem[4 * D + c] + ln_smem[6 * D + c];
    }
}

__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int nrows, int D,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    // block = 16 columns x 64 row groups; fixed summation order => bitwise reproducible
    __shared__ float red[64][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float s = 0.f;
    if (c < 2 * D)
        for (int r = ry; r < nrows; r += 64) s += partial[(size_t)r * 2 * D + c];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < 2 * D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) t += red[k][cx];
        float* dst = (c < D) ? (dgamma + c) : (dbeta + (c - D));
        *dst = accumulate ? (*dst + t) : t;
    }
}

// Scratch of vitk_layernorm_bwd_noalloc: the dgamma / dbeta partial rows [blocks <= 1024][2][D <= 1024] that the other
// backward entry points take from the caller's workspace.  One copy per GPU the library runs on (8 MB each, held until the
// process ends).  In host code the array's NAME is only a stand-in: its device address comes from hipGetSymbolAddress, once
// per device, and that pointer is what kernels receive.
#define LN_SCRATCH_FLOATS (1024 * 2 * 1024)
static __device__ __align__(256) float ln_scratch[LN_SCRATCH_FLOATS];

static std::mutex g_ln_scratch_mu;
static float* g_ln_scratch_ptr[64] = {nullptr};

// device address of ln_scratch on the current device (looked up on the first call per device); null + error text on failure
static float* ln_scratch_ptr() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        vitk_set_error("layernorm_bwd_noalloc: cannot query the current device");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_ln_scratch_mu);
    if (g_ln_scratch_ptr[dev] == nullptr) {
        float* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(ln_scratch));
        if (e != hipSuccess || ptr == nullptr) {
            vitk_set_error("layernorm_bwd_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return nullptr;
        }
        g_ln_scratch_ptr[dev] = ptr;
    }
    return g_ln_scratch_ptr[dev];
}

static int ln_grid(int M) {
    int g = (M + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK;
    return g < LN_MAX_BLOCKS ? g : LN_MAX_BLOCKS;
}
static int ln_bwd_grid(int M) {
    int g = (M + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK;
    return g < 1024 ? g : 1024;
}

extern "C" {

static int ln_fwd_impl(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int D, float eps,
                       const int* sample_map, int rows_per_sample, void* stream) {
    VITK_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
    VITK_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && D <= 1024, "layernorm_fwd: need M>0, D%%8==0, D<=1024 (M=%d D=%d)", M, D);
    VITK_REQUIRE(vitk_aligned(x, 16) && vitk_aligned(y, 16) && vitk_aligned(gamma, 16) && vitk_aligned(beta, 16),
                 "layernorm_fwd: pointers must be 16-byte aligned");
    VITK_REQUIRE(!sample_map || (rows_per_sample > 0 && M % rows_per_sample == 0),
                 "layernorm_fwd: a sample map needs rows_per_sample > 0 dividing M (M=%d rows_per_sample=%d)", M, rows_per_sample);
    hipStream_t s = (hipStream_t)stream;
    const int grid = ln_grid(M);
    const bool pr = vitk_prof_begin(stream, 4.0 * (double)M * (double)D, VITK_PROF_LN_FWD);
    if (D <= 512)
        ln_fwd_kernel<1><<<grid, 256, 0, s>>>((const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D, eps, sample_map, rows_per_sample);
    else
        ln_fwd_kernel<2><<<grid, 256, 0, s>>>((const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D, eps, sample_map, rows_per_sample);
    if (pr) vitk_prof_end(stream);
    return vitk_check_launch("layernorm_fwd");
}

int vitk_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M,
                       int D, float eps, void* stream) {
    return ln_fwd_impl(x, gamma, beta, y, mean, rstd, M, D, eps, nullptr, 0, stream);
}

int vitk_layernorm_fwd_gather(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M,
                              int D, float eps, const int* sample_map, int rows_per_sample, void* stream) {
    VITK_REQUIRE(sample_map, "layernorm_fwd_gather: sample_map is required");
    return ln_fwd_impl(x, gamma, beta, y, mean, rstd, M, D, eps, sample_map, rows_per_sample, stream);
}

size_t vitk_layernorm_bwd_workspace(int M, int D) {
    if (M <= 0 || D <= 0) return 0;
    return (size_t)ln_bwd_grid(M) * 2 * (size_t)D * sizeof(float);
}

static int ln_bwd_impl(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const void* dres,
                       void* dx, const float* rowscale, int rows_per_group, void* dxs, float* dgamma, float* dbeta, int accumulate,
                       int M, int D, void* workspace, size_t workspace_bytes, void* stream, const int* in_inv = nullptr,
                       const int* out_inv = nullptr) {
    VITK_REQUIRE(dy && x && mean && rstd && gamma && dx && workspace, "layernorm_bwd: null pointer");
    VITK_REQUIRE((dgamma != nullptr) == (dbeta != nullptr), "layernorm_bwd: dgamma and dbeta must both be given or both be null");
    VITK_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && D <= 1024, "layernorm_bwd: need M>0, D%%8==0, D<=1024 (M=%d D=%d)", M, D);
    VITK_REQUIRE(vitk_aligned(dy, 16) && vitk_aligned(x, 16) && vitk_aligned(dx, 16) && (!dres || vitk_aligned(dres, 16)),
                 "layernorm_bwd: pointers must be 16-byte aligned");
    VITK_REQUIRE(!rowscale || (dxs && rows_per_group > 0 && vitk_aligned(dxs, 16)),
                 "layernorm_bwd: rowscale needs dx_scaled (16-byte aligned) and rows_per_group > 0");
    if (workspace_bytes < vitk_layernorm_bwd_workspace(M, D)) {
        vitk_set_error("layernorm_bwd: workspace %zu < %zu", workspace_bytes, vitk_layernorm_bwd_workspace(M, D));
        return VITK_ENOSPC;
    }
    hipStream_t s = (hipStream_t)stream;
    const int grid = ln_bwd_grid(M);
    const size_t smem = (size_t)LN_ROWS_PER_BLOCK * 2 * D * sizeof(float);
    float* partial = (float*)workspace;
#define VITK_LN_BWD(NV, R, S)                                                                                                  \
    ln_bwd_kernel<NV, R, S><<<grid, 256, smem, s>>>((const bf16_t*)dy, (const bf16_t*)x, mean, rstd, gamma, (const bf16_t*)dres, \
                                                    (bf16_t*)dx, rowscale, rows_per_group, (bf16_t*)dxs, partial, M, D, in_inv, out_inv)
    const int sel = (D <= 512 ? 0 : 4) + (dres ? 2 : 0) + (rowscale ? 1 : 0);
    const bool pr = vitk_prof_begin(stream, (6.0 + (dres ? 2.0 : 0.0) + (rowscale ? 2.0 : 0.0)) * (double)M * (double)D, VITK_PROF_LN_BWD);
    switch (sel) {
        case 0: VITK_LN_BWD(1, false, false); break;
        case 1: VITK_LN_BWD(1, false, true); break;
        case 2: VITK_LN_BWD(1, true, false); break;
        case 3: VITK_LN_BWD(1, true, true); break;
        case 4: VITK_LN_BWD(2, false, false); break;
        case 5: VITK_LN_BWD(2, false, true); break;
        case 6: VITK_LN_BWD(2, true, false); break;
        default: VITK_LN_BWD(2, true, true); break;
    }
#undef VITK_LN_BWD
    if (pr) vitk_prof_end(stream);
    int rc = vitk_check_launch("layernorm_bwd");
    if (rc) return rc;
    if (dgamma == nullptr) return 0;      // the partial rows stay in the workspace for vitk_layernorm_bwd_reduce
    ln_bwd_reduce_kernel<<<(2 * D + 15) / 16, 1024, 0, s>>>(partial, grid, D, dgamma, dbeta, accumulate);
    return vitk_check_launch("layernorm_bwd_reduce");
}

int vitk_layernorm_bwd_reduce(const void* workspace, size_t workspace_bytes, int M, int D, float* dgamma, float* dbeta,
                              int accumulate, void* stream) {
    VITK_REQUIRE(workspace && dgamma && dbeta, "layernorm_bwd_reduce: null pointer");
    VITK_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && D <= 1024, "layernorm_bwd_reduce: need M>0, D%%8==0, D<=1024 (M=%d D=%d)", M, D);
    if (workspace_bytes < vitk_layernorm_bwd_workspace(M, D)) {
        vitk_set_error("layernorm_bwd_reduce: workspace %zu < %zu", workspace_bytes, vitk_layernorm_bwd_workspace(M, D));
        return VITK_ENOSPC;
    }
    ln_bwd_reduce_kernel<<<(2 * D + 15) / 16, 1024, 0, (hipStream_t)stream>>>((const float*)workspace, ln_bwd_grid(M), D, dgamma, dbeta,
                                                                               accumulate);
    return vitk_check_launch("layernorm_bwd_reduce");
}

int vitk_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                       const void* dres, void* dx, float* dgamma, float* dbeta, int accumulate, int M, int D,
                       void* workspace, size_t workspace_bytes, void* stream) {
    return ln_bwd_impl(dy, x, mean, rstd, gamma, dres, dx, nullptr, 0, nullptr, dgamma, dbeta, accumulate, M, D, workspace,
                       workspace_bytes, stream);
}

int vitk_layernorm_bwd_noalloc(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                               const void* dres, void* dx, float* dgamma, float* dbeta, int accumulate, int M, int D,
                               void* stream) {
    // sizes first: nothing is looked up or launched for a shape the scratch array cannot hold
    VITK_REQUIRE(M > 0 && D > 0 && D % 8 == 0 && D <= 1024,
                 "layernorm_bwd_noalloc: need M>0, D%%8==0 and D <= 1024, the row width the library's own scratch is sized for (M=%d D=%d)", M, D);
    const size_t need = vitk_layernorm_bwd_workspace(M, D);
    if (need > sizeof(float) * (size_t)LN_SCRATCH_FLOATS) {
        vitk_set_error("layernorm_bwd_noalloc: %zu bytes of partial sums exceed the %zu-byte scratch", need, sizeof(float) * (size_t)LN_SCRATCH_FLOATS);
        return VITK_ENOSPC;
    }
    float* scratch = ln_scratch_ptr();
    if (scratch == nullptr) return VITK_ELAUNCH;
    // every partial row the reduction reads is written by the first kernel of this call: nothing has to start at zero
    return ln_bwd_impl(dy, x, mean, rstd, gamma, dres, dx, nullptr, 0, nullptr, dgamma, dbeta, accumulate, M, D, scratch,
                       sizeof(float) * (size_t)LN_SCRATCH_FLOATS, stream);
}

int vitk_layernorm_bwd_scaled(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                              const void* dres, void* dx, const float* rowscale, int rows_per_group, void* dx_scaled,
                              float* dgamma, float* dbeta, int accumulate, int M, int D, void* workspace,
                              size_t workspace_bytes, void* stream) {
    VITK_REQUIRE(rowscale && dx_scaled, "layernorm_bwd_scaled: rowscale and dx_scaled are required");
    return ln_bwd_impl(dy, x, mean, rstd, gamma, dres, dx, rowscale, rows_per_group, dx_scaled, dgamma, dbeta, accumulate, M, D,
                       workspace, workspace_bytes, stream);
}

int vitk_layernorm_bwd_gather(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                              const void* dres, void* dx, const float* rowscale, int rows_per_sample, void* dx_scaled,
                              const int* in_inv, const int* out_inv, float* dgamma, float* dbeta, int accumulate, int M, int D,
                              void* workspace, size_t workspace_bytes, void* stream) {
    VITK_REQUIRE(rows_per_sample > 0 && M % rows_per_sample == 0, "layernorm_bwd_gather: rows_per_sample must divide M (M=%d, %d)", M,
                 rows_per_sample);
    VITK_REQUIRE(!rowscale == !dx_scaled, "layernorm_bwd_gather: rowscale and dx_scaled come together");
    VITK_REQUIRE(!out_inv || rowscale, "layernorm_bwd_gather: out_inv needs rowscale / dx_scaled");
    return ln_bwd_impl(dy, x, mean, rstd, gamma, dres, dx, rowscale, rows_per_sample, dx_scaled, dgamma, dbeta, accumulate, M, D,
                       workspace, workspace_bytes, stream, in_inv, out_inv);
}

}  // extern "C"
