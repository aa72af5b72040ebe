// Fused multi-head self-attention (head dim 64, no mask, no dropout) forward and backward for gfx950.
//
// Token counts are small (197 / 198 / 577), so one workgroup owns a whole (batch, head): K and V of the head
// are staged once into LDS (<= 608 keys), and the softmax over all keys of a 16-query tile is done in
// registers in one pass (no online rescaling).  All matrix products use v_mfma_f32_16x16x32_bf16:
//   forward : S^T = K Q^T (key rows, query on the lane)  ->  softmax along registers + 2 shuffles
//             O^T = V^T P^T with the S^T accumulators reused in place as the B operand (bf16-packed),
//             V^T fragments by the transposing LDS read ds_read_b64_tr_b16.
//   backward: S = Q K^T and dP = dO V^T with the key on the lane; P and dS (bf16-packed accumulators) feed
//             dV^T += dO^T P and dK^T += Q^T dS directly; only dS crosses LDS once for dQ^T = K^T dS^T.
// qkv is nn.Linear(D,3D)'s output [B*N, 3*H*64]; column = which*H*64 + head*64 + d.
#include "common.h"

#define HD 64   // head dim

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

static __device__ __forceinline__ bf16x8 tr_frag2(const bf16_t* lo, const bf16_t* hi) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)lo);
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)hi);
    return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
static __device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const u4 r = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, r);
}
// [rows][64] bf16 image with the 16-byte chunk index XOR-swizzled by (row & 7): element offset of (row, col)
static __device__ __forceinline__ int swz(int row, int col) {
    return row * HD + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7));
}

// ------------------------------------------------------------------------------------------------ forward
// One workgroup per (batch, head).  K and V go global -> LDS by DMA (no staging registers), both as [key][64] with the
// 16-byte chunk XOR-swizzled by (key & 7): conflict-free for the ds_read_b128 fragments of K and for the transposing reads of
// V alike.  Rows past N repeat row N-1 (finite data): their scores are forced to -inf and 0 * finite = 0 in P V.
// MAXT = 2 * ceil(N / 32) key tiles of 16, all held in registers, is a template parameter, so the loops carry no run-time
// bounds: tiles below MAXT-2 are entirely real keys; the last two start their score accumulators at -inf where the key is
// padding (C = -inf + finite stays -inf), which is the whole masking cost.
// WAVES waves per workgroup and WPE waves per SIMD asked of the compiler: 7 and 4 up to 224 keys (13 query tiles at 197
// tokens = at most 2 per wave; two workgroups per CU), 4 and 1 beyond.
template <int MAXT, int WAVES, int WPE>
__global__ __launch_bounds__(WAVES * 64, WPE) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                   float* __restrict__ lse, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
    constexpr int NPAD = MAXT * 16;
    bf16_t* Ks = reinterpret_cast<bf16_t*>(attn_smem);
    bf16_t* Vs = Ks + NPAD * HD;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4, lq = lr >> 2, lp = lr & 3;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * HD, ld = 3 * D;
    const bf16_t* base = qkv + (size_t)b * N * ld + h * HD;

    {   // DMA fill: a piece = 8 rows x 128 B; pieces [0, NPAD/8) are K, the rest V
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(Ks));
        const int prow = lane >> 3, ppos = lane & 7;
        for (int pc = wave; pc < 2 * (NPAD / 8); pc += WAVES) {
            const int which = pc >= NPAD / 8 ? 1 : 0;
            const int row = (pc - which * (NPAD / 8)) * 8 + prow;
            const int srow = row < N ? row : N - 1;
            const bf16_t* src = base + (size_t)srow * ld + (1 + which) * D + ((ppos ^ (row & 7)) << 3);
            glds16_v(src, lds0 + (unsigned)pc * 1024u);
        }
    }

    const int nqt = (N + 15) / 16;
    const float c2 = scale * 1.44269504088896340736f;      // softmax in base 2
    f32x4 tail0[2];     // accumulator start values of the last two key tiles: 0 for a real key, -inf for padding
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) tail0[i][r] = ((MAXT - 2 + i) * 16 + 4 * lg + r < N) ? 0.f : -INFINITY;
    const int koff0 = lr * HD + ((lg ^ (lr & 7)) << 3), koff1 = lr * HD + (((4 + lg) ^ (lr & 7)) << 3);   // (16t + lr) & 7 == lr & 7
    int voff[4];        // (32u + 4lg + lq) & 7 == (4lg + lq) & 7, and the same 16 rows further: one offset per d tile serves all u
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) voff[dt] = swz(4 * lg + lq, dt * 16 + 4 * lp);

    auto load_q = [&](int qt, bf16x8 (&qf)[2]) {
        int q = qt * 16 + lr;
        q = q < N ? q : N - 1;
        const bf16_t* qp = base + (size_t)q * ld + 8 * lg;
        qf[0] = *reinterpret_cast<const bf16x8*>(qp);
        qf[1] = *reinterpret_cast<const bf16x8*>(qp + 32);
    };
    bf16x8 qf[2], qn[2];
    if (wave < nqt) load_q(wave, qf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int qt = wave; qt < nqt; qt += WAVES) {
        if (qt + WAVES < nqt) load_q(qt + WAVES, qn);      // the next tile's queries travel under this tile's work
        // K fragments are fetched one key tile ahead by hand and the scheduler is fenced per tile: left alone, hipcc hoists every
        // fragment load of the unrolled loop to the top and spills
        f32x4 s[MAXT];
        bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(&Ks[koff0]);
        bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(&Ks[koff1]);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const bf16x8 a0 = kf0, a1 = kf1;
            if (t + 1 < MAXT) {
                kf0 = *reinterpret_cast<const bf16x8*>(&Ks[(t + 1) * 16 * HD + koff0]);
                kf1 = *reinterpret_cast<const bf16x8*>(&Ks[(t + 1) * 16 * HD + koff1]);
            }
            s[t] = t < MAXT - 2 ? (f32x4){0.f, 0.f, 0.f, 0.f} : tail0[t - (MAXT - 2)];
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[0], s[t], 0, 0, 0);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[1], s[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // s[t][r] = S^T[key = 16t + 4lg + r][q = lr] (unscaled)
        float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
        for (int t = 1; t < MAXT; ++t) mx = fmaxf(fmaxf(mx, s[t][0]), fmaxf(fmaxf(s[t][1], s[t][2]), s[t][3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nmc = -mx * c2;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], c2, nmc));
                s[t][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < MAXT / 2; ++u) {
            const bf16x8 pb = pack8(s[2 * u], s[2 * u + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16_t* vp = Vs + u * 32 * HD + voff[dt];
                const bf16x8 va = tr_frag2(vp, vp + 16 * HD);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // o[dt][r] = O^T[d = 16dt + 4lg + r][q = lr]
        const int q = qt * 16 + lr;
        if (q < N) {
            bf16_t* op = out + ((size_t)b * N + q) * D + h * HD + 4 * lg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 w;
                w.x = pack2bf(o[dt][0] * inv, o[dt][1] * inv);
                w.y = pack2bf(o[dt][2] * inv, o[dt][3] * inv);
                *reinterpret_cast<uint2*>(op + dt * 16) = w;
            }
            if (lg == 0) lse[((size_t)b * H + h) * N + q] = (mx * c2 + __builtin_amdgcn_logf(sum)) * 0.69314718055994530942f;
        }
        qf[0] = qn[0];
        qf[1] = qn[1];
    }
}

// ----------------------------------------------------------------------------------------------- backward
#define KB_MAX 224    // keys per workgroup (14 tiles of 16)
#define KB_TILES 14
#define QSTEP 32

// one workgroup = (batch*head, key block).  dq_acc != nullptr: several key blocks add their dQ into fp32.
__global__ __launch_bounds__(256) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                        const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                        bf16_t* __restrict__ dqkv, float* __restrict__ dq_acc, int N, int H,
                                                        float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(attn_smem);            // [224][64] swizzled
    bf16_t* Vs = Ks + KB_MAX * HD;                                 // [224][64] swizzled
    bf16_t* Qs = Vs + KB_MAX * HD;                                 // [32][64]  swizzled
    bf16_t* dOs = Qs + QSTEP * HD;                                 // [32][64]  swizzled
    bf16_t* dSs = dOs + QSTEP * HD;                                // [224][32] (key rows, 32 queries)
    float* lse_s = reinterpret_cast<float*>(dSs + KB_MAX * QSTEP); // [32]
    float* dl_s = lse_s + QSTEP;                                   // [32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4, lq = lr >> 2, lp = lr & 3;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * HD, ld = 3 * D;
    const int key0 = blockIdx.y * KB_MAX;
    int nkeys = N - key0; nkeys = nkeys < KB_MAX ? nkeys : KB_MAX;
    const int nkt = (nkeys + 15) / 16;            // real key tiles in this block
    const int nk32 = (nkeys + 31) / 32;           // 32-key contraction steps for dQ
    const bf16_t* base = qkv + (size_t)b * N * ld + h * HD;

    for (int ci = tid; ci < KB_MAX * 8; ci += 256) {
        const int key = ci >> 3, ch = ci & 7;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < nkeys) {
            const bf16_t* rowp = base + (size_t)(key0 + key) * ld + ch * 8;
            kv = *reinterpret_cast<const uint4*>(rowp + D);
            vv = *reinterpret_cast<const uint4*>(rowp + 2 * D);
        }
        const int off = key * HD + ((ch ^ (key & 7)) << 3);
        *reinterpret_cast<uint4*>(&Ks[off]) = kv;
        *reinterpret_cast<uint4*>(&Vs[off]) = vv;
    }

    f32x4 dv[4][4], dk[4][4];   // [owned key tile][d tile]: D[d = 4lg+r][key = lr]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { dv[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int nsteps = (N + QSTEP - 1) / QSTEP;
    const int srow = tid >> 3, sch = tid & 7;     // staging: 32 rows x 8 chunks
    for (int st = 0; st < nsteps; ++st) {
        const int q0 = st * QSTEP;
        {   // stage Q, dO (swizzled) and delta = rowsum(dO * O), lse
            const int q = q0 + srow;
            uint4 qv = make_uint4(0, 0, 0, 0), dov = make_uint4(0, 0, 0, 0);
            float part = 0.f;
            if (q < N) {
                qv = *reinterpret_cast<const uint4*>(base + (size_t)q * ld + sch * 8);
                const size_t oo = ((size_t)b * N + q) * D + h * HD + sch * 8;
                dov = *reinterpret_cast<const uint4*>(dout + oo);
                float a[8], c[8];
                load8(dout + oo, a);
                load8(out + oo, c);
#pragma unroll
                for (int j = 0; j < 8; ++j) part += a[j] * c[j];
            }
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            const int off = srow * HD + ((sch ^ (srow & 7)) << 3);
            *reinterpret_cast<uint4*>(&Qs[off]) = qv;
            *reinterpret_cast<uint4*>(&dOs[off]) = dov;
            if (sch == 0) {
                dl_s[srow] = part;
                lse_s[srow] = (q < N) ? lse[((size_t)b * H + h) * N + q] : 0.f;
            }
        }
        __syncthreads();

        // per-step fragments shared by all of the wave's key tiles
        bf16x8 qf[2][2], dof[2][2], qtr[4], dotr[4];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row = qi * 16 + lr;
                const int off = row * HD + (((ks * 4 + lg) ^ (row & 7)) << 3);
                qf[qi][ks] = *reinterpret_cast<const bf16x8*>(&Qs[off]);
                dof[qi][ks] = *reinterpret_cast<const bf16x8*>(&dOs[off]);
            }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int rlo = 4 * lg + lq, rhi = 16 + 4 * lg + lq, col = dt * 16 + 4 * lp;
            qtr[dt] = tr_frag2(&Qs[swz(rlo, col)], &Qs[swz(rhi, col)]);
            dotr[dt] = tr_frag2(&dOs[swz(rlo, col)], &dOs[swz(rhi, col)]);
        }
        float lrow[2][4], drow[2][4];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lrow[qi][r] = lse_s[qi * 16 + 4 * lg + r];
                drow[qi][r] = dl_s[qi * 16 + 4 * lg + r];
            }

#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int kt = wave + 4 * ti;
            if (kt < nkt) {
                const int key = kt * 16 + lr;
                bf16x8 kf[2], vf[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int off = key * HD + (((ks * 4 + lg) ^ (key & 7)) << 3);
                    kf[ks] = *reinterpret_cast<const bf16x8*>(&Ks[off]);
                    vf[ks] = *reinterpret_cast<const bf16x8*>(&Vs[off]);
                }
                f32x4 pt[2], ds[2];
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f}, dpacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qi][ks], kf[ks], sacc, 0, 0, 0);
                        dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[qi][ks], vf[ks], dpacc, 0, 0, 0);
                    }
                    // sacc[r] = S[q = 16qi + 4lg + r][key = lr]
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __expf(sacc[r] * scale - lrow[qi][r]);
                        pt[qi][r] = p;
                        ds[qi][r] = p * (dpacc[r] - drow[qi][r]) * scale;
                    }
                }
                const bf16x8 pa = pack8(pt[0], pt[1]);     // k-slot (lg, j): q = 4lg+j (j<4) | 16+4lg+j-4
                const bf16x8 dsa = pack8(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[ti][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotr[dt], pa, dv[ti][dt], 0, 0, 0);
                    dk[ti][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr[dt], dsa, dk[ti][dt], 0, 0, 0);
                }
                // dS^T image [key][32 q]: lane writes q = 16qi + 4lg .. +3 for its key
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    uint2 w;
                    w.x = pack2bf(ds[qi][0], ds[qi][1]);
                    w.y = pack2bf(ds[qi][2], ds[qi][3]);
                    *reinterpret_cast<uint2*>(&dSs[key * QSTEP + qi * 16 + 4 * lg]) = w;
                }
            } else if (kt < 2 * nk32) {
                // padded key tile inside the last 32-key contraction step: its dS rows must read as zero
                const int key = kt * 16 + lr;
#pragma unroll
                for (int qi = 0; qi < 2; ++qi)
                    *reinterpret_cast<uint2*>(&dSs[key * QSTEP + qi * 16 + 4 * lg]) = make_uint2(0, 0);
            }
        }
        __syncthreads();

        // dQ^T[d][q] = sum_key K[key][d] * dS[q][key] : wave -> (qi, two d tiles)
        {
            const int qi = wave >> 1;
            f32x4 dq[2];
            dq[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dq[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int kk = 0; kk < nk32; ++kk) {
                const int rlo = kk * 32 + 8 * lg + lq, rhi = rlo + 4;
                const bf16x8 dst = tr_frag2(&dSs[rlo * QSTEP + qi * 16 + 4 * lp], &dSs[rhi * QSTEP + qi * 16 + 4 * lp]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int col = (2 * (wave & 1) + e) * 16 + 4 * lp;
                    const bf16x8 ktr = tr_frag2(&Ks[swz(rlo, col)], &Ks[swz(rhi, col)]);
                    dq[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr, dst, dq[e], 0, 0, 0);
                }
            }
            // dq[e][r] = dQ[q = q0 + 16qi + lr][d = 16*(2*(wave&1)+e) + 4lg + r]
            const int q = q0 + qi * 16 + lr;
            if (q < N) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int d = (2 * (wave & 1) + e) * 16 + 4 * lg;
                    if (dq_acc != nullptr) {
                        float* ap = dq_acc + ((size_t)b * N + q) * D + h * HD + d;
#pragma unroll
                        for (int r = 0; r < 4; ++r) atomicAdd(ap + r, dq[e][r]);
                    } else {
                        uint2 w;
                        w.x = pack2bf(dq[e][0], dq[e][1]);
                        w.y = pack2bf(dq[e][2], dq[e][3]);
                        *reinterpret_cast<uint2*>(dqkv + ((size_t)b * N + q) * ld + h * HD + d) = w;
                    }
                }
            }
        }
        // the next step's staging only touches Qs/dOs/lse_s/dl_s, which nobody reads after the barrier above;
        // its first barrier orders this dQ phase before the next dSs writes.
    }

    // dK, dV of the owned key tiles: lane holds [key = lr][d = 16dt + 4lg + r]
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const int kt = wave + 4 * ti;
        const int key = kt * 16 + lr;
        if (kt < nkt && key < nkeys) {
            bf16_t* rowp = dqkv + ((size_t)b * N + key0 + key) * ld + h * HD + 4 * lg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 wk, wv;
                wk.x = pack2bf(dk[ti][dt][0], dk[ti][dt][1]); wk.y = pack2bf(dk[ti][dt][2], dk[ti][dt][3]);
                wv.x = pack2bf(dv[ti][dt][0], dv[ti][dt][1]); wv.y = pack2bf(dv[ti][dt][2], dv[ti][dt][3]);
                *reinterpret_cast<uint2*>(rowp + D + dt * 16) = wk;
                *reinterpret_cast<uint2*>(rowp + 2 * D + dt * 16) = wv;
            }
        }
    }
}

// ------------------------------------------------------------------------------- backward, resident variant
// For heads of at most 224 tokens.  Two barrier-free phases; the scores are recomputed in the second instead of passing dS
// between waves:
//   phase 1, a wave owns key tiles:   S = Q K^T, dP = dO V^T (key on the lane), P = 2^(S c - lse'), dS = P (dP - delta) scale,
//                                     dV^T += dO^T P, dK^T += Q^T dS   (Q^T, dO^T by transposing reads)
//   phase 2, a wave owns query tiles: S^T = K Q^T, dP^T = V dO^T (query on the lane), same P and dS, dQ^T += K^T dS^T.
// What all waves share goes through LDS ([NPAD][64] images, chunk XOR-swizzled by row & 7, filled by DMA): Q and dO in
// phase 1, K and V in phase 2 - the SAME 56 KB, refilled between the phases - while what only one wave needs (its own K, V
// rows in phase 1, its own Q, dO rows in phase 2) comes straight from global memory into registers.  58 KB of LDS and 128
// registers let two workgroups (14 waves) share a CU, so one head's loads and barriers hide under the other's MFMAs.
// The owned key tiles of phase 1 are processed one after the other, each step in two stages that reuse the same registers
// (b128 fragments of Q, dO for S and dP; then their transposed fragments for dV, dK).
// Padding: query rows >= N carry lse' = +inf, so P = dS = 0 there; key rows >= N start the S^T accumulators of phase 2 at
// -inf (tiles below MAXT-2 are all real keys); in phase 1 padded keys only feed dK/dV rows that are never stored.
// MAXT = 2 * ceil(N / 32) key tiles is a template parameter: no run-time loop bounds.
template <int MAXT, int WAVES, int WPE>
__global__ __launch_bounds__(WAVES * 64, WPE) void attn_bwd_res_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                       const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                       bf16_t* __restrict__ dqkv, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
    constexpr int NPAD = MAXT * 16;
    constexpr int TPW = (MAXT + WAVES - 1) / WAVES;        // key tiles (phase 1) / query tiles (phase 2) per wave
    bf16_t* R0 = reinterpret_cast<bf16_t*>(attn_smem);     // Q, then K
    bf16_t* R1 = R0 + NPAD * HD;                            // dO, then V
    float* lse_s = reinterpret_cast<float*>(R1 + NPAD * HD);   // [NPAD] lse * log2(e), +inf past N
    float* dl_s = lse_s + NPAD;                                // [NPAD] delta

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4, lq = lr >> 2, lp = lr & 3;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * HD, ld = 3 * D;
    const bf16_t* base = qkv + (size_t)b * N * ld + h * HD;
    const bf16_t* dobase = dout + (size_t)b * N * D + h * HD;
    const bf16_t* obase = out + (size_t)b * N * D + h * HD;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(R0));

    // DMA fill of the two images, a piece = 8 rows x 128 B: first = false: (Q, dO), true: (K, V)
    auto fill = [&](bool second) {
        const int prow = lane >> 3, ppos = lane & 7;
        for (int pc = wave; pc < 2 * (NPAD / 8); pc += WAVES) {
            const int which = pc >= NPAD / 8 ? 1 : 0;
            const int row = (pc - which * (NPAD / 8)) * 8 + prow;
            const int srow = row < N ? row : N - 1;
            const int ch = (ppos ^ (row & 7)) << 3;
            const bf16_t* src = second ? base + (size_t)srow * ld + (1 + which) * D + ch
                                       : (which ? dobase + (size_t)srow * D + ch : base + (size_t)srow * ld + ch);
            glds16_v(src, lds0 + (unsigned)pc * 1024u);
        }
    };
    fill(false);
    // delta and lse: item = (query row, 16-byte chunk), 8 neighbouring lanes share a row; NPAD * 8 items are an exact multiple
    // of the block size.  All loads of a thread are issued before the first is used.
    {
        constexpr int ITEMS = NPAD * 8 / (WAVES * 64);
        static_assert(ITEMS * WAVES * 64 == NPAD * 8, "items must divide evenly");
        u32x4_t ra[ITEMS], rc[ITEMS];
        float lv[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + i * WAVES * 64;
            int q = it >> 3;
            q = q < N ? q : N - 1;
            ra[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(dobase + (size_t)q * D + (it & 7) * 8));
            rc[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(obase + (size_t)q * D + (it & 7) * 8));
            lv[i] = lse[((size_t)b * H + h) * N + q];
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + i * WAVES * 64;
            const int q = it >> 3;
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part += __uint_as_float(ra[i][j] << 16) * __uint_as_float(rc[i][j] << 16);
                part += __uint_as_float(ra[i][j] & 0xffff0000u) * __uint_as_float(rc[i][j] & 0xffff0000u);
            }
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            if ((it & 7) == 0) {
                dl_s[q] = part;
                lse_s[q] = q < N ? lv[i] * 1.44269504088896340736f : INFINITY;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float c2 = scale * 1.44269504088896340736f;
    const int foff0 = lr * HD + ((lg ^ (lr & 7)) << 3), foff1 = lr * HD + (((4 + lg) ^ (lr & 7)) << 3);   // b128 fragment of row 16t + lr
    int toff[4];        // transposing-read offset of rows 4lg + lq (+16), columns 16dt + 4lp; same for every 32-row group
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) toff[dt] = swz(4 * lg + lq, dt * 16 + 4 * lp);

    // ---------------------------------------------------------------- phase 1: dK, dV of the owned key tiles (R0 = Q, R1 = dO)
#pragma unroll 1
    for (int kt = wave; kt < MAXT; kt += WAVES) {
        bf16x8 kf[2], vf[2];
        {
            int key = kt * 16 + lr;
            key = key < N ? key : N - 1;
            const bf16_t* kp = base + (size_t)key * ld + D + 8 * lg;
            kf[0] = *reinterpret_cast<const bf16x8*>(kp);
            kf[1] = *reinterpret_cast<const bf16x8*>(kp + 32);
            vf[0] = *reinterpret_cast<const bf16x8*>(kp + D);
            vf[1] = *reinterpret_cast<const bf16x8*>(kp + D + 32);
        }
        f32x4 dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
        for (int st = 0; st < MAXT / 2; ++st) {
            const bf16_t* Qp = R0 + st * 32 * HD;
            const bf16_t* dOp = R1 + st * 32 * HD;
            bf16x8 pa, dsa;
            {   // stage 1: scores and their gradient for 32 queries x 16 keys
                f32x4 pt[2], ds[2];
                const f32x4 l0 = *reinterpret_cast<const f32x4*>(&lse_s[st * 32 + 4 * lg]);
                const f32x4 l1 = *reinterpret_cast<const f32x4*>(&lse_s[st * 32 + 16 + 4 * lg]);
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(&dl_s[st * 32 + 4 * lg]);
                const f32x4 d1 = *reinterpret_cast<const f32x4*>(&dl_s[st * 32 + 16 + 4 * lg]);
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    const bf16x8 q0 = *reinterpret_cast<const bf16x8*>(Qp + qi * 16 * HD + foff0);
                    const bf16x8 q1 = *reinterpret_cast<const bf16x8*>(Qp + qi * 16 * HD + foff1);
                    const bf16x8 o0 = *reinterpret_cast<const bf16x8*>(dOp + qi * 16 * HD + foff0);
                    const bf16x8 o1 = *reinterpret_cast<const bf16x8*>(dOp + qi * 16 * HD + foff1);
                    f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f}, dpacc = (f32x4){0.f, 0.f, 0.f, 0.f};
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, kf[0], sacc, 0, 0, 0);
                    dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o0, vf[0], dpacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1, kf[1], sacc, 0, 0, 0);
                    dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o1, vf[1], dpacc, 0, 0, 0);
                    // sacc[r] = S[q = 32st + 16qi + 4lg + r][key = lr]
                    const f32x4 lq4 = qi ? l1 : l0, dq4 = qi ? d1 : d0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pp = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, -lq4[r]));
                        pt[qi][r] = pp;
                        ds[qi][r] = pp * (dpacc[r] - dq4[r]) * scale;
                    }
                }
                pa = pack8(pt[0], pt[1]);     // contraction slot (lg, j): q = 4lg + j | 16 + 4lg + j - 4
                dsa = pack8(ds[0], ds[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // stage 2: dV^T += dO^T P, dK^T += Q^T dS
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 dotr = tr_frag2(dOp + toff[dt], dOp + 16 * HD + toff[dt]);
                const bf16x8 qtr = tr_frag2(Qp + toff[dt], Qp + 16 * HD + toff[dt]);
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotr, pa, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr, dsa, dk[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // lane holds dK^T / dV^T [d = 16dt + 4lg + r][key = lr]
        const int key = kt * 16 + lr;
        if (key < N) {
            bf16_t* rowp = dqkv + ((size_t)b * N + key) * ld + h * HD + 4 * lg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 wk, wv;
                wk.x = pack2bf(dk[dt][0], dk[dt][1]); wk.y = pack2bf(dk[dt][2], dk[dt][3]);
                wv.x = pack2bf(dv[dt][0], dv[dt][1]); wv.y = pack2bf(dv[dt][2], dv[dt][3]);
                *reinterpret_cast<uint2*>(rowp + D + dt * 16) = wk;
                *reinterpret_cast<uint2*>(rowp + 2 * D + dt * 16) = wv;
            }
        }
    }
    (void)TPW;

    // every wave is done with Q and dO: refill the same LDS with K and V
    __syncthreads();
    fill(true);
    // this wave's first query tile travels with the refill
    auto load_q = [&](int qt, bf16x8 (&qf)[2], bf16x8 (&dof)[2]) {
        int q = qt * 16 + lr;
        q = q < N ? q : N - 1;
        const bf16_t* qp = base + (size_t)q * ld + 8 * lg;
        const bf16_t* dp = dobase + (size_t)q * D + 8 * lg;
        qf[0] = *reinterpret_cast<const bf16x8*>(qp);
        qf[1] = *reinterpret_cast<const bf16x8*>(qp + 32);
        dof[0] = *reinterpret_cast<const bf16x8*>(dp);
        dof[1] = *reinterpret_cast<const bf16x8*>(dp + 32);
    };
    bf16x8 qf[2], dof[2];
    load_q(wave, qf, dof);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---------------------------------------------------------------- phase 2: dQ of the owned query tiles (R0 = K, R1 = V)
    f32x4 tail0[2];     // S^T accumulator start values of the last two key tiles: 0 for a real key, -inf for padding
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) tail0[i][r] = ((MAXT - 2 + i) * 16 + 4 * lg + r < N) ? 0.f : -INFINITY;
#pragma unroll 1
    for (int qt = wave; qt * 16 < N; qt += WAVES) {
        const int q = qt * 16 + lr;
        const float nl = -lse_s[q], dl = dl_s[q];
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(&R0[foff0]), kf1 = *reinterpret_cast<const bf16x8*>(&R0[foff1]);
        bf16x8 vf0 = *reinterpret_cast<const bf16x8*>(&R1[foff0]), vf1 = *reinterpret_cast<const bf16x8*>(&R1[foff1]);
#pragma unroll
        for (int u = 0; u < MAXT / 2; ++u) {
            f32x4 ds[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int t = 2 * u + e;
                const bf16x8 a0 = kf0, a1 = kf1, b0 = vf0, b1 = vf1;
                if (t + 1 < MAXT) {
                    kf0 = *reinterpret_cast<const bf16x8*>(&R0[(t + 1) * 16 * HD + foff0]);
                    kf1 = *reinterpret_cast<const bf16x8*>(&R0[(t + 1) * 16 * HD + foff1]);
                    vf0 = *reinterpret_cast<const bf16x8*>(&R1[(t + 1) * 16 * HD + foff0]);
                    vf1 = *reinterpret_cast<const bf16x8*>(&R1[(t + 1) * 16 * HD + foff1]);
                }
                f32x4 sacc = t < MAXT - 2 ? (f32x4){0.f, 0.f, 0.f, 0.f} : tail0[t - (MAXT - 2)];
                f32x4 dpacc = (f32x4){0.f, 0.f, 0.f, 0.f};
                sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[0], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, dof[0], dpacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[1], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, dof[1], dpacc, 0, 0, 0);
                // sacc[r] = S^T[key = 16t + 4lg + r][q = lr]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pp = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], c2, nl));
                    ds[e][r] = pp * (dpacc[r] - dl) * scale;
                }
            }
            const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16_t* kp = R0 + u * 32 * HD + toff[dt];
                const bf16x8 ktr = tr_frag2(kp, kp + 16 * HD);
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr, dsb, dq[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((qt + WAVES) * 16 < N) load_q(qt + WAVES, qf, dof);
        // dq[dt][r] = dQ^T[d = 16dt + 4lg + r][q = lr]
        if (q < N) {
            bf16_t* rowp = dqkv + ((size_t)b * N + q) * ld + h * HD + 4 * lg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 w;
                w.x = pack2bf(dq[dt][0], dq[dt][1]);
                w.y = pack2bf(dq[dt][2], dq[dt][3]);
                *reinterpret_cast<uint2*>(rowp + dt * 16) = w;
            }
        }
    }
}

// dqkv[:, q columns] = bf16(dq_acc)
__global__ void attn_dq_convert_kernel(const float* __restrict__ dq_acc, bf16_t* __restrict__ dqkv, size_t rows, int D) {
    const size_t total = rows * (size_t)(D / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (D / 4);
        const int c = (int)(i % (D / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(dq_acc + r * D + c);
        uint2 w;
        w.x = pack2bf(v.x, v.y);
        w.y = pack2bf(v.z, v.w);
        *reinterpret_cast<uint2*>(dqkv + r * 3 * D + c) = w;
    }
}

static const size_t kBwdSmem = (size_t)(2 * KB_MAX * HD + 2 * QSTEP * HD + KB_MAX * QSTEP) * sizeof(bf16_t) + 2 * QSTEP * sizeof(float);

extern "C" {

int vitk_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, void* stream) {
    VITK_REQUIRE(qkv && out && lse, "attention_fwd: null pointer");
    VITK_REQUIRE(B > 0 && N > 0 && H > 0 && N <= 608, "attention_fwd: need B,H>0 and 0<N<=608 (B=%d N=%d H=%d)", B, N, H);
    VITK_REQUIRE(vitk_aligned(qkv, 16) && vitk_aligned(out, 16), "attention_fwd: pointers must be 16-byte aligned");
    const int maxt = 2 * ((N + 31) / 32);
    const size_t smem = (size_t)2 * maxt * 16 * HD * sizeof(bf16_t);
    hipStream_t s = (hipStream_t)stream;
    static bool attr_done[40] = {false};
#define VITK_FWD_CASE(T, W, E)                                                                                              \
    case T: {                                                                                                               \
        if (!attr_done[T]) {                                                                                                \
            hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<T, W, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            if (e != hipSuccess) { vitk_set_error("attention_fwd: cannot set LDS size %zu: %s", smem, hipGetErrorString(e)); return VITK_ELAUNCH; } \
            attr_done[T] = true;                                                                                            \
        }                                                                                                                   \
        attn_fwd_kernel<T, W, E><<<B * H, W * 64, smem, s>>>((const bf16_t*)qkv, (bf16_t*)out, lse, N, H, scale);          \
    } break;
    switch (maxt) {
        VITK_FWD_CASE(2, 4, 4) VITK_FWD_CASE(4, 4, 4) VITK_FWD_CASE(6, 4, 4) VITK_FWD_CASE(8, 4, 4) VITK_FWD_CASE(10, 7, 4)
        VITK_FWD_CASE(12, 7, 4) VITK_FWD_CASE(14, 7, 4) VITK_FWD_CASE(16, 4, 2) VITK_FWD_CASE(18, 4, 2) VITK_FWD_CASE(20, 4, 2)
        VITK_FWD_CASE(22, 4, 2) VITK_FWD_CASE(24, 4, 2) VITK_FWD_CASE(26, 4, 1) VITK_FWD_CASE(28, 4, 1) VITK_FWD_CASE(30, 4, 1)
        VITK_FWD_CASE(32, 4, 1) VITK_FWD_CASE(34, 4, 1) VITK_FWD_CASE(36, 4, 1) VITK_FWD_CASE(38, 4, 1)
        default: vitk_set_error("attention_fwd: unsupported token count %d", N); return VITK_EINVAL;
    }
#undef VITK_FWD_CASE
    return vitk_check_launch("attention_fwd");
}

size_t vitk_attention_bwd_workspace(int B, int N, int H) {
    if (B <= 0 || N <= 0 || H <= 0 || N <= KB_MAX) return 0;
    return (size_t)B * N * H * HD * sizeof(float);
}

int vitk_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N,
                       int H, float scale, void* workspace, size_t workspace_bytes, void* stream) {
    VITK_REQUIRE(qkv && out && dout && lse && dqkv, "attention_bwd: null pointer");
    VITK_REQUIRE(B > 0 && N > 0 && H > 0, "attention_bwd: bad shape B=%d N=%d H=%d", B, N, H);
    VITK_REQUIRE(vitk_aligned(qkv, 16) && vitk_aligned(out, 16) && vitk_aligned(dout, 16) && vitk_aligned(dqkv, 16),
                 "attention_bwd: pointers must be 16-byte aligned");
    if (N <= KB_MAX) {
        const int maxt = 2 * ((N + 31) / 32);
        const size_t smem = (size_t)2 * maxt * 16 * HD * sizeof(bf16_t) + (size_t)2 * maxt * 16 * sizeof(float);
        hipStream_t s = (hipStream_t)stream;
        static bool attr_done[16] = {false};
#define VITK_BWD_CASE(T, W, E)                                                                                                 \
    case T: {                                                                                                               \
        if (!attr_done[T]) {                                                                                                \
            hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_res_kernel<T, W, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            if (e != hipSuccess) { vitk_set_error("attention_bwd: cannot set LDS size %zu: %s", smem, hipGetErrorString(e)); return VITK_ELAUNCH; } \
            attr_done[T] = true;                                                                                            \
        }                                                                                                                   \
        attn_bwd_res_kernel<T, W, E><<<B * H, W * 64, smem, s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, \
                                                              (bf16_t*)dqkv, N, H, scale);                                  \
    } break;
        switch (maxt) {
            VITK_BWD_CASE(2, 2, 4) VITK_BWD_CASE(4, 4, 4) VITK_BWD_CASE(6, 6, 4) VITK_BWD_CASE(8, 4, 4) VITK_BWD_CASE(10, 5, 4)
            VITK_BWD_CASE(12, 6, 4) VITK_BWD_CASE(14, 7, 4)
            default: vitk_set_error("attention_bwd: unsupported token count %d", N); return VITK_EINVAL;
        }
#undef VITK_BWD_CASE
        return vitk_check_launch("attention_bwd");
    }
    const int nkb = (N + KB_MAX - 1) / KB_MAX;
    const size_t need = vitk_attention_bwd_workspace(B, N, H);
    float* dq_acc = nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (nkb > 1) {
        if (!workspace || workspace_bytes < need) {
            vitk_set_error("attention_bwd: workspace %zu < %zu", workspace_bytes, need);
            return VITK_ENOSPC;
        }
        VITK_REQUIRE(vitk_aligned(workspace, 16), "attention_bwd: workspace must be 16-byte aligned");
        dq_acc = (float*)workspace;
        hipError_t me = hipMemsetAsync(dq_acc, 0, need, s);
        if (me != hipSuccess) { vitk_set_error("attention_bwd: memset failed: %s", hipGetErrorString(me)); return VITK_ELAUNCH; }
    }
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdSmem);
    if (e != hipSuccess) { vitk_set_error("attention_bwd: cannot set LDS size %zu: %s", kBwdSmem, hipGetErrorString(e)); return VITK_ELAUNCH; }
    attn_bwd_kernel<<<dim3(B * H, nkb), 256, kBwdSmem, s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse,
                                                           (bf16_t*)dqkv, dq_acc, N, H, scale);
    int rc = vitk_check_launch("attention_bwd");
    if (rc) return rc;
    if (nkb > 1) {
        const size_t rows = (size_t)B * N;
        attn_dq_convert_kernel<<<2048, 256, 0, s>>>(dq_acc, (bf16_t*)dqkv, rows, H * HD);
        rc = vitk_check_launch("attention_bwd_convert");
    }
    return rc;
}

}  // extern "C"
