// Fused multi-head self-attention (head dims 64 and 96 on MFMA, others up to 96 on vector arithmetic; no mask, no dropout) forward
// and backward for gfx950.
//
// Token counts are small (197 / 198 / 577), so one workgroup owns a whole (batch, head): K and V of the head
// are staged once into LDS (<= 608 keys), and the softmax over all keys of a 16-query tile is done in
// registers in one pass (no online rescaling).  All matrix products use v_mfma_f32_16x16x32_bf16:
//   forward : S^T = K Q^T (key rows, query on the lane)  ->  softmax along registers + 2 shuffles
//             O^T = V^T P^T with the S^T accumulators reused in place as the B operand (bf16-packed),
//             V^T fragments by the transposing LDS read ds_read_b64_tr_b16.
//   backward: two phases over one LDS image (Q, dO first; then K, V): key-owner waves form dK, dV, query-owner waves
//             recompute the scores and form dQ; nothing crosses between waves (see attn_bwd_res_kernel).
// qkv is nn.Linear(D,3D)'s output [B*N, 3*H*hd]; column = which*H*hd + head*hd + d (hd = 64, or 96: AttnGeo below).
#include "common.h"

bool vitk_prof_begin(void* stream, double flop, int kind);
void vitk_prof_end(void* stream);

#define HD 64   // head dim
#ifndef VITK_ATTN_OWN
#define VITK_ATTN_OWN 1
#endif
#ifndef VITK_ATTN_ODD
#define VITK_ATTN_ODD 1
#endif

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
// Geometry of the MFMA kernels for a head of HDP columns in memory (a multiple of 32): 64, or 96 for the heads between 64 and 96
// wide, which the model pads with zeros (my_vit_xs: 72).  LDS rows are HDW elements: 64 (128 B; chunk index XOR row & 7, rows
// alternate between the two halves of the banks) or 128 (256 B: every row starts at bank 0, so the chunk index is XOR-ed with
// 2 * (row & 7), which keeps the 16-lane groups of ds_read_b128 and the 8 rows x 32 B of a transposing read on distinct banks).
// A 1 KB DMA piece covers RPP rows.
template <int HDP>
struct AttnGeo {
    static constexpr int HDW = HDP == 64 ? 64 : 128;
    static constexpr int NKS = HDP / 32, NDT = HDP / 16, NCH = HDP / 8, CPR = HDW / 8, RPP = 512 / HDW;
    static __device__ __forceinline__ int sig(int row) { return HDW == 64 ? (row & 7) : ((row & 7) << 1); }
    static __device__ __forceinline__ int swz(int row, int col) { return row * HDW + ((((col >> 3) ^ sig(row)) << 3) | (col & 7)); }
};

// Output rows.  After the transposed products a lane holds, for row lr of a 16-row tile, four 8-byte pieces: 4 consecutive bf16
// at column 16*dt + 4*lg.  Stores cost per instruction, not per byte, so pieces are paired first: two exchanges between
// neighbouring 16-lane rows (v_permlane16_swap: rows 1 and 3 of the first operand <-> rows 0 and 2 of the second) turn the pieces
// of column tiles dtA and dtB into ONE 16-byte piece per lane - lanes with even lg get columns 16*dtA + 8*(lg/2) .. +7, lanes with
// odd lg columns 16*dtB + 8*(lg/2) .. +7.  Every lane of the wave must take part in the exchange (call it outside row guards).
static __device__ __forceinline__ u32x4_t pair_pieces(uint2 pA, uint2 pB) {
    const auto rx = __builtin_amdgcn_permlane16_swap(pA.x, pB.x, false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(pA.y, pB.y, false, false);
    return (u32x4_t){rx[0], ry[0], rx[1], ry[1]};
}
static __device__ __forceinline__ int pair_col(int dtA, int dtB, int lg) { return 16 * ((lg & 1) ? dtB : dtA) + 8 * (lg >> 1); }

// ------------------------------------------------------------------------------------------------ forward
// One workgroup per (batch, head).  K and V go global -> LDS by DMA (no staging registers), both as [key][64] with the
// 16-byte chunk XOR-swizzled by (key & 7): conflict-free for the ds_read_b128 fragments of K and for the transposing reads of
// V alike.  Rows past N repeat row N-1 (finite data): their scores are forced to -inf and 0 * finite = 0 in P V.
// MAXT = 2 * ceil(N / 32) key tiles of 16, all held in registers, is a template parameter, so the loops carry no run-time
// bounds: tiles below MAXT-2 are entirely real keys; the last two start their score accumulators at -inf where the key is
// padding (C = -inf + finite stays -inf), which is the whole masking cost.
// WAVES waves per workgroup and WPE waves per SIMD asked of the compiler: 7 and 4 up to 224 keys (13 query tiles at 197
// tokens = at most 2 per wave; two workgroups per CU), 4 and 1 beyond.
// ODD: N <= (MAXT - 1) * 16, the last key tile is all padding (197 tokens are 13 tiles): its scores are not computed and it enters
// the 32-key P V products as zeros.  The images keep MAXT * 16 rows (the transposing reads take 32 rows at a time).
template <int MAXT, int WAVES, int WPE, int HDP = 64, bool ODD = false>
__global__ __launch_bounds__(WAVES * 64, WPE) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                   float* __restrict__ lse, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
    using G = AttnGeo<HDP>;
    constexpr int HDW = G::HDW, NKS = G::NKS, NDT = G::NDT, NCH = G::NCH, CPR = G::CPR, RPP = G::RPP;
    constexpr int NPAD = MAXT * 16;
    constexpr int KT = MAXT - (ODD ? 1 : 0);               // key tiles that hold real keys
    bf16_t* Ks = reinterpret_cast<bf16_t*>(attn_smem);
    bf16_t* Vs = Ks + NPAD * HDW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4, lq = lr >> 2, lp = lr & 3;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * HDP, ld = 3 * D;
    const bf16_t* base = qkv + (size_t)b * N * ld + h * HDP;

    {   // DMA fill: a piece = RPP rows x (HDW * 2) B; pieces [0, NPAD/RPP) are K, the rest V
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(Ks));
        const int prow = lane / CPR, ppos = lane % CPR;
        for (int pc = wave; pc < 2 * (NPAD / RPP); pc += WAVES) {
            const int which = pc >= NPAD / RPP ? 1 : 0;
            const int row = (pc - which * (NPAD / RPP)) * RPP + prow;
            const int srow = row < N ? row : N - 1;
            int ch = ppos ^ G::sig(row);
            if (CPR > NCH && ch >= NCH) ch -= 8;          // slots past the head's columns: any valid chunk, never read
            const bf16_t* src = base + (size_t)srow * ld + (1 + which) * D + (ch << 3);
            glds16_v(src, lds0 + (unsigned)pc * 1024u);
        }
    }

    const int nqt = (N + 15) / 16;
    const float c2 = scale * 1.44269504088896340736f;      // softmax in base 2
    f32x4 tail0[2];     // accumulator start values of the last two key tiles: 0 for a real key, -inf for padding
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) tail0[i][r] = ((KT - 2 + i) * 16 + 4 * lg + r < N) ? 0.f : -INFINITY;
    int koff[NKS];      // b128 fragment of row 16t + lr, K-slice s; (16t + lr) & 7 == lr & 7
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) koff[ks] = lr * HDW + (((4 * ks + lg) ^ G::sig(lr)) << 3);
    int voff[NDT];      // (32u + 4lg + lq) & 7 == (4lg + lq) & 7, and the same 16 rows further: one offset per d tile serves all u
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) voff[dt] = G::swz(4 * lg + lq, dt * 16 + 4 * lp);

    auto load_q = [&](int qt, bf16x8 (&qf)[NKS]) {
        int q = qt * 16 + lr;
        q = q < N ? q : N - 1;
        const bf16_t* qp = base + (size_t)q * ld + 8 * lg;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
    };
    bf16x8 qf[NKS], qn[NKS];
    if (wave < nqt) load_q(wave, qf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int qt = wave; qt < nqt; qt += WAVES) {
        if (qt + WAVES < nqt) load_q(qt + WAVES, qn);      // the next tile's queries travel under this tile's work
        // K fragments are fetched one key tile ahead by hand and the scheduler is fenced per tile: left alone, hipcc hoists every
        // fragment load of the unrolled loop to the top and spills
        f32x4 s[MAXT];
        bf16x8 kf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(&Ks[koff[ks]]);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            bf16x8 a[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) a[ks] = kf[ks];
            if (t + 1 < KT) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(&Ks[(t + 1) * 16 * HDW + koff[ks]]);
            }
            s[t] = t < KT - 2 ? (f32x4){0.f, 0.f, 0.f, 0.f} : tail0[t - (KT - 2)];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], qf[ks], s[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // s[t][r] = S^T[key = 16t + 4lg + r][q = lr] (unscaled)
        float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
        for (int t = 1; t < KT; ++t) mx = fmaxf(fmaxf(mx, s[t][0]), fmaxf(fmaxf(s[t][1], s[t][2]), s[t][3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nmc = -mx * c2;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], c2, nmc));
                s[t][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if constexpr (ODD) s[MAXT - 1] = (f32x4){0.f, 0.f, 0.f, 0.f};

        f32x4 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < MAXT / 2; ++u) {
            const bf16x8 pb = pack8(s[2 * u], s[2 * u + 1]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16_t* vp = Vs + u * 32 * HDW + voff[dt];
                const bf16x8 va = tr_frag2(vp, vp + 16 * HDW);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // o[dt][r] = O^T[d = 16dt + 4lg + r][q = lr]
        const int q = qt * 16 + lr;
        uint2 pc[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            pc[dt].x = pack2bf(o[dt][0] * inv, o[dt][1] * inv);
            pc[dt].y = pack2bf(o[dt][2] * inv, o[dt][3] * inv);
        }
        u32x4_t w[NDT / 2];
#pragma unroll
        for (int j = 0; j < NDT / 2; ++j) w[j] = pair_pieces(pc[2 * j], pc[2 * j + 1]);
        if (q < N) {
            bf16_t* op = out + ((size_t)b * N + q) * D + h * HDP;
#pragma unroll
            for (int j = 0; j < NDT / 2; ++j) *reinterpret_cast<u32x4_t*>(op + pair_col(2 * j, 2 * j + 1, lg)) = w[j];
            if (lg == 0) lse[((size_t)b * H + h) * N + q] = (mx * c2 + __builtin_amdgcn_logf(sum)) * 0.69314718055994530942f;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = qn[ks];
    }
}

// ----------------------------------------------------------------------------------------------- backward
// Two barrier-free phases; the scores are recomputed in the second instead of passing dS
// between waves:
//   phase 1, a wave owns key tiles:   S = Q K^T, dP = dO V^T (key on the lane), P = 2^((S - lse / scale) c), dS = P (dP - delta) scale
//                                     (-lse / scale and -delta are the start values of the S and dP accumulators, scale is applied to dK, dQ),
//                                     dV^T += dO^T P, dK^T += Q^T dS   (Q^T, dO^T by transposing reads)
//   phase 2, a wave owns query tiles: S^T = K Q^T, dP^T = V dO^T (query on the lane), same P and dS, dQ^T += K^T dS^T.
// What all waves share goes through LDS ([NPAD][64] images, chunk XOR-swizzled by row & 7, filled by DMA): Q and dO in
// phase 1, K and V in phase 2 - the SAME 56 KB, refilled between the phases - while what only one wave needs (its own K, V
// rows in phase 1, its own Q, dO rows in phase 2) comes straight from global memory into registers.  Up to 224 tokens 58 KB
// of LDS and 128 registers let two workgroups (14 waves) share a CU, so one head's loads and barriers hide under the other's
// MFMAs; beyond (up to 608 tokens = 157 KB) one 8-wave workgroup owns the CU.
// The owned key tiles of phase 1 are processed one after the other, each step in two stages that reuse the same registers
// (b128 fragments of Q, dO for S and dP; then their transposed fragments for dV, dK).
// Padding: query rows >= N carry -lse / scale = -inf, so P = dS = 0 there; key rows >= N start the S^T accumulators of phase 2 at
// -inf (tiles below MAXT-2 are all real keys); in phase 1 padded keys only feed dK/dV rows that are never stored.
// MAXT = 2 * ceil(N / 32) key tiles is a template parameter: no run-time loop bounds.
// ODD: N <= (MAXT - 1) * 16 - tile MAXT-1 is all padding as a key tile and as a query tile: no wave owns it, and where it is the
// second half of a 32-row step its scores are not computed and it enters the 32-deep products as zeros.
template <int MAXT, int WAVES, int WPE, int HDP = 64, bool ODD = false>
__global__ __launch_bounds__(WAVES * 64, WPE) void attn_bwd_res_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                       const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                       bf16_t* __restrict__ dqkv, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
    using G = AttnGeo<HDP>;
    constexpr int HDW = G::HDW, NKS = G::NKS, NDT = G::NDT, NCH = G::NCH, CPR = G::CPR, RPP = G::RPP;
    constexpr int NPAD = MAXT * 16;
    constexpr int KT = MAXT - (ODD ? 1 : 0);               // tiles that hold real tokens
    constexpr int TPW = (MAXT + WAVES - 1) / WAVES;        // key tiles (phase 1) / query tiles (phase 2) per wave
    bf16_t* R0 = reinterpret_cast<bf16_t*>(attn_smem);     // Q, then K
    bf16_t* R1 = R0 + NPAD * HDW;                           // dO, then V
    float* lse_s = reinterpret_cast<float*>(R1 + NPAD * HDW);  // [NPAD] -lse / scale, -inf past N
    float* dl_s = lse_s + NPAD;                                // [NPAD] -delta
    const float inv_scale = 1.0f / scale;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4, lq = lr >> 2, lp = lr & 3;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * HDP, ld = 3 * D;
    const bf16_t* base = qkv + (size_t)b * N * ld + h * HDP;
    const bf16_t* dobase = dout + (size_t)b * N * D + h * HDP;
    const bf16_t* obase = out + (size_t)b * N * D + h * HDP;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(R0));

    // DMA fill of the two images, a piece = RPP rows x (HDW * 2) B: first = false: (Q, dO), true: (K, V)
    auto fill = [&](bool second) {
        const int prow = lane / CPR, ppos = lane % CPR;
        for (int pc = wave; pc < 2 * (NPAD / RPP); pc += WAVES) {
            const int which = pc >= NPAD / RPP ? 1 : 0;
            const int row = (pc - which * (NPAD / RPP)) * RPP + prow;
            const int srow = row < N ? row : N - 1;
            int c = ppos ^ G::sig(row);
            if (CPR > NCH && c >= NCH) c -= 8;            // slots past the head's columns: any valid chunk, never read
            const int ch = c << 3;
            const bf16_t* src = second ? base + (size_t)srow * ld + (1 + which) * D + ch
                                       : (which ? dobase + (size_t)srow * D + ch : base + (size_t)srow * ld + ch);
            glds16_v(src, lds0 + (unsigned)pc * 1024u);
        }
    };
    // this wave's own K, V rows of a key tile (phase 1), straight from global memory as MFMA fragments
    auto load_kv = [&](int kt, bf16x8 (&kf)[NKS], bf16x8 (&vf)[NKS]) {
        int key = kt * 16 + lr;
        key = key < N ? key : N - 1;
        const bf16_t* kp = base + (size_t)key * ld + D + 8 * lg;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 32 * ks);
            vf[ks] = *reinterpret_cast<const bf16x8*>(kp + D + 32 * ks);
        }
    };
    // Up to 224 tokens a wave owns at most two tiles per phase (OWN): nothing that sits in LDS is fetched from global memory a
    // second time - delta takes dO from the LDS image, and the wave's own Q, dO fragments of phase 2 are read out of the
    // images before K and V replace them - and the first own K, V fragments travel with the fill.
    // delta items: (query row, 16-byte chunk slot), LPQ neighbouring lanes share a row (8 at 64 columns; 16 at 96, where the
    // slots past the head contribute nothing)
    constexpr int LPQ = CPR, LPQ_SH = HDW == 64 ? 3 : 4;
    constexpr int TOT = NPAD * LPQ, STR = WAVES * 64;
    constexpr bool OWN = VITK_ATTN_OWN && TOT % STR == 0 && TOT / STR <= (HDW == 64 ? 4 : 8) && TPW <= 2;
    bf16x8 kf[NKS], vf[NKS];
    if constexpr (OWN) load_kv(wave, kf, vf);
    fill(false);
    auto row_dot = [&](const u32x4_t ra, const u32x4_t rc) {
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            part += __uint_as_float(ra[j] << 16) * __uint_as_float(rc[j] << 16);
            part += __uint_as_float(ra[j] & 0xffff0000u) * __uint_as_float(rc[j] & 0xffff0000u);
        }
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (LPQ == 16) part += __shfl_xor(part, 8, 64);
        return part;
    };
    // delta and lse.  Up to 224 tokens the items are an exact multiple of the block size and all loads of a thread are issued
    // before the first is used; beyond, a plain loop (every trip is taken by whole lane groups, so the shuffles stay inside
    // active lanes).
    if constexpr (OWN) {
        constexpr int ITEMS = TOT / STR;
        u32x4_t rc[ITEMS];
        float lv[ITEMS];
        const u32x4_t zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + i * STR;
            int q = it >> LPQ_SH;
            q = q < N ? q : N - 1;
            const int c = it & (LPQ - 1);
            const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(obase + (size_t)q * D + (c < NCH ? c : 0) * 8));
            rc[i] = c < NCH ? t : zero4;
            lv[i] = lse[((size_t)b * H + h) * N + q];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + i * STR;
            const int q = it >> LPQ_SH;
            const int c = it & (LPQ - 1);
            const u32x4_t ra = *reinterpret_cast<const u32x4_t*>(R1 + q * HDW + ((c ^ G::sig(q)) << 3));
            const float part = row_dot(ra, rc[i]);
            if (c == 0) {
                dl_s[q] = -part;
                lse_s[q] = q < N ? -lv[i] * inv_scale : -INFINITY;
            }
        }
    } else if constexpr (HDW == 64 && TOT % STR == 0 && TOT / STR <= 4) {
        constexpr int ITEMS = TOT / STR;
        u32x4_t ra[ITEMS], rc[ITEMS];
        float lv[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + i * STR;
            int q = it >> 3;
            q = q < N ? q : N - 1;
            ra[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(dobase + (size_t)q * D + (it & 7) * 8));
            rc[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(obase + (size_t)q * D + (it & 7) * 8));
            lv[i] = lse[((size_t)b * H + h) * N + q];
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + i * STR;
            const int q = it >> 3;
            const float part = row_dot(ra[i], rc[i]);
            if ((it & 7) == 0) {
                dl_s[q] = -part;
                lse_s[q] = q < N ? -lv[i] * inv_scale : -INFINITY;
            }
        }
    } else {
        const u32x4_t zero4 = {0u, 0u, 0u, 0u};
#pragma unroll 2
        for (int it = tid; it < TOT; it += STR) {
            const int q = it >> LPQ_SH;
            const int qc = q < N ? q : N - 1;
            const int c = it & (LPQ - 1), cc = c < NCH ? c : 0;
            u32x4_t ra = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(dobase + (size_t)qc * D + cc * 8));
            const u32x4_t rc = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(obase + (size_t)qc * D + cc * 8));
            if (c >= NCH) ra = zero4;
            const float lv = lse[((size_t)b * H + h) * N + qc];
            const float part = row_dot(ra, rc);
            if (c == 0) {
                dl_s[q] = -part;
                lse_s[q] = q < N ? -lv * inv_scale : -INFINITY;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float c2 = scale * 1.44269504088896340736f;
    int foff[NKS];      // b128 fragment of row 16t + lr, K-slice s
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) foff[ks] = lr * HDW + (((4 * ks + lg) ^ G::sig(lr)) << 3);
    int toff[NDT];      // transposing-read offset of rows 4lg + lq (+16), columns 16dt + 4lp; same for every 32-row group
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) toff[dt] = G::swz(4 * lg + lq, dt * 16 + 4 * lp);

    // ---------------------------------------------------------------- phase 1: dK, dV of the owned key tiles (R0 = Q, R1 = dO)
    constexpr bool UNR = MAXT <= 14 && HDP == 64;           // the 32-query steps of phase 1 are unrolled
#pragma unroll 1
    for (int kt = wave; kt < KT; kt += WAVES) {
        if (!OWN || kt != wave) load_kv(kt, kf, vf);
        f32x4 dv[NDT], dk[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) { dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        // up to 224 tokens the 32-query steps are unrolled: every LDS address becomes base + immediate (the rolled loop spends
        // a third of its vector instructions on address arithmetic)
#pragma unroll(UNR ? MAXT / 2 : 1)
        for (int st = 0; st < MAXT / 2; ++st) {
            const bf16_t* Qp = R0 + st * 32 * HDW;
            const bf16_t* dOp = R1 + st * 32 * HDW;
            bf16x8 pa, dsa;
            {   // stage 1: scores and their gradient for 32 queries x 16 keys
                f32x4 pt[2], ds[2];
                const f32x4 l0 = *reinterpret_cast<const f32x4*>(&lse_s[st * 32 + 4 * lg]);
                const f32x4 l1 = *reinterpret_cast<const f32x4*>(&lse_s[st * 32 + 16 + 4 * lg]);
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(&dl_s[st * 32 + 4 * lg]);
                const f32x4 d1 = *reinterpret_cast<const f32x4*>(&dl_s[st * 32 + 16 + 4 * lg]);
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    if (ODD && UNR && st == MAXT / 2 - 1 && qi == 1) {      // the padding tile: a constant after unrolling
                        pt[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        ds[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        continue;
                    }
                    bf16x8 qv[NKS], ov[NKS];
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        qv[ks] = *reinterpret_cast<const bf16x8*>(Qp + qi * 16 * HDW + foff[ks]);
                        ov[ks] = *reinterpret_cast<const bf16x8*>(dOp + qi * 16 * HDW + foff[ks]);
                    }
                    // the row constants are the accumulators' start values: S - lse / scale and dP - delta come out of the chains
                    f32x4 sacc = qi ? l1 : l0, dpacc = qi ? d1 : d0;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qv[ks], kf[ks], sacc, 0, 0, 0);
                        dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ov[ks], vf[ks], dpacc, 0, 0, 0);
                    }
                    // sacc[r] = S[q = 32st + 16qi + 4lg + r][key = lr] - lse[q] / scale
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pp = __builtin_amdgcn_exp2f(sacc[r] * c2);
                        pt[qi][r] = pp;
                        ds[qi][r] = pp * dpacc[r];           // dS / scale: dK is scaled once, below
                    }
                }
                pa = pack8(pt[0], pt[1]);     // contraction slot (lg, j): q = 4lg + j | 16 + 4lg + j - 4
                dsa = pack8(ds[0], ds[1]);
            }
            // stage 2: dV^T += dO^T P, dK^T += Q^T dS
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16x8 dotr = tr_frag2(dOp + toff[dt], dOp + 16 * HDW + toff[dt]);
                const bf16x8 qtr = tr_frag2(Qp + toff[dt], Qp + 16 * HDW + toff[dt]);
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotr, pa, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr, dsa, dk[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // keeps the unrolled steps from being hoisted into each other (spills)
        }
        // lane holds dK^T / dV^T [d = 16dt + 4lg + r][key = lr]
        const int key = kt * 16 + lr;
        uint2 pk[NDT], pv[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            dk[dt] *= scale;
            pk[dt].x = pack2bf(dk[dt][0], dk[dt][1]); pk[dt].y = pack2bf(dk[dt][2], dk[dt][3]);
            pv[dt].x = pack2bf(dv[dt][0], dv[dt][1]); pv[dt].y = pack2bf(dv[dt][2], dv[dt][3]);
        }
        u32x4_t kw[NDT / 2], vw[NDT / 2];
#pragma unroll
        for (int j = 0; j < NDT / 2; ++j) { kw[j] = pair_pieces(pk[2 * j], pk[2 * j + 1]); vw[j] = pair_pieces(pv[2 * j], pv[2 * j + 1]); }
        if (key < N) {
            bf16_t* rowp = dqkv + ((size_t)b * N + key) * ld + h * HDP;
#pragma unroll
            for (int j = 0; j < NDT / 2; ++j) *reinterpret_cast<u32x4_t*>(rowp + D + pair_col(2 * j, 2 * j + 1, lg)) = kw[j];
#pragma unroll
            for (int j = 0; j < NDT / 2; ++j) *reinterpret_cast<u32x4_t*>(rowp + 2 * D + pair_col(2 * j, 2 * j + 1, lg)) = vw[j];
        }
    }

    // this wave's own Q, dO fragments of phase 2: out of the LDS images while they still hold Q and dO (OWN), else from global
    auto load_q = [&](int qt, bf16x8 (&qf)[NKS], bf16x8 (&dof)[NKS]) {
        int q = qt * 16 + lr;
        q = q < N ? q : N - 1;
        const bf16_t* qp = base + (size_t)q * ld + 8 * lg;
        const bf16_t* dp = dobase + (size_t)q * D + 8 * lg;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
            dof[ks] = *reinterpret_cast<const bf16x8*>(dp + 32 * ks);
        }
    };
    auto lds_q = [&](int qt, bf16x8 (&qf)[NKS], bf16x8 (&dof)[NKS]) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(R0 + qt * 16 * HDW + foff[ks]);
            dof[ks] = *reinterpret_cast<const bf16x8*>(R1 + qt * 16 * HDW + foff[ks]);
        }
    };
    bf16x8 qf[NKS], dof[NKS], qf2[NKS], dof2[NKS];
    if constexpr (OWN) {
        lds_q(wave, qf, dof);
        if constexpr (TPW == 2) lds_q(wave + WAVES < MAXT ? wave + WAVES : wave, qf2, dof2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // every wave is done with Q and dO: refill the same LDS with K and V
    __syncthreads();
    fill(true);
    if constexpr (!OWN) load_q(wave, qf, dof);      // the first query tile travels with the refill
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---------------------------------------------------------------- phase 2: dQ of the owned query tiles (R0 = K, R1 = V)
    f32x4 tail0[2];     // S^T accumulator start values of the last two key tiles: 0 for a real key, -inf for padding
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) tail0[i][r] = ((KT - 2 + i) * 16 + 4 * lg + r < N) ? 0.f : -INFINITY;
#pragma unroll 1
    for (int qt = wave; qt * 16 < N; qt += WAVES) {
        const int q = qt * 16 + lr;
        const float nl = lse_s[q], dl = dl_s[q];          // -lse / scale and -delta of this lane's query: accumulator start values
        const f32x4 s0 = {nl, nl, nl, nl}, d0 = {dl, dl, dl, dl};
        f32x4 tl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) tl[i] = tail0[i] + nl;
        f32x4 dq[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 kn[NKS], vn[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            kn[ks] = *reinterpret_cast<const bf16x8*>(&R0[foff[ks]]);
            vn[ks] = *reinterpret_cast<const bf16x8*>(&R1[foff[ks]]);
        }
#pragma unroll
        for (int u = 0; u < MAXT / 2; ++u) {
            f32x4 ds[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int t = 2 * u + e;
                if (t >= KT) {                                  // the padding tile
                    ds[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    continue;
                }
                bf16x8 ka[NKS], va[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) { ka[ks] = kn[ks]; va[ks] = vn[ks]; }
                if (t + 1 < KT) {
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        kn[ks] = *reinterpret_cast<const bf16x8*>(&R0[(t + 1) * 16 * HDW + foff[ks]]);
                        vn[ks] = *reinterpret_cast<const bf16x8*>(&R1[(t + 1) * 16 * HDW + foff[ks]]);
                    }
                }
                f32x4 sacc = t < KT - 2 ? s0 : tl[t - (KT - 2)];
                f32x4 dpacc = d0;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[ks], qf[ks], sacc, 0, 0, 0);
                    dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[ks], dof[ks], dpacc, 0, 0, 0);
                }
                // sacc[r] = S^T[key = 16t + 4lg + r][q = lr]
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[e][r] = __builtin_amdgcn_exp2f(sacc[r] * c2) * dpacc[r];      // dS / scale
            }
            const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16_t* kp = R0 + u * 32 * HDW + toff[dt];
                const bf16x8 ktr = tr_frag2(kp, kp + 16 * HDW);
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr, dsb, dq[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!OWN) {
            if ((qt + WAVES) * 16 < N) load_q(qt + WAVES, qf, dof);
        }
        // dq[dt][r] = dQ^T[d = 16dt + 4lg + r][q = lr]
        uint2 pq[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            dq[dt] *= scale;
            pq[dt].x = pack2bf(dq[dt][0], dq[dt][1]);
            pq[dt].y = pack2bf(dq[dt][2], dq[dt][3]);
        }
        u32x4_t qw[NDT / 2];
#pragma unroll
        for (int j = 0; j < NDT / 2; ++j) qw[j] = pair_pieces(pq[2 * j], pq[2 * j + 1]);
        if (q < N) {
            bf16_t* rowp = dqkv + ((size_t)b * N + q) * ld + h * HDP;
#pragma unroll
            for (int j = 0; j < NDT / 2; ++j) *reinterpret_cast<u32x4_t*>(rowp + pair_col(2 * j, 2 * j + 1, lg)) = qw[j];
        }
        if constexpr (OWN && TPW == 2) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) { qf[ks] = qf2[ks]; dof[ks] = dof2[ks]; }
        }
    }
}

extern "C" {

int vitk_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, void* stream) {
    VITK_REQUIRE(qkv && out && lse, "attention_fwd: null pointer");
    VITK_REQUIRE(B > 0 && N > 0 && H > 0 && N <= 608, "attention_fwd: need B,H>0 and 0<N<=608 (B=%d N=%d H=%d)", B, N, H);
    VITK_REQUIRE(vitk_aligned(qkv, 16) && vitk_aligned(out, 16), "attention_fwd: pointers must be 16-byte aligned");
    const int maxt = 2 * ((N + 31) / 32);
    const size_t smem = (size_t)2 * maxt * 16 * HD * sizeof(bf16_t);
    hipStream_t s = (hipStream_t)stream;
    static VitkLdsOnce attr_done[40] = {};
    const bool pr = vitk_prof_begin(stream, 8.0 * (double)B * (double)N * (double)H * HD, VITK_PROF_ATTN_FWD);
    if (VITK_ATTN_ODD && maxt == 14 && N <= 13 * 16) {       // 193 ... 208 tokens (ViT at 224 px: 197, distilled 198): 13 tiles, the 14th is padding
        static VitkLdsOnce odd_done = {};
        if (int rc_ = vitk_lds_once(odd_done, (const void*)attn_fwd_kernel<14, 7, 4, 64, true>, smem, "attention_fwd")) return rc_;
        attn_fwd_kernel<14, 7, 4, 64, true><<<B * H, 7 * 64, smem, s>>>((const bf16_t*)qkv, (bf16_t*)out, lse, N, H, scale);
        if (pr) vitk_prof_end(stream);
        return vitk_check_launch("attention_fwd");
    }
#define VITK_FWD_CASE(T, W, E)                                                                                              \
    case T: {                                                                                                               \
        if (int rc_ = vitk_lds_once(attr_done[T], (const void*)attn_fwd_kernel<T, W, E>, smem, "attention_fwd")) return rc_;          \
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
    if (pr) vitk_prof_end(stream);
    return vitk_check_launch("attention_fwd");
}

size_t vitk_attention_bwd_workspace(int B, int N, int H) {
    (void)B; (void)N; (void)H;
    return 0;       // the backward keeps everything in LDS and registers
}

int vitk_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N,
                       int H, float scale, void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    VITK_REQUIRE(qkv && out && dout && lse && dqkv, "attention_bwd: null pointer");
    VITK_REQUIRE(B > 0 && N > 0 && H > 0 && N <= 608, "attention_bwd: need B,H>0 and 0<N<=608 (B=%d N=%d H=%d)", B, N, H);
    VITK_REQUIRE(vitk_aligned(qkv, 16) && vitk_aligned(out, 16) && vitk_aligned(dout, 16) && vitk_aligned(dqkv, 16),
                 "attention_bwd: pointers must be 16-byte aligned");
    const int maxt = 2 * ((N + 31) / 32);
    const size_t smem = (size_t)2 * maxt * 16 * HD * sizeof(bf16_t) + (size_t)2 * maxt * 16 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    static VitkLdsOnce attr_done[40] = {};
    const bool pr = vitk_prof_begin(stream, 16.0 * (double)B * (double)N * (double)H * HD, VITK_PROF_ATTN_BWD);
    if (VITK_ATTN_ODD && maxt == 14 && N <= 13 * 16) {       // 13 tiles of real tokens (see vitk_attention_fwd)
        static VitkLdsOnce odd_done = {};
        if (int rc_ = vitk_lds_once(odd_done, (const void*)attn_bwd_res_kernel<14, 7, 4, 64, true>, smem, "attention_bwd")) return rc_;
        attn_bwd_res_kernel<14, 7, 4, 64, true><<<B * H, 7 * 64, smem, s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse,
                                                                          (bf16_t*)dqkv, N, H, scale);
        if (pr) vitk_prof_end(stream);
        return vitk_check_launch("attention_bwd");
    }
#define VITK_BWD_CASE(T, W, E)                                                                                                 \
    case T: {                                                                                                               \
        if (int rc_ = vitk_lds_once(attr_done[T], (const void*)attn_bwd_res_kernel<T, W, E>, smem, "attention_bwd")) return rc_;          \
        attn_bwd_res_kernel<T, W, E><<<B * H, W * 64, smem, s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, \
                                                              (bf16_t*)dqkv, N, H, scale);                                  \
    } break;
    switch (maxt) {
        VITK_BWD_CASE(2, 2, 4) VITK_BWD_CASE(4, 4, 4) VITK_BWD_CASE(6, 6, 4) VITK_BWD_CASE(8, 4, 4) VITK_BWD_CASE(10, 5, 4)
        VITK_BWD_CASE(12, 6, 4) VITK_BWD_CASE(14, 7, 4)
        VITK_BWD_CASE(16, 8, 2) VITK_BWD_CASE(18, 8, 2) VITK_BWD_CASE(20, 8, 2) VITK_BWD_CASE(22, 8, 2) VITK_BWD_CASE(24, 8, 2)
        VITK_BWD_CASE(26, 8, 2) VITK_BWD_CASE(28, 8, 2) VITK_BWD_CASE(30, 8, 2) VITK_BWD_CASE(32, 8, 2) VITK_BWD_CASE(34, 8, 2)
        VITK_BWD_CASE(36, 8, 2) VITK_BWD_CASE(38, 8, 2)
        default: vitk_set_error("attention_bwd: unsupported token count %d", N); return VITK_EINVAL;
    }
#undef VITK_BWD_CASE
    if (pr) vitk_prof_end(stream);
    return vitk_check_launch("attention_bwd");
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------------
// Attention for head dims other than 64 (48 and 72 in the reference's my_vit_mini / my_vit_xs, models/my_vit.py:84-165): any
// multiple of 8 up to 96, up to 256 tokens.  These are the small models, so this path is plain vector arithmetic, one thread
// per query (or key) row with the whole row in registers and the other side's rows broadcast out of LDS - the same two-phase
// plan and the same bf16 storage points as the MFMA kernels above (P and dS rounded before their products, O rounded on
// output), fp32 everywhere else.  One workgroup of 256 threads per (batch, head); one wave per SIMD, so a thread may hold the
// four rows of the backward's key phase (4 x 96 floats).
#define GHDMAX 96       // largest head dim; register rows are sized by the template parameter GHD (48 or 96) >= the run-time head dim
#define GNMAX 256

template <int GHD>
static __device__ __forceinline__ void g_load_row(const bf16_t* p, int hd, float (&r)[GHD]) {
#pragma unroll
    for (int c = 0; c < GHD / 8; ++c) {
        if (c * 8 < hd) {
            float t[8];
            load8(p + c * 8, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[c * 8 + j] = t[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[c * 8 + j] = 0.f;
        }
    }
}
template <int GHD>
static __device__ __forceinline__ void g_store_row(bf16_t* p, int hd, const float (&r)[GHD]) {
#pragma unroll
    for (int c = 0; c < GHD / 8; ++c)
        if (c * 8 < hd) {
            const float t[8] = {r[c * 8], r[c * 8 + 1], r[c * 8 + 2], r[c * 8 + 3], r[c * 8 + 4], r[c * 8 + 5], r[c * 8 + 6], r[c * 8 + 7]};
            store8(p + c * 8, t);
        }
}
// dot of a register row with an LDS row (all lanes read the same address: broadcast), and axpy from an LDS row
template <int GHD>
static __device__ __forceinline__ float g_dot(const float (&r)[GHD], const bf16_t* s, int hd) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < GHD / 8; ++c)
        if (c * 8 < hd) {
            float t[8];
            load8(s + c * 8, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_fmaf(r[c * 8 + j], t[j], acc);
        }
    return acc;
}
template <int GHD>
static __device__ __forceinline__ void g_axpy(float a, const bf16_t* s, int hd, float (&r)[GHD]) {
#pragma unroll
    for (int c = 0; c < GHD / 8; ++c)
        if (c * 8 < hd) {
            float t[8];
            load8(s + c * 8, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[c * 8 + j] = __builtin_fmaf(a, t[j], r[c * 8 + j]);
        }
}
static __device__ __forceinline__ float g_round_bf16(float x) { return bf2f(f2bf(x)); }

// rows [N][hd] of one (batch, head) slice of a [B*N, ld] tensor -> LDS [N][hd] (16-byte pieces)
static __device__ __forceinline__ void g_fill(bf16_t* dst, const bf16_t* src, int ld, int N, int hd) {
    const int cpr = hd / 8;
    for (int i = threadIdx.x; i < N * cpr; i += blockDim.x) {
        const int row = i / cpr, c = i - row * cpr;
        *reinterpret_cast<uint4*>(dst + row * hd + c * 8) = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + c * 8);
    }
}

template <int GHD>
__global__ __launch_bounds__(256) void attn_fwd_generic_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse,
                                                                  int N, int H, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(attn_smem);
    bf16_t* Vs = Ks + N * hd;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * hd, ld = 3 * D;
    const bf16_t* base = qkv + (size_t)b * N * ld + h * hd;
    g_fill(Ks, base + D, ld, N, hd);
    g_fill(Vs, base + 2 * D, ld, N, hd);
    __syncthreads();
    const int q = threadIdx.x;
    if (q >= N) return;
    float qr[GHD], o[GHD];
    g_load_row(base + (size_t)q * ld, hd, qr);
    float mx = -INFINITY;
#pragma unroll 4
    for (int j = 0; j < N; ++j) mx = fmaxf(mx, g_dot(qr, Ks + j * hd, hd));      // unrolled: four keys' LDS reads in flight per thread
    const float c2 = scale * 1.44269504088896340736f;
#pragma unroll
    for (int d = 0; d < GHD; ++d) o[d] = 0.f;
    float sum = 0.f;
#pragma unroll 4
    for (int j = 0; j < N; ++j) {
        const float p = __builtin_amdgcn_exp2f((g_dot(qr, Ks + j * hd, hd) - mx) * c2);
        sum += p;
        g_axpy(g_round_bf16(p), Vs + j * hd, hd, o);
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int d = 0; d < GHD; ++d) o[d] *= inv;
    g_store_row(out + ((size_t)b * N + q) * D + h * hd, hd, o);
    lse[((size_t)b * H + h) * N + q] = (mx * c2 + __builtin_amdgcn_logf(sum)) * 0.69314718055994530942f;
}

template <int GHD>
__global__ __launch_bounds__(256) void attn_bwd_generic_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                  const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                  bf16_t* __restrict__ dqkv, int N, int H, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_smem[];
    bf16_t* R0 = reinterpret_cast<bf16_t*>(attn_smem);       // Q, then K
    bf16_t* R1 = R0 + N * hd;                                 // dO, then V
    float* lse_s = reinterpret_cast<float*>(R1 + N * hd);     // [N] lse * log2(e)
    float* dl_s = lse_s + N;                                  // [N] rowsum(dO * O)
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * hd, ld = 3 * D;
    const bf16_t* base = qkv + (size_t)b * N * ld + h * hd;
    const bf16_t* dobase = dout + (size_t)b * N * D + h * hd;
    const bf16_t* obase = out + (size_t)b * N * D + h * hd;
    const int t = threadIdx.x;
    const float c2 = scale * 1.44269504088896340736f;
    g_fill(R0, base, ld, N, hd);
    g_fill(R1, dobase, D, N, hd);
    if (t < N) {
        float a[GHD], c[GHD];
        g_load_row(dobase + (size_t)t * D, hd, a);
        g_load_row(obase + (size_t)t * D, hd, c);
        float part = 0.f;
#pragma unroll
        for (int d = 0; d < GHD; ++d) part = __builtin_fmaf(a[d], c[d], part);
        dl_s[t] = part;
        lse_s[t] = lse[((size_t)b * H + h) * N + t] * 1.44269504088896340736f;
    }
    __syncthreads();
    // phase 1: thread = key j.  dV[j] = sum_i bf16(P[i,j]) dO[i];  dK[j] = scale * sum_i bf16(dS[i,j] / scale) Q[i]
    // (dS is rounded before the scale, as in the MFMA kernels, where the scale is applied once to the dK / dQ accumulators)
    if (t < N) {
        float kr[GHD], vr[GHD], dk[GHD], dv[GHD];
        g_load_row(base + (size_t)t * ld + D, hd, kr);
        g_load_row(base + (size_t)t * ld + 2 * D, hd, vr);
#pragma unroll
        for (int d = 0; d < GHD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
#pragma unroll 2
        for (int i = 0; i < N; ++i) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(g_dot(kr, R0 + i * hd, hd), c2, -lse_s[i]));
            const float ds = p * (g_dot(vr, R1 + i * hd, hd) - dl_s[i]);
            g_axpy(g_round_bf16(p), R1 + i * hd, hd, dv);
            g_axpy(g_round_bf16(ds), R0 + i * hd, hd, dk);
        }
#pragma unroll
        for (int d = 0; d < GHD; ++d) dk[d] *= scale;
        bf16_t* rowp = dqkv + ((size_t)b * N + t) * ld + h * hd;
        g_store_row(rowp + D, hd, dk);
        g_store_row(rowp + 2 * D, hd, dv);
    }
    __syncthreads();
    g_fill(R0, base + D, ld, N, hd);
    g_fill(R1, base + 2 * D, ld, N, hd);
    __syncthreads();
    // phase 2: thread = query i.  dQ[i] = sum_j bf16(dS[i,j]) K[j]
    if (t < N) {
        float qr[GHD], dor[GHD], dq[GHD];
        g_load_row(base + (size_t)t * ld, hd, qr);
        g_load_row(dobase + (size_t)t * D, hd, dor);
#pragma unroll
        for (int d = 0; d < GHD; ++d) dq[d] = 0.f;
        const float nl = -lse_s[t], dl = dl_s[t];
#pragma unroll 4
        for (int j = 0; j < N; ++j) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(g_dot(qr, R0 + j * hd, hd), c2, nl));
            const float ds = p * (g_dot(dor, R1 + j * hd, hd) - dl);
            g_axpy(g_round_bf16(ds), R0 + j * hd, hd, dq);
        }
#pragma unroll
        for (int d = 0; d < GHD; ++d) dq[d] *= scale;
        g_store_row(dqkv + ((size_t)b * N + t) * ld + h * hd, hd, dq);
    }
}

extern "C" {

int vitk_attention_generic_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int head_dim, float scale, void* stream) {
    VITK_REQUIRE(qkv && out && lse, "attention_generic_fwd: null pointer");
    VITK_REQUIRE(B > 0 && H > 0 && N > 0 && N <= GNMAX, "attention_generic_fwd: need B,H>0 and 0<N<=%d (B=%d N=%d H=%d)", GNMAX, B, N, H);
    VITK_REQUIRE(head_dim >= 8 && head_dim <= GHDMAX && head_dim % 8 == 0, "attention_generic_fwd: head dim %d must be a multiple of 8 in [8, %d]",
                 head_dim, GHDMAX);
    VITK_REQUIRE(vitk_aligned(qkv, 16) && vitk_aligned(out, 16), "attention_generic_fwd: pointers must be 16-byte aligned");
    if (head_dim == 96 && N <= 224) {
        // heads of 96 columns (the model zero-pads 72 to 96) on the MFMA kernels: 256-byte LDS rows, one workgroup per compute unit
        const int maxt = 2 * ((N + 31) / 32);
        const size_t wsmem = (size_t)2 * maxt * 16 * 128 * sizeof(bf16_t);
        static VitkLdsOnce wdone[16] = {};
        hipStream_t s = (hipStream_t)stream;
#define VITK_WFWD_CASE(T, W)                                                                                                \
    case T: {                                                                                                               \
        if (int rc_ = vitk_lds_once(wdone[T], (const void*)attn_fwd_kernel<T, W, 2, 96>, wsmem, "attention_fwd(96)")) return rc_; \
        attn_fwd_kernel<T, W, 2, 96><<<B * H, W * 64, wsmem, s>>>((const bf16_t*)qkv, (bf16_t*)out, lse, N, H, scale);         \
    } break;
        if (VITK_ATTN_ODD && maxt == 14 && N <= 13 * 16) {       // 13 tiles of real tokens (see vitk_attention_fwd)
            static VitkLdsOnce odd_done = {};
            if (int rc_ = vitk_lds_once(odd_done, (const void*)attn_fwd_kernel<14, 7, 2, 96, true>, wsmem, "attention_fwd(96)")) return rc_;
            attn_fwd_kernel<14, 7, 2, 96, true><<<B * H, 7 * 64, wsmem, s>>>((const bf16_t*)qkv, (bf16_t*)out, lse, N, H, scale);
            return vitk_check_launch("attention_fwd(96)");
        }
        switch (maxt) {
            VITK_WFWD_CASE(2, 4) VITK_WFWD_CASE(4, 4) VITK_WFWD_CASE(6, 4) VITK_WFWD_CASE(8, 4) VITK_WFWD_CASE(10, 7) VITK_WFWD_CASE(12, 7)
            VITK_WFWD_CASE(14, 7)
            default: vitk_set_error("attention_generic_fwd: unsupported token count %d", N); return VITK_EINVAL;
        }
#undef VITK_WFWD_CASE
        return vitk_check_launch("attention_fwd(96)");
    }
    const size_t smem = (size_t)2 * N * head_dim * sizeof(bf16_t);
#define VITK_GFWD(G)                                                                                                                  \
    {                                                                                                                                 \
        static VitkLdsOnce once = {};                                                                                                 \
        if (int rc_ = vitk_lds_once(once, (const void*)attn_fwd_generic_kernel<G>, (size_t)2 * GNMAX * GHDMAX * sizeof(bf16_t),       \
                                    "attention_generic_fwd"))                                                                        \
            return rc_;                                                                                                               \
        attn_fwd_generic_kernel<G><<<B * H, 256, smem, (hipStream_t)stream>>>((const bf16_t*)qkv, (bf16_t*)out, lse, N, H, head_dim, scale); \
    }
    if (head_dim <= 48) VITK_GFWD(48) else VITK_GFWD(96)      // (a 72-wide instantiation measured slower than the 96-wide one: 1 wave per SIMD either way)
#undef VITK_GFWD
    return vitk_check_launch("attention_generic_fwd");
}

int vitk_attention_generic_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N, int H,
                               int head_dim, float scale, void* stream) {
    VITK_REQUIRE(qkv && out && dout && lse && dqkv, "attention_generic_bwd: null pointer");
    VITK_REQUIRE(B > 0 && H > 0 && N > 0 && N <= GNMAX, "attention_generic_bwd: need B,H>0 and 0<N<=%d (B=%d N=%d H=%d)", GNMAX, B, N, H);
    VITK_REQUIRE(head_dim >= 8 && head_dim <= GHDMAX && head_dim % 8 == 0, "attention_generic_bwd: head dim %d must be a multiple of 8 in [8, %d]",
                 head_dim, GHDMAX);
    VITK_REQUIRE(vitk_aligned(qkv, 16) && vitk_aligned(out, 16) && vitk_aligned(dout, 16) && vitk_aligned(dqkv, 16),
                 "attention_generic_bwd: pointers must be 16-byte aligned");
    if (head_dim == 96 && N <= 224) {
        const int maxt = 2 * ((N + 31) / 32);
        const size_t wsmem = (size_t)2 * maxt * 16 * 128 * sizeof(bf16_t) + (size_t)2 * maxt * 16 * sizeof(float);
        static VitkLdsOnce wdone[16] = {};
        hipStream_t s = (hipStream_t)stream;
#define VITK_WBWD_CASE(T, W)                                                                                                   \
    case T: {                                                                                                               \
        if (int rc_ = vitk_lds_once(wdone[T], (const void*)attn_bwd_res_kernel<T, W, 2, 96>, wsmem, "attention_bwd(96)")) return rc_; \
        attn_bwd_res_kernel<T, W, 2, 96><<<B * H, W * 64, wsmem, s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, \
                                                                   (bf16_t*)dqkv, N, H, scale);                             \
    } break;
        if (VITK_ATTN_ODD && maxt == 14 && N <= 13 * 16) {
            static VitkLdsOnce odd_done = {};
            if (int rc_ = vitk_lds_once(odd_done, (const void*)attn_bwd_res_kernel<14, 7, 2, 96, true>, wsmem, "attention_bwd(96)")) return rc_;
            attn_bwd_res_kernel<14, 7, 2, 96, true><<<B * H, 7 * 64, wsmem, s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse,
                                                                              (bf16_t*)dqkv, N, H, scale);
            return vitk_check_launch("attention_bwd(96)");
        }
        switch (maxt) {
            VITK_WBWD_CASE(2, 2) VITK_WBWD_CASE(4, 4) VITK_WBWD_CASE(6, 6) VITK_WBWD_CASE(8, 4) VITK_WBWD_CASE(10, 5) VITK_WBWD_CASE(12, 6)
            VITK_WBWD_CASE(14, 7)
            default: vitk_set_error("attention_generic_bwd: unsupported token count %d", N); return VITK_EINVAL;
        }
#undef VITK_WBWD_CASE
        return vitk_check_launch("attention_bwd(96)");
    }
    const size_t smem = (size_t)2 * N * head_dim * sizeof(bf16_t) + (size_t)2 * N * sizeof(float);
#define VITK_GBWD(G)                                                                                                                  \
    {                                                                                                                                 \
        static VitkLdsOnce once = {};                                                                                                 \
        if (int rc_ = vitk_lds_once(once, (const void*)attn_bwd_generic_kernel<G>,                                                    \
                                    (size_t)2 * GNMAX * GHDMAX * sizeof(bf16_t) + 2 * GNMAX * sizeof(float), "attention_generic_bwd"))  \
            return rc_;                                                                                                               \
        attn_bwd_generic_kernel<G><<<B * H, 256, smem, (hipStream_t)stream>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, \
                                                                             (bf16_t*)dqkv, N, H, head_dim, scale);                  \
    }
    if (head_dim <= 48) VITK_GBWD(48) else VITK_GBWD(96)
#undef VITK_GBWD
    return vitk_check_launch("attention_generic_bwd");
}

}  // extern "C"
