// bf16 MFMA GEMMs (fp32 accumulate, v_mfma_f32_16x16x32_bf16) for the ViT linears, gfx950.
//
//   gemm_nt : C[M,N] = A[M,K] * B[N,K]^T + fused epilogue   (forward linears; data gradients with a transposed weight copy)
//   gemm_tn : C[N,K] = A[M,N]^T * B[M,K]                    (weight gradients: contraction over the M token rows)
//
// Each has two kernels.
//   gemm_ntp_kernel / gemm_tnb_kernel: the large-tile kernels every big linear takes - one 512-thread workgroup per compute
//     unit, 256 x 256 tiles, 8 waves of 128 x 64 (128 accumulator registers), operands global -> LDS by LDS-DMA into a
//     double-buffered 128 KB image, fragments read one 16-MFMA phase ahead.  gemm_ntp is persistent (a workgroup claims tile
//     after tile, the K-steps of all of them one continuous request stream) and transposes its accumulators through LDS so that
//     every global access of the epilogue is a full line; gemm_tnb runs one (tile, row range) task per workgroup and leaves
//     fp32 partial tiles for a fixed-order reduction.  Details at each kernel.
//   gemm_nt_kernel / gemm_tn_kernel: 128 x 128 tiles, 4 waves, operands staged through registers; they take what the large
//     kernels do not (K tails with zero fill, small M / N / K: the heads, unit tests).
// gemm_nt keeps K contiguous in LDS ([row][64], 16-byte chunks XOR-swizzled: conflict-free ds_read_b128 fragments); gemm_tn
// keeps the output dimension contiguous and builds its fragments with the transposing read ds_read_b64_tr_b16.
// The MFMA operands are swapped (first operand = the matrix whose rows become output columns) so that a lane ends up holding
// 4 consecutive output columns of one row.
#include "common.h"
#include <cstdlib>
#include <mutex>

#include "gemm_nt.h"

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmNTParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[GBM * GBK];
    __shared__ __attribute__((aligned(16))) bf16_t sB[GBN * GBK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;

    const int tile = xcd_remap(blockIdx.x, p.num_tiles);
    const int m0 = (tile / p.tiles_n) * GBM;
    const int n0 = (tile % p.tiles_n) * GBN;

    // staging: thread handles 16-byte chunks ci = tid + 256*i ; row = ci/8, chunk = ci%8
    const int srow = tid >> 3, sch = tid & 7;
    const bf16_t* gA[4];
    const bf16_t* gB[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + 32 * i;
        int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
        gA[i] = p.A + (size_t)gm * p.lda + sch * 8;
        gB[i] = p.B + (size_t)gn * p.ldb + sch * 8;
        soff[i] = row * GBK + ((sch ^ (row & 7)) << 3);
    }
    u32x4 ra[4], rb[4];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + GBK - 1) / GBK;
    // loads are unconditional (clamped address) and zeroed by a value select, which keeps ra/rb in registers
    {
        const bool kin = sch * 8 < p.K;
        const int ko = kin ? 0 : -(sch * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 ta = *reinterpret_cast<const u32x4*>(gA[i] + ko);
            const u32x4 tb = *reinterpret_cast<const u32x4*>(gB[i] + ko);
            ra[i] = kin ? ta : zero4;
            rb[i] = kin ? tb : zero4;
        }
    }
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(&sA[soff[i]]) = ra[i];
            *reinterpret_cast<u32x4*>(&sB[soff[i]]) = rb[i];
        }
        __syncthreads();
        if (kt + 1 < nk) {
            const int kofs = (kt + 1) * GBK;
            const bool kin = kofs + sch * 8 < p.K;
            const int ko = kin ? kofs : -(sch * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 ta = *reinterpret_cast<const u32x4*>(gA[i] + ko);
                const u32x4 tb = *reinterpret_cast<const u32x4*>(gB[i] + ko);
                ra[i] = kin ? ta : zero4;
                rb[i] = kin ? tb : zero4;
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
            const int ch = ks * 4 + lg;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rowA = wm * 64 + t * 16 + lr;
                const int rowB = wn * 64 + t * 16 + lr;
                fa[t] = *reinterpret_cast<const bf16x8*>(&sA[rowA * GBK + ((ch ^ (rowA & 7)) << 3)]);
                fb[t] = *reinterpret_cast<const bf16x8*>(&sB[rowB * GBK + ((ch ^ (rowB & 7)) << 3)]);
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn], fa[tm], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: lane holds C[m][n..n+3], m = tile row (lane&15), n = 4*(lane>>4)
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const int m = m0 + wm * 64 + tm * 16 + lr;
        if (m >= p.M) continue;
        float rs = 1.0f;
        int orow = m;
        const float* posrow = nullptr;
        if (EPI == VITK_EPI_BIAS_RES) {
            if (p.rowscale != nullptr) rs = p.rowscale[m / p.rows_per_group];
            if (p.smap != nullptr) { const int c = m / p.rps; orow = p.smap[c] * p.rps + (m - c * p.rps); }
        }
        if (EPI == VITK_EPI_PATCH) {
            const int b = m / p.patches, pi = m - b * p.patches;
            orow = b * (p.patches + p.prefix) + p.prefix + pi;
            posrow = p.pos + (size_t)(p.prefix + pi) * p.N;
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int n = n0 + wn * 64 + tn * 16 + 4 * lg;
            if (n >= p.N) continue;
            float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
            nt_epilogue<EPI, 4>(p, v, m, n, orow, rs, posrow);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// gemm_nt, large-tile persistent variant.  One 512-thread workgroup per CU walks over 256x256 output tiles; 8 waves
// (2 x 4), each a 128x64 sub-tile = 8x4 MFMA tiles = 128 accumulator registers; BK = 64.
//  * Operands go global -> LDS directly (LDS-DMA, 16 B per lane, no register stop) into a double-buffered 128 KB image.
//    The DMA writes LDS lane-linearly (wave base + 16*lane), so the bank-conflict XOR swizzle is applied to the per-lane
//    SOURCE chunk and again on the fragment read.
//  * The K-steps of all tiles of a workgroup form ONE stream: K-step s+2 is requested in the middle of step s, also across
//    a tile boundary, so a tile's first operands are already in LDS when its first MFMA issues and the previous tile's
//    output stores drain in the background (no prologue, no store tail between tiles).
//  * Fragments are read from LDS one phase (16 MFMAs) before the MFMAs that use them.
// Needs K % 64 == 0 and K >= 128 (no zero-fill on this path).
#define NTP_DRAIN 16                 // output stores a wave issues per full tile, at least (16 row groups, one each)
#define NTP_LDS (4 * LTILE * 2 + 8 * 16 * 64 * 4)   // bytes: operand image + 4 KB of epilogue staging per wave

// GMAP (bias + residual epilogue only): the stochastic-depth tables are looked up per ROW inside the store loop - for samples
// of fewer than 128 rows, or tables with different group sizes.  It is an instantiation of its own because a load under a
// run-time condition inside that loop makes hipcc wait for EVERY outstanding memory operation (s_waitcnt vmcnt(0)) where the
// two paths join, also on the path that loaded nothing: 75 such waits per tile instead of 9 in the common instantiation
// (2-4 % of a bias + residual launch).
template <int EPI, bool GMAP = false>
__global__ __launch_bounds__(512) void gemm_ntp_kernel(const GemmNTParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lsm[];   // [2 buffers][A tile | B tile], [8 waves][16][64] fp32

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;
    float* stg = reinterpret_cast<float*>(lsm + 4 * LTILE) + wave * (16 * 64);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr(lsm));

    // staging: wave w, piece j covers rows (w*4+j)*8 .. +7 of a tile; lane -> row += lane/8, LDS position lane%8, which
    // receives source chunk (lane%8) ^ (row&7); row&7 == lane/8 for every piece
    const int srow = lane >> 3, sch = (lane & 7) ^ srow;
    const unsigned lds_stage = (unsigned)(wave * 4 * 8 * GBK * sizeof(bf16_t));
    const bf16_t* sA = p.A;          // staging state: operand panels of the tile being requested
    const bf16_t* sB = p.B;
    unsigned voA[4], voB[4];
    auto aim = [&](int v, int& m0, int& n0) {
        // tile order: blocks of cb column tiles, row-major inside a block (the last block may be narrower)
        const int tile = xcd_remap(v, p.num_tiles);
        const int full = p.tiles_m * p.cb;
        const int blk = tile / full;
        const int rem = tile - blk * full;
        const int c0 = blk * p.cb;
        const int width = p.tiles_n - c0 < p.cb ? p.tiles_n - c0 : p.cb;
        const int tm_ = rem / width;
        m0 = tm_ * LBM;
        n0 = (c0 + rem - tm_ * width) * LBN;
        sA = p.A + (size_t)m0 * p.lda;
        sB = p.B + (size_t)n0 * p.ldb;
        const int ra_max = p.M - 1 - m0, rb_max = p.N - 1 - n0;       // rows past the edge re-read the last valid row
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (wave * 4 + j) * 8 + srow;
            const int ra = row < ra_max ? row : ra_max, rb = row < rb_max ? row : rb_max;
            voA[j] = (unsigned)((ra * p.lda + sch * 8) * (int)sizeof(bf16_t));
            voB[j] = (unsigned)((rb * p.ldb + sch * 8) * (int)sizeof(bf16_t));
        }
    };
    // K-step kstep of the staging tile -> LDS buffer par (asynchronous), in two halves: the 8 DMA instructions of a step
    // are issued 4 + 4, a phase apart (the activation panel first: it comes from HBM, the weight panel from L2), because a burst
    // of 8 holds up the issuing wave for most of a phase (4 + 2 + 2 over three phases measured no better)
    auto stageA = [&](int kstep, int par) {
        const unsigned dA = lds_base + (unsigned)(par * 2 * LTILE * sizeof(bf16_t)) + lds_stage;
        const bf16_t* a = sA + kstep * GBK;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(a, voA[j], dA + j * 8 * GBK * (unsigned)sizeof(bf16_t));
    };
    auto stageB = [&](int kstep, int par) {
        const unsigned dB = lds_base + (unsigned)((par * 2 + 1) * LTILE * sizeof(bf16_t)) + lds_stage;
        const bf16_t* b = sB + kstep * GBK;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(b, voB[j], dB + j * 8 * GBK * (unsigned)sizeof(bf16_t));
    };

    // fragment read offsets (elements) inside a tile: row*64 + ((chunk ^ (row&7)) << 3); row&7 == lr&7 for every tile row
    int offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int x = (((ks * 4 + lg) ^ (lr & 7)) << 3);
        offA[ks] = (wm * 128 + lr) * GBK + x;
        offB[ks] = (wn * 64 + lr) * GBK + x;
    }
    f32x4 acc[8][4];
    bf16x8 faE[4], faO[4], fbE[4], fbO[4];
    auto ldA = [&](bf16x8 (&f)[4], const bf16_t* cA, int ks, int h) {
#pragma unroll
        for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const bf16x8*>(cA + offA[ks] + (4 * h + t) * 16 * GBK);
    };
    auto ldB = [&](bf16x8 (&f)[4], const bf16_t* cB, int ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const bf16x8*>(cB + offB[ks] + t * 16 * GBK);
    };
    auto mma = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4], int h) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[4 * h + tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn], fa[tm], acc[4 * h + tm][tn], 0, 0, 0);
    };

    const int nk = p.K / GBK;
    const int G = gridDim.x;
    // Tiles are dealt dynamically: a workgroup's first tile is its block index, every further one is claimed from its XCD's
    // counter (claim c of XCD x is tile index G + 8 c + x, so an XCD keeps its contiguous share of the tile order).  A workgroup
    // that reaches its compute unit late - another stream's kernel or a collective was holding it - finds the tiles already taken by
    // the others instead of delaying the launch by its whole static share.  The claim for the tile after next leaves one tile
    // ahead (wave 0, before the epilogue's stores) and reaches the other waves through LDS at the next tile's first barrier.
    const int xcd = blockIdx.x & 7;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    lds_u32* mailbox = (lds_u32*)(__UINTPTR_TYPE__)(lds_base + (unsigned)(4 * LTILE * sizeof(bf16_t)));   // first word of wave 0's epilogue staging
    unsigned claim = 0;
    auto issue_claim = [&](unsigned inc) {
        if (wave == 0 && lane == 0) claim = atomicAdd(&p.sched[xcd], inc);
    };
    bool entry = true;      // the claim in flight is the entry atomic: ticket + first claim, checked where it is first used
    int v = blockIdx.x, vn = 0;
    int m0, n0, nm0 = 0, nn0 = 0;
    issue_claim(NTP_ARRIVE + 1u);
    aim(v, m0, n0);
    stageA(0, 0);
    stageB(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stageA(1, 1);
    stageB(1, 1);
    ldB(fbE, lsm + LTILE, 0);
    ldA(faE, lsm, 0, 0);
    int par = 0;            // LDS buffer of the current K-step
    int pend_k = -1, pend_par = 0;      // second half (B operand) of a request whose first half has been issued
    bool loose = false;     // the last VMEM operations of this wave are >= NTP_DRAIN output stores younger than every request

    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bool has_next = false;       // known from the first barrier of the tile on

        for (int kt = 0; kt < nk; ++kt) {
            const bf16_t* cA = lsm + par * 2 * LTILE;
            const bf16_t* cB = cA + LTILE;
            const bf16_t* nA = lsm + (par ^ 1) * 2 * LTILE;
            const bf16_t* nB = nA + LTILE;
            // Each phase: 16 MFMAs on fragments fetched a phase ago, interleaved (MFMAs first) with the reads for the next
            // phase, N-side fragments first because the next phase's first four MFMAs need all of them.
            // phase (0,0): fetch A rows 64..127 of ks 0
            if (pend_k >= 0) { stageB(pend_k, pend_par); pend_k = -1; }
            mma(faE, fbE, 0);
            ldA(faO, cA, 0, 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            // phase (0,1): fetch ks 1's N side and A rows 0..63
            mma(faO, fbE, 1);
            ldB(fbO, cB, 1);
            ldA(faE, cA, 1, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            // phase (1,0)
            mma(faE, fbO, 0);
            ldA(faO, cA, 1, 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            // Every wave has now read all of the current buffer; the next K-step, requested a full step ago, must have
            // landed: publish it, then refill the current buffer with the K-step after that.  vmcnt counts in issue order,
            // so in a tile's first step the previous tile's output stores (all younger than the request) may stay pending.
            if (kt == 0 && loose) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NTP_DRAIN) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (kt == 0) {
                if (wave == 0 && lane == 0) {
                    if (entry) ntp_check_entry(p, claim, xcd, G);
                    *mailbox = (unsigned)G + 8u * (claim & NTP_CLAIM_MASK) + (unsigned)xcd;
                }
                entry = false;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt == 0) {
                vn = __builtin_amdgcn_readfirstlane((int)*mailbox);
                has_next = (unsigned)vn < (unsigned)p.num_tiles;
            }
            if (kt + 2 < nk) {
                stageA(kt + 2, par);
                pend_k = kt + 2; pend_par = par;
            } else if (has_next) {
                if (kt + 2 == nk) aim(vn, nm0, nn0);
                stageA(kt + 2 - nk, par);
                pend_k = kt + 2 - nk; pend_par = par;
            }
            // phase (1,1): fetch the first fragments of the next K-step (after the last tile: stale LDS, never used)
            mma(faO, fbO, 1);
            ldB(fbE, nB, 0);
            ldA(faE, nA, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            par ^= 1;
        }

        if (pend_k >= 0) { stageB(pend_k, pend_par); pend_k = -1; }      // before the stores: vmcnt counts in issue order
        if (has_next) issue_claim(1u);                                    // likewise: the claim is older than every store
        // Epilogue.  The MFMA layout (lane = row lr, 4 columns) would store 32-byte pieces of 16 rows per instruction, and
        // the store path is paid per lane-access, not per byte.  Transpose through this wave's private 4 KB of LDS instead
        // (fp32, so the arithmetic is unchanged): 16 rows x 64 columns at a time, read back as lane = (row lane/8, 8
        // columns), so every global access is 16 bytes per lane and 8 lanes cover one full 128-byte line.
        // 16-byte chunk c of staging row r sits at chunk c ^ (r & 7): conflict-free both ways.
        // The side operand (residual / pre-activation) may be the output buffer itself, so hipcc keeps every load behind
        // the previous store: fetch it by hand, 8 pieces (64 rows) ahead, to keep HBM latency out of the store loop.
        constexpr bool SIDE = EPI == VITK_EPI_BIAS_RES || EPI == VITK_EPI_DGELU || EPI == VITK_EPI_MUL;
        const int ecol = n0 + wn * 64 + (lane & 7) * 8;
        // Stochastic depth (BIAS_RES): GEMM row m belongs to kept sample m / rps, whose rows live at sample smap[.] of the residual
        // stream and are scaled by rowscale[.].  A wave's 128 rows span at most two samples when a sample has >= 128 rows: the two
        // samples' row offsets and scales are looked up ONCE per tile (wave-uniform), and a row picks between them with one
        // compare - instead of two integer divisions and two dependent loads in front of every side fetch and every store.
        constexpr bool fastmap = EPI == VITK_EPI_BIAS_RES && !GMAP;      // without tables: one "sample" that never ends, offset 0, scale 1
        int map_bnd = 0x7fffffff, map_off0 = 0, map_off1 = 0;
        float map_rs0 = 1.0f, map_rs1 = 1.0f;
        if (EPI == VITK_EPI_BIAS_RES && (p.smap != nullptr || p.rowscale != nullptr)) {
            const int per = p.smap != nullptr ? p.rps : p.rows_per_group;
            if (fastmap) {
                int mw = __builtin_amdgcn_readfirstlane(m0) + wm * 128;
                mw = mw < p.M ? mw : p.M - 1;
                const int c0 = mw / per;
                map_bnd = (c0 + 1) * per;
                const int c1 = map_bnd < p.M ? c0 + 1 : c0;
                // The tables are read through the scalar cache (constant address space, wave-uniform index): a vector load here
                // would be waited for with vmcnt, which counts in issue order - i.e. together with the next tile's operand
                // requests issued a moment ago, a full memory round trip in front of every epilogue.  Nothing in this kernel
                // writes the tables.
                typedef const __attribute__((address_space(4))) int* ktab_i;
                typedef const __attribute__((address_space(4))) float* ktab_f;
                if (p.smap != nullptr) {
                    const ktab_i sm = (ktab_i)p.smap;
                    map_off0 = (sm[c0] - c0) * per;
                    map_off1 = (sm[c1] - (c0 + 1)) * per;
                }
                if (p.rowscale != nullptr) {
                    const ktab_f rsc = (ktab_f)p.rowscale;
                    map_rs0 = rsc[c0];
                    map_rs1 = rsc[c1];
                }
            }
        }
        u32x4 side[SIDE ? 8 : 1];
        // the two pieces (16 rows) of row group tm, into the ring slots of tm & 3
        auto fetch_side = [&](int tm) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int m = m0 + wm * 128 + tm * 16 + i * 8 + srow;
                m = m < p.M ? m : p.M - 1;
                if (EPI == VITK_EPI_BIAS_RES) {
                    if (fastmap) m += m >= map_bnd ? map_off1 : map_off0;
                    else if (p.smap != nullptr) { const int c = m / p.rps; m = p.smap[c] * p.rps + (m - c * p.rps); }
                }
                const int n = ecol < p.N ? ecol : 0;
                side[SIDE ? (tm & 3) * 2 + i : 0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(nt_side_ptr<EPI>(p, m, n)));
            }
        };
        // the bias depends on the lane's columns only: one fetch per tile (behind the stores hipcc would re-load it 16 times)
        constexpr bool BIAS = EPI != VITK_EPI_PLAIN && EPI != VITK_EPI_DGELU && EPI != VITK_EPI_MUL;
        float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (SIDE) { fetch_side(0); fetch_side(1); fetch_side(2); fetch_side(3); }
        if (BIAS) add_f32<8>(p.bias + (ecol < p.N ? ecol : 0), bias8);
#pragma unroll
        for (int tm = 0; tm < 8; ++tm) {
            float sv[2][8];
            if (SIDE) {
                unpack_bf16x8(side[(tm & 3) * 2], sv[0]);
                unpack_bf16x8(side[(tm & 3) * 2 + 1], sv[1]);
                if (tm == 3) { asm volatile("" ::: "memory"); fetch_side(4); fetch_side(5); fetch_side(6); fetch_side(7); }
            }
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                *reinterpret_cast<f32x4*>(stg + lr * 64 + (((tn * 4 + lg) ^ (lr & 7)) << 2)) = acc[tm][tn];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int r = half * 8 + srow;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + r * 64 + (((2 * (lane & 7)) ^ srow) << 2));
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + r * 64 + (((2 * (lane & 7) + 1) ^ srow) << 2));
                const int m = m0 + wm * 128 + tm * 16 + r;
                const int n = ecol;
                if (m < p.M && n < p.N) {
                    float rs = 1.0f;
                    int orow = m;
                    const float* posrow = nullptr;
                    if (EPI == VITK_EPI_BIAS_RES) {
                        if (fastmap) {
                            const bool nx = m >= map_bnd;
                            rs = nx ? map_rs1 : map_rs0;
                            orow = m + (nx ? map_off1 : map_off0);
                        } else {
                            if (p.rowscale != nullptr) rs = p.rowscale[m / p.rows_per_group];
                            if (p.smap != nullptr) { const int c = m / p.rps; orow = p.smap[c] * p.rps + (m - c * p.rps); }
                        }
                    }
                    if (EPI == VITK_EPI_PATCH) {
                        const int b = m / p.patches, pi = m - b * p.patches;
                        orow = b * (p.patches + p.prefix) + p.prefix + pi;
                        posrow = p.pos + (size_t)(p.prefix + pi) * p.N;
                    }
                    float o[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    nt_epilogue<EPI, 8>(p, o, m, n, orow, rs, posrow, SIDE ? sv[half] : nullptr, BIAS ? bias8 : nullptr);
                }
            }
        }
        if (!has_next) break;
        loose = m0 + LBM <= p.M && n0 + LBN <= p.N;      // a full tile: every lane stored all 16 pieces
        v = vn;
        m0 = nm0;
        n0 = nn0;
    }
    // the last workgroup to leave zeroes the counters for the next launch that uses them
    if (wave == 0 && lane == 0) {
        const unsigned left = atomicAdd(&p.sched[8], 1u);
        if (left == (unsigned)G - 1u) {
#pragma unroll
            for (int i = 0; i < 9; ++i) p.sched[i] = 0u;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
struct GemmTNParams {
    const bf16_t* A; int lda;
    const bf16_t* B; int ldb;
    float* slab;        // [splits][N][K]
    float* colslab;     // [splits][N] or null
    int M, N, K;
    int m_per_split;
    int tiles, splits_per_xcd;
    int tiles_k;
};

#define TBM 64   // contraction rows staged per step

// gemm_tn LDS image: [64 m][128 cols] bf16, 256-byte rows = one LDS bank row.  A ds_read_b64_tr_b16 is serviced per
// 32-lane half = 2 lane groups x 4 rows (m = 8g+q) x 4 eight-byte pieces; unswizzled, the 8 rows hit the same 32 bytes
// (8-way conflict).  XOR the 16-byte chunk index with 2*(4*((m>>3)&1) + (m&3)): the 8 rows of a half land on 8 disjoint
// chunk pairs -> conflict-free.
static __device__ __forceinline__ int tn_swz(int m) { return (((m >> 3) & 1) << 3) | ((m & 3) << 1); }

static __device__ __forceinline__ bf16x8 tr_frag(const bf16_t* lo, const bf16_t* hi) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)lo);
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)hi);
    return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

template <bool COLSUM>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTNParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[TBM * GBN];   // [m][n]
    __shared__ __attribute__((aligned(16))) bf16_t sB[TBM * GBN];   // [m][k]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;    // wave's 64 n-rows / 64 k-cols of the output tile
    const int lr = lane & 15, lg = lane >> 4;
    const int lq = lr >> 2, lp = lr & 3;

    // Blocks are dealt round-robin over the 8 XCDs.  All blocks of one XCD walk the SAME contraction (m) range at about
    // the same time, tile after tile, so each activation row is fetched from HBM into one L2 only (instead of into all 8):
    // xcd = bid % 8 owns splits [xcd*spx, (xcd+1)*spx); its blocks take (split, tile) pairs split-major, tile n-major.
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile_id = local % p.tiles;
    const int split = xcd * p.splits_per_xcd + local / p.tiles;
    const int n0 = (tile_id / p.tiles_k) * GBN;
    const int k0 = (tile_id % p.tiles_k) * GBN;
    const int ms = split * p.m_per_split;
    int me = ms + p.m_per_split; me = me < p.M ? me : p.M;

    // staging: chunk ci = tid + 256*i ; row = ci/16 (0..63), chunk = ci%16 (8 columns each)
    const int srow = tid >> 4, sch = tid & 15;
    const bool a_col_ok = n0 + sch * 8 < p.N;
    const bool b_col_ok = k0 + sch * 8 < p.K;
    const bf16_t* gA = p.A + (size_t)(a_col_ok ? n0 + sch * 8 : 0);
    const bf16_t* gB = p.B + (size_t)(b_col_ok ? k0 + sch * 8 : 0);
    u32x4 ra[4], rb[4];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    float colacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) colacc[j] = 0.f;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool do_colsum = COLSUM && (tile_id % p.tiles_k) == 0;

    // Loads are unconditional, from clamped (always valid) addresses, and nothing consumes a loaded register before the
    // next step's LDS store: a "valid ? load : 0" select makes hipcc branch around every load and wait for each one in
    // turn, and an AND placed beside the load makes it wait right there.  Validity travels as one mask word per load
    // (column in range && row in range), applied at the LDS store.
    const unsigned cmA = a_col_ok ? ~0u : 0u, cmB = b_col_ok ? ~0u : 0u;
    unsigned mkA[4], mkB[4];
    auto gload = [&](int mbase) {
        const bool full = mbase + TBM <= me;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = mbase + srow + 16 * i;
            const bool rok = full || gm < me;
            const int gmc = rok ? gm : me - 1;
            ra[i] = *reinterpret_cast<const u32x4*>(gA + (size_t)gmc * p.lda);
            rb[i] = *reinterpret_cast<const u32x4*>(gB + (size_t)gmc * p.ldb);
            mkA[i] = rok ? cmA : 0u;
            mkB[i] = rok ? cmB : 0u;
        }
    };

#pragma unroll
    for (int i = 0; i < 4; ++i) { ra[i] = zero4; rb[i] = zero4; mkA[i] = 0u; mkB[i] = 0u; }
    if (ms < me) gload(ms);
    for (int mb = ms; mb < me; mb += TBM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 16 * i;
            const int so = row * GBN + ((sch ^ tn_swz(row)) << 3);
            const u32x4 va = ra[i] & (u32x4){mkA[i], mkA[i], mkA[i], mkA[i]};
            const u32x4 vb = rb[i] & (u32x4){mkB[i], mkB[i], mkB[i], mkB[i]};
            *reinterpret_cast<u32x4*>(&sA[so]) = va;
            *reinterpret_cast<u32x4*>(&sB[so]) = vb;
            if (do_colsum) {
                const u32x4 r = va;
                colacc[0] += __uint_as_float(r.x << 16); colacc[1] += __uint_as_float(r.x & 0xffff0000u);
                colacc[2] += __uint_as_float(r.y << 16); colacc[3] += __uint_as_float(r.y & 0xffff0000u);
                colacc[4] += __uint_as_float(r.z << 16); colacc[5] += __uint_as_float(r.z & 0xffff0000u);
                colacc[6] += __uint_as_float(r.w << 16); colacc[7] += __uint_as_float(r.w & 0xffff0000u);
            }
        }
        __syncthreads();
        if (mb + TBM < me) gload(mb + TBM);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
            const int r0 = ks * 32 + 8 * lg + lq;   // rows of the first 4-row block; second block is +4
            const int sw = tn_swz(r0);               // tn_swz(r0 + 4) == tn_swz(r0)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // 16-byte chunk index of the lane's 4 columns, swizzled; (lp & 1) picks the 8-byte half of the chunk
                const int ca = ((((wr * 64 + t * 16) >> 3) + (lp >> 1)) ^ sw) * 8 + (lp & 1) * 4;
                const int cb = ((((wc * 64 + t * 16) >> 3) + (lp >> 1)) ^ sw) * 8 + (lp & 1) * 4;
                fa[t] = tr_frag(&sA[r0 * GBN + ca], &sA[(r0 + 4) * GBN + ca]);
                fb[t] = tr_frag(&sB[r0 * GBN + cb], &sB[(r0 + 4) * GBN + cb]);
            }
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tk = 0; tk < 4; ++tk)
                    acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tk], fa[tn], acc[tn][tk], 0, 0, 0);
        }
        __syncthreads();
    }

    float* slab = p.slab + (size_t)split * p.N * p.K;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int n = n0 + wr * 64 + tn * 16 + lr;
        if (n >= p.N) continue;
#pragma unroll
        for (int tk = 0; tk < 4; ++tk) {
            const int k = k0 + wc * 64 + tk * 16 + 4 * lg;
            if (k >= p.K) continue;
            *reinterpret_cast<float4*>(slab + (size_t)n * p.K + k) =
                make_float4(acc[tn][tk][0], acc[tn][tk][1], acc[tn][tk][2], acc[tn][tk][3]);
        }
    }
    if (do_colsum) {
        // fold the 16 row groups (tid>>4) that share a column chunk; reuse sA as fp32 scratch [16][128]
        float* red = reinterpret_cast<float*>(sA);
#pragma unroll
        for (int j = 0; j < 8; ++j) red[srow * GBN + sch * 8 + j] = colacc[j];
        __syncthreads();
        if (tid < GBN && n0 + tid < p.N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * GBN + tid];
            p.colslab[(size_t)split * p.N + n0 + tid] = s;
        }
    }
}

// out[r*ldo + c] = (accumulate ? out : 0) + sum_s slab[s][r][c], fixed order; cols % 4 == 0
__global__ void reduce_slabs_kernel(const float* __restrict__ slab, int splits, int rows, int cols, float* __restrict__ out,
                                    int ldo, int accumulate) {
    const size_t total4 = (size_t)rows * cols / 4;
    const size_t stride = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        float4 s = *reinterpret_cast<const float4*>(slab + e);
        for (int k = 1; k < splits; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(slab + k * stride + e);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const int r = (int)(e / cols), c = (int)(e % cols);
        float4* dst = reinterpret_cast<float4*>(out + (size_t)r * ldo + c);
        if (accumulate) { const float4 o = *dst; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        *dst = s;
    }
}
__global__ void reduce_vec_kernel(const float* __restrict__ slab, int splits, int n, float* __restrict__ out, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += slab[(size_t)k * n + i];
    out[i] = accumulate ? out[i] + s : s;
}

static int tn_splits(int M, int N, int K) {
    // a multiple of 8 (one group of splits per XCD); enough (split, tile) pairs per XCD to fill its 32 CUs several
    // times over, but at least 512 contraction rows per split and at most 4 splits per XCD (slab traffic)
    const int tiles = ((N + GBN - 1) / GBN) * ((K + GBN - 1) / GBN);
    int spx = (256 + tiles - 1) / tiles;
    const int max_by_rows = (M / 8) / 512;
    if (spx > max_by_rows) spx = max_by_rows;
    if (spx > 4) spx = 4;
    if (spx < 1) spx = 1;
    return 8 * spx;
}


// ------------------------------------------------------------------------------------------------------------
// gemm_tn, large-tile variant for the weight gradients of the big linears.  The contraction runs over the M token rows, which
// are the slow dimension of both operands, so there is plenty of M to split: the work is T output tiles (256 x 256) times R
// row ranges with T * R just under the CU count, ONE task per workgroup and no second round (no tail), each task one long
// pipelined loop of 64-row steps.  Tasks are numbered range-major and an XCD takes a contiguous block of them, so the
// workgroups that share an L2 read (almost) the same rows.  Partial tiles go to fp32 slabs [task][256][256], summed in a
// fixed order by reduce_tiles_kernel.
//  * 8 waves (2 x 4), each 128 (n) x 64 (k) = 8 x 4 MFMA tiles; operands global -> LDS by DMA into a double-buffered
//    128 KB image [m][256 columns]; fragments come from the transposing read (ds_read_b64_tr_b16), one phase ahead.
//  * rows past M (last step of the last range) are fetched from a zero row, so they add nothing.
//  * the bias gradient (column sums of A) rides along as MFMAs against an all-ones fragment, shared out over the waves of the
//    tasks that read the same A columns.
struct GemmTNBigParams {
    const bf16_t* A; int lda;
    const bf16_t* B; int ldb;
    float* slab;        // [R * T][256][256]
    float* colslab;     // [R][N] or null
    int M, N, K;
    int T, tiles_k, R, steps_total, ntasks;
};

__device__ __attribute__((aligned(512))) unsigned vitk_zero_row[128];     // 512 bytes of zeros (never written)


#define TNB_TILE (64 * 256)      // elements per operand tile: 64 rows x 256 columns

template <bool COLSUM>
__global__ __launch_bounds__(512) void gemm_tnb_kernel(const GemmTNBigParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lsm[];   // [2 buffers][A tile | B tile]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;
    const int lq = lr >> 2, lp = lr & 3;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr(lsm));

    // Execution order (an XCD takes a contiguous stretch of it): range-major; inside a range the tile index of the NARROWER
    // operand runs fastest, so the tasks that share a column panel of the wider one (most of the bytes) are neighbours and sit
    // on one XCD.  Tasks keep their number range * T + n tile * tiles_k + k tile for the slabs.
    const int pos = xcd_remap(blockIdx.x, p.ntasks);
    const int range = pos / p.T, idx = pos - range * p.T;
    const int tiles_n = p.T / p.tiles_k;
    const int tn_ = p.tiles_k > tiles_n ? idx % tiles_n : idx / p.tiles_k;
    const int tile = p.tiles_k > tiles_n ? tn_ * p.tiles_k + idx / tiles_n : idx;
    const int task = range * p.T + tile;
    const int n0 = tn_ * 256, k0 = (tile - tn_ * p.tiles_k) * 256;
    const int step0 = (int)(((long long)range * p.steps_total) / p.R);
    const int nsteps = (int)(((long long)(range + 1) * p.steps_total) / p.R) - step0;      // >= 2 (host)
    // Bias gradient = column sums of A over this range's rows, as MFMAs against an all-ones fragment.  The 8 row tiles (16
    // columns of A each) of a wave row wm are shared out over the 8 waves of the first TWO k-tile tasks of an n tile (tiles_k >= 2,
    // host): wave wn of task k-tile j sums row tile 4j + wn.  A wave walks its row tiles in a permuted order - tile index XOR
    // (4j + wn), folded into the fragment address - so the tile it sums is always fragment 0 of the first half: one copy of
    // the loop with two extra MFMAs per step, one without.  (All of it in the k-tile-0 tasks - four extra MFMAs per step -
    // made those the stragglers of every launch: +6 %.)
    const int ktile = tile - tn_ * p.tiles_k;
    const bool cs_role = COLSUM && ktile < 2;
    // Where the k tile runs fastest, the tasks that share a column panel of A are the k tiles of one n tile - two of them with
    // the sums, the rest without, i.e. 3 % faster per step: they drift apart until the panel has left L2 between the first and
    // the last reader.  The rest run the sums too (results dropped): a launch ends with its slowest task anyway.
    const bool cs_pace = COLSUM && (cs_role || p.tiles_k <= tiles_n);
    const int tperm = COLSUM ? (((ktile & 1) << 2) | wn) : 0;       // accumulator tile tm holds row tile tm ^ tperm

    // staging: a piece (one wave-instruction, 1 KB) is 2 rows x 512 B; wave w, piece j -> local rows (4w+j)*2 + lane/32;
    // LDS position lane%32 of row m receives source chunk (lane%32) ^ tn_swz(m)
    const int shalf = lane >> 5, spos = lane & 31;
    unsigned voA[4], voB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lrow = (wave * 4 + j) * 2 + shalf;
        const int ch = spos ^ tn_swz(lrow);
        int ca = n0 + ch * 8; ca = ca < p.N ? ca : p.N - 8;         // columns past the edge only feed outputs nobody reads
        int cb = k0 + ch * 8; cb = cb < p.K ? cb : p.K - 8;
        voA[j] = (unsigned)((lrow * p.lda + ca) * (int)sizeof(bf16_t));
        voB[j] = (unsigned)((lrow * p.ldb + cb) * (int)sizeof(bf16_t));
    }
    const unsigned lds_stage = (unsigned)(wave * 4 * 1024);
    // 64-row step st of this range -> LDS buffer par (asynchronous); which = 0: the A half, 1: the B half (issued a phase
    // apart: a burst of 8 DMA instructions holds up the issuing wave for most of a phase)
    auto stage = [&](int st, int par, int which) {
        const unsigned dst = lds_base + (unsigned)((par * 2 + which) * TNB_TILE * sizeof(bf16_t)) + lds_stage;
        const int mrow = (step0 + st) * 64;
        const bf16_t* src = which ? p.B + (size_t)mrow * p.ldb : p.A + (size_t)mrow * p.lda;
        if (mrow + 64 <= p.M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) glds16(src, which ? voB[j] : voA[j], dst + j * 1024);
        } else {       // ragged last step: rows >= M come from the zero row
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = mrow + (wave * 4 + j) * 2 + shalf < p.M;
                const void* z = (const char*)vitk_zero_row + spos * 16;
                glds16_v(ok ? (const void*)((const char*)src + (which ? voB[j] : voA[j])) : z, dst + j * 1024);
            }
        }
    };

    // fragment addresses: rows r0 = ks*32 + 8*lg + lq (and +4), tn_swz(r0) is the same for both ks; the lane's 4 columns of
    // MFMA tile t are half (lp&1) of chunk (tile base chunk | lp>>1), XOR-swizzled
    const int sw = tn_swz(8 * lg + lq);
    const int preA = ((wm * 16) | (lp >> 1)) ^ sw ^ (tperm << 1), preB = ((wn * 8) | (lp >> 1)) ^ sw;
    const int rbase = (8 * lg + lq) * 256 + (lp & 1) * 4;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 cs = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    bf16x8 faE[4], faO[4], fbE[4], fbO[4];
    auto ldA = [&](bf16x8 (&f)[4], const bf16_t* cA, int ks, int h) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16_t* q = cA + ks * 32 * 256 + rbase + ((preA ^ (2 * (4 * h + t))) << 3);
            f[t] = tr_frag(q, q + 4 * 256);
        }
    };
    auto ldB = [&](bf16x8 (&f)[4], const bf16_t* cB, int ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16_t* q = cB + ks * 32 * 256 + rbase + ((preB ^ (2 * t)) << 3);
            f[t] = tr_frag(q, q + 4 * 256);
        }
    };
    // CS = 1: this wave sums the columns of its fragment 0.  CS and h are compile-time so that the pipelined loop has no branch
    // in it: with a run-time test around the extra MFMAs hipcc lost the interleaving of the phases.
    auto mma = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4], auto htag, auto cstag) {
        constexpr int h = decltype(htag)::value;
        constexpr int CS = decltype(cstag)::value;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[4 * h + tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn], fa[tm], acc[4 * h + tm][tn], 0, 0, 0);
        if constexpr (CS > 0 && h == 0) cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[0], cs, 0, 0, 0);
    };

    stage(0, 0, 0);
    stage(0, 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage(1, 1, 0);
    stage(1, 1, 1);
    ldB(fbE, lsm + TNB_TILE, 0);
    ldA(faE, lsm, 0, 0);

    auto run = [&](auto cstag) {
    for (int st = 0; st < nsteps; ++st) {
        const int par = st & 1;
        const bf16_t* cA = lsm + par * 2 * TNB_TILE;
        const bf16_t* cB = cA + TNB_TILE;
        const bf16_t* nA = lsm + (par ^ 1) * 2 * TNB_TILE;
        const bf16_t* nB = nA + TNB_TILE;
        // same four-phase pipeline as the persistent gemm_nt kernel: 16 MFMAs on fragments fetched a phase ago, interleaved
        // with the transposing reads for the next phase
        if (st >= 1 && st + 1 < nsteps) stage(st + 1, par ^ 1, 1);      // second half of the request made in the previous step
        mma(faE, fbE, itag<0>{}, cstag);
        ldA(faO, cA, 0, 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        mma(faO, fbE, itag<1>{}, cstag);
        ldB(fbO, cB, 1);
        ldA(faE, cA, 1, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        mma(faE, fbO, itag<0>{}, cstag);
        ldA(faO, cA, 1, 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 2 < nsteps) stage(st + 2, par, 0);
        mma(faO, fbO, itag<1>{}, cstag);
        ldB(fbE, nB, 0);
        ldA(faE, nA, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
    }
    };
    // one copy of the loop with the column sums, one without (wave-uniform choice; both meet the same barriers)
    if (cs_pace) run(itag<1>{});
    else run(itag<0>{});

    // partial tile -> slab (streaming): lane = row lr of MFMA tile tm, 4 consecutive k
    float* slab = p.slab + (size_t)task * (256 * 256);
#pragma unroll
    for (int tm = 0; tm < 8; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
            __builtin_nontemporal_store(acc[tm][tn], reinterpret_cast<f32x4*>(slab + (wm * 128 + (tm ^ tperm) * 16 + lr) * 256 + wn * 64 + tn * 16 + 4 * lg));
    if (cs_role && lg == 0) {
        const int n = n0 + wm * 128 + tperm * 16 + lr;
        if (n < p.N) p.colslab[(size_t)range * p.N + n] = cs[0];
    }
}

// out[n*ldo + k] = (accumulate ? out : 0) + sum over the R row ranges of the tile slabs, fixed order; K % 4 == 0.
// The blocks past tile_blocks sum the bias-gradient partials colslab [R][N] the same way (one launch instead of two).
__global__ void reduce_tiles_kernel(const float* __restrict__ slab, int R, int T, int tiles_k, int N, int K, float* __restrict__ out,
                                    int ldo, int accumulate, int tile_blocks, const float* __restrict__ colslab,
                                    float* __restrict__ dbias) {
    if ((int)blockIdx.x >= tile_blocks) {
        const int i = ((int)blockIdx.x - tile_blocks) * blockDim.x + threadIdx.x;
        if (i < N) {
            float sum = 0.f;
            for (int r = 0; r < R; ++r) sum += colslab[(size_t)r * N + i];
            dbias[i] = accumulate ? dbias[i] + sum : sum;
        }
        return;
    }
    const size_t total4 = (size_t)N * K / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)tile_blocks * blockDim.x) {
        const size_t e = i * 4;
        const int n = (int)(e / K), k = (int)(e % K);
        const int tile = (n >> 8) * tiles_k + (k >> 8);
        const float* src = slab + (size_t)tile * 65536 + (n & 255) * 256 + (k & 255);
        f32x4 s = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
        for (int r = 1; r < R; ++r) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)r * T * 65536));
        f32x4* dst = reinterpret_cast<f32x4*>(out + (size_t)n * ldo + k);
        if (accumulate) s += *dst;
        *dst = s;
    }
}

// plan of the large-tile variant: false -> use the 128x128 kernel
struct TnPlan { bool big; int T, tiles_k, R, steps_total; };
static TnPlan tn_plan(int M, int N, int K) {
    TnPlan q = {false, 0, 0, 0, 0};
    static const bool force_small = getenv("VITK_GEMM_SMALL") != nullptr;
    if (force_small || M < 8192 || N < 256 || K <= 256) return q;      // two k tiles at least: they share the bias gradient
    q.tiles_k = (K + 255) / 256;
    q.T = ((N + 255) / 256) * q.tiles_k;
    q.steps_total = (M + 63) / 64;
    static const char* cap_s = getenv("VITK_GEMM_CUS");      // experiments: workgroups a GEMM launch may use
    static const int cap = cap_s && atoi(cap_s) > 0 ? atoi(cap_s) : 256;
    if (q.T > cap) return q;
    q.R = cap / q.T;
    if (q.R > q.steps_total / 2) q.R = q.steps_total / 2;
    q.big = q.R >= 1;
    return q;
}

// Tile counters of the persistent NT kernels: one set of 16 words per (device, stream) - launches of one stream run one after the
// other and a kernel leaves its set zeroed, launches of different streams may overlap and must not share one.  In host code the
// array's name is only a stand-in: its device address comes from hipGetSymbolAddress, once per device.  A process that has used
// more streams than there are sets recycles the set whose stream launched a GEMM longest ago (a finished launch leaves it clean).
// A graph that captured a GEMM holds its capture stream's set: replay it on that stream, or at least never beside eager GEMMs of
// that stream.
#define NTP_SCHED_SLOTS 256
static __device__ __align__(256) unsigned ntp_sched[NTP_SCHED_SLOTS * 16 + 256];      // counter sets, then the 1 KB store sink of gemm_ntq_kernel
struct NtpSchedDev { unsigned* base = nullptr; unsigned* err_dev = nullptr; int used = 0; void* streams[NTP_SCHED_SLOTS]; unsigned long long tick[NTP_SCHED_SLOTS]; };
static std::mutex g_ntp_sched_mu;
static NtpSchedDev g_ntp_sched[64];
static unsigned long long g_ntp_tick = 0;
// the sticky error word of ntp_check_entry: host memory that the devices write directly - one page of the library's own data,
// page-locked and mapped once (no allocation); where the driver refuses to register that page, a page of pinned memory from
// hipHostMalloc, once per process
alignas(4096) static unsigned g_ntp_err_page[1024];
static unsigned* g_ntp_err_host = g_ntp_err_page;
static int g_ntp_err_state = 0;       // 0: not set up yet, 1: registered page, 2: hipHostMalloc page, -1: neither (no reporting)

static NtpSchedDev* ntp_sched_dev() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        vitk_set_error("gemm_nt: cannot query the current device");
        return nullptr;
    }
    NtpSchedDev& d = g_ntp_sched[dev];
    if (d.base == nullptr) {
        unsigned* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(ntp_sched));
        if (e != hipSuccess || ptr == nullptr) {
            vitk_set_error("gemm_nt: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return nullptr;
        }
        d.base = ptr;
        if (g_ntp_err_state == 0) {
            if (hipHostRegister(g_ntp_err_page, sizeof(g_ntp_err_page), hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess) {
                g_ntp_err_state = 1;
            } else {
                void* hp = nullptr;
                if (hipHostMalloc(&hp, 4096, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && hp != nullptr) {
                    g_ntp_err_host = (unsigned*)hp;
                    g_ntp_err_host[0] = 0u;
                    g_ntp_err_state = 2;
                } else {
                    g_ntp_err_state = -1;
                }
            }
        }
        void* dp = nullptr;
        if (g_ntp_err_state > 0 && hipHostGetDevicePointer(&dp, g_ntp_err_host, 0) == hipSuccess) d.err_dev = (unsigned*)dp;
        (void)hipGetLastError();      // a failed set-up only switches the reporting off
    }
    return &d;
}

// the counter set of `stream` on the current device (recycling the least recently used one when all are taken); *slot_index: its number
static unsigned* ntp_sched_slot(void* stream, unsigned** err_dev, int* slot_index = nullptr) {
    std::lock_guard<std::mutex> lk(g_ntp_sched_mu);
    NtpSchedDev* d = ntp_sched_dev();
    if (d == nullptr) return nullptr;
    if (err_dev) *err_dev = d->err_dev;
    int at = -1;
    for (int i = 0; i < d->used; ++i)
        if (d->streams[i] == stream) { at = i; break; }
    if (at < 0) {
        if (d->used < NTP_SCHED_SLOTS) {
            at = d->used++;
        } else {
            at = 0;
            for (int i = 1; i < NTP_SCHED_SLOTS; ++i)
                if (d->tick[i] < d->tick[at]) at = i;
        }
        d->streams[at] = stream;
    }
    d->tick[at] = ++g_ntp_tick;
    if (slot_index) *slot_index = at;
    return d->base + (size_t)at * 16;
}

// 0, or VITK_ELAUNCH with the error text set when a kernel has reported counters that were not zero at its start
static int ntp_sched_check() {
    const unsigned code = __atomic_load_n(&g_ntp_err_host[0], __ATOMIC_RELAXED);
    if (code == 0) return 0;
    vitk_set_error("gemm_nt: the tile counters of a stream were not zero when a GEMM started (workgroup residue %u saw %u claims): "
                   "an earlier launch on that stream did not finish, or a captured graph is replayed beside launches of its capture "
                   "stream; results of that GEMM are invalid.  vitk_gemm_nt_sched_reset() clears the state",
                   (code >> 24) & 7u, code & 0xffffffu);
    return VITK_ELAUNCH;
}

// compute units of the current device (cached per device id); <= 0 on failure
static int vitk_num_cus() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    if (cache[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return -1;
        // A multiple of 8, and at least 8: claim c of residue x is tile G + 8 c + x, so a grid that does not hold every residue
        // would leave tiles unclaimed (a device with fewer than 8 compute units takes the 128 x 128 kernel: see gemm_nt_impl)
        cache[dev] = n & ~7;
        if (cache[dev] == 0) cache[dev] = n;
        const char* cap_s = getenv("VITK_GEMM_CUS");
        if (cap_s && atoi(cap_s) > 0 && atoi(cap_s) < cache[dev]) cache[dev] = atoi(cap_s) >= 8 ? atoi(cap_s) & ~7 : 8;
    }
    return cache[dev];
}

// Tile order of the persistent NT kernel.  An XCD walks a contiguous range of the order, so row-major order makes every XCD
// sweep ALL column tiles: the whole weight matrix cycles through its 4 MB L2 once per tile round, and when it does not fit
// (N x K x 2 B above about 2.5 MB beside the streaming operands) every round fetches it again from the Infinity Cache.  Cutting
// the order into blocks of cb column tiles keeps a slice of the weights resident at the price of fetching the activation panel
// once per block; cb minimises the estimate A x blocks + (slice resident ? 8 slices : rounds x 8 slices).
static int ntp_column_block(int tiles_m, int tiles_n, int K, int cus) {
    static const char* bud_s = getenv("VITK_NT_SLICE_KB");
    static const double budget = (bud_s ? atof(bud_s) : 2560.0) * 1024.0;
    const double a_bytes = (double)tiles_m * LBM * K * 2.0, col_bytes = (double)LBN * K * 2.0;
    const double rounds = (double)tiles_m * tiles_n / (double)cus;
    int best = tiles_n;
    double best_cost = 0.0;
    for (int nb = 1; nb <= tiles_n && nb <= 6; ++nb) {
        const int cb = (tiles_n + nb - 1) / nb;
        if (nb > 1 && (tiles_n + cb - 1) / cb != nb) continue;
        const double slice = cb * col_bytes;
        const double cost = a_bytes * nb + (slice <= budget ? 8.0 * slice * nb : rounds * 8.0 * slice);
        if (nb == 1 || cost < best_cost) { best = cb; best_cost = cost; }
    }
    return best;
}

extern "C" {

static int gemm_nt_impl(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int epilogue,
                        const float* bias, const void* res, int ldres, const float* rowscale, int rows_per_group, void* out2,
                        int ldout2, const void* aux, int ldaux, const float* pos, int patches, int prefix, const int* sample_map,
                        int rows_per_sample, void* stream) {
    VITK_REQUIRE(A && B && C, "gemm_nt: null operand");
    VITK_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    VITK_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K,
                 "gemm_nt: K, lda, ldb must be multiples of 8 and lda,ldb >= K (K=%d lda=%d ldb=%d)", K, lda, ldb);
    VITK_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && ldc >= N, "gemm_nt: N and ldc must be multiples of 4, ldc >= N (N=%d ldc=%d)", N, ldc);
    VITK_REQUIRE(vitk_aligned(A, 16) && vitk_aligned(B, 16) && vitk_aligned(C, 16), "gemm_nt: operands must be 16-byte aligned");
    VITK_REQUIRE(epilogue >= VITK_EPI_PLAIN && epilogue <= VITK_EPI_MUL, "gemm_nt: unknown epilogue %d", epilogue);
    if (epilogue != VITK_EPI_PLAIN && epilogue != VITK_EPI_DGELU && epilogue != VITK_EPI_MUL)
        VITK_REQUIRE(bias && vitk_aligned(bias, 16), "gemm_nt: epilogue %d needs a 16-byte aligned bias", epilogue);
    if (epilogue == VITK_EPI_BIAS_GELU || epilogue == VITK_EPI_BIAS_GELU_DG)
        VITK_REQUIRE(out2 && ldout2 % 4 == 0 && ldout2 >= N && vitk_aligned(out2, 16), "gemm_nt: BIAS_GELU needs out2 (ld%%4==0)");
    if (epilogue == VITK_EPI_BIAS_RES) {
        VITK_REQUIRE(res && ldres % 4 == 0 && ldres >= N && vitk_aligned(res, 16), "gemm_nt: BIAS_RES needs res (ld%%4==0)");
        VITK_REQUIRE(!rowscale || rows_per_group > 0, "gemm_nt: rowscale needs rows_per_group > 0");
    }
    if (epilogue == VITK_EPI_DGELU || epilogue == VITK_EPI_MUL)
        VITK_REQUIRE(aux && ldaux % 4 == 0 && ldaux >= N && vitk_aligned(aux, 16), "gemm_nt: DGELU needs aux (ld%%4==0)");
    if (epilogue == VITK_EPI_PATCH)
        VITK_REQUIRE(pos && patches > 0 && prefix >= 0 && M % patches == 0 && vitk_aligned(pos, 16),
                     "gemm_nt: PATCH needs pos, patches>0, M%%patches==0 (M=%d patches=%d)", M, patches);

    GemmNTParams p;
    p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.bias = bias; p.res = (const bf16_t*)res; p.ldres = ldres;
    p.rowscale = rowscale; p.rows_per_group = rows_per_group; p.out2 = (bf16_t*)out2; p.ldout2 = ldout2;
    p.aux = (const bf16_t*)aux; p.ldaux = ldaux; p.pos = pos; p.patches = patches; p.prefix = prefix;
    p.smap = sample_map; p.rps = rows_per_sample;
    hipStream_t s = (hipStream_t)stream;
    // large problems with K a multiple of 64 take the persistent 256x256 direct-to-LDS kernel; the 128x128 register-staged kernel
    // handles K tails (zero fill) and small shapes.  VITK_GEMM_SMALL=1 forces the latter (A/B testing).
    static const bool force_small = getenv("VITK_GEMM_SMALL") != nullptr;
    const bool ld8 = ldc % 8 == 0 && (!res || ldres % 8 == 0) && (!out2 || ldout2 % 8 == 0) && (!aux || ldaux % 8 == 0);   // 16-byte row pieces
    bool big = !force_small && K % GBK == 0 && K >= 2 * GBK && M >= 512 && N >= 128 && N % 8 == 0 && ld8;
    const int cus = big ? vitk_num_cus() : 1;
    if (cus <= 0) { vitk_set_error("gemm_nt: cannot query the compute-unit count"); return VITK_ELAUNCH; }
    if (cus < 8 || (long long)((M + LBM - 1) / LBM) * ((N + LBN - 1) / LBN) / 8 >= (long long)NTP_CLAIM_MASK - 64) big = false;
    p.sched = nullptr;
    p.sched_err = nullptr;
    p.sink = nullptr;
    if (big) {
        if (int rc_ = ntp_sched_check()) return rc_;
        int slot_index = 0;
        p.sched = ntp_sched_slot(stream, &p.sched_err, &slot_index);
        if (p.sched == nullptr) return VITK_ELAUNCH;
        p.sink = p.sched - (size_t)slot_index * 16 + (size_t)NTP_SCHED_SLOTS * 16;
    }
    const bool pr = vitk_prof_begin(stream, 2.0 * (double)M * (double)N * (double)K, VITK_PROF_GEMM_NT);       // every return below closes the bracket
    if (big) {
        p.tiles_n = (N + LBN - 1) / LBN;
        p.tiles_m = (M + LBM - 1) / LBM;
        p.num_tiles = p.tiles_m * p.tiles_n;
        p.cb = ntp_column_block(p.tiles_m, p.tiles_n, K, cus);
        const size_t smem = NTP_LDS;
        const int grid = p.num_tiles < cus ? p.num_tiles : cus;      // one persistent workgroup per CU
        if (vitk_gemm_ntq_takes(p, epilogue)) {      // four waves, output stores and side loads under the next tile's MFMAs
            const int rc_ = vitk_gemm_ntq_launch(p, epilogue, grid, s);
            if (pr) vitk_prof_end(stream);
            return rc_ ? rc_ : vitk_check_launch("gemm_nt");
        }
#define VITK_LAUNCH_BIG(E)                                                                                              \
    case E: {                                                                                                           \
        static VitkLdsOnce once = {};                                                                                   \
        if (int rc_ = vitk_lds_once(once, (const void*)gemm_ntp_kernel<E>, smem, "gemm_nt")) {                          \
            if (pr) vitk_prof_end(stream);                                                                              \
            return rc_;                                                                                                 \
        }                                                                                                               \
        gemm_ntp_kernel<E><<<grid, 512, smem, s>>>(p);                                                         \
    } break;
        // bias + residual with stochastic-depth tables the per-tile lookup cannot take (see gemm_ntp_kernel): the per-row instantiation
        bool per_row_tables = false;
        if (epilogue == VITK_EPI_BIAS_RES && (p.smap != nullptr || p.rowscale != nullptr)) {
            const int per = p.smap != nullptr ? p.rps : p.rows_per_group;
            per_row_tables = !(per >= 128 && (p.smap == nullptr || p.rowscale == nullptr || p.rps == p.rows_per_group));
        }
        if (per_row_tables) {
            static VitkLdsOnce once = {};
            if (int rc_ = vitk_lds_once(once, (const void*)gemm_ntp_kernel<VITK_EPI_BIAS_RES, true>, smem, "gemm_nt")) {
                if (pr) vitk_prof_end(stream);
                return rc_;
            }
            gemm_ntp_kernel<VITK_EPI_BIAS_RES, true><<<grid, 512, smem, s>>>(p);
            if (pr) vitk_prof_end(stream);
            return vitk_check_launch("gemm_nt");
        }
        switch (epilogue) {
            VITK_LAUNCH_BIG(VITK_EPI_PLAIN)
            VITK_LAUNCH_BIG(VITK_EPI_BIAS)
            VITK_LAUNCH_BIG(VITK_EPI_BIAS_F32)
            VITK_LAUNCH_BIG(VITK_EPI_BIAS_GELU)
            VITK_LAUNCH_BIG(VITK_EPI_BIAS_RES)
            VITK_LAUNCH_BIG(VITK_EPI_PATCH)
            VITK_LAUNCH_BIG(VITK_EPI_DGELU)
            VITK_LAUNCH_BIG(VITK_EPI_BIAS_GELU_DG)
            VITK_LAUNCH_BIG(VITK_EPI_MUL)
        }
#undef VITK_LAUNCH_BIG
    } else {
        p.tiles_n = (N + GBN - 1) / GBN;
        p.num_tiles = ((M + GBM - 1) / GBM) * p.tiles_n;
        switch (epilogue) {
            case VITK_EPI_PLAIN: gemm_nt_kernel<VITK_EPI_PLAIN><<<p.num_tiles, 256, 0, s>>>(p); break;
            case VITK_EPI_BIAS: gemm_nt_kernel<VITK_EPI_BIAS><<<p.num_tiles, 256, 0, s>>>(p); break;
            case VITK_EPI_BIAS_F32: gemm_nt_kernel<VITK_EPI_BIAS_F32><<<p.num_tiles, 256, 0, s>>>(p); break;
            case VITK_EPI_BIAS_GELU: gemm_nt_kernel<VITK_EPI_BIAS_GELU><<<p.num_tiles, 256, 0, s>>>(p); break;
            case VITK_EPI_BIAS_RES: gemm_nt_kernel<VITK_EPI_BIAS_RES><<<p.num_tiles, 256, 0, s>>>(p); break;
            case VITK_EPI_PATCH: gemm_nt_kernel<VITK_EPI_PATCH><<<p.num_tiles, 256, 0, s>>>(p); break;
            case VITK_EPI_DGELU: gemm_nt_kernel<VITK_EPI_DGELU><<<p.num_tiles, 256, 0, s>>>(p); break;
            case VITK_EPI_BIAS_GELU_DG: gemm_nt_kernel<VITK_EPI_BIAS_GELU_DG><<<p.num_tiles, 256, 0, s>>>(p); break;
            case VITK_EPI_MUL: gemm_nt_kernel<VITK_EPI_MUL><<<p.num_tiles, 256, 0, s>>>(p); break;
        }
    }
    if (pr) vitk_prof_end(stream);
    return vitk_check_launch("gemm_nt");
}

int vitk_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int epilogue,
                 const float* bias, const void* res, int ldres, const float* rowscale, int rows_per_group, void* out2,
                 int ldout2, const void* aux, int ldaux, const float* pos, int patches, int prefix, void* stream) {
    return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, epilogue, bias, res, ldres, rowscale, rows_per_group, out2, ldout2, aux, ldaux,
                        pos, patches, prefix, nullptr, 0, stream);
}

int vitk_gemm_nt_res_scatter(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                             const void* res, int ldres, const float* rowscale, int rows_per_group, const int* sample_map,
                             int rows_per_sample, void* stream) {
    VITK_REQUIRE(sample_map && rows_per_sample > 0 && M % rows_per_sample == 0,
                 "gemm_nt_res_scatter: sample_map and rows_per_sample dividing M are required (M=%d rows_per_sample=%d)", M, rows_per_sample);
    return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, VITK_EPI_BIAS_RES, bias, res, ldres, rowscale, rows_per_group, nullptr, 0, nullptr,
                        0, nullptr, 0, 0, sample_map, rows_per_sample, stream);
}

int vitk_gemm_nt_sched_reset(void* stream) {
    unsigned* err = nullptr;
    unsigned* slot = ntp_sched_slot(stream, &err);
    if (slot == nullptr) return VITK_ELAUNCH;
    const hipError_t e = hipMemsetAsync(slot, 0, 16 * sizeof(unsigned), (hipStream_t)stream);
    if (e != hipSuccess) { vitk_set_error("gemm_nt_sched_reset: %s", hipGetErrorString(e)); return VITK_ELAUNCH; }
    __atomic_store_n(&g_ntp_err_host[0], 0u, __ATOMIC_RELAXED);
    return VITK_OK;
}

int vitk_gemm_nt_sched_state(void* stream, unsigned* words16, int* reporting) {
    unsigned* err = nullptr;
    unsigned* slot = ntp_sched_slot(stream, &err);
    if (slot == nullptr) return VITK_ELAUNCH;
    if (reporting) *reporting = err != nullptr ? g_ntp_err_state : 0;
    if (words16) {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e == hipSuccess) e = hipMemcpy(words16, slot, 16 * sizeof(unsigned), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { vitk_set_error("gemm_nt_sched_state: %s", hipGetErrorString(e)); return VITK_ELAUNCH; }
    }
    return (int)__atomic_load_n(&g_ntp_err_host[0], __ATOMIC_RELAXED) != 0 ? 1 : VITK_OK;
}

int vitk_gemm_nt_sched_poke(void* stream, int word, unsigned value) {
    VITK_REQUIRE(word >= 0 && word < 16, "gemm_nt_sched_poke: word %d outside the 16-word counter set", word);
    unsigned* slot = ntp_sched_slot(stream, nullptr);
    if (slot == nullptr) return VITK_ELAUNCH;
    const hipError_t e = hipMemsetD32Async((hipDeviceptr_t)(slot + word), (int)value, 1, (hipStream_t)stream);
    if (e != hipSuccess) { vitk_set_error("gemm_nt_sched_poke: %s", hipGetErrorString(e)); return VITK_ELAUNCH; }
    return VITK_OK;
}

size_t vitk_gemm_tn_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const TnPlan q = tn_plan(M, N, K);
    if (q.big) return ((size_t)q.R * q.T * 65536 + (size_t)q.R * N) * sizeof(float);
    const int s = tn_splits(M, N, K);
    return (size_t)s * ((size_t)N * K + (size_t)N) * sizeof(float);
}

int vitk_gemm_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K, float* dbias,
                 int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    VITK_REQUIRE(A && B && C && workspace, "gemm_tn: null operand");
    VITK_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_tn: bad shape M=%d N=%d K=%d", M, N, K);
    VITK_REQUIRE(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= N && ldb >= K,
                 "gemm_tn: N, K, lda, ldb must be multiples of 8 (N=%d K=%d lda=%d ldb=%d)", N, K, lda, ldb);
    VITK_REQUIRE(ldc % 4 == 0 && ldc >= K, "gemm_tn: ldc must be a multiple of 4 and >= K");
    VITK_REQUIRE(vitk_aligned(A, 16) && vitk_aligned(B, 16) && vitk_aligned(C, 16) && vitk_aligned(workspace, 16),
                 "gemm_tn: operands must be 16-byte aligned");
    if (workspace_bytes < vitk_gemm_tn_workspace(M, N, K)) {
        vitk_set_error("gemm_tn: workspace %zu < %zu", workspace_bytes, vitk_gemm_tn_workspace(M, N, K));
        return VITK_ENOSPC;
    }
    hipStream_t s = (hipStream_t)stream;
    const TnPlan q = tn_plan(M, N, K);
    if (q.big) {
        GemmTNBigParams b;
        b.A = (const bf16_t*)A; b.lda = lda; b.B = (const bf16_t*)B; b.ldb = ldb;
        b.slab = (float*)workspace;
        b.colslab = dbias ? b.slab + (size_t)q.R * q.T * 65536 : nullptr;
        b.M = M; b.N = N; b.K = K; b.T = q.T; b.tiles_k = q.tiles_k; b.R = q.R; b.steps_total = q.steps_total;
        b.ntasks = q.R * q.T;
        const size_t smem = (size_t)4 * TNB_TILE * sizeof(bf16_t);
        static VitkLdsOnce once_t = {}, once_f = {};
        if (int rc_ = vitk_lds_once(once_t, (const void*)gemm_tnb_kernel<true>, smem, "gemm_tn")) return rc_;
        if (int rc_ = vitk_lds_once(once_f, (const void*)gemm_tnb_kernel<false>, smem, "gemm_tn")) return rc_;
        const bool pr = vitk_prof_begin(stream, 2.0 * (double)M * (double)N * (double)K, VITK_PROF_GEMM_TN);
        if (dbias) gemm_tnb_kernel<true><<<b.ntasks, 512, smem, s>>>(b);
        else gemm_tnb_kernel<false><<<b.ntasks, 512, smem, s>>>(b);
        if (pr) vitk_prof_end(stream);
        int rc = vitk_check_launch("gemm_tn");
        if (rc) return rc;
        const size_t total4 = (size_t)N * K / 4;
        int rgrid = (int)((total4 + 255) / 256);
        if (rgrid > 2048) rgrid = 2048;
        const int bias_blocks = dbias ? (N + 255) / 256 : 0;
        reduce_tiles_kernel<<<rgrid + bias_blocks, 256, 0, s>>>(b.slab, q.R, q.T, q.tiles_k, N, K, C, ldc, accumulate, rgrid, b.colslab,
                                                                dbias);
        return vitk_check_launch("gemm_tn_reduce");
    }
    const int splits = tn_splits(M, N, K);
    GemmTNParams p;
    p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb;
    p.slab = (float*)workspace;
    p.colslab = dbias ? p.slab + (size_t)splits * N * K : nullptr;
    p.M = M; p.N = N; p.K = K;
    int mps = (M + splits - 1) / splits;
    mps = ((mps + TBM - 1) / TBM) * TBM;
    p.m_per_split = mps;
    const int tiles_n = (N + GBN - 1) / GBN;
    p.tiles_k = (K + GBN - 1) / GBN;
    p.tiles = tiles_n * p.tiles_k;
    p.splits_per_xcd = splits / 8;
    dim3 grid(p.tiles * splits);
    const bool pr = vitk_prof_begin(stream, 2.0 * (double)M * (double)N * (double)K, VITK_PROF_GEMM_TN);
    if (dbias) gemm_tn_kernel<true><<<grid, 256, 0, s>>>(p);
    else gemm_tn_kernel<false><<<grid, 256, 0, s>>>(p);
    if (pr) vitk_prof_end(stream);
    int rc = vitk_check_launch("gemm_tn");
    if (rc) return rc;
    const size_t total4 = (size_t)N * K / 4;
    int rgrid = (int)((total4 + 255) / 256);
    if (rgrid > 2048) rgrid = 2048;
    reduce_slabs_kernel<<<rgrid, 256, 0, s>>>(p.slab, splits, N, K, C, ldc, accumulate);
    rc = vitk_check_launch("gemm_tn_reduce");
    if (rc) return rc;
    if (dbias) {
        reduce_vec_kernel<<<(N + 255) / 256, 256, 0, s>>>(p.colslab, splits, N, dbias, accumulate);
        rc = vitk_check_launch("gemm_tn_reduce_bias");
    }
    return rc;
}

}  // extern "C"
