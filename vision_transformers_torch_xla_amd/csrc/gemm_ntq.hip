// gemm_nt, large-tile persistent variant with FOUR waves and a DEFERRED epilogue (gfx950).
//
// Same contract, 256 x 256 tiles, LDS image, LDS-DMA staging, K-step stream, tile order and tile claims as gemm_ntp_kernel
// (gemm.hip), but one wave per SIMD, each a 128 x 128 sub-tile = 8 x 8 MFMA tiles = 256 accumulator registers.  With the whole
// register file of a SIMD to itself a wave has room for what the eight-wave kernel has not: 128 registers that hold a finished
// tile's OUTPUT (bf16, in the layout of its global stores) while the next tile is being computed.  So a tile's 32 output stores
// per wave leave under the next tile's MFMAs (one per 8-MFMA phase, eight per K-step, over its first four K-steps) instead of in a store burst
// that no MFMA overlaps, and the side operand of an epilogue (residual rows, kept GELU derivative) arrives in the same registers
// during the tile's last K-steps (they are free once the previous tile's output has left) instead of being waited for.  What is
// left between two tiles is the transposition of the fp32 accumulators through LDS and the epilogue arithmetic, on fp32 values
// exactly as in gemm_ntp_kernel: results are bit-identical to that kernel.
//
// MEASURED (MI355X, 95 % of 100 864 rows, same process, microseconds eight-wave / this kernel): plain K = 768 99 / 123, K = 2304
// 281 / 331, K = 3072 387 / 454; bias N = 2304 302 / 383; bias + residual K = 768 145 / 227, K = 3072 422 / 549; multiply N = 3072
// 442 / 752 - slower everywhere, so the launcher takes it only with VITK_GEMM_NTQ=1.  Per tile round its K-step costs 1.73-2.05 us
// against 1.51-1.78 and its tile boundary 3.8 us against 1.7: with one wave per SIMD nothing covers the issue time of a wave's 16
// LDS-DMA instructions per K-step (eight per wave, and the partner wave's MFMAs beside them, in the eight-wave kernel), nor the
// LDS round trip of the epilogue's transposition; the stores and side loads it moves under the MFMAs were costing the
// eight-wave kernel less than that.  Kept as the tested record of the experiment (bit-identical results).
//
// Vector-memory bookkeeping.  The LDS-DMA requests are inline assembly and counted by hand; the output stores, side loads and
// bias loads are ordinary (the compiler counts them, but never has to wait for one inside the K loop: their first use is the
// epilogue).  All of them share the wave's in-order vmcnt counter.  A K-step's wait must cover the request of the NEXT K-step,
// whose last quarter is issued at the start of phase 4 of 16; everything ordinary is issued in phases 6..14, i.e. younger than that
// quarter, and a step issues a FIXED number of them (template constant EXTRA), so the wait is `vmcnt(EXTRA)`.  That is why only
// FULL tiles defer their stores (every lane stores: the count is exact); a ragged tile or a workgroup's last tile stores at once.
#include "gemm_nt.h"
#include <cstdlib>

#define NTQ_STG_FLOATS 2048                                   // per wave: two staging buffers of 16 rows x 64 fp32 (4 KB each)
#define NTQ_LDS (4 * LTILE * 2 + 4 * NTQ_STG_FLOATS * 4)      // operand image 128 KB + 32 KB of staging = 160 KB
#define NTQ_MIN_KSTEPS 12                                     // drain (4 steps) + plain + intake (4 steps) + last
#ifndef NTQ_FRAGS
#define NTQ_FRAGS 48            // fragment registers of the K loop: 48 (16 phases of 8 MFMAs) or 80 (8 phases of 16 MFMAs)
#endif

struct NtqRowMap { int bnd, off0, off1; float rs0, rs1; };    // stochastic-depth row mapping of a wave's 128 rows (see gemm_ntp_kernel)

// f(itag<I>{}) for I = FROM .. TO-1: every index a compile-time constant (register arrays must never be indexed at run time)
template <int FROM, int TO, class F>
static __device__ __forceinline__ void ntq_static_for(F&& f) {
    if constexpr (FROM < TO) {
        f(itag<FROM>{});
        ntq_static_for<FROM + 1, TO>(f);
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_ntq_kernel(const GemmNTParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t lsm[];   // [2 buffers][A tile | B tile], [4 waves][2][16][64] fp32
    constexpr bool RES = EPI == VITK_EPI_BIAS_RES;
    constexpr bool SIDE = EPI == VITK_EPI_BIAS_RES || EPI == VITK_EPI_MUL;
    constexpr bool BIAS = EPI == VITK_EPI_BIAS || EPI == VITK_EPI_BIAS_RES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;
    float* stg = reinterpret_cast<float*>(lsm + 4 * LTILE) + wave * NTQ_STG_FLOATS;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr(lsm));

    // staging: wave w, piece j (0..7) covers rows (w*8+j)*8 .. +7 of a tile; lane -> row += lane/8, LDS position lane%8, which
    // receives source chunk (lane%8) ^ (row&7); row&7 == lane/8 for every piece
    const int srow = lane >> 3, sch = (lane & 7) ^ srow;
    const unsigned lds_stage = (unsigned)(wave * 8 * 8 * GBK * sizeof(bf16_t));
    const int vrow0 = wave * 64 + srow;
    const bf16_t* sA = p.A;          // staging state: operand panels of the tile being requested
    const bf16_t* sB = p.B;
    int ra_max = 0, rb_max = 0;      // rows past the edge re-read the last valid row
    auto aim = [&](int v, int& m0, int& n0) __attribute__((always_inline)) {
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
        ra_max = p.M - 1 - m0;
        rb_max = p.N - 1 - n0;
    };
    // quarter q of the request for K-step kstep -> LDS buffer par: q = 0, 1 the activation panel (it comes from HBM), 2, 3 the
    // weight panel (L2); 4 DMA instructions each
    auto stage_q = [&](int kstep, int par, auto qtag) __attribute__((always_inline)) {
        constexpr int q = decltype(qtag)::value;
        constexpr int which = q >> 1, j0 = (q & 1) * 4;
        const unsigned d = lds_base + (unsigned)((par * 2 + which) * LTILE * sizeof(bf16_t)) + lds_stage;
        const bf16_t* src = (which ? sB : sA) + kstep * GBK;
        const int rmax = which ? rb_max : ra_max, ld = which ? p.ldb : p.lda;
#pragma unroll
        for (int j = j0; j < j0 + 4; ++j) {
            const int row = vrow0 + 8 * j;
            const unsigned vo = (unsigned)(((row < rmax ? row : rmax) * ld + sch * 8) * (int)sizeof(bf16_t));
            glds16(src, vo, d + j * 8 * GBK * (unsigned)sizeof(bf16_t));
        }
    };

    // fragment read offsets (elements) inside a tile: row*64 + ((chunk ^ (row&7)) << 3); row&7 == lr&7 for every tile row
    int offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int x = (((ks * 4 + lg) ^ (lr & 7)) << 3);
        offA[ks] = (wm * 128 + lr) * GBK + x;
        offB[ks] = (wn * 128 + lr) * GBK + x;
    }
    f32x4 acc[8][8];
    // Fragments: a K-step is 16 phases of 8 MFMAs, ordered (K-slice ks, column half nh, row pair q): the 4 B fragments of a
    // (ks, nh) group stay for its 4 phases while the A fragments come in pairs, one phase ahead, and the next group's B fragments
    // arrive one per phase - 48 fragment registers (a scheme that holds a slice's 8 B fragments and the next slice's needs 80,
    // which the 256 accumulator + 128 held-tile registers do not leave).  Each A fragment is read twice per slice.
#if NTQ_FRAGS == 80
    bf16x8 fb0[8], fb1[8], faE[2], faO[2];
#else
    bf16x8 fbC[4], fbN[4], faE[2], faO[2];
#endif
    auto ldA2 = [&](bf16x8 (&f)[2], const bf16_t* cA, int ks, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) f[t] = *reinterpret_cast<const bf16x8*>(cA + offA[ks] + (2 * q + t) * 16 * GBK);
    };
    auto ldB1 = [&](bf16x8& f, const bf16_t* cB, int ks, int t) __attribute__((always_inline)) { f = *reinterpret_cast<const bf16x8*>(cB + offB[ks] + t * 16 * GBK); };
    // fresh: the tile's first K-slice starts every accumulator from a constant zero (no pass that clears 256 registers)
    auto mma = [&](const bf16x8 (&fa)[2], const bf16x8 (&fb)[4], int q, int nh, bool fresh) __attribute__((always_inline)) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[2 * q + tm][4 * nh + tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn], fa[tm], fresh ? zero : acc[2 * q + tm][4 * nh + tn], 0, 0, 0);
    };
    [[maybe_unused]] auto mma8 = [&](const bf16x8 (&fa)[2], const bf16x8 (&fb)[8], int q, bool fresh) __attribute__((always_inline)) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 8; ++tn)
                acc[2 * q + tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn], fa[tm], fresh ? zero : acc[2 * q + tm][tn], 0, 0, 0);
    };
    // scheduling of a phase: `mf` MFMAs then one LDS read, `n` times (the rest of the MFMAs follow)
#define NTQ_SCHED(n, mf)                                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, (mf), 0);                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                      \
    }                                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);

    // ---- the held tile.  Piece pi (0..31) of a wave's 128 x 128 sub-tile = 8 rows x 64 columns: unit u = pi >> 1 = (16-row tile
    // tm = u >> 1, column half h = u & 1), row half q2 = pi & 1; lane -> row tm*16 + q2*8 + lane/8, 8 columns from h*64 + (lane%8)*8:
    // eight lanes cover one 128-byte line, an instruction eight rows.  D[pi] holds, in turn: the tile's side operand piece
    // (intake, SIDE epilogues), then its output piece (from the epilogue until its store).
    u32x4 D[32];
    float bias16[2][8];                                         // this lane's bias entries: column half h, 8 columns
    int m0 = 0, n0 = 0, nm0 = 0, nn0 = 0, pm0 = 0, pn0 = 0;     // tile being computed, next tile, tile whose output D holds
    NtqRowMap cmap = {0x7fffffff, 0, 0, 1.0f, 1.0f}, pmap = cmap;
    auto make_map = [&](int tile_m0) __attribute__((always_inline)) {
        // BIAS_RES with stochastic depth: GEMM row m belongs to kept sample m / per, whose rows live at sample smap[.] of the residual
        // stream and are scaled by rowscale[.]; a wave's 128 rows span at most two samples (per >= 128, checked by the launcher)
        NtqRowMap mp = {0x7fffffff, 0, 0, 1.0f, 1.0f};
        if (RES && (p.smap != nullptr || p.rowscale != nullptr)) {
            const int per = p.smap != nullptr ? p.rps : p.rows_per_group;
            int mw = __builtin_amdgcn_readfirstlane(tile_m0) + wm * 128;
            mw = mw < p.M ? mw : p.M - 1;
            const int c0 = mw / per;
            mp.bnd = (c0 + 1) * per;
            const int c1 = mp.bnd < p.M ? c0 + 1 : c0;
            if (p.smap != nullptr) {
                mp.off0 = __builtin_amdgcn_readfirstlane((p.smap[c0] - c0) * per);
                mp.off1 = __builtin_amdgcn_readfirstlane((p.smap[c1] - (c0 + 1)) * per);
            }
            if (p.rowscale != nullptr) {
                mp.rs0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.rowscale[c0])));
                mp.rs1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.rowscale[c1])));
            }
        }
        return mp;
    };
    const int prow = wm * 128 + srow, pcol = wn * 128 + (lane & 7) * 8;     // a lane's row / column inside a tile, before the piece's own
    // output piece pi of the held tile (a FULL tile: no lane is masked, the instruction count per K-step is exact)
    auto drain_full = [&](auto pitag) __attribute__((always_inline)) {
        constexpr int pi = decltype(pitag)::value;
        const int m = pm0 + prow + (pi >> 2) * 16 + (pi & 1) * 8;
        const int n = pn0 + pcol + ((pi >> 1) & 1) * 64;
        const int orow = RES ? m + (m >= pmap.bnd ? pmap.off1 : pmap.off0) : m;
        __builtin_nontemporal_store(D[pi], reinterpret_cast<u32x4*>((bf16_t*)p.C + (size_t)orow * p.ldc + n));
    };
    // the same for any tile, at once (ragged tiles, a workgroup's last tile)
    auto drain_now = [&](int tile_m0, int tile_n0, const NtqRowMap& mp) __attribute__((always_inline)) {
        // No branch around a lane's store: a lane whose piece lies past the edge of the matrix writes into a 1 KB sink of the
        // library instead (never read).  A divergent branch here, between the tile loop's body and its exit, makes hipcc treat
        // the loop's exit as divergent and every value the loop carries - LDS addresses, operand bases, K-step numbers - as
        // per-lane: vector arithmetic for all of it, and no scalar registers to hand to the LDS-DMA instructions.
        u32x4* sink = reinterpret_cast<u32x4*>(p.sink) + lane;
        ntq_static_for<0, 32>([&](auto pitag) __attribute__((always_inline)) {
            constexpr int pi = decltype(pitag)::value;
            const int m = tile_m0 + prow + (pi >> 2) * 16 + (pi & 1) * 8;
            const int n = tile_n0 + pcol + ((pi >> 1) & 1) * 64;
            const int orow = RES ? m + (m >= mp.bnd ? mp.off1 : mp.off0) : m;
            u32x4* dst = reinterpret_cast<u32x4*>((bf16_t*)p.C + (size_t)orow * p.ldc + n);
            __builtin_nontemporal_store(D[pi], (m < p.M && n < p.N) ? dst : sink);
        });
    };
    // side operand piece pi of the tile being computed (rows / columns past the edge re-read valid ones: never stored)
    auto intake = [&](auto pitag) __attribute__((always_inline)) {
        constexpr int pi = decltype(pitag)::value;
        int m = m0 + prow + (pi >> 2) * 16 + (pi & 1) * 8;
        m = m < p.M ? m : p.M - 1;
        int n = n0 + pcol + ((pi >> 1) & 1) * 64;
        n = n < p.N ? n : 0;
        if (RES) m += m >= cmap.bnd ? cmap.off1 : cmap.off0;
        D[pi] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(nt_side_ptr<EPI>(p, m, n)));
    };
    auto load_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int n = n0 + pcol + h * 64;
            n = n < p.N ? n : 0;
            const float4 lo = *reinterpret_cast<const float4*>(p.bias + n), hi = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bias16[h][0] = lo.x; bias16[h][1] = lo.y; bias16[h][2] = lo.z; bias16[h][3] = lo.w;
            bias16[h][4] = hi.x; bias16[h][5] = hi.y; bias16[h][6] = hi.z; bias16[h][7] = hi.w;
        }
    };

    const int nk = p.K / GBK;
    const int G = gridDim.x;
    // tile claims: as gemm_ntp_kernel (first tile = block index, every further one claimed from the XCD's counter one tile ahead)
    const int xcd = blockIdx.x & 7;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    lds_u32* mailbox = (lds_u32*)(__UINTPTR_TYPE__)(lds_base + (unsigned)(4 * LTILE * sizeof(bf16_t)));   // first word of wave 0's staging
    unsigned claim = 0;
    auto issue_claim = [&](unsigned inc) __attribute__((always_inline)) {
        if (wave == 0 && lane == 0) claim = atomicAdd(&p.sched[xcd], inc);
    };
    bool entry = true;
    int vn = 0;
    bool has_next = false, have_prev = false;
    int par = 0;                       // LDS buffer of the current K-step
    int pend_k = -1, pend_par = 0;     // request whose first quarter has been issued

    // One K-step.  MODE 0: nothing beside it; 1: stores of held pieces BASE .. BASE+7; 2: side loads into pieces BASE .. BASE+7.
    // FLAGS bit 0: this step loads the bias; bit 1: the tile's last step (the request for the step after next is left to the epilogue);
    // bit 2: the tile's first step (its first K-slice starts the accumulators).
    auto kstep = [&](int kt, auto mode_t, auto base_t, auto flags_t) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_t)::value, BASE = decltype(base_t)::value, FLAGS = decltype(flags_t)::value;
        constexpr bool BLD = (FLAGS & 1) != 0, LAST = (FLAGS & 2) != 0, FIRST = (FLAGS & 4) != 0;
        constexpr int EXTRA = (MODE != 0 ? 8 : 0) + (BLD ? 4 : 0);
        auto beside = [&](auto itag_) __attribute__((always_inline)) {
            constexpr int i = decltype(itag_)::value;
            if (MODE == 1) drain_full(itag<BASE + i>{});
            if (MODE == 2) intake(itag<BASE + i>{});
            if (MODE != 0) __builtin_amdgcn_sched_barrier(0);
        };
        const bf16_t* cA = lsm + par * 2 * LTILE;
        const bf16_t* cB = cA + LTILE;
        const bf16_t* nA = lsm + (par ^ 1) * 2 * LTILE;
        const bf16_t* nB = nA + LTILE;
#if NTQ_FRAGS == 80
        // 8 phases of 16 MFMAs: phase (ks, q) multiplies A row tiles 2q, 2q+1 of K-slice ks with the 8 B fragments of the slice; the A
        // fragments of the next phase and, through slice 0, the B fragments of slice 1 are read while it runs
        if (pend_k >= 0) stage_q(pend_k, pend_par, itag<1>{});
        mma8(faE, fb0, 0, FIRST);
        ldA2(faO, cA, 0, 1); ldB1(fb1[0], cB, 1, 0); ldB1(fb1[1], cB, 1, 1);
        NTQ_SCHED(4, 4)
        if (pend_k >= 0) stage_q(pend_k, pend_par, itag<2>{});
        mma8(faO, fb0, 1, FIRST);
        ldA2(faE, cA, 0, 2); ldB1(fb1[2], cB, 1, 2); ldB1(fb1[3], cB, 1, 3);
        NTQ_SCHED(4, 4)
        if (pend_k >= 0) { stage_q(pend_k, pend_par, itag<3>{}); pend_k = -1; }
        mma8(faE, fb0, 2, FIRST);
        ldA2(faO, cA, 0, 3); ldB1(fb1[4], cB, 1, 4); ldB1(fb1[5], cB, 1, 5);
        NTQ_SCHED(4, 4)
        // everything ordinary from here to the wait is younger than the request's last quarter: two instructions per phase
        beside(itag<0>{}); beside(itag<1>{});
        mma8(faO, fb0, 3, FIRST);
        ldA2(faE, cA, 1, 0); ldB1(fb1[6], cB, 1, 6); ldB1(fb1[7], cB, 1, 7);
        NTQ_SCHED(4, 4)
        beside(itag<2>{}); beside(itag<3>{});
        mma8(faE, fb1, 0, false);
        ldA2(faO, cA, 1, 1);
        NTQ_SCHED(2, 8)
        beside(itag<4>{}); beside(itag<5>{});
        mma8(faO, fb1, 1, false);
        ldA2(faE, cA, 1, 2);
        NTQ_SCHED(2, 8)
        beside(itag<6>{}); beside(itag<7>{});
        if (BLD) { load_bias(); __builtin_amdgcn_sched_barrier(0); }
        mma8(faE, fb1, 2, false);
        ldA2(faO, cA, 1, 3);
        NTQ_SCHED(2, 8)
#else
        // phase (g, q), g = 2 ks + nh: 8 MFMAs of row pair q with the group's B fragments (fbC for even g, fbN for odd g); beside
        // them the reads of the next phase's A pair and of one B fragment of the next group
#define NTQ_PHASE(G_, Q_, FA_CUR, FA_NXT, FB_CUR, FB_NXT)                                                                      \
        mma(FA_CUR, FB_CUR, Q_, (G_) & 1, FIRST && (G_) < 2);                                                                   \
        ldA2(FA_NXT, cA, ((G_) * 4 + (Q_) + 1) >> 3, ((Q_) + 1) & 3);                                                           \
        if ((G_) < 3) ldB1(FB_NXT[Q_], cB, ((G_) + 1) >> 1, (((G_) + 1) & 1) * 4 + (Q_));                                        \
        NTQ_SCHED(((G_) < 3 ? 3 : 2), ((G_) < 3 ? 2 : 3))
        // ---- group 0 (K-slice 0, columns 0..63): the rest of the pending request leaves beside its first phases
        if (pend_k >= 0) stage_q(pend_k, pend_par, itag<1>{});
        NTQ_PHASE(0, 0, faE, faO, fbC, fbN)
        NTQ_PHASE(0, 1, faO, faE, fbC, fbN)
        if (pend_k >= 0) stage_q(pend_k, pend_par, itag<2>{});
        NTQ_PHASE(0, 2, faE, faO, fbC, fbN)
        NTQ_PHASE(0, 3, faO, faE, fbC, fbN)
        // ---- group 1 (K-slice 0, columns 64..127)
        if (pend_k >= 0) { stage_q(pend_k, pend_par, itag<3>{}); pend_k = -1; }
        NTQ_PHASE(1, 0, faE, faO, fbN, fbC)
        NTQ_PHASE(1, 1, faO, faE, fbN, fbC)
        // everything ordinary from here to the wait is younger than the request's last quarter: one instruction per phase
        beside(itag<0>{});
        NTQ_PHASE(1, 2, faE, faO, fbN, fbC)
        beside(itag<1>{});
        NTQ_PHASE(1, 3, faO, faE, fbN, fbC)
        // ---- group 2 (K-slice 1, columns 0..63)
        beside(itag<2>{});
        NTQ_PHASE(2, 0, faE, faO, fbC, fbN)
        beside(itag<3>{});
        NTQ_PHASE(2, 1, faO, faE, fbC, fbN)
        beside(itag<4>{});
        NTQ_PHASE(2, 2, faE, faO, fbC, fbN)
        beside(itag<5>{});
        NTQ_PHASE(2, 3, faO, faE, fbC, fbN)
        // ---- group 3 (K-slice 1, columns 64..127)
        beside(itag<6>{});
        NTQ_PHASE(3, 0, faE, faO, fbN, fbC)
        beside(itag<7>{});
        NTQ_PHASE(3, 1, faO, faE, fbN, fbC)
        if (BLD) { load_bias(); __builtin_amdgcn_sched_barrier(0); }
        NTQ_PHASE(3, 2, faE, faO, fbN, fbC)
#undef NTQ_PHASE
#endif
        // every wave has issued its last read of the current buffer; the next K-step, requested a full step ago, must have landed:
        // publish it, then refill the current buffer with the K-step after that
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(EXTRA) : "memory");
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
        if (!LAST) {
            if (kt + 2 < nk) {
                pend_k = kt + 2; pend_par = par;
            } else if (has_next) {
                if (kt + 2 == nk) aim(vn, nm0, nn0);
                pend_k = kt + 2 - nk; pend_par = par;
            }
            if (pend_k >= 0) stage_q(pend_k, pend_par, itag<0>{});
        }
        // last phase: the first fragments of the next K-step out of the other buffer (after the last tile: stale LDS, never used)
#if NTQ_FRAGS == 80
        mma8(faO, fb1, 3, false);
#pragma unroll
        for (int t = 0; t < 8; ++t) ldB1(fb0[t], nB, 0, t);
        ldA2(faE, nA, 0, 0);
        NTQ_SCHED(10, 1)
#else
        mma(faO, fbN, 3, 1, false);
#pragma unroll
        for (int t = 0; t < 4; ++t) ldB1(fbC[t], nB, 0, t);
        ldA2(faE, nA, 0, 0);
        NTQ_SCHED(6, 1)
#endif
        par ^= 1;
    };

    // ---- epilogue of the tile in acc: fp32 accumulators -> (bias, side operand, scale) -> bf16 pieces in D.  The accumulators
    // go through this wave's staging 16 rows x 64 columns at a time (fp32, so the arithmetic is that of gemm_ntp_kernel), two
    // buffers in turn; 16-byte chunk c of staging row r sits at chunk c ^ (r & 7): conflict-free both ways.  When a next tile
    // follows, the request for its second K-step leaves from in here (its first one is already in LDS).
    auto epilogue = [&]() __attribute__((always_inline)) {
        auto put = [&](auto utag) __attribute__((always_inline)) {
            constexpr int u = decltype(utag)::value;
            float* b = stg + (u & 1) * 1024;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                *reinterpret_cast<f32x4*>(b + lr * 64 + (((t * 4 + lg) ^ (lr & 7)) << 2)) = acc[u >> 1][(u & 1) * 4 + t];
        };
        // The side operand and the bias were loaded by ordinary loads a K-step or more ago.  Naming the youngest of them here makes
        // the compiler place its one wait for them now, while nothing else is outstanding (the last K-step waited for everything and
        // requested nothing) - not behind the LDS-DMA requests issued further down, which it cannot see and would then wait for too.
        if (SIDE) asm volatile("" ::"v"(D[31]), "v"(D[30]), "v"(D[0]));
        if (BIAS) asm volatile("" ::"v"(bias16[1][7]), "v"(bias16[1][0]), "v"(bias16[0][0]));
        if (has_next) issue_claim(1u);          // the claim for the tile after next: used a whole tile from now
        put(itag<0>{});
        ntq_static_for<0, 16>([&](auto utag) __attribute__((always_inline)) {
            constexpr int u = decltype(utag)::value;
            const float* b = stg + (u & 1) * 1024;
            f32x4 lo[2], hi[2];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int r = q2 * 8 + srow;
                lo[q2] = *reinterpret_cast<const f32x4*>(b + r * 64 + (((2 * (lane & 7)) ^ srow) << 2));
                hi[q2] = *reinterpret_cast<const f32x4*>(b + r * 64 + (((2 * (lane & 7) + 1) ^ srow) << 2));
            }
            // the next unit's accumulators go into the other buffer while this unit's arithmetic runs
            if constexpr (u + 1 < 16) put(itag<u + 1>{});
            ntq_static_for<0, 2>([&](auto qtag) __attribute__((always_inline)) {
                constexpr int q2 = decltype(qtag)::value;
                constexpr int pi = u * 2 + q2;
                const int m = m0 + prow + (u >> 1) * 16 + q2 * 8;
                float o[8] = {lo[q2][0], lo[q2][1], lo[q2][2], lo[q2][3], hi[q2][0], hi[q2][1], hi[q2][2], hi[q2][3]};
                if (BIAS) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += bias16[u & 1][j];
                }
                if (SIDE) {
                    float sv[8];
                    unpack_bf16x8(D[pi], sv);
                    if (RES) {
                        const float rs = m >= cmap.bnd ? cmap.rs1 : cmap.rs0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = sv[j] + rs * o[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] *= sv[j];
                    }
                }
                D[pi] = (u32x4){pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
            });
            if constexpr (u < 4) {       // the request for the next tile's second K-step, a quarter behind each of the first units
                if (has_next) stage_q(1, par ^ 1, itag<u>{});
            }
        });
    };

    // ---- prologue: the first tile's first two K-steps
    issue_claim(NTP_ARRIVE + 1u);
    aim(blockIdx.x, m0, n0);
    stage_q(0, 0, itag<0>{}); stage_q(0, 0, itag<1>{}); stage_q(0, 0, itag<2>{}); stage_q(0, 0, itag<3>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage_q(1, 1, itag<0>{}); stage_q(1, 1, itag<1>{}); stage_q(1, 1, itag<2>{}); stage_q(1, 1, itag<3>{});
#if NTQ_FRAGS == 80
#pragma unroll
    for (int t = 0; t < 8; ++t) ldB1(fb0[t], lsm + LTILE, 0, t);
#else
#pragma unroll
    for (int t = 0; t < 4; ++t) ldB1(fbC[t], lsm + LTILE, 0, t);
#endif
    ldA2(faE, lsm, 0, 0);

    for (;;) {
        has_next = false;            // known from the first barrier of the tile on
        cmap = make_map(m0);
        // the held tile's stores beside steps 0..3, the side operand beside steps nk-5 .. nk-2 (the bias with the last of them), the
        // last step without a request; every other step has nothing beside it
        if (have_prev) {
            kstep(0, itag<1>{}, itag<0>{}, itag<4>{});
            kstep(1, itag<1>{}, itag<8>{}, itag<0>{});
            kstep(2, itag<1>{}, itag<16>{}, itag<0>{});
            kstep(3, itag<1>{}, itag<24>{}, itag<0>{});
        } else {
            kstep(0, itag<0>{}, itag<0>{}, itag<4>{});
#pragma unroll 1
            for (int kt = 1; kt < 4; ++kt) kstep(kt, itag<0>{}, itag<0>{}, itag<0>{});
        }
#pragma unroll 1
        for (int kt = 4; kt < nk - 5; ++kt) kstep(kt, itag<0>{}, itag<0>{}, itag<0>{});
        if constexpr (SIDE) {
            kstep(nk - 5, itag<2>{}, itag<0>{}, itag<0>{});
            kstep(nk - 4, itag<2>{}, itag<8>{}, itag<0>{});
            kstep(nk - 3, itag<2>{}, itag<16>{}, itag<0>{});
            kstep(nk - 2, itag<2>{}, itag<24>{}, itag<(BIAS ? 1 : 0)>{});
        } else {
#pragma unroll 1
            for (int kt = nk - 5; kt < nk - 2; ++kt) kstep(kt, itag<0>{}, itag<0>{}, itag<0>{});
            kstep(nk - 2, itag<0>{}, itag<0>{}, itag<(BIAS ? 1 : 0)>{});
        }
        kstep(nk - 1, itag<0>{}, itag<0>{}, itag<2>{});
        epilogue();
        const bool full = m0 + LBM <= p.M && n0 + LBN <= p.N;
        if (!has_next) break;
        if (full) {
            have_prev = true;
            pm0 = m0; pn0 = n0; pmap = cmap;
        } else {
            drain_now(m0, n0, cmap);
            have_prev = false;
        }
        m0 = nm0;
        n0 = nn0;
    }
    drain_now(m0, n0, cmap);          // the workgroup's last tile
    // the last workgroup to leave zeroes the counters for the next launch that uses them
    if (wave == 0 && lane == 0) {
        const unsigned left = atomicAdd(&p.sched[8], 1u);
        if (left == (unsigned)G - 1u) {
#pragma unroll
            for (int i = 0; i < 9; ++i) p.sched[i] = 0u;
        }
    }
#undef NTQ_SCHED
}

// the problems this kernel takes (the others stay with gemm_ntp_kernel): the four epilogues with at most one bf16 output, K long
// enough for the drain / intake schedule, and - bias + residual under stochastic depth - samples of 128 rows or more
bool vitk_gemm_ntq_takes(const GemmNTParams& p, int epilogue) {
    // Off unless VITK_GEMM_NTQ=1: measured on MI355X this kernel is 8-40 % slower than gemm_ntp_kernel on the ViT shapes (see the
    // header).  The variable is read at every call: the parity test runs both kernels in one process.
    const char* env = getenv("VITK_GEMM_NTQ");
    if (env == nullptr || env[0] != '1') return false;
    if (epilogue != VITK_EPI_PLAIN && epilogue != VITK_EPI_BIAS && epilogue != VITK_EPI_BIAS_RES && epilogue != VITK_EPI_MUL) return false;
    if (p.K / GBK < NTQ_MIN_KSTEPS) return false;
    if (epilogue == VITK_EPI_BIAS_RES && (p.smap != nullptr || p.rowscale != nullptr)) {
        const int per = p.smap != nullptr ? p.rps : p.rows_per_group;
        if (per < 128) return false;
        if (p.smap != nullptr && p.rowscale != nullptr && p.rps != p.rows_per_group) return false;
    }
    return true;
}

int vitk_gemm_ntq_launch(const GemmNTParams& p, int epilogue, int grid, hipStream_t stream) {
    const size_t smem = NTQ_LDS;
#define VITK_LAUNCH_NTQ(E)                                                                                              \
    case E: {                                                                                                           \
        static VitkLdsOnce once = {};                                                                                   \
        if (int rc_ = vitk_lds_once(once, (const void*)gemm_ntq_kernel<E>, smem, "gemm_nt")) return rc_;                \
        gemm_ntq_kernel<E><<<grid, 256, smem, stream>>>(p);                                                             \
    } break;
    switch (epilogue) {
        VITK_LAUNCH_NTQ(VITK_EPI_PLAIN)
        VITK_LAUNCH_NTQ(VITK_EPI_BIAS)
        VITK_LAUNCH_NTQ(VITK_EPI_BIAS_RES)
        VITK_LAUNCH_NTQ(VITK_EPI_MUL)
        default:
            vitk_set_error("gemm_nt: epilogue %d has no four-wave kernel", epilogue);
            return VITK_EINVAL;
    }
#undef VITK_LAUNCH_NTQ
    return VITK_OK;
}
