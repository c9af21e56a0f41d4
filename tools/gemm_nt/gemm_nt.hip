// Pipelined bf16 NT GEMM for gfx950: C[m][n] = epilogue(alpha * sum_k A[m][k] * B[n][k]), both operands K-contiguous (nn.Linear forward
// as is; the input-gradient products run on transposed shadow weights, runtime.FlatParams.w16t, so they are NT as well).
//
// Round-2 cycle stamps of the register-staged kernel (gemm.hip) showed its K-tile at 4 445 cycles for 1 024 cycles of MFMA: the operand
// round trip under load (> 2.7 k cycles) is longer than one K-tile of MFMAs and a register-staged prefetch distance of more than one
// tile fits neither the register file nor the LDS next to the ds_write pass.  This kernel removes the registers from the load path:
//
//   * operands go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction, no VGPR round trip, no ds_write
//     pass, no per-thread address arithmetic: per-lane voffset is constant, the K-tile / row-piece offset is a scalar soffset);
//   * NSTAGE LDS stages form a ring; the loads of tile t + NSTAGE - 1 are issued as soon as tile t - 1's stage is free, and a wave only
//     ever waits with a COUNTED s_waitcnt vmcnt((NSTAGE - 2) * G): NSTAGE - 1 tiles stay in flight across the (raw) barrier;
//   * ONE s_barrier per K-tile: it publishes tile t (every wave has waited for its own pieces) and at the same time retires tile t - 1
//     (every wave has finished reading it), which frees that stage for the loads issued right behind the barrier;
//   * the LDS image of a tile is lane-linear per 1 KiB piece (the DMA writes base + lane * 16), so the bank-conflict swizzle sits on the
//     SOURCE side: lane l fetches the 16-byte chunk (l % CPR) ^ swz(row) of its row, and the fragment reads apply the same XOR -
//     conflict-free ds_read_b128 for both tile shapes (128-byte rows: chunk ^ ((row >> 1) & 7); 64-byte rows: chunk ^ ((row >> 2) & 3));
//   * big tiles: the CU's vector-memory path moves 64 B / clk, a 128 x 128 x 64 tile needs 512 of its 515 MFMA cycles worth of it, a
//     256 x 256 tile half - 256 x 256 (8 waves, wave tile 128 x 64) wherever the grid still fills the chip, 256 x 128 / 128 x 128 below.
//
// Epilogue: the shared fused 8-wide epilogue (gemm_epilogue.h: bias, activation, pre-activation side output, activation backward,
// dropout, output scale, residual), accumulators transposed through the (now idle) pipeline LDS so every global access is a 16-byte
// piece of a contiguous row.
#include "gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_void_ptr;

// LDS-DMA of one 1 KiB piece: lds_base (wave-uniform) + lane * 16  <-  rsrc base + voff (per lane) + soff (scalar)
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, uint16_t* lds_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)lds_base, 16, voff, soff, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int WM, int WN, int BKT, int NSTAGE, typename TC>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(GemmArgs g) {
    constexpr int NT = 64 * WM * WN, NW = WM * WN;
    constexpr int ROWB = BKT * 2;                       // bytes per tile row
    constexpr int CPR = ROWB / 16;                      // 16-byte chunks per row (8 | 4)
    constexpr int RPP = 1024 / ROWB;                    // rows per 1 KiB DMA piece (8 | 16)
    constexpr int PA = BM / RPP, PB = BN / RPP;         // pieces per stage
    constexpr int GA = PA / NW, GB = PB / NW, G = GA + GB;      // pieces per wave and stage
    static_assert(PA % NW == 0 && PB % NW == 0, "pieces must divide over the waves");
    static_assert((NW & 1) == 0, "even wave count (piece parity is wave-constant)");
    static_assert(NSTAGE >= 3 && (BKT / 16) % 2 == 0, "the one-tile-ahead pipeline needs three stages");
    constexpr int STAGE_ELEMS = (BM + BN) * BKT;        // bf16 elements per stage
    constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
    constexpr int KS = BKT / 16;                        // k-steps (MFMA K = 16) per stage
    constexpr int PC = BN + 4;                          // f32 staging pitch of the epilogue
    constexpr int EPI_ELEMS = WM * 32 * PC * 2;         // in 16-bit units
    constexpr int LDS_ELEMS = NSTAGE * STAGE_ELEMS > EPI_ELEMS ? NSTAGE * STAGE_ELEMS : EPI_ELEMS;
    __shared__ __attribute__((aligned(16))) uint16_t smem[LDS_ELEMS];          // (ONE LDS object: a second one makes hipcc drain the DMA queue)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
#ifdef GEMM_STAMPS
#define KSTAMP(k) do { if (g.stamps && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) g.stamps[(wave * 24 + 23) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KSTAMP(k) do {} while (0)
#endif
    KSTAMP(0);
    // XCD-aware tile order (see gemm.hip): XCD x owns row panels x, x + 8, ... and walks their column tiles back to back
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = lin & 7, j = lin >> 3;
        by = xcd + 8 * (j / gridDim.x);
        bx = j % gridDim.x;
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int nt = g.K / BKT;

    // ---- DMA addressing: lane-constant voffsets, scalar soffsets
    const int lr = lane / CPR, slot = lane % CPR;                               // row within the piece, LDS chunk slot
    // piece p covers tile rows p * RPP ...; p = wave + NW * q, so (p & 1) = (wave & 1): the swizzle term is lane-constant
    const int swz_row = (CPR == 8) ? ((4 * (wave & 1) + (lr >> 1)) & 7) : ((lr >> 2) & 3);
    const unsigned voffA = (unsigned)(lr * g.lda * 2 + ((slot ^ swz_row) << 4));
    const unsigned voffB = (unsigned)(lr * g.ldb * 2 + ((slot ^ swz_row) << 4));
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, 0x7fffffff, 0x00020000);
    const unsigned sA0 = (unsigned)(((long)(m0 + wave * RPP) * g.lda) * 2), sB0 = (unsigned)(((long)(n0 + wave * RPP) * g.ldb) * 2);
    const unsigned sAstep = (unsigned)(NW * RPP * g.lda * 2), sBstep = (unsigned)(NW * RPP * g.ldb * 2);
    // one 1 KiB piece of tile t into `stage`: q < GA -> rows (wave + NW q) RPP.. of A, else of B
    auto issue_piece = [&](int q, int t, int stage) {
        uint16_t* sb = smem + stage * STAGE_ELEMS;
        const unsigned kb = (unsigned)t * ROWB;
        if (q < GA) glds16(rsA, sb + (wave + NW * q) * 512, voffA, sA0 + q * sAstep + kb);
        else glds16(rsB, sb + (PA + wave + NW * (q - GA)) * 512, voffB, sB0 + (q - GA) * sBstep + kb);
    };
    auto issue = [&](int t, int stage) {
#pragma unroll
        for (int q = 0; q < G; ++q) issue_piece(q, t, stage);
    };

    // ---- fragment read addressing (element offsets inside a stage): row (lane & 31) of a 32-row fragment, chunk (2 kk + (lane >> 5)) ^ swz
    int lo[KS];
    {
        const int r = lane & 31;
        const int sw = (CPR == 8) ? ((r >> 1) & 7) : ((r >> 2) & 3);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) lo[kk] = r * BKT + (((2 * kk + (lane >> 5)) ^ sw) << 3);
    }
    constexpr int A_ROW0 = 0, B_ROW0 = BM;                                      // tile rows of the B operand follow A's

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: NSTAGE - 1 tiles in flight
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue(s, s);                           // (nt >= NSTAGE: checked by the launcher)

    // bias: a thread keeps the same 8 output columns in every epilogue slice (ordinary loads, oldest in the vmcnt queue)
    constexpr int CHN = BN / 8;                                                 // 8-column chunks per output row
    const int ch = tid % CHN, n = n0 + ch * 8;
    float bias8[8];
    if (g.bias) {
        const float4 b0 = *(const float4*)(g.bias + n), b1 = *(const float4*)(g.bias + n + 4);
        bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    }

    // Pipeline invariant at the barrier of iteration t: tile t is visible to every wave (published by the previous barrier), tile t + 1
    // has landed for the waiting wave (its counted vmcnt) and becomes visible with THIS barrier, tiles t + 2 .. t + NSTAGE - 2 are in
    // flight, tile t - 1's stage is free behind the barrier and takes tile t + NSTAGE - 1, whose pieces are issued between the MFMAs
    // of tile t.  Publishing one tile ahead lets a wave read the first fragments of tile t + 1 BEFORE the next barrier, behind its
    // last MFMAs of tile t: the matrix pipe starts on tile t + 1 right behind that barrier instead of waiting one LDS round trip with
    // both waves of the SIMD in lockstep (round-3 stamps: ~300-400 idle cycles per 1 024-cycle tile).
    int stage = 0, nstage_issue = NSTAGE - 1;                                   // stage of tile t; stage the next issue goes to
    bf16x8 fa0[FM], fb0[FN], fa1[FM], fb1[FN];                                  // two fragment register sets, alive across the tiles
    auto load_frags = [&](int stg, int kk, bf16x8 (&fa)[FM], bf16x8 (&fb)[FN]) {
        const uint16_t* tA = smem + stg * STAGE_ELEMS + (A_ROW0 + wm * FM * 32) * BKT;
        const uint16_t* tB = smem + stg * STAGE_ELEMS + (B_ROW0 + wn * FN * 32) * BKT;
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i] = *(const bf16x8*)&tA[i * 32 * BKT + lo[kk]];
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[j] = *(const bf16x8*)&tB[j * 32 * BKT + lo[kk]];
    };
    // One K-tile: fragments of k-step kk + 1 are read while the MFMAs of k-step kk run; with ISSUE the G DMA pieces of tile t_issue are
    // issued BETWEEN the MFMAs, evenly spaced.  Round-3 stamps of the first version (all pieces issued in one burst behind the barrier):
    // the CU's vector-memory path takes ~16 cycles per piece, so the burst of 32-64 pieces kept every wave out of its MFMA phase for
    // 250-1100 cycles per tile and the matrix pipe ran 53 % of the time; spread out, a piece every ~4 MFMAs is half the path's rate.
    auto compute_tile = [&](auto issue_tag, auto next_tag, int t_issue) {
        constexpr bool ISSUE = decltype(issue_tag)::value, NEXT = decltype(next_tag)::value;
        const int cur = stage, nxt = (stage + 1 == NSTAGE) ? 0 : stage + 1;
        constexpr int NMF = KS * FM * FN;                                      // MFMAs per tile and wave; piece q goes in front of MFMA q NMF / G
        auto mfmas = [&](int kk, const bf16x8 (&fa)[FM], const bf16x8 (&fb)[FN]) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int mf = (kk * FM + i) * FN + j;
                    if (ISSUE) {
#pragma unroll
                        for (int q = 0; q < G; ++q)
                            if ((q * NMF) / G == mf) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_piece(q, t_issue, nstage_issue);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
        };
#pragma unroll
        for (int kk = 0; kk < KS; kk += 2) {
            load_frags(cur, kk + 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(kk, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < KS) load_frags(cur, kk + 2, fa0, fb0);
            else if (NEXT) load_frags(nxt, 0, fa0, fb0);                        // first fragments of the NEXT tile (visible since this tile's barrier)
            __builtin_amdgcn_sched_barrier(0);
            mfmas(kk + 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage = nxt;
        nstage_issue = (nstage_issue + 1 == NSTAGE) ? 0 : nstage_issue + 1;
    };
#ifdef GEMM_STAMPS
    // probe build (tools/gemm_nt_stamps.py): s_memtime per loop phase of workgroup (0,0), all waves, first 24 K-tiles
#define NSTAMP(k) do { if (g.stamps && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && t < 24) g.stamps[(wave * 24 + t) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NSTAMP(k) do {} while (0)
#endif
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    wait_vmcnt<(NSTAGE - 2) * G>();                                             // tile 0 has landed
    __builtin_amdgcn_s_barrier();                                               // ... for every wave
    load_frags(0, 0, fa0, fb0);
    KSTAMP(1);
    // (separate loops instead of branches around the waits / the DMA: hipcc 7.2's simplifycfg crashes when it tries to merge blocks that
    //  hold inline-asm statements differing in an immediate operand, and straight-line bodies keep the placement of the pieces exact)
    int t = 0;
    for (; t + NSTAGE - 1 < nt; ++t) {                                          // steady state
        NSTAMP(0);
        wait_vmcnt<(NSTAGE - 3) * G>();                                         // tile t + 1 has landed; up to NSTAGE - 3 later tiles in flight
        NSTAMP(1);
        __builtin_amdgcn_s_barrier();                                           // tile t + 1 visible to all, tile t - 1 retired by all
        NSTAMP(2);
        compute_tile(yes(), yes(), t + NSTAGE - 1);
        NSTAMP(4);
    }
    for (; t + 1 < nt; ++t) {                                                   // tail: nothing left to issue, drain
        __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        compute_tile(no(), yes(), 0);
    }
    compute_tile(no(), no(), 0);                                                // last tile: published by the previous barrier
    __builtin_amdgcn_s_barrier();                                               // every wave is done with the last stage: LDS is free
    KSTAMP(2);

    // ---- epilogue (see gemm.hip): transposed through LDS in slices of one fragment row per wave row
    TC* C = (TC*)g.C;
    const TC* Rz = (const TC*)g.resid;
    TC* P = (TC*)g.preact;
    const TC* Xa = (const TC*)g.aux;
    const float inv_keep = g.p_drop > 0.f ? 1.0f / (1.0f - g.p_drop) : 1.0f;
    float* sC = (float*)smem;
    constexpr int ROWS_SL = WM * 32;                                            // rows per slice
    constexpr int RSTEP = NT / CHN;                                             // rows covered by one pass of the threads
    constexpr int NK = ROWS_SL / RSTEP;                                         // passes per slice
    static_assert(NT % CHN == 0 && ROWS_SL % RSTEP == 0, "epilogue mapping");
    const TC* Ex = (sizeof(TC) == 2) ? (Rz && !Xa ? Rz : (Xa && !Rz ? Xa : nullptr)) : nullptr;
    const long ldex = (Ex == Rz) ? g.ldr : g.ldc;
    const bool has_ex = Ex != nullptr;
    f8 exa[NK], exb[NK];
    auto load_ex = [&](int i, f8 (&dst)[NK]) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int r = tid / CHN + RSTEP * k;
            const int m = m0 + (r >> 5) * (FM * 32) + i * 32 + (r & 31);
            dst[k] = ld8(Ex + (long)m * ldex + n);
        }
    };
    if (has_ex) load_ex(0, exa);
    auto slice = [&](const int i, f8 (&cur)[NK], f8 (&nxt)[NK]) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *(float4*)&sC[(wm * 32 + (lane & 31)) * PC + wn * (FN * 32) + j * 32 + 8 * gq + 4 * (lane >> 5)] =
                    make_float4(acc[i][j][gq * 4 + 0], acc[i][j][gq * 4 + 1], acc[i][j][gq * 4 + 2], acc[i][j][gq * 4 + 3]);
        if (has_ex && i + 1 < FM) load_ex(i + 1, nxt);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int r = tid / CHN + RSTEP * k;                                 // wave row r >> 5, row r & 31 of fragment i
            const int m = m0 + (r >> 5) * (FM * 32) + i * 32 + (r & 31);
            f8 v;
            const float4 a0 = *(const float4*)&sC[r * PC + ch * 8], a1 = *(const float4*)&sC[r * PC + ch * 8 + 4];
            v.v[0] = g.alpha * a0.x; v.v[1] = g.alpha * a0.y; v.v[2] = g.alpha * a0.z; v.v[3] = g.alpha * a0.w;
            v.v[4] = g.alpha * a1.x; v.v[5] = g.alpha * a1.y; v.v[6] = g.alpha * a1.z; v.v[7] = g.alpha * a1.w;
            epilogue8<TC, false>(g, v, 0, m, n, C, Rz, P, Xa, nullptr, nullptr, bias8, true, inv_keep, has_ex, cur[k]);
        }
        if (i + 1 < FM) __syncthreads();
    };
    slice(0, exa, exb);
    if constexpr (FM >= 2) slice(1, exb, exa);
    if constexpr (FM >= 4) { slice(2, exa, exb); slice(3, exb, exa); }
    KSTAMP(3);
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
// Tile configurations (K-tile ring of >= 3 stages):
//   0: 256 x 256, 8 waves (2 x 4), BK 32, 4 stages (128 KiB)      2: 256 x 128, 8 waves (4 x 2), BK 64, 3 stages (144 KiB)
//   3: 128 x 128, 4 waves (2 x 2), BK 32, 4 stages (64 KiB: 2 workgroups / CU)
//   6: 256 x 256, 4 waves (2 x 2, one wave per SIMD, 128 x 128 per wave, accumulators in AGPRs), BK 32, 4 stages - probe only
// Measured on MI355X (tools/gemm_nt_sweep.py, M = 16384, round 3) against the register-staged kernel of gemm.hip ("old") and the library:
//   N x K        old     cfg0    cfg2    library        N x K        old     cfg0    cfg2    library
//   1024 x 3072  105 us   96      103     75 us          512 x 2048  45.7 us  60      39      37 us
//    768 x 3072   94      86       97     69             512 x 1024  26.4     39      24      23
//   3072 x 1024  121     126      128     85            2048 x  512  54.6     62      57      42
//   3072 x  768  105     107      109     69            1536 x  512  43.3     55      44      42
// and where its cycles go (tools/gemm_nt_stamps.py, 256 x 256 tile): a 32-deep K-tile takes ~1 640 cycles for 1 024 cycles of MFMA per
// SIMD (the two co-resident waves of a SIMD run their MFMA phases one after the other: the older wave's DMA issue and LDS waits are not
// filled by the younger one's MFMAs), and the epilogue of a 256 x 256 tile is ~20 000 cycles - every workgroup of the single round writes
// its 128 KiB at the same time while the matrix cores idle; with K = 512 that is 40 % of a workgroup's life.  So the deep pipeline pays
// only for LONG K; short-K products are bound by output traffic that many small co-resident workgroups (gemm.hip: 2-3 per CU, desynchronised)
// overlap with each other's K loops.  The launcher therefore takes this kernel for K >= 2048, or K >= 1024 with N <= 512.
template <typename TC>
static int launch_cfg(const GemmArgs& g, int cfg, hipStream_t st) {
    switch (cfg) {
        case 0: gemm_nt_kernel<256, 256, 2, 4, 32, 4, TC><<<dim3(g.N / 256, g.M / 256), 512, 0, st>>>(g); break;
        case 2: gemm_nt_kernel<256, 128, 4, 2, 64, 3, TC><<<dim3(g.N / 128, g.M / 256), 512, 0, st>>>(g); break;
        case 3: gemm_nt_kernel<128, 128, 2, 2, 32, 4, TC><<<dim3(g.N / 128, g.M / 128), 256, 0, st>>>(g); break;
        case 6: gemm_nt_kernel<256, 256, 2, 2, 32, 4, TC><<<dim3(g.N / 256, g.M / 256), 256, 0, st>>>(g); break;
        default: return 1;
    }
    SARSSL_CHECK_LAUNCH("gemm_nt_kernel");
    return 0;
}

static bool cfg_fits(const GemmArgs& g, int cfg) {
    int bm, bn, bk, ns;
    switch (cfg) {
        case 0: case 6: bm = 256; bn = 256; bk = 32; ns = 4; break;
        case 2: bm = 256; bn = 128; bk = 64; ns = 3; break;
        case 3: bm = 128; bn = 128; bk = 32; ns = 4; break;
        default: return false;
    }
    return g.M % bm == 0 && g.N % bn == 0 && g.K % bk == 0 && g.K >= bk * ns;
}

// sarssl_gemm's fast path: returns 0 when the product was launched, 1 when the shape / epilogue is not this kernel's (the caller then
// takes the general register-staged kernel), < 0 on errors.  dtC: SARSSL_BF16 | SARSSL_F32.
int sarssl_gemm_nt_try(const GemmArgs& g, int dtC, void* stream) {
    static const int enabled = getenv("SARSSL_GEMM_NT") ? atoi(getenv("SARSSL_GEMM_NT")) : 1;          // A/B switch: 0 = general kernel everywhere
    static const int forced = getenv("SARSSL_GEMM_NT_CFG") ? atoi(getenv("SARSSL_GEMM_NT_CFG")) : -1;  // tools/gemm_nt_sweep.py: one configuration for every shape it fits
    if (!enabled) return 1;
    if (g.split_k > 0 || g.row_shift != 0 || g.acc_in || g.acc_out || g.partA || g.partB) return 1;
    if ((g.N & 7) || (g.ldc & 7) || (g.resid && (g.ldr & 7)) || (g.lda & 7) || (g.ldb & 7)) return 1;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((uintptr_t)g.C & 15)) return 1;
    if ((long)g.M * g.lda * 2 >= 0x7fffffffL || (long)g.N * g.ldb * 2 >= 0x7fffffffL) return 1;      // 32-bit buffer offsets
    int cfg = -1;
    if (forced >= 0) { if (cfg_fits(g, forced)) cfg = forced; }
    else if (g.K >= 2048 || (g.K >= 1024 && g.N <= 512)) {
        const long cus = sarssl_cu_count();
        const long t256 = (long)(g.M / 256) * (g.N / 256), t2128 = (long)(g.M / 256) * (g.N / 128);
        if (cfg_fits(g, 0) && t256 * 10 >= cus * 7) cfg = 0;
        else if (cfg_fits(g, 2) && t2128 * 10 >= cus * 9) cfg = 2;
    }
    if (cfg < 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtC == SARSSL_BF16) return launch_cfg<bf16>(g, cfg, st);
    if (dtC == SARSSL_F32) return launch_cfg<float>(g, cfg, st);
    return 1;
}
