// Batched bf16-MFMA GEMM with fused epilogue for gfx950.
//
//   C[z][m][n] = epilogue( alpha * sum_k opA(A)[z][m][k] * opB(B)[z][n][k] )
//
// Layout flags say which dimension of each operand is contiguous in memory:
//   a_kc = 1: A is [M][K] (K contiguous, row stride lda);   a_kc = 0: A is [K][M] (M contiguous)
//   b_kc = 1: B is [N][K] (nn.Linear weight layout);          b_kc = 0: B is [K][N]
// so Linear forward is (1,1), dX = dY*W is (1,0), dW = dY^T*X is (0,0) with A=dY, B=X.
//
// Storage types: f32, bf16 or fp16 per tensor.  MFMA inputs are bf16 (v_mfma_f32_32x32x16_bf16, f32 accumulate) - or fp16
// (v_mfma_f32_32x32x16_f16, same rate) when BOTH operands are stored in fp16: the forward products of the fp16-forward mode.  For f32 storage the "precise" mode splits every operand x = hi + lo
// (hi = bf16(x), lo = bf16(x - hi)) at LDS-staging time and the host runs three passes
// (hi*hi, hi*lo, lo*hi) accumulating in an f32 workspace: ~2^-16 relative error per product,
// on the same matrix cores and the same code path as the fast bf16 mode.
//
// Tiling: (64*FM) x 128 x 64 per 256-thread workgroup, 4 waves as 2 x 2; each wave owns FM x 2 MFMA fragments of 32 x 32
// (FM = 4: 256 x 128 tiles, 128 x 64 per wave, 128 accumulator registers; FM = 2: 128 x 128 tiles for grids that would not fill
// the chip with the large tile).  Round-2 counters (profiles/r02_gemm_before_counters.json) showed the round-1 kernel issue-bound,
// not LDS- or HBM-bound: per K-tile a wave spent ~400 cycles issuing address arithmetic / bounds checks / staging instructions
// next to 512 cycles of MFMA, and the fused epilogue another ~14 k cycles per 128 x 128 tile.  Hence: twice the MFMAs per staged
// byte and per barrier (wave tile 128 x 64), per-thread base pointers advanced by a constant per K-tile, no bounds logic unless the
// shape is ragged (EDGE instantiation), and an epilogue whose row / column bookkeeping is hoisted out of the element loop.
// K-contiguous operands are staged through registers into LDS as [row][k] with a 144-byte pitch (conflict-free ds_read_b128
// fragment reads); operands whose contraction index is the SLOW memory dimension (weight gradients, dX = dY*W) are staged
// untransposed as [k][row] (pitch rows + 32 elements) and their MFMA fragments are fetched with the gfx950 transpose read
// ds_read_b64_tr_b16 - no transposition pass, no bank conflicts.  The next K-tile is always prefetched into registers behind the
// MFMAs.  Operand roles are swapped in the MFMA (weights as the "A"/row operand) so each lane ends up with 4 consecutive n for one
// m; the epilogue transposes the accumulators through LDS so every thread stores 16-byte pieces of contiguous rows.
#include "gemm_epilogue.h"
#include <stdlib.h>

#define BN 128
#define BK 64
#define PITCH (BK + 8)

// MFMA operand type of a product: fp16 when both operands are stored in fp16 (the forward GEMMs of the fp16-forward mode), bf16 otherwise -
// f32 operands are split into bf16 parts, and a saved fp16 activation that meets a bf16 gradient (weight-gradient / unfused attention
// backward products of the mixed mode) is re-encoded to bf16 while it is staged (load_chunk)
template <typename TA, typename TB> struct MfmaT { typedef bf16 type; };
template <> struct MfmaT<f16, f16> { typedef f16 type; };
template <typename T, typename TM>
__device__ __forceinline__ uint4 load_chunk(const T* __restrict__ p, int part) {
    if constexpr (sizeof(T) == 2) return recode8<T, TM>(*(const uint4*)p);
    else { const f8 v = ld8(p); return pack8_part(v, part); }
}

// Register-staged operand tile: ROWS x 64 elements, 16-byte chunks, NCH = ROWS / 32 chunks per thread.
//   KC  ([row][k] in memory):  chunk i of thread t = row (t >> 3) + 32 i, k-chunk t & 7
//   !KC ([k][row] in memory):  CPR = ROWS / 8 chunks per k-row; chunk i of thread t = k (t / CPR) + i * (256 / CPR), row-chunk t % CPR
// `p` points at this thread's chunk 0 of the current K-tile; EDGE instantiations zero-fill rows >= R and k >= Kend.
template <typename T, typename TM, bool KC, int ROWS, bool EDGE>
__device__ __forceinline__ void tile_load(const T* __restrict__ p, long ld, int r0, int k0, int R, int Kend, int part, int tid,
                                          uint4 (&regs)[ROWS / 32]) {
    constexpr int CPR = ROWS / 8, KPP = 256 / CPR;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const T* q = KC ? p + (long)(32 * i) * ld : p + (long)(KPP * i) * ld;
        if constexpr (EDGE) {
            const int gr = KC ? r0 + (tid >> 3) + 32 * i : r0 + (tid % CPR) * 8;
            const int gk = KC ? k0 + (tid & 7) * 8 : k0 + tid / CPR + KPP * i;
            regs[i] = (gr < R && gk < Kend) ? load_chunk<T, TM>(q, part) : make_uint4(0, 0, 0, 0);
        } else regs[i] = load_chunk<T, TM>(q, part);
    }
}
template <bool KC, int ROWS>
__device__ __forceinline__ void tile_store(uint16_t* __restrict__ s, int tid, const uint4 (&regs)[ROWS / 32]) {
    constexpr int CPR = ROWS / 8, KPP = 256 / CPR, PT = ROWS + 32;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        if (KC) *(uint4*)&s[((tid >> 3) + 32 * i) * PITCH + (tid & 7) * 8] = regs[i];
        else *(uint4*)&s[(tid / CPR + KPP * i) * PT + (tid % CPR) * 8] = regs[i];
    }
}

// MFMA fragment (8 consecutive k for row r0 + (lane&31)) from a [k][row] tile (pitch PT) via two transpose reads
template <int PT>
__device__ __forceinline__ bf16x8 frag_tr(const uint16_t* s, int kbase, int r0, int lane) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const int col = r0 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
    union { s16x4 v[2]; bf16x8 b; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = kbase + (lane >> 5) * 8 + h * 4 + ((lane & 15) >> 2);
        u.v[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + k * PT + col));
    }
    return u.b;
}

// NSEG (sarssl_gemm_split; NT layout, fp16 operands): 2 = A B^T + A B2^T, 3 = A B^T + A2 B^T + A B2^T - the lo tiles (A2, B2) are staged
// next to the hi tiles and every fragment read feeds two MFMAs (fp16 pairs of an f32 activation / weight: hi hi + lo hi + hi lo)
template <typename TA, typename TB, typename TC, bool AKC, bool BKC, int FM, bool EDGE, bool CSUM = false, int NSEG = 0>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bid_x, const int bid_y, const int bid_z, const int grid_x,
                                          const int grid_y, const int grid_z) {
    typedef typename MfmaT<TA, TB>::type TM;
    static_assert(sizeof(TA) == 2 || (sizeof(TA) == 4 && sizeof(TB) == 4), "f32 operands come in pairs (split-bf16 passes)");
    static_assert(NSEG == 0 || ((NSEG == 2 || NSEG == 3) && AKC && BKC && sizeof(TA) == 2 && sizeof(TB) == 2 && FM <= 2), "pair products: NT layout, 16-bit operands");
    constexpr int BM = 64 * FM;
    constexpr int PTA = BM + 32, PTB = BN + 32;
    constexpr int A_ELEMS = AKC ? BM * PITCH : BK * PTA;
    constexpr int B_ELEMS = BKC ? BN * PITCH : BK * PTB;
    constexpr int STAGE = A_ELEMS + B_ELEMS + (NSEG >= 2 ? B_ELEMS : 0) + (NSEG == 3 ? A_ELEMS : 0);
    constexpr int PC = BN + 4;                                  // f32 staging pitch of the epilogue (conflict-free 16-byte LDS writes)
    constexpr int EPI_ELEMS = 64 * PC * 2;                      // 64 x 132 f32, in 16-bit units
    constexpr int LDS_ELEMS = STAGE > EPI_ELEMS ? STAGE : EPI_ELEMS;      // (a double-buffered variant - two LDS stages, two register
                                                                           //  stages - measured neutral inside the step: NOTES.md 4.2)
    __shared__ __attribute__((aligned(16))) uint16_t smem[LDS_ELEMS];    // FM = 2: 2 x 32-37 KiB (2 workgroups / CU), FM = 4: 56 KiB (2 / CU)
    uint16_t* sA = smem;
    uint16_t* sB = smem + A_ELEMS;
    uint16_t* sB2 = sB + B_ELEMS;                 // NSEG >= 2: the lo tile of B
    uint16_t* sA2 = sB2 + B_ELEMS;                // NSEG == 3: the lo tile of A
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nsplit = g.split_k > 0 ? g.split_k : 1;
    // Split-K products whose slice count is a multiple of 8 (one batch matrix): slice-major order - workgroup `lin` (the hardware
    // dispatches linear id % 8 to XCD % 8) takes K-slice lin % nsplit of tile lin / nsplit, so XCD x only ever touches the K-slices
    // congruent to x and each XCD's L2 reads ITS eighth of both operands once.  With the row-panel order below every XCD streams the
    // whole B operand: the round-2 counters show 203 MB fetched for 84 MB of operands on the FFN weight gradient and 862 MB for
    // 134 MB on the decoder's - those launches were HBM-bound at 4.4-5.8 TB/s of mostly repeated reads.
    const bool slice_major = g.split_k > 0 && ((nsplit & 7) == 0 || nsplit == 4 || nsplit == 2) && grid_z == nsplit;   // (2, 4: two / four XCDs share a slice)
    const int lin_all = (bid_z * grid_y + bid_y) * grid_x + bid_x;
    const int z = slice_major ? 0 : bid_z / nsplit, ks = slice_major ? lin_all % nsplit : bid_z % nsplit;
    const int z0 = z / g.batch_inner, z1 = z % g.batch_inner;
    const int k_begin = g.split_k > 0 ? ks * g.k_per_split : 0;
    const int k_end = g.split_k > 0 ? min(g.K, k_begin + g.k_per_split) : g.K;
    // XCD-aware tile order.  Workgroups are dispatched round-robin over the 8 XCDs (linear id % 8), each with its own L2: in the
    // natural order the column tiles of one A row panel land on different XCDs and the panel is fetched from HBM by every one of
    // them.  Remap so that XCD x owns row panels x, x+8, ... and walks all their column tiles back to back - the A panel is then
    // served by that XCD's L2.
    int bx = bid_x, by = bid_y;
    if (slice_major) {
        const int tile = lin_all / nsplit;
        bx = tile % grid_x; by = tile / grid_x;
    } else if ((grid_y & 7) == 0) {
        const int lin = bid_y * grid_x + bid_x;
        const int xcd = lin & 7, j = lin >> 3;
        by = xcd + 8 * (j / grid_x);
        bx = j % grid_x;
    }
    const int m0 = by * BM, n0 = bx * BN;
    // this thread's chunk 0 of the first K-tile; advanced by a constant per K-tile
    constexpr int CPRA = BM / 8, CPRB = BN / 8;
    const TA* pa = (const TA*)g.A + z0 * g.sA0 + z1 * g.sA1 +
                   (AKC ? (long)(m0 + (tid >> 3)) * g.lda + k_begin + (tid & 7) * 8 : (long)(k_begin + tid / CPRA) * g.lda + m0 + (tid % CPRA) * 8);
    const TB* pb = (const TB*)g.B + z0 * g.sB0 + z1 * g.sB1 +
                   (BKC ? (long)(n0 + (tid >> 3)) * g.ldb + k_begin + (tid & 7) * 8 : (long)(k_begin + tid / CPRB) * g.ldb + n0 + (tid % CPRB) * 8);
    const long stepA = AKC ? BK : (long)BK * g.lda, stepB = BKC ? BK : (long)BK * g.ldb;

    f32x16 acc[FM][2];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // CSUM (grouped weight-gradient launch): the column sums of the A operand over this K-slice - the bias gradient that autograd
    // computes next to every nn.Linear weight gradient - ride on the matrix cores: one extra MFMA per A fragment against an all-ones
    // fragment (every row of the 32 x 32 result then holds sum_k A[m][k]), in the workgroups of column tile 0 only.  Round 2 read every dY
    // a second time for these sums (colsum_multi: 235 us per step inside the graph).
    // ONE extra accumulator for both A fragments (FM = 2): the ones sit in rows 0-15 of the selector fragment used with fragment 0 and in
    // rows 16-31 of the one used with fragment 1, so rows 0-15 of the result collect fragment 0's sums and rows 16-31 fragment 1's
    // (two accumulators did not fit next to the 3-waves-per-SIMD register cap: 108 bytes of scratch per lane).
    static_assert(!CSUM || FM == 2, "column-sum epilogue: two A fragments share one accumulator");
    const bool do_cs = CSUM && g.csum_ws != nullptr && bx == 0 && wn == 0;
    f32x16 accs;
#pragma unroll
    for (int r = 0; r < 16; ++r) accs[r] = 0.f;
    // lane = row (lane & 31) of the selector: all ones (0x3F80 pairs) or all zeros; fragment 1's selector is its complement
    const uint32_t selw = ((lane & 31) < 16) ? 0x3F803F80u : 0u;

    // epilogue operands that do not depend on the accumulators are requested early: a thread keeps the same 8 output columns in
    // every epilogue slice, so its 8 bias values are loaded here, behind the whole K loop
    const int ch = tid & 15, n = n0 + ch * 8;
    float bias8[8];
    if (g.bias && (!EDGE || n + 8 <= g.N)) {
        const float4 b0 = *(const float4*)(g.bias + n), b1 = *(const float4*)(g.bias + n + 4);
        bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[e] = (g.bias && n + e < g.N) ? g.bias[n + e] : 0.f;
    }

    // MFMA phase of one K-tile, software-pipelined by hand: the fragments of k-step kk+1 are read from LDS BEFORE the MFMAs of step kk
    // are issued (two register sets).  Left alone, hipcc sinks every ds_read next to its first use ("2 reads, wait, 2 MFMAs"), so the
    // matrix pipe idles for one LDS round trip per pair of MFMAs; the sched_barriers pin the order.
    auto mfma_phase = [&](const uint16_t* tA, const uint16_t* tB) {
        auto load_frags = [&](int kk, bf16x8 (&fa)[FM], bf16x8 (&fb)[2]) {
            const int koff = kk * 16 + (lane >> 5) * 8;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if (AKC) fa[i] = *(const bf16x8*)&tA[(wm * (FM * 32) + i * 32 + (lane & 31)) * PITCH + koff];
                else fa[i] = frag_tr<PTA>(tA, kk * 16, wm * (FM * 32) + i * 32, lane);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (BKC) fb[j] = *(const bf16x8*)&tB[(wn * 64 + j * 32 + (lane & 31)) * PITCH + koff];
                else fb[j] = frag_tr<PTB>(tB, kk * 16, wn * 64 + j * 32, lane);
            }
        };
        auto mfmas = [&](const bf16x8 (&fa)[FM], const bf16x8 (&fb)[2]) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = mfma16<TM>(fb[j], fa[i], acc[i][j]);
            if constexpr (CSUM) {
                if (do_cs) {
                    union { uint32_t w[4]; bf16x8 b; } s0, s1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s0.w[e] = selw; s1.w[e] = selw ^ 0x3F803F80u; }
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s0.b, fa[0], accs, 0, 0, 0);
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s1.b, fa[FM - 1], accs, 0, 0, 0);
                }
            }
        };
        bf16x8 fa0[FM], fb0[2], fa1[FM], fb1[2];
        if (g.prio) __builtin_amdgcn_s_setprio(2);      // over the co-resident workgroups' waves that are staging / storing
        load_frags(0, fa0, fb0);
#pragma unroll
        for (int kk = 0; kk < BK / 16; kk += 2) {
            load_frags(kk + 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < BK / 16) load_frags(kk + 2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g.prio) __builtin_amdgcn_s_setprio(0);
    };

    if constexpr (NSEG >= 2) {
        // pair products: per K-tile the hi and lo tiles of both operands are staged once; a k-step reads FM (+ FM) A fragments and 2 + 2
        // B fragments for 2 FM x NSEG MFMAs (hi hi, hi lo_B, lo_A hi) - two thirds of the fragment reads, one third of the barriers and
        // tile stores per MFMA of a plain product
        const TA* pa2 = (const TA*)g.A2 + (pa - (const TA*)g.A);
        const TB* pb2 = (const TB*)g.B2 + (pb - (const TB*)g.B);
        uint4 ra[BM / 32], rb[BN / 32], rb2[BN / 32], ra2[NSEG == 3 ? BM / 32 : 1];
        auto load_all = [&](int k0) {
            tile_load<TA, TM, AKC, BM, EDGE>(pa, g.lda, m0, k0, g.M, k_end, 0, tid, ra);
            tile_load<TB, TM, BKC, BN, EDGE>(pb, g.ldb, n0, k0, g.N, k_end, 0, tid, rb);
            tile_load<TB, TM, BKC, BN, EDGE>(pb2, g.ldb, n0, k0, g.N, k_end, 0, tid, rb2);
            if constexpr (NSEG == 3) tile_load<TA, TM, AKC, BM, EDGE>(pa2, g.lda, m0, k0, g.M, k_end, 0, tid, ra2);
        };
        auto phase = [&]() {
            auto load_frags = [&](int kk, bf16x8 (&fa)[FM], bf16x8 (&fa2)[FM], bf16x8 (&fb)[2], bf16x8 (&fb2)[2]) {
                const int koff = kk * 16 + (lane >> 5) * 8;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int o = (wm * (FM * 32) + i * 32 + (lane & 31)) * PITCH + koff;
                    fa[i] = *(const bf16x8*)&sA[o];
                    if constexpr (NSEG == 3) fa2[i] = *(const bf16x8*)&sA2[o];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int o = (wn * 64 + j * 32 + (lane & 31)) * PITCH + koff;
                    fb[j] = *(const bf16x8*)&sB[o];
                    fb2[j] = *(const bf16x8*)&sB2[o];
                }
            };
            auto mfmas = [&](const bf16x8 (&fa)[FM], const bf16x8 (&fa2)[FM], const bf16x8 (&fb)[2], const bf16x8 (&fb2)[2]) {
                // (one pass over the accumulators per product type: consecutive MFMAs never wait on each other's result)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<TM>(fb[j], fa[i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<TM>(fb2[j], fa[i], acc[i][j]);
                if constexpr (NSEG == 3) {
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<TM>(fb[j], fa2[i], acc[i][j]);
                }
            };
            bf16x8 fa0[FM], fa20[FM], fb0[2], fb20[2], fa1[FM], fa21[FM], fb1[2], fb21[2];
            if (g.prio) __builtin_amdgcn_s_setprio(2);
            load_frags(0, fa0, fa20, fb0, fb20);
#pragma unroll
            for (int kk = 0; kk < BK / 16; kk += 2) {
                load_frags(kk + 1, fa1, fa21, fb1, fb21);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(fa0, fa20, fb0, fb20);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 2 < BK / 16) load_frags(kk + 2, fa0, fa20, fb0, fb20);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(fa1, fa21, fb1, fb21);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g.prio) __builtin_amdgcn_s_setprio(0);
        };
        load_all(k_begin);
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            tile_store<AKC, BM>(sA, tid, ra);
            tile_store<BKC, BN>(sB, tid, rb);
            tile_store<BKC, BN>(sB2, tid, rb2);
            if constexpr (NSEG == 3) tile_store<AKC, BM>(sA2, tid, ra2);
            __syncthreads();
            if (k0 + BK < k_end) {
                pa += stepA; pb += stepB; pa2 += stepA; pb2 += stepB;
                load_all(k0 + BK);
            }
            phase();
            __syncthreads();
        }
    } else {
        uint4 ra[BM / 32], rb[BN / 32];
        tile_load<TA, TM, AKC, BM, EDGE>(pa, g.lda, m0, k_begin, g.M, k_end, g.partA, tid, ra);
        tile_load<TB, TM, BKC, BN, EDGE>(pb, g.ldb, n0, k_begin, g.N, k_end, g.partB, tid, rb);
#ifdef GEMM_STAMPS
#define GSTAMP(k) do { if (g.stamps && bid_x == 0 && bid_y == 0 && bid_z == 0 && lane == 0 && kt < 16) g.stamps[((tid >> 6) * 16 + kt) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
        int kt = 0;
#else
#define GSTAMP(k) do {} while (0)
#endif
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            GSTAMP(0);
            tile_store<AKC, BM>(sA, tid, ra);
            tile_store<BKC, BN>(sB, tid, rb);
            GSTAMP(1);
            __syncthreads();
            GSTAMP(2);
            if (k0 + BK < k_end) {                           // next K-tile in flight behind the MFMAs
                pa += stepA; pb += stepB;
                tile_load<TA, TM, AKC, BM, EDGE>(pa, g.lda, m0, k0 + BK, g.M, k_end, g.partA, tid, ra);
                tile_load<TB, TM, BKC, BN, EDGE>(pb, g.ldb, n0, k0 + BK, g.N, k_end, g.partB, tid, rb);
            }
            GSTAMP(3);
            mfma_phase(sA, sB);
            GSTAMP(4);
            __syncthreads();
            GSTAMP(5);
#ifdef GEMM_STAMPS
            ++kt;
#endif
        }
    }

    if constexpr (CSUM) {
        if (do_cs && lane < 32) {                       // column lane of the result = row m of A; register 0 = result row 0 (fragment 0's
            float* q = g.csum_ws + (long)ks * g.M + m0 + wm * (FM * 32) + lane;     // sums), register 8 = result row 16 (fragment 1's)
            q[0] = accs[0];
            q[32] = accs[8];
        }
    }
    // ---- epilogue.  The MFMA leaves each lane with 4 consecutive n for ONE row m (32 different rows per wave instruction):
    // storing that directly is store-issue bound (every instruction touches 32 cache lines).  Instead the f32 accumulators are
    // transposed through LDS, 64 rows at a time (fragment row i of both wave rows), so each thread owns 8 consecutive columns of
    // one row: bias / activation / dropout / residual are applied on 8-wide vectors and every global access is a full 16/32-byte
    // piece of a contiguous row.  A thread keeps the same 8 columns in every slice, so its bias values are loaded once.
    TC* C = (TC*)g.C + z0 * g.sC0 + z1 * g.sC1;
    const TC* Rz = g.resid ? (const TC*)g.resid + z0 * g.sR0 + z1 * g.sR1 : nullptr;
    TC* P = g.preact ? (TC*)g.preact + z0 * g.sC0 + z1 * g.sC1 : nullptr;
    const TC* Xa = g.aux ? (const TC*)g.aux + z0 * g.sC0 + z1 * g.sC1 : nullptr;
    float* W = g.acc_ws ? g.acc_ws + (long)z * g.M * g.N : nullptr;
    float* Wp = (g.split_k > 0) ? g.acc_ws + ((long)z * nsplit + ks) * g.M * g.N : nullptr;
    const bool vec_ok = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && (!g.resid || (g.ldr & 7) == 0);
    DropCtx dc;
    dc.init(g);
    float* sC = (float*)smem;
    // residual (forward GEMMs) or saved pre-activation (activation-backward GEMMs) rows of a slice are requested one slice ahead, so
    // their latency hides behind the LDS transposition and the arithmetic of the previous slice instead of stalling every tile
    // (bf16 outputs only: with f32 outputs the two register sets did not survive hipcc's register allocation - they went to scratch)
    const TC* Ex = (!EDGE && sizeof(TC) == 2 && g.split_k <= 0 && !g.acc_out) ? (Rz && !Xa ? Rz : (Xa && !Rz ? Xa : nullptr)) : nullptr;
    const long ldex = (Ex == Rz) ? g.ldr : g.ldc;
    const bool ex_f16 = g.aux_f16 && Ex == Xa && Ex != nullptr;      // the prefetched rows are the saved fp16 pre-activation
    // (two register sets addressed with compile-time indices only: an array that is copied / passed by pointer here ends up in
    //  scratch memory - 144 bytes per lane of private memory traffic per tile, which tripled the HBM writes of the residual /
    //  activation-backward GEMMs in the round-2 counters)
    f8 exa[4], exb[4];
    auto load_ex = [&](int i, f8 (&dst)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (tid >> 4) + 16 * k;
            const int m = m0 + (r >> 5) * (FM * 32) + i * 32 + (r & 31);
            dst[k] = ld8_as(Ex + (long)m * ldex + n, ex_f16);
        }
    };
    const bool has_ex = !EDGE && sizeof(TC) == 2 && Ex != nullptr;
    if (has_ex) load_ex(0, exa);
    auto slice = [&](const int i, f8 (&cur)[4], f8 (&nxt)[4]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *(float4*)&sC[(wm * 32 + (lane & 31)) * PC + wn * 64 + j * 32 + 8 * gq + 4 * (lane >> 5)] =
                    make_float4(acc[i][j][gq * 4 + 0], acc[i][j][gq * 4 + 1], acc[i][j][gq * 4 + 2], acc[i][j][gq * 4 + 3]);
        if (has_ex && i + 1 < FM) load_ex(i + 1, nxt);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (tid >> 4) + 16 * k;                   // 0..63: wave row r >> 5, row r & 31 of fragment i
            const int m = m0 + (r >> 5) * (FM * 32) + i * 32 + (r & 31);
            if (EDGE && (m >= g.M || n >= g.N)) continue;
            f8 v;
            const float4 a0 = *(const float4*)&sC[r * PC + ch * 8], a1 = *(const float4*)&sC[r * PC + ch * 8 + 4];
            v.v[0] = g.alpha * a0.x; v.v[1] = g.alpha * a0.y; v.v[2] = g.alpha * a0.z; v.v[3] = g.alpha * a0.w;
            v.v[4] = g.alpha * a1.x; v.v[5] = g.alpha * a1.y; v.v[6] = g.alpha * a1.z; v.v[7] = g.alpha * a1.w;
            epilogue8<TC, EDGE>(g, v, z, m, n, C, Rz, P, Xa, W, Wp, bias8, vec_ok, dc, has_ex, cur[k]);
        }
        if (i + 1 < FM) __syncthreads();
    };
    slice(0, exa, exb);
    if constexpr (FM >= 2) slice(1, exb, exa);
    if constexpr (FM >= 4) { slice(2, exa, exb); slice(3, exb, exa); }
}

#ifdef GEMM_STAMPS
static unsigned long long* g_gemm_stamps_host = nullptr;
extern "C" int sarssl_gemm_stamp_buffer(void* p) { g_gemm_stamps_host = (unsigned long long*)p; return 0; }
#endif
template <typename TA, typename TB, typename TC, bool AKC, bool BKC, int FM, bool EDGE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FM == 4 ? 2 : 3))) void gemm_kernel(GemmArgs g) {
    gemm_body<TA, TB, TC, AKC, BKC, FM, EDGE>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z);
}
// pair products (sarssl_gemm_split): fp16 operands, NT layout, NSEG = 2 | 3 (see gemm_body); two workgroups per CU (55-74 KB of LDS)
template <typename TC, int FM, bool EDGE, int NSEG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_seg_kernel(GemmArgs g) {
    gemm_body<f16, f16, TC, true, true, FM, EDGE, false, NSEG>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z);
}

// ---- grouped launch: up to GROUP_MAXP independent split-K weight-gradient products (A = dY [K][M], B = X [K][N], both with the
// contraction index slow: layout (0,0); bf16 in, f32 partial sums out) in ONE grid.  The ~9 products of a Conformer block's backward
// are each too small to fill the chip (82-440 TFLOP/s alone, ~14 us of launch / fill / drain per launch); together they are one
// launch of a few thousand workgroups.  Each product keeps its own tile grid (first[q] .. first[q+1]) and its own partial buffer,
// folded into the gradient buffers afterwards by sarssl_splitk_reduce_multi.
#define GROUP_MAXP 12
struct GemmGroup {
    GemmArgs p[GROUP_MAXP];                      // (12 x 272 bytes of kernel arguments)
    int gx[GROUP_MAXP], gy[GROUP_MAXP], first[GROUP_MAXP + 1];
    int n;
};
template <typename TB>       // TB = f16: the saved activations of the fp16-forward mode, re-encoded to bf16 while staged
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void gemm_group_tn_kernel(GemmGroup a) {
    int q = 0;
    while (q + 1 < a.n && (int)blockIdx.x >= a.first[q + 1]) ++q;
    const int local = blockIdx.x - a.first[q];
    const int gx = a.gx[q], gy = a.gy[q], ns = a.p[q].split_k;
    if (local >= gx * gy * ns) return;               // (first[] is padded to multiples of 8: linear id % 8 stays the XCD inside a product)
    const int bz = local / (gx * gy), rem = local - bz * (gx * gy);
    gemm_body<bf16, TB, float, false, false, 2, false, true>(a.p[q], rem % gx, rem / gx, bz, gx, gy, ns);
}

// C[z][m][n] += sum_s ws[z][s][m][n]   (second stage of split-K weight-gradient GEMMs; C is f32)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int nsplit, int M, int N, float* __restrict__ C, long ldc,
                                     int batch_inner, long sC0, long sC1) {
    const long mn = (long)M * N;
    const int z = blockIdx.y;
    float* Cz = C + (z / batch_inner) * sC0 + (z % batch_inner) * sC1;
    const float* wz = ws + (long)z * nsplit * mn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += wz[(long)k * mn + i];
        const long m = i / N, n = i - m * N;
        Cz[m * ldc + n] += s;
    }
}

// The same reduction for several split-K products in one launch (the weight-gradient GEMMs of one backward stage are reduced
// together when the stage ends: 40 launches of ~12 us per step otherwise).
#define SPLITK_MAXP 32
struct SplitKMulti {
    const float* ws[SPLITK_MAXP]; float* C[SPLITK_MAXP]; long ldc[SPLITK_MAXP];
    int nsplit[SPLITK_MAXP], M[SPLITK_MAXP], N[SPLITK_MAXP], first[SPLITK_MAXP + 1];
    int n;
};
__global__ void splitk_reduce_multi_kernel(SplitKMulti a) {
    int q = 0;
    while (q + 1 < a.n && (int)blockIdx.x >= a.first[q + 1]) ++q;
    const int nblk = a.first[q + 1] - a.first[q], local = blockIdx.x - a.first[q];
    const long mn = (long)a.M[q] * a.N[q];
    const float* wz = a.ws[q];
    float* Cz = a.C[q];
    const int N = a.N[q], nsplit = a.nsplit[q];
    const long ldc = a.ldc[q];
    for (long i = ((long)local * blockDim.x + threadIdx.x) * 4; i < mn; i += (long)nblk * blockDim.x * 4) {      // N % 4 == 0
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < nsplit; ++k) {
            const float4 v = *(const float4*)(wz + (long)k * mn + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const long m = i / N, n = i - m * N;
        float4* dst = (float4*)(Cz + m * ldc + n);
        float4 c = *dst;
        c.x += s.x; c.y += s.y; c.z += s.z; c.w += s.w;
        *dst = c;
    }
}
// C_q[m][n] += sum_s ws_q[s][m][n] for n_prob <= 32 products (also the bias-gradient column sums' slices: M = 1) (N_q % 4 == 0, ldc_q % 4 == 0)
extern "C" int sarssl_splitk_reduce_multi(const float* const* ws, const int* nsplit, const int* M, const int* N, float* const* C,
                                          const long* ldc, int n_prob, void* stream) {
    SARSSL_REQUIRE(n_prob > 0 && n_prob <= SPLITK_MAXP, "sarssl_splitk_reduce_multi");
    SplitKMulti a;
    a.n = n_prob;
    int total = 0;
    for (int q = 0; q < n_prob; ++q) {
        SARSSL_REQUIRE(N[q] % 4 == 0 && ldc[q] % 4 == 0 && nsplit[q] > 0, "sarssl_splitk_reduce_multi(shape)");
        a.ws[q] = ws[q]; a.C[q] = C[q]; a.ldc[q] = ldc[q]; a.nsplit[q] = nsplit[q]; a.M[q] = M[q]; a.N[q] = N[q];
        const long mn4 = (long)M[q] * N[q] / 4;
        int nb = (int)((mn4 + 255) / 256); if (nb > 512) nb = 512; if (nb < 1) nb = 1;
        a.first[q] = total; total += nb;
    }
    a.first[n_prob] = total;
    splitk_reduce_multi_kernel<<<total, 256, 0, (hipStream_t)stream>>>(a);
    SARSSL_CHECK_LAUNCH("splitk_reduce_multi_kernel");
    return 0;
}

template <typename TA, typename TB, typename TC, int FM, bool EDGE>
static void launch_fm(const GemmArgs& g, int a_kc, int b_kc, dim3 grid, hipStream_t st) {
    if (a_kc && b_kc) gemm_kernel<TA, TB, TC, true, true, FM, EDGE><<<grid, 256, 0, st>>>(g);
    else if (a_kc && !b_kc) gemm_kernel<TA, TB, TC, true, false, FM, EDGE><<<grid, 256, 0, st>>>(g);
    else if (!a_kc && b_kc) gemm_kernel<TA, TB, TC, false, true, FM, EDGE><<<grid, 256, 0, st>>>(g);
    else gemm_kernel<TA, TB, TC, false, false, FM, EDGE><<<grid, 256, 0, st>>>(g);
}

// Tile choice: 256 x 128 tiles (FM = 4) halve the staged bytes and barriers per MFMA but also the number of workgroups; they are
// used when the grid still has at least ~one workgroup per CU, otherwise 128 x 128 (FM = 2).  big = allowed for this dtype combo.
static int pick_fm(const GemmArgs& g, int nbatch, bool big) {
    if (!big) return 2;
#ifdef SARSSL_PROBE_ENV          // probe builds only: force a tile height (tools/step_gemm_table.py A/B runs)
    { const char* e = getenv("SARSSL_GEMM_FM"); if (e && atoi(e) > 0) return atoi(e); }
#endif
    const long nsplit = g.split_k > 0 ? g.split_k : 1;
    const long wg4 = (long)((g.M + 255) / 256) * ((g.N + BN - 1) / BN) * nbatch * nsplit;
    // 64 x 128 tiles (FM = 1) when 128 x 128 tiles give at most ~one workgroup per CU (N <= 256 at M = 16384): nothing else on the
    // CU hides a K-tile's memory round trip then (0.7-0.8 us per K-tile measured against 0.25 us of MFMA)
    const long wg2 = (long)((g.M + 127) / 128) * ((g.N + BN - 1) / BN) * nbatch * nsplit;
    constexpr int fm1_pct = 112;
    if (g.split_k <= 0 && g.M >= 128 && wg2 * 100 <= (long)sarssl_cu_count() * fm1_pct) return 1;
    const int k_len = g.split_k > 0 ? g.k_per_split : g.K;
    // round 4 (tools/step_gemm_table.py with a forced tile height, in-step launches at M = 16384): with K <= 256 a 128 x 128 tile is
    // four K-tiles of main loop behind a full prologue / epilogue - the 64-row tile's extra workgroups hide each other's latencies
    // (N = 1024, K = 256: 57.8 -> 48.4 us with the fused activation backward, 50.7 -> 46.1 us with Swish + dropout + pre-activation)
    if (g.split_k <= 0 && g.M >= 4096 && k_len <= 256 && g.N >= 1024) return 1;
    // measured on MI355X (tools/bench_kernels.py, tools/gemm_diag.py, M = 16384): the large tile wins from ~2 workgroups per CU and
    // K >= 768 on (decoder 3072 x 768 / 1024 x 3072: -4 ... -15 %); with one workgroup per CU (N = 512) or short K it loses 5-20 %
    // (round 4: K = 768 / 1024 at N = 3072 - the decoder's first layer and its data gradient with the fused ReLU backward - run 5-10 %
    //  faster on 128 x 128 tiles inside the step: 185.8 -> 168.3 us; the long-K products keep the large tile: 117 vs 139 us)
    return (g.M >= 192 && k_len >= 1536 && wg4 >= 2L * sarssl_cu_count()) ? 4 : 2;
}

template <typename TA, typename TB, typename TC, bool BIG>
static int launch_layout(const GemmArgs& g, int a_kc, int b_kc, int nbatch, hipStream_t st) {
    const int fm = pick_fm(g, nbatch, BIG);
    const int bm = 64 * fm;
    const int k_len = g.split_k > 0 ? g.k_per_split : g.K;
    const bool vec_ok = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && (!g.resid || (g.ldr & 7) == 0);
    const bool edge = (g.M % bm) != 0 || (g.N % BN) != 0 || (k_len % BK) != 0 || (g.K % BK) != 0 || !vec_ok || g.row_shift != 0;
    dim3 grid((g.N + BN - 1) / BN, (g.M + bm - 1) / bm, nbatch * (g.split_k > 0 ? g.split_k : 1));
    if constexpr (BIG) {
        if (fm == 4) {
            if (edge) launch_fm<TA, TB, TC, 4, true>(g, a_kc, b_kc, grid, st);
            else launch_fm<TA, TB, TC, 4, false>(g, a_kc, b_kc, grid, st);
            SARSSL_CHECK_LAUNCH("sarssl_gemm");
            return 0;
        }
    }
    if constexpr (BIG) {
        if (fm == 1) {
            if (edge) launch_fm<TA, TB, TC, 1, true>(g, a_kc, b_kc, grid, st);
            else launch_fm<TA, TB, TC, 1, false>(g, a_kc, b_kc, grid, st);
            SARSSL_CHECK_LAUNCH("sarssl_gemm");
            return 0;
        }
    }
    if (edge) launch_fm<TA, TB, TC, 2, true>(g, a_kc, b_kc, grid, st);
    else launch_fm<TA, TB, TC, 2, false>(g, a_kc, b_kc, grid, st);
    SARSSL_CHECK_LAUNCH("sarssl_gemm");
    return 0;
}

// C ABI ------------------------------------------------------------------------------------
// dtypes: 0 = f32, 1 = bf16.  Supported (A,B,C) combinations: (1,1,1) (1,1,0) (0,0,0).
// precise != 0 (f32 operands only) runs the 3-pass split and needs ws (f32, nbatch*M*N).
extern "C" int sarssl_gemm(const void* A, const void* B, void* C, int dtA, int dtB, int dtC,
                           int a_kc, int b_kc, int M, int N, int K, long lda, long ldb, long ldc,
                           int nbatch, int batch_inner, long sA0, long sA1, long sB0, long sB1, long sC0, long sC1,
                           float alpha, float out_scale, const float* bias, int act,
                           const void* resid, long ldr, long sR0, long sR1, float res_scale,
                           void* preact, const void* aux, int aux_act, int aux_dtype, float p_drop, unsigned long long seed,
                           int precise, float* ws, int split_k, int c_row_shift, void* stream) {
    SARSSL_REQUIRE(M > 0 && N > 0 && K > 0 && nbatch > 0 && batch_inner > 0, "sarssl_gemm");
    SARSSL_REQUIRE(a_kc ? (K % 8 == 0 && lda % 8 == 0) : (M % 8 == 0 && lda % 8 == 0), "sarssl_gemm(A alignment)");
    SARSSL_REQUIRE(b_kc ? (K % 8 == 0 && ldb % 8 == 0) : (N % 8 == 0 && ldb % 8 == 0), "sarssl_gemm(B alignment)");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.batch_inner = batch_inner; g.sA0 = sA0; g.sA1 = sA1; g.sB0 = sB0; g.sB1 = sB1; g.sC0 = sC0; g.sC1 = sC1;
    g.alpha = alpha; g.out_scale = out_scale; g.bias = bias; g.act = act;
    g.resid = resid; g.ldr = ldr; g.sR0 = sR0; g.sR1 = sR1; g.res_scale = res_scale;
    g.preact = preact; g.aux = aux; g.aux_act = aux_act; g.acc_ws = nullptr; g.acc_in = 0; g.acc_out = 0; g.partA = 0; g.partB = 0;
    // aux (the saved pre-activation of a fused activation backward) has C's dtype, or fp16 next to a bf16 C (fp16-forward mode)
    SARSSL_REQUIRE(!aux || aux_dtype == dtC || (aux_dtype == SARSSL_F16 && dtC == SARSSL_BF16), "sarssl_gemm(aux dtype)");
    g.aux_f16 = (aux && aux_dtype == SARSSL_F16 && dtC == SARSSL_BF16) ? 1 : 0;
    g.p_drop = p_drop; g.seed = seed; g.salt = sarssl_dropout_salt(); g.prio = (sarssl_mfma_prio() == 1 || sarssl_mfma_prio() == 3);
    g.split_k = 0; g.k_per_split = K;
    g.row_shift = 0; g.csum_ws = nullptr; g.A2 = nullptr; g.B2 = nullptr; g.nseg = 0;
#ifdef GEMM_STAMPS
    g.stamps = g_gemm_stamps_host;
#endif
    if (c_row_shift) {
        SARSSL_REQUIRE(M == N && ldc == N && split_k <= 0 && !precise && !resid && !preact && !aux, "sarssl_gemm(c_row_shift)");
        g.row_shift = N;
    }
    if (split_k > 0) {
        // accumulate mode: C (f32) += alpha * A*B, no other epilogue; K split over split_k workgroups per tile, partials
        // go to ws (f32, nbatch*split_k*M*N) and a second kernel folds them into C (deterministic, no atomics)
        SARSSL_REQUIRE(dtC == SARSSL_F32 && !bias && !resid && !preact && !aux && act == 0 && p_drop == 0.f && ws != nullptr,
                       "sarssl_gemm(split_k epilogue)");
        g.acc_ws = ws;
        int per = ((K + split_k - 1) / split_k + BK - 1) / BK * BK;
        g.split_k = (K + per - 1) / per; g.k_per_split = per;
    }
    hipStream_t st = (hipStream_t)stream;
    auto reduce = [&]() -> int {
        const long mn = (long)M * N;
        int bx = (int)((mn + 255) / 256); if (bx > 1024) bx = 1024;
        splitk_reduce_kernel<<<dim3(bx, nbatch), 256, 0, st>>>(ws, g.split_k, M, N, (float*)C, ldc, batch_inner, sC0, sC1);
        SARSSL_CHECK_LAUNCH("splitk_reduce_kernel");
        return 0;
    };
#ifdef SARSSL_WITH_GEMM_NT
    // probe builds only (tools/gemm_nt/): the direct-to-LDS pipelined NT kernel measured in round 3 - faster than this file's kernel on
    // long-K products alone (dec2 105 -> 96 us), neutral inside the step (its 128-144 KiB of LDS keep the other encoder's stream off the
    // CU), so it is not part of the product library
    if (dtA == SARSSL_BF16 && dtB == SARSSL_BF16 && a_kc && b_kc && nbatch == 1 && g.split_k <= 0 && !c_row_shift) {
        const int rc = sarssl_gemm_nt_try(g, dtC, stream);
        if (rc <= 0) return rc;
    }
#endif
    if (dtA == SARSSL_BF16 && dtB == SARSSL_BF16 && dtC == SARSSL_BF16)
        return launch_layout<bf16, bf16, bf16, true>(g, a_kc, b_kc, nbatch, st);
    if (dtA == SARSSL_BF16 && dtB == SARSSL_BF16 && dtC == SARSSL_F32) {
        int rc = launch_layout<bf16, bf16, float, true>(g, a_kc, b_kc, nbatch, st);
        if (rc || g.split_k <= 0 || C == nullptr) return rc;          // C == null: partials only, reduced later (sarssl_splitk_reduce_multi)
        return reduce();
    }
    // fp16-forward mode: forward products on fp16 operands ...
    if (dtA == SARSSL_F16 && dtB == SARSSL_F16 && dtC == SARSSL_F16)
        return launch_layout<f16, f16, f16, true>(g, a_kc, b_kc, nbatch, st);
    if (dtA == SARSSL_F16 && dtB == SARSSL_F16 && dtC == SARSSL_F32) {
        int rc = launch_layout<f16, f16, float, false>(g, a_kc, b_kc, nbatch, st);
        if (rc || g.split_k <= 0 || C == nullptr) return rc;
        return reduce();
    }
    // ... backward products of a bf16 gradient with a saved fp16 activation (either side), contracted in bf16
    if (dtA == SARSSL_BF16 && dtB == SARSSL_F16 && dtC == SARSSL_BF16)
        return launch_layout<bf16, f16, bf16, false>(g, a_kc, b_kc, nbatch, st);
    if (dtA == SARSSL_BF16 && dtB == SARSSL_F16 && dtC == SARSSL_F32) {
        int rc = launch_layout<bf16, f16, float, false>(g, a_kc, b_kc, nbatch, st);
        if (rc || g.split_k <= 0 || C == nullptr) return rc;
        return reduce();
    }
    if (dtA == SARSSL_F16 && dtB == SARSSL_BF16 && dtC == SARSSL_BF16)
        return launch_layout<f16, bf16, bf16, false>(g, a_kc, b_kc, nbatch, st);
    if (dtA == SARSSL_F32 && dtB == SARSSL_F32 && dtC == SARSSL_F32) {
        if (g.split_k > 0) {          // the accumulate epilogue is linear: the split-precision passes just add up
            GemmArgs p = g;
            const int npass = precise ? 3 : 1;
            for (int pass = 0; pass < npass; ++pass) {
                p.partA = precise ? (pass == 1) : 0; p.partB = precise ? (pass == 0) : 0;      // hi*lo, lo*hi, hi*hi
                int rc = launch_layout<float, float, float, false>(p, a_kc, b_kc, nbatch, st); if (rc) return rc;
                rc = reduce(); if (rc) return rc;
            }
            return 0;
        }
        if (!precise) return launch_layout<float, float, float, false>(g, a_kc, b_kc, nbatch, st);
        SARSSL_REQUIRE(ws != nullptr, "sarssl_gemm(precise needs workspace)");
        g.acc_ws = ws;
        GemmArgs p = g;
        p.partA = 0; p.partB = 1; p.acc_in = 0; p.acc_out = 1;          // hi*lo
        int rc = launch_layout<float, float, float, false>(p, a_kc, b_kc, nbatch, st); if (rc) return rc;
        p.partA = 1; p.partB = 0; p.acc_in = 1; p.acc_out = 1;          // + lo*hi
        rc = launch_layout<float, float, float, false>(p, a_kc, b_kc, nbatch, st); if (rc) return rc;
        p.partA = 0; p.partB = 0; p.acc_in = 1; p.acc_out = 0;          // + hi*hi, then epilogue
        return launch_layout<float, float, float, false>(p, a_kc, b_kc, nbatch, st);
    }
    sarssl_set_error("sarssl_gemm: unsupported dtype combination (%d,%d,%d)", dtA, dtB, dtC);
    return -1;
}

// K-segment NT product on fp16 operands (the "hybrid" numeric mode's forward Linear layers): C = epilogue(A B^T [+ A_lo B^T] [+ A B_lo^T])
// with f32 accumulation across the segments - an f32 activation given as its fp16 pair (hi = fp16(x), lo = fp16(x - hi): 22 significant
// bits) against a weight given as its pair contracts to ~2^-21 relative per product (the lo lo term is dropped), on the fp16 matrix
// cores.  The segments are a longer K loop of the plain kernel: same tiles, same epilogue, the prologue / epilogue of a short-K product
// amortised over 2-3 times the MFMAs.  A, A_lo: [M][K] (row stride lda), B, B_lo: [N][K] (row stride ldb); A_lo / B_lo may be null.
// C / resid / preact: dtC (fp16 | f32).
template <typename TC, int NSEG>
static int launch_seg(const GemmArgs& g, hipStream_t st) {
    // 64-row tiles when 128-row tiles would leave CUs without a second workgroup (the rule of pick_fm), else 128 x 128
    const long wg2 = (long)((g.M + 127) / 128) * ((g.N + BN - 1) / BN);
    // measured on MI355X (tools/hybrid_fm_ab.sh, M = 16384 / 8192): with at most one 128 x 128 tile per CU (N = 256) the 64-row tile wins at
    // any K (50.6 vs 59.1 us at K = 1024); with up to two per CU only at K <= 256 (four K-tiles: 34 vs 42 us) - at K = 512 .. 3072 the larger
    // tile's operand reuse wins (N = 512: 64 vs 68 us at K = 512, 106 vs 143 us at K = 2048; the decoder's second layer 127 vs 190 us)
    const long cus = sarssl_cu_count();
    static const int force_fm = [] { const char* e = getenv("SARSSL_SPLIT_FM"); return e ? atoi(e) : 0; }();      // A/B runs: 1 | 2
    const int fm = force_fm ? force_fm : ((g.M >= 128 && (wg2 <= cus || (wg2 <= 2 * cus && g.K <= 256))) ? 1 : 2);
    const int bm = 64 * fm;
    const bool vec_ok = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && (!g.resid || (g.ldr & 7) == 0);
    const bool edge = (g.M % bm) != 0 || (g.N % BN) != 0 || (g.K % BK) != 0 || !vec_ok;
    dim3 grid((g.N + BN - 1) / BN, (g.M + bm - 1) / bm, 1);
    if (fm == 1) { if (edge) gemm_seg_kernel<TC, 1, true, NSEG><<<grid, 256, 0, st>>>(g); else gemm_seg_kernel<TC, 1, false, NSEG><<<grid, 256, 0, st>>>(g); }
    else { if (edge) gemm_seg_kernel<TC, 2, true, NSEG><<<grid, 256, 0, st>>>(g); else gemm_seg_kernel<TC, 2, false, NSEG><<<grid, 256, 0, st>>>(g); }
    SARSSL_CHECK_LAUNCH("sarssl_gemm_split");
    return 0;
}
extern "C" int sarssl_gemm_split(const void* A, const void* A_lo, const void* B, const void* B_lo, void* C, int dtC, int M, int N, int K,
                                 long lda, long ldb, long ldc, float out_scale, const float* bias, int act, const void* resid, long ldr,
                                 float res_scale, void* preact, float p_drop, unsigned long long seed, void* stream) {
    SARSSL_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "sarssl_gemm_split");
    SARSSL_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "sarssl_gemm_split(alignment)");
    SARSSL_REQUIRE(dtC == SARSSL_F16 || dtC == SARSSL_F32, "sarssl_gemm_split(dtC)");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.batch_inner = 1; g.sA0 = g.sA1 = g.sB0 = g.sB1 = g.sC0 = g.sC1 = 0;
    g.alpha = 1.f; g.out_scale = out_scale; g.bias = bias; g.act = act;
    g.resid = resid; g.ldr = ldr; g.sR0 = g.sR1 = 0; g.res_scale = res_scale;
    g.preact = preact; g.aux = nullptr; g.aux_act = 0; g.aux_f16 = 0; g.acc_ws = nullptr; g.acc_in = 0; g.acc_out = 0; g.partA = 0; g.partB = 0;
    g.p_drop = p_drop; g.seed = seed; g.salt = sarssl_dropout_salt(); g.prio = (sarssl_mfma_prio() == 1 || sarssl_mfma_prio() == 3);
    g.split_k = 0; g.k_per_split = K; g.row_shift = 0; g.csum_ws = nullptr;
    // segments: (A, B) [, (A_lo, B)] [, (A, B_lo)] - the kernel reads A2 in segment 1 of 3 and B2 in the last of 2 or 3
    SARSSL_REQUIRE(!A_lo || B_lo, "sarssl_gemm_split(A_lo needs B_lo)");
    if (A_lo && B_lo) { g.A2 = A_lo; g.B2 = B_lo; g.nseg = 3; }
    else if (B_lo) { g.A2 = nullptr; g.B2 = B_lo; g.nseg = 2; }
    else { g.A2 = nullptr; g.B2 = nullptr; g.nseg = 1; }
#ifdef GEMM_STAMPS
    g.stamps = nullptr;
#endif
    hipStream_t st = (hipStream_t)stream;
    if (g.nseg == 1) {                   // no lo parts: the plain fp16 product
        if (dtC == SARSSL_F16) return launch_layout<f16, f16, f16, true>(g, 1, 1, 1, st);
        return launch_layout<f16, f16, float, false>(g, 1, 1, 1, st);
    }
    if (g.nseg == 2) return dtC == SARSSL_F16 ? launch_seg<f16, 2>(g, st) : launch_seg<float, 2>(g, st);
    return dtC == SARSSL_F16 ? launch_seg<f16, 3>(g, st) : launch_seg<float, 3>(g, st);
}

// Grouped split-K weight-gradient products: ws[q] (f32, split_q * M_q * N_q, see `split_out`) receives the partial sums of
// dY_q^T X_q with A_q = dY [K_q][M_q] (row stride lda), B_q = X [K_q][N_q] (row stride ldb), bf16; fold with
// sarssl_splitk_reduce_multi.  split_k[q] is the requested split; the effective number of partial slices (ceil(K / per), per a
// multiple of 64) is returned in split_out[q] - size ws[q] for split_k[q] slices.  Returns 1 without launching when any shape is
// ragged (M % 128, N % 128 or the K slices % 64).
extern "C" int sarssl_gemm_group_tn(const void* const* A, const void* const* B, float* const* ws, const int* M, const int* N,
                                    const int* K, const long* lda, const long* ldb, const int* split_k, int* split_out,
                                    float* const* csum_ws, int n_prob, int dtB, void* stream) {
    SARSSL_REQUIRE(n_prob > 0 && n_prob <= GROUP_MAXP && (dtB == SARSSL_BF16 || dtB == SARSSL_F16), "sarssl_gemm_group_tn");
    GemmGroup a;
    a.n = n_prob;
    int total = 0;
    bool edge = false;
    for (int q = 0; q < n_prob; ++q) {
        SARSSL_REQUIRE(M[q] > 0 && N[q] > 0 && K[q] > 0 && split_k[q] > 0 && M[q] % 8 == 0 && N[q] % 8 == 0 && lda[q] % 8 == 0 &&
                       ldb[q] % 8 == 0 && ws[q] != nullptr, "sarssl_gemm_group_tn(shape)");
        const int per = ((K[q] + split_k[q] - 1) / split_k[q] + BK - 1) / BK * BK;
        const int ns = (K[q] + per - 1) / per;
        GemmArgs& g = a.p[q];
        g.A = A[q]; g.B = B[q]; g.C = nullptr; g.M = M[q]; g.N = N[q]; g.K = K[q]; g.lda = lda[q]; g.ldb = ldb[q]; g.ldc = N[q];
        g.batch_inner = 1; g.sA0 = g.sA1 = g.sB0 = g.sB1 = g.sC0 = g.sC1 = 0;
        g.alpha = 1.f; g.out_scale = 1.f; g.bias = nullptr; g.act = 0;
        g.resid = nullptr; g.ldr = 0; g.sR0 = g.sR1 = 0; g.res_scale = 0.f;
        g.preact = nullptr; g.aux = nullptr; g.aux_act = 0; g.aux_f16 = 0; g.acc_ws = ws[q]; g.acc_in = 0; g.acc_out = 0; g.partA = 0; g.partB = 0;
        g.p_drop = 0.f; g.seed = 0; g.salt = nullptr; g.prio = (sarssl_mfma_prio() == 1 || sarssl_mfma_prio() == 3);
        g.split_k = ns; g.k_per_split = per; g.row_shift = 0; g.A2 = nullptr; g.B2 = nullptr; g.nseg = 0;
        g.csum_ws = csum_ws ? csum_ws[q] : nullptr;
        a.gx[q] = (N[q] + BN - 1) / BN; a.gy[q] = (M[q] + 127) / 128;
        a.first[q] = total; total += (a.gx[q] * a.gy[q] * ns + 7) / 8 * 8;
        split_out[q] = ns;
        edge = edge || (M[q] % 128) != 0 || (N[q] % BN) != 0 || (per % BK) != 0 || (K[q] % BK) != 0;
    }
    a.first[n_prob] = total;
    if (edge) return 1;            // ragged shapes: not taken (the caller launches the products one by one); the EDGE instantiation of
                                   // the grouped kernel crashes hipcc 7.2's simplifycfg
    if (dtB == SARSSL_F16) gemm_group_tn_kernel<f16><<<total, 256, 0, (hipStream_t)stream>>>(a);
    else gemm_group_tn_kernel<bf16><<<total, 256, 0, (hipStream_t)stream>>>(a);
    SARSSL_CHECK_LAUNCH("gemm_group_tn_kernel");
    return 0;
}
