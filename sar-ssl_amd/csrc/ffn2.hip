// Fused feed-forward module for gfx950: two chained GEMMs per 64-row tile with the hidden tile resident on the CU.
//
//   forward  (feed_forward.py:47-54, Conformer.py:60-67):  y = x + f * drop2( W2 . drop1( swish( W1 . ln + b1 ) ) + b2 )
//   backward (same launch shape, transposed weights):      dh = (dz2 . W2) * drop1' * swish'(hpre);   dln = dh . W1
//
// Why: the round-4 counters put the GEMM family furthest below its roofline, and inside it the d = 256 / 512 FFN pairs: FFN-1 wrote a
// [M, 4d] hidden tensor and a [M, 4d] pre-activation, FFN-2 read the hidden tensor back (~200 MB moved for an algorithmic 50 MB at
// d = 256), each behind its own prologue / epilogue with four K-tiles of main loop in between.  Here a workgroup owns 64 rows for the
// whole module: the [64, d] input tile stays in LDS, the hidden dimension is walked in chunks of 256, and each chunk's [64, 256] hidden
// tile goes matrix cores -> registers (bias / Swish / dropout or their backward) -> LDS -> matrix cores without touching HBM as an
// operand; the second product accumulates in registers over all chunks.  What still leaves the chip is what the backward pass needs
// (forward: the pre-activation and the dropped hidden tile, saved once, row-contiguous from the LDS tile; backward: dh for the two
// weight-gradient products) and the [M, d] result.
//
// Weights: every workgroup streams both matrices (1 MB at d = 256) from L2.  They are PRE-PACKED in MFMA fragment order
// (sarssl_ffn_pack: block (n / 32, k / 16) = 1 KiB, lane l holds row n = l & 31, k = 8 (l >> 5) .. + 7), so a wave fetches a fragment
// with ONE fully coalesced 1-KiB global_load_dwordx4 straight into the registers the MFMA reads - no LDS staging, no transposition,
// no barrier on the weight path, each weight element loaded exactly once per workgroup (waves are laid out 1 x 8 over the output
// columns: a wave owns 32 hidden units of the chunk in the first product and d / 8 output columns in the second, and all 64 rows,
// so every weight fragment feeds two MFMAs).  A 16-deep register queue keeps 16 KiB per wave in flight across both products and
// across chunk boundaries.  The backward launch reads packs of the TRANSPOSED matrices (same kernel, other epilogues).
//
// Roles in the MFMA are swapped like in gemm.hip (weights = row operand): a lane ends up with 4 consecutive hidden units / output
// columns of ONE row, which is what the 8-byte LDS tile writes and the dropout pair hashes want.  Dropout decisions are the same pure
// function of (seed, row * N + column) as in gemm_epilogue.h: masks are identical to the unfused kernels', forward and backward.
#include "common.h"

#ifndef FFN_ABL
#define FFN_ABL 0          // probe builds (tools/bench_ffn2.py --ablate): 1 no activation / dropout math, 2 no tile stores, 4 no weight refills, 8 no MFMAs
#endif
#define FFN_NT 512
#define FFN_HC 256
#define FFN_PH (FFN_HC + 8)

struct Ffn2Args {
    const void* A; long lda;                   // [M][D] input rows (forward: LayerNorm output; backward: gradient of the second Linear's output)
    const void* W1p; const void* W2p;          // packed [4D x D] and [D x 4D] (see sarssl_ffn_pack)
    const float* b1; const float* b2;          // forward only
    void* P;                                   // [M][4D]: forward OUT pre-activation, backward IN pre-activation
    void* Hs;                                  // [M][4D]: forward OUT dropped hidden activations, backward OUT dh
    void* Y; long ldy;                         // [M][D] result
    const void* R; long ldr;                   // forward: residual rows (or null)
    float p1, p2; unsigned long long s1, s2; const unsigned long long* salt;
    float out_scale;
    int M;
    // LayerNorm fused into the launch (either may be unused):
    //   forward  (X != null): A is ignored - the input tile is LayerNorm(X rows) (feed_forward.py:48), written to LNout [M][D] with its
    //            statistics (saved for the backward pass) and kept in LDS as the first product's operand; R is normally X itself
    //   backward (X != null): the second product's result is not stored - it is the gradient w.r.t. the LayerNorm output, and the
    //            epilogue runs the LayerNorm backward on it: Y = dx (+ R), Y2 (optional) = dx * dropmask(p2, s2) * out_scale (the dropout
    //            backward of the next module of the chain), ln_partial [grid][2][D] = per-workgroup sums for dgamma | dbeta
    const void* X; long ldx;
    const float* ln_g; const float* ln_b; float ln_eps;
    void* LNout; float* ln_mean; float* ln_rstd;
    void* Y2; float* ln_partial;
    int rot;                                   // chunk rotation per workgroup (see the weight queue)
};

#include "ffn_common.h"

template <typename T, typename TP, int D, bool BWD>
__global__ __launch_bounds__(FFN_NT) void ffn2_kernel(Ffn2Args g) {
    constexpr int H = 4 * D, NCH = H / FFN_HC, PA = D + 8, KS1 = D / 16, DB = D / 256;
    constexpr int SA_ELEMS = 64 * PA, ST_ELEMS = 64 * FFN_PH, PY = D + 4;
    constexpr int TILE_ELEMS = SA_ELEMS + 2 * ST_ELEMS, YST_ELEMS = 64 * PY * 2;
    constexpr int LDS_ELEMS = TILE_ELEMS > YST_ELEMS ? TILE_ELEMS : YST_ELEMS;
    static_assert(KS1 % 16 == 0 && (16 * DB) % 16 == 0, "weight queue: segments of 16 pieces");
    __shared__ __attribute__((aligned(16))) uint16_t smem[LDS_ELEMS];
    uint16_t* sA = smem;
    uint16_t* sH = smem + SA_ELEMS;
    uint16_t* sP = sH + ST_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long m0 = (long)blockIdx.x * 64;
    const T* A = (const T*)g.A;
    const uint4* W1 = (const uint4*)g.W1p + lane;
    const uint4* W2 = (const uint4*)g.W2p + lane;

    auto w1_piece = [&](int c, int ks) -> const uint4* { return W1 + ((long)((c * 8 + w) * KS1 + ks) << 6); };
    auto w2_piece = [&](int c, int ksl, int dbi) -> const uint4* { return W2 + ((long)((w * DB + dbi) * (H / 16) + c * 16 + ksl) << 6); };

    // weight queue: 16 fragments (16 KiB per wave) in flight; position p of a product uses q[p % 16] and refills it with the fragment
    // 16 positions further down the wave's stream (first product of chunk c, second product of chunk c, first product of chunk c + 1, ...)
    // Every workgroup streams the same weight packs: started on the same chunk, the 32 CUs of an XCD (workgroups b, b + 8, ...) ask its L2
    // for the same lines at the same time.  Workgroup b walks the chunks from (b / 8) % NCH instead - the second product's sum over the
    // chunks has a fixed order per workgroup either way.  -5 % on the launch (tools/bench_ffn2.py, SARSSL_FFN_ROT=0 for the old order).
    static_assert((NCH & (NCH - 1)) == 0, "chunk rotation");
    const int c0 = g.rot ? (int)((blockIdx.x >> 3) & (NCH - 1)) : 0;
    uint4 q[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) q[j] = *w1_piece(c0, j);

    // ---- input tile -> LDS (padded pitch: conflict-free ds_read_b128 fragment reads)
    if (!BWD && g.X) {
        // LayerNorm of the tile's rows, the arithmetic of layernorm_fwd_kernel (csrc/elementwise.hip) operation for operation - one wave per
        // row, lane = 4 consecutive columns (+ 256 i), two-pass mean / variance through the same xor-shuffle tree - so that the fused launch
        // reproduces the stand-alone kernel bit for bit: rows w * 8 .. w * 8 + 7 of the tile belong to wave w
        constexpr int NVL = D / 256;
        const T* X = (const T*)g.X;
        T* LN = (T*)g.LNout;
        float4 v[8][NVL];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < NVL; ++i) v[j][i] = ld4(X + (m0 + w * 8 + j) * g.ldx + (lane + i * 64) * 4);
        float4 gam[NVL], bet[NVL];
#pragma unroll
        for (int i = 0; i < NVL; ++i) { gam[i] = *(const float4*)(g.ln_g + (lane + i * 64) * 4); bet[i] = *(const float4*)(g.ln_b + (lane + i * 64) * 4); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = w * 8 + j;
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < NVL; ++i) sm += v[j][i].x + v[j][i].y + v[j][i].z + v[j][i].w;
            const float mu = wave_sum(sm) / (float)D;
            float qv = 0.f;
#pragma unroll
            for (int i = 0; i < NVL; ++i) {
                const float a = v[j][i].x - mu, b = v[j][i].y - mu, c = v[j][i].z - mu, e = v[j][i].w - mu;
                qv += a * a + b * b + c * c + e * e;
            }
            const float rs = rsqrtf(wave_sum(qv) / (float)D + g.ln_eps);
#pragma unroll
            for (int i = 0; i < NVL; ++i) {
                const int c4 = lane + i * 64;
                const float4 o = make_float4((v[j][i].x - mu) * rs * gam[i].x + bet[i].x, (v[j][i].y - mu) * rs * gam[i].y + bet[i].y,
                                             (v[j][i].z - mu) * rs * gam[i].z + bet[i].z, (v[j][i].w - mu) * rs * gam[i].w + bet[i].w);
                st4(LN + (m0 + row) * (long)D + c4 * 4, o);
                const float ov[4] = {o.x, o.y, o.z, o.w};
                *(uint2*)&sA[row * PA + c4 * 4] = pack4<T>(ov);
            }
            if (lane == 0) { g.ln_mean[m0 + row] = mu; g.ln_rstd[m0 + row] = rs; }
        }
    } else {
        constexpr int CPRW = D / 8, RPI = FFN_NT / CPRW;            // 16-byte chunks per row, rows per pass
        const int r = tid / CPRW, ch = tid % CPRW;
#pragma unroll
        for (int j = 0; j < 64 / RPI; ++j) {
            const int row = r + RPI * j;
            *(uint4*)&sA[row * PA + ch * 8] = *(const uint4*)(A + (m0 + row) * g.lda + ch * 8);
        }
    }
    FfnDrop d1;
    d1.init(g.p1, g.s1, g.salt);

    f32x16 Y[DB][2];
#pragma unroll
    for (int a = 0; a < DB; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[a][i][r] = 0.f;

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const int hl0 = w * 32 + 4 * (lane >> 5);                      // + 8 g + e: the lane's hidden units inside the chunk
    // cooperative row-contiguous tile <-> global mapping: 32 lanes x 16 bytes = one 512-byte tile row
    const int trow = tid >> 5, tch = tid & 31;
    __syncthreads();

    for (int it = 0; it < NCH; ++it) {
        const int c = (it + c0) & (NCH - 1);
        const int cn = it + 1 < NCH ? ((c + 1) & (NCH - 1)) : c;
        uint4 hp_regs[4];
        if constexpr (BWD) {        // the chunk's saved pre-activation tile: requested now, parked in LDS behind the first product
            const TP* P = (const TP*)g.P;
#pragma unroll
            for (int j = 0; j < 4; ++j) hp_regs[j] = *(const uint4*)(P + (m0 + trow + 16 * j) * H + c * FFN_HC + tch * 8);
        }
        // ---- first product: S^T[h][m] = sum_k W1[h][k] A[m][k], wave = 32 hidden units x 64 rows
        f32x16 S[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[i][r] = 0.f;
        {
            bf16x8 fa[2][2];
            fa[0][0] = *(const bf16x8*)&sA[frow * PA + fk];
            fa[0][1] = *(const bf16x8*)&sA[(32 + frow) * PA + fk];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int cur = ks & 1, nx = cur ^ 1;
                if (ks + 1 < KS1) {
                    fa[nx][0] = *(const bf16x8*)&sA[frow * PA + (ks + 1) * 16 + fk];
                    fa[nx][1] = *(const bf16x8*)&sA[(32 + frow) * PA + (ks + 1) * 16 + fk];
                }
                const bf16x8 wf = __builtin_bit_cast(bf16x8, q[ks & 15]);
                if (!(FFN_ABL & 8)) {
                    S[0] = mfma16<T>(wf, fa[cur][0], S[0]);
                    S[1] = mfma16<T>(wf, fa[cur][1], S[1]);
                } else { S[0][ks & 15] += (float)wf[0] + (float)fa[cur][0][0]; S[1][ks & 15] += (float)fa[cur][1][0]; }
                const int nxt = ks + 16;                            // refill: 16 positions ahead in the wave's weight stream
                if (!(FFN_ABL & 4)) {
                    if (nxt < KS1) q[ks & 15] = *w1_piece(c, nxt);
                    else q[ks & 15] = *w2_piece(c, (nxt - KS1) / DB, (nxt - KS1) % DB);
                }
            }
        }
        if constexpr (BWD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *(uint4*)&sP[(trow + 16 * j) * FFN_PH + tch * 8] = hp_regs[j];
            __syncthreads();
        }
        // ---- epilogue of the first product, in the accumulators' own layout: lane = row m, registers = 4 x 4 consecutive hidden units
        {
            float4 bq[4];
            if constexpr (!BWD) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) bq[gq] = *(const float4*)(g.b1 + c * FFN_HC + hl0 + 8 * gq);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = i * 32 + frow;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int hl = hl0 + 8 * gq;
                    float v[4] = {S[i][4 * gq + 0], S[i][4 * gq + 1], S[i][4 * gq + 2], S[i][4 * gq + 3]};
                    if constexpr (!BWD) {
                        v[0] += bq[gq].x; v[1] += bq[gq].y; v[2] += bq[gq].z; v[3] += bq[gq].w;
                        *(uint2*)&sP[m * FFN_PH + hl] = pack4<TP>(v);                 // pre-activation as saved for backward
                        if (!(FFN_ABL & 1)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] * sigmoidf_(v[e]);
                        }
                    } else {
                        const uint2 hu = *(const uint2*)&sP[m * FFN_PH + hl];
                        const float hp[4] = {H16<TP>::lo(hu.x), H16<TP>::hi(hu.x), H16<TP>::lo(hu.y), H16<TP>::hi(hu.y)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float sg = sigmoidf_(hp[e]); v[e] *= sg * (1.f + hp[e] * (1.f - sg)); }
                    }
                    if (d1.p > 0.f && !(FFN_ABL & 1)) {
                        float k[4];
                        d1.scale4((unsigned long long)(m0 + m) * H + (unsigned long long)(c * FFN_HC + hl), k);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= k[e];
                    }
                    *(uint2*)&sH[m * FFN_PH + hl] = pack4<T>(v);
                }
            }
        }
        __syncthreads();
        // ---- the tiles leave for HBM row by row (what the backward pass / the weight-gradient products read) ...
        if (!(FFN_ABL & 2)) {
            T* Hs = (T*)g.Hs;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = trow + 16 * j;
                *(uint4*)(Hs + (m0 + row) * H + c * FFN_HC + tch * 8) = *(const uint4*)&sH[row * FFN_PH + tch * 8];
            }
            if constexpr (!BWD) {
                TP* P = (TP*)g.P;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = trow + 16 * j;
                    *(uint4*)(P + (m0 + row) * H + c * FFN_HC + tch * 8) = *(const uint4*)&sP[row * FFN_PH + tch * 8];
                }
            }
        }
        // ---- ... while the second product contracts the hidden tile: Y^T[d][m] += sum_h W2[d][h] Hc[m][h], wave = 32 DB columns x 64 rows
        {
            bf16x8 fh[2][2];
            fh[0][0] = *(const bf16x8*)&sH[frow * FFN_PH + fk];
            fh[0][1] = *(const bf16x8*)&sH[(32 + frow) * FFN_PH + fk];
#pragma unroll
            for (int ksl = 0; ksl < 16; ++ksl) {
                const int cur = ksl & 1, nx = cur ^ 1;
                if (ksl + 1 < 16) {
                    fh[nx][0] = *(const bf16x8*)&sH[frow * FFN_PH + (ksl + 1) * 16 + fk];
                    fh[nx][1] = *(const bf16x8*)&sH[(32 + frow) * FFN_PH + (ksl + 1) * 16 + fk];
                }
#pragma unroll
                for (int dbi = 0; dbi < DB; ++dbi) {
                    const int p = ksl * DB + dbi;
                    const bf16x8 wf = __builtin_bit_cast(bf16x8, q[p & 15]);
                    if (!(FFN_ABL & 8)) {
                        Y[dbi][0] = mfma16<T>(wf, fh[cur][0], Y[dbi][0]);
                        Y[dbi][1] = mfma16<T>(wf, fh[cur][1], Y[dbi][1]);
                    } else { Y[dbi][0][ksl & 15] += (float)wf[0] + (float)fh[cur][0][0]; Y[dbi][1][ksl & 15] += (float)fh[cur][1][0]; }
                    const int nxt = p + 16;
                    if (!(FFN_ABL & 4)) {
                        if (nxt < 16 * DB) q[p & 15] = *w2_piece(c, nxt / DB, nxt % DB);
                        else q[p & 15] = *w1_piece(cn, nxt - 16 * DB);          // (last chunk: a harmless re-read of its own fragments)
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- result: accumulators -> f32 LDS staging (all operand tiles are dead) -> 8-wide row pieces: bias, dropout, scale, residual
    float* sY = (float*)smem;
#pragma unroll
    for (int dbi = 0; dbi < DB; ++dbi)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *(float4*)&sY[(i * 32 + frow) * PY + (w * DB + dbi) * 32 + 8 * gq + 4 * (lane >> 5)] =
                    make_float4(Y[dbi][i][4 * gq + 0], Y[dbi][i][4 * gq + 1], Y[dbi][i][4 * gq + 2], Y[dbi][i][4 * gq + 3]);
    __syncthreads();
    {
        constexpr int CPRW = D / 8, RPI = FFN_NT / CPRW;
        const int r = tid / CPRW, ch = tid % CPRW, n = ch * 8;
        T* Yo = (T*)g.Y;
        float bias8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
        FfnDrop d2;
        d2.init(BWD ? 0.f : g.p2, g.s2, g.salt);
        if constexpr (!BWD) {
            const float4 b0 = *(const float4*)(g.b2 + n), b1v = *(const float4*)(g.b2 + n + 4);
            bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1v.x; bias8[5] = b1v.y; bias8[6] = b1v.z; bias8[7] = b1v.w;
        }
        if (BWD && g.X) {
            // LayerNorm backward on the rows of the tile (layernorm_bwd_kernel, csrc/elementwise.hip): dx = rstd (g - mean(g) - xhat mean(g xhat))
            // + resid with g = dln * gamma, dln rounded to the gradient dtype as the stand-alone sequence stores it; a row's CPRW threads fold
            // their sums by xor shuffles; dgamma / dbeta partial sums over the tile's 64 rows go to ln_partial (folded by the block's reduce)
            const TP* Xs = (const TP*)g.X;
            float gam8[8], ag[8], ab[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { gam8[e] = g.ln_g[n + e]; ag[e] = 0.f; ab[e] = 0.f; }
            FfnDrop dq;
            dq.init(g.Y2 ? g.p2 : 0.f, g.s2, g.salt);
#pragma unroll
            for (int j = 0; j < 64 / RPI; ++j) {
                const int row = r + RPI * j;
                const long m = m0 + row;
                const float4 a0 = *(const float4*)&sY[row * PY + n], a1 = *(const float4*)&sY[row * PY + n + 4];
                const float av[8] = {round_as<T>(a0.x), round_as<T>(a0.y), round_as<T>(a0.z), round_as<T>(a0.w),
                                     round_as<T>(a1.x), round_as<T>(a1.y), round_as<T>(a1.z), round_as<T>(a1.w)};
                const f8 xx = ld8(Xs + m * g.ldx + n);
                const float mu = g.ln_mean[m], rs = g.ln_rstd[m];
                float xh[8], gq[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[e] = (xx.v[e] - mu) * rs;
                    gq[e] = av[e] * gam8[e];
                    s1 += gq[e]; s2 += gq[e] * xh[e];
                    ag[e] += av[e] * xh[e]; ab[e] += av[e];
                }
#pragma unroll
                for (int o = CPRW / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                s1 /= (float)D; s2 /= (float)D;
                f8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] = rs * (gq[e] - s1 - xh[e] * s2);
                if (g.R) {
                    const f8 rr = ld8((const T*)g.R + m * g.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.v[e] += rr.v[e];
                }
                st8(Yo + m * g.ldy + n, v);
                if (g.Y2) {
                    if (dq.p > 0.f) {
                        float k0[4], k1[4];
                        const unsigned long long base = (unsigned long long)m * D + n;
                        dq.scale4(base, k0);
                        dq.scale4(base + 4, k1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v.v[e] *= k0[e]; v.v[4 + e] *= k1[e]; }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.v[e] *= g.out_scale;
                    st8((T*)g.Y2 + m * (long)D + n, v);
                }
            }
            if (g.ln_partial) {
                __syncthreads();                                   // every thread is done with the staged accumulators
                float* sR = (float*)smem;                          // [RPI][2][D]
#pragma unroll
                for (int e = 0; e < 8; ++e) { sR[(r * 2 + 0) * D + n + e] = ag[e]; sR[(r * 2 + 1) * D + n + e] = ab[e]; }
                __syncthreads();
                float* P = g.ln_partial + (long)blockIdx.x * 2 * D;
                for (int c = tid; c < 2 * D; c += FFN_NT) {
                    float acc_ = 0.f;
#pragma unroll
                    for (int rr = 0; rr < RPI; ++rr) acc_ += sR[rr * 2 * D + c];
                    P[c] = acc_;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 64 / RPI; ++j) {
            const int row = r + RPI * j;
            const long m = m0 + row;
            const float4 a0 = *(const float4*)&sY[row * PY + n], a1 = *(const float4*)&sY[row * PY + n + 4];
            f8 v;
            v.v[0] = a0.x; v.v[1] = a0.y; v.v[2] = a0.z; v.v[3] = a0.w; v.v[4] = a1.x; v.v[5] = a1.y; v.v[6] = a1.z; v.v[7] = a1.w;
            if constexpr (!BWD) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] += bias8[e];
                if (d2.p > 0.f) {
                    float k0[4], k1[4];
                    const unsigned long long base = (unsigned long long)m * D + n;
                    d2.scale4(base, k0);
                    d2.scale4(base + 4, k1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v.v[e] *= k0[e]; v.v[4 + e] *= k1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] *= g.out_scale;
                if (g.R) {
                    const f8 rr = ld8((const T*)g.R + m * g.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.v[e] += rr.v[e];
                }
            }
            st8(Yo + m * g.ldy + n, v);
        }
    }
}

// ---- weight packs ---------------------------------------------------------------------------------------------------------------------
// dst block (n / 32, k / 16) = 64 lanes x 8 elements: lane l <- src(n = 32 nb + (l & 31), k = 16 ks + 8 (l >> 5) + 0..7),
// src(n, k) = src[n * rs + k * cs] (rs / cs select the matrix or its transpose).  One launch for up to FFN_PACK_MAX matrices.
#define FFN_PACK_MAX 64
struct FfnPackJobs {
    const void* src[FFN_PACK_MAX]; void* dst[FFN_PACK_MAX];
    int N[FFN_PACK_MAX], K[FFN_PACK_MAX]; long rs[FFN_PACK_MAX], cs[FFN_PACK_MAX];
    int first[FFN_PACK_MAX + 1];
    int n;
};
__global__ void ffn_pack_kernel(FfnPackJobs a) {
    int jb = 0;
    while (jb + 1 < a.n && (int)blockIdx.x >= a.first[jb + 1]) ++jb;
    const long piece = (long)(blockIdx.x - a.first[jb]) * 256 + threadIdx.x;
    const int K = a.K[jb], N = a.N[jb];
    const int ksn = K / 16;
    if (piece >= (long)(N / 32) * ksn * 64) return;
    const int l = (int)(piece & 63);
    const long blk = piece >> 6;
    const int ks = (int)(blk % ksn), nb = (int)(blk / ksn);
    const long n = (long)nb * 32 + (l & 31), k0 = (long)ks * 16 + (l >> 5) * 8;
    const uint16_t* s = (const uint16_t*)a.src[jb];
    const long rs = a.rs[jb], cs = a.cs[jb];
    uint4 u;
    if (cs == 1) u = *(const uint4*)(s + n * rs + k0);
    else {
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = s[n * rs + (k0 + j) * cs];
        u.x = e[0] | (e[1] << 16); u.y = e[2] | (e[3] << 16); u.z = e[4] | (e[5] << 16); u.w = e[6] | (e[7] << 16);
    }
    ((uint4*)a.dst[jb])[piece] = u;
}

extern "C" int sarssl_ffn_pack(const void* const* src, void* const* dst, const int* N, const int* K, const long* rs, const long* cs,
                               int n_mat, void* stream) {
    SARSSL_REQUIRE(n_mat > 0 && n_mat <= FFN_PACK_MAX, "sarssl_ffn_pack");
    FfnPackJobs a;
    a.n = n_mat;
    int total = 0;
    for (int j = 0; j < n_mat; ++j) {
        SARSSL_REQUIRE(N[j] > 0 && K[j] > 0 && N[j] % 32 == 0 && K[j] % 16 == 0 && (cs[j] != 1 || rs[j] % 8 == 0), "sarssl_ffn_pack(shape)");
        a.src[j] = src[j]; a.dst[j] = dst[j]; a.N[j] = N[j]; a.K[j] = K[j]; a.rs[j] = rs[j]; a.cs[j] = cs[j];
        a.first[j] = total;
        total += (int)(((long)N[j] * K[j] / 8 + 255) / 256);
    }
    a.first[n_mat] = total;
    ffn_pack_kernel<<<total, 256, 0, (hipStream_t)stream>>>(a);
    SARSSL_CHECK_LAUNCH("ffn_pack_kernel");
    return 0;
}

template <typename T, typename TP, bool BWD>
static int ffn2_launch(const Ffn2Args& g, int d, hipStream_t st) {
    const int grid = g.M / 64;
    if (d == 256) ffn2_kernel<T, TP, 256, BWD><<<grid, FFN_NT, 0, st>>>(g);
    else ffn2_kernel<T, TP, 512, BWD><<<grid, FFN_NT, 0, st>>>(g);
    SARSSL_CHECK_LAUNCH("ffn2_kernel");
    return 0;
}

extern "C" int sarssl_ffn2_supported(long M, int d) { return (M > 0 && M % 64 == 0 && (d == 256 || d == 512)) ? 1 : 0; }

// forward: y[M][d] = resid + out_scale * drop(p2, s2)( W2 drop(p1, s1)( swish(W1 ln + b1) ) + b2 );  preact / hidden [M][4d] are written
// for the backward pass.  dtype: SARSSL_F16 | SARSSL_BF16 (ln, packs, preact, hidden, y, resid).  x_ln != null: ln = LayerNorm(x_ln) is
// formed in the launch (gamma, beta, eps), stored to ln_out [M][d] with ln_mean / ln_rstd [M]; the `ln` argument is ignored.
extern "C" int sarssl_ffn2_fwd(const void* ln, long ldln, const void* w1p, const void* w2p, const float* b1, const float* b2, void* preact,
                               void* hidden, void* y, long ldy, const void* resid, long ldr, long M, int d, float p1,
                               unsigned long long s1, float p2, unsigned long long s2, float out_scale, const void* x_ln, long ldx,
                               const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out, float* ln_mean, float* ln_rstd,
                               int dtype, void* stream) {
    SARSSL_REQUIRE(sarssl_ffn2_supported(M, d) && ldy % 8 == 0 && (!resid || ldr % 8 == 0) && b1 && b2 && preact && hidden, "sarssl_ffn2_fwd");
    SARSSL_REQUIRE(x_ln ? (ldx % 8 == 0 && ln_gamma && ln_beta && ln_out && ln_mean && ln_rstd) : (ln != nullptr && ldln % 8 == 0), "sarssl_ffn2_fwd(layernorm)");
    Ffn2Args g;
    g.X = x_ln; g.ldx = ldx; g.ln_g = ln_gamma; g.ln_b = ln_beta; g.ln_eps = ln_eps; g.LNout = ln_out; g.ln_mean = ln_mean; g.ln_rstd = ln_rstd;
    g.Y2 = nullptr; g.ln_partial = nullptr;
    g.A = ln; g.lda = ldln; g.W1p = w1p; g.W2p = w2p; g.b1 = b1; g.b2 = b2; g.P = preact; g.Hs = hidden; g.Y = y; g.ldy = ldy;
    g.R = resid; g.ldr = ldr; g.p1 = p1; g.p2 = p2; g.s1 = s1; g.s2 = s2; g.salt = sarssl_dropout_salt(); g.rot = ffn_rot(); g.out_scale = out_scale; g.M = (int)M;
    if (dtype == SARSSL_F16) return ffn2_launch<f16, f16, false>(g, d, (hipStream_t)stream);
    if (dtype == SARSSL_BF16) return ffn2_launch<bf16, bf16, false>(g, d, (hipStream_t)stream);
    sarssl_set_error("sarssl_ffn2_fwd: dtype %d", dtype);
    return -1;
}

// backward: dh[M][4d] = (dz2 W2) * dropmask(p1, s1) * swish'(preact) (written: operand of both weight-gradient products),
// dln[M][d] = dh W1.  w2tp / w1tp: packs of W2^T ([4d x d]) and W1^T ([d x 4d]).  dtype: SARSSL_BF16 (all 16-bit tensors bf16) or
// SARSSL_MIX16 (bf16 gradients, fp16 saved pre-activation / x_ln).  x_ln != null: the LayerNorm backward runs in the epilogue -
// dln receives dx = LN'(dh W1) + resid, dx2 (optional) = dx * dropmask(p2, s2) * gscale2, ln_partial [M / 64][2][d] the dgamma | dbeta partials.
extern "C" int sarssl_ffn2_bwd(const void* dz2, long lddz, const void* w2tp, const void* w1tp, const void* preact, void* dh, void* dln,
                               long lddln, long M, int d, float p1, unsigned long long s1, const void* x_ln, long ldx, const float* ln_gamma,
                               const float* ln_mean, const float* ln_rstd, const void* resid, long ldr, void* dx2, float p2,
                               unsigned long long s2, float gscale2, float* ln_partial, int dtype, void* stream) {
    SARSSL_REQUIRE(sarssl_ffn2_supported(M, d) && lddz % 8 == 0 && lddln % 8 == 0 && preact && dh, "sarssl_ffn2_bwd");
    SARSSL_REQUIRE(!x_ln || (ldx % 8 == 0 && ln_gamma && ln_mean && ln_rstd && (!resid || ldr % 8 == 0)), "sarssl_ffn2_bwd(layernorm)");
    Ffn2Args g;
    g.A = dz2; g.lda = lddz; g.W1p = w2tp; g.W2p = w1tp; g.b1 = nullptr; g.b2 = nullptr; g.P = const_cast<void*>(preact); g.Hs = dh; g.Y = dln;
    g.ldy = lddln; g.R = x_ln ? resid : nullptr; g.ldr = ldr; g.p1 = p1; g.p2 = p2; g.s1 = s1; g.s2 = s2; g.salt = sarssl_dropout_salt(); g.rot = ffn_rot();
    g.out_scale = x_ln ? gscale2 : 1.f;
    g.M = (int)M;
    g.X = x_ln; g.ldx = ldx; g.ln_g = ln_gamma; g.ln_b = nullptr; g.ln_eps = 0.f; g.LNout = nullptr; g.ln_mean = const_cast<float*>(ln_mean);
    g.ln_rstd = const_cast<float*>(ln_rstd); g.Y2 = x_ln ? dx2 : nullptr; g.ln_partial = x_ln ? ln_partial : nullptr;
    if (dtype == SARSSL_BF16) return ffn2_launch<bf16, bf16, true>(g, d, (hipStream_t)stream);
    if (dtype == SARSSL_MIX16) return ffn2_launch<bf16, f16, true>(g, d, (hipStream_t)stream);
    sarssl_set_error("sarssl_ffn2_bwd: dtype %d", dtype);
    return -1;
}
