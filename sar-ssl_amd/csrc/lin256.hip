// Row-tile-resident Linear layers of the d = 256 Conformer blocks for gfx950: Y[M][N] = epilogue( A[M][K] . W[N][K]^T ), N and K multiples
// of 256 (K <= 768), one 512-thread workgroup per 64 rows.
//
// Why: the per-shape table of the step (profiles/r05_step_gemm_table.txt) lists the d = 256 projections of the spat encoder - q/k/v
// (N = 768), the two pointwise convolutions (N = 512 / 256), the attention output projection and their data gradients - at 25-35 us for
// 2-6 GFLOP (77-200 TFLOP/s, 17-50 MB of HBM traffic: neither pipe is busy).  They are four K-tiles of main loop behind a launch, a
// prologue and an epilogue, each in front or behind a LayerNorm launch of its own.  The fused feed-forward kernel (ffn2.hip) runs twice
// such a product in the same time; this is its single-product form with the same building blocks:
//   * the 64 x K input tile is staged ONCE (optionally through the module's LayerNorm: arithmetic of layernorm_fwd_kernel, bit-identical)
//     and stays in LDS; weights come from fragment-order packs (sarssl_ffn_pack) straight into MFMA operand registers through a 16-deep
//     queue, every element once per workgroup; waves are laid out 1 x 8 over 256 output columns, all 64 rows each;
//   * per 256-column chunk the accumulators go through an f32 LDS staging tile to 8-wide row pieces: bias, dropout (the same function of
//     (seed, row * N + column) as gemm_epilogue.h), output scale, residual - or, for the data gradient that ends in a LayerNorm
//     (q/k/v and the first pointwise convolution), that LayerNorm's backward (arithmetic of layernorm_bwd_kernel) with the dropped copy
//     for the next module of the chain and the dgamma / dbeta partials.
// Roles in the MFMA are swapped (weights = row operand) as in gemm.hip / ffn2.hip.
#include "ffn_common.h"

#define LIN_NT 512

struct LinArgs {
    const void* A; long lda;                   // [M][K] input rows (ignored when X drives a LayerNorm prologue)
    const void* Wp;                            // pack of W [N x K] (block (n / 32, k / 16))
    const float* bias;                         // [N] or null
    void* Y; long ldy;                         // [M][N]
    const void* R; long ldr;                   // residual rows [M][N] or null
    float p; unsigned long long seed; const unsigned long long* salt; float out_scale;
    int M, N, K;
    // forward LayerNorm prologue (K == 256): A tile = LayerNorm(X rows), written to LNout [M][256] with ln_mean / ln_rstd
    const void* X; long ldx; const float* ln_g; const float* ln_b; float ln_eps; void* LNout; float* ln_mean; float* ln_rstd;
    // LayerNorm backward epilogue (N == 256): Y = dx = LN'(product) (+ R), Y2 (optional) = dx * dropmask(p, seed) * out_scale,
    // ln_partial [grid][2][256]; XS = the LayerNorm's saved input rows (type TX), statistics in ln_mean / ln_rstd
    const void* XS; long ldxs; void* Y2; float* ln_partial;
};

// T: 16-bit type of A, the pack, Y, R (MFMA operand type); TX: type of XS.  KMAX: the input tile's LDS capacity (256: any N - the staging
// tile sits behind it; 768: N == 256 - the staging tile reuses the input tile's memory once the products are done).
template <typename T, typename TX, int KMAX>
__global__ __launch_bounds__(LIN_NT) void lin256_kernel(LinArgs g) {
    constexpr int D = 256, PY = D + 4;
    constexpr int SA_ELEMS = 64 * (KMAX + 8), YST_ELEMS = 64 * PY * 2;
    constexpr int LDS_ELEMS = KMAX == 256 ? SA_ELEMS + YST_ELEMS : (SA_ELEMS > YST_ELEMS ? SA_ELEMS : YST_ELEMS);
    __shared__ __attribute__((aligned(16))) uint16_t smem[LDS_ELEMS];
    uint16_t* sA = smem;
    float* sY = (float*)(KMAX == 256 ? smem + SA_ELEMS : smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long m0 = (long)blockIdx.x * 64;
    const int K = g.K, N = g.N, PA = K + 8, KS = K >> 4, NCH = N >> 8, KCH = K >> 8;
    const uint4* Wq = (const uint4*)g.Wp + lane;
    // the wave's weight stream: for every 256-column chunk nc the fragments of n-block nc * 8 + w, k-steps 0 .. KS - 1; a 16-step segment
    // (nc, kc) is one contiguous 16-KiB run of the pack
    auto segment = [&](int nc, int kc) -> const uint4* { return Wq + ((long)((nc * 8 + w) * KS + kc * 16) << 6); };
    uint4 q[16];
    {
        const uint4* s0 = segment(0, 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) q[j] = s0[j << 6];
    }

    // ---- input tile -> LDS
    if (g.X) {      // LayerNorm prologue (K == 256): see ffn2.hip - wave w normalises rows 8 w .. 8 w + 7 exactly like layernorm_fwd_kernel
        const T* X = (const T*)g.X;
        T* LN = (T*)g.LNout;
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ld4(X + (m0 + w * 8 + j) * g.ldx + lane * 4);
        const float4 gam = *(const float4*)(g.ln_g + lane * 4), bet = *(const float4*)(g.ln_b + lane * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = w * 8 + j;
            float sm = 0.f;
            sm += v[j].x + v[j].y + v[j].z + v[j].w;
            const float mu = wave_sum(sm) / (float)D;
            float qv = 0.f;
            {
                const float a = v[j].x - mu, b = v[j].y - mu, c = v[j].z - mu, e = v[j].w - mu;
                qv += a * a + b * b + c * c + e * e;
            }
            const float rs = rsqrtf(wave_sum(qv) / (float)D + g.ln_eps);
            const float4 o = make_float4((v[j].x - mu) * rs * gam.x + bet.x, (v[j].y - mu) * rs * gam.y + bet.y,
                                         (v[j].z - mu) * rs * gam.z + bet.z, (v[j].w - mu) * rs * gam.w + bet.w);
            st4(LN + (m0 + row) * (long)D + lane * 4, o);
            const float ov[4] = {o.x, o.y, o.z, o.w};
            *(uint2*)&sA[row * PA + lane * 4] = pack4<T>(ov);
            if (lane == 0) { g.ln_mean[m0 + row] = mu; g.ln_rstd[m0 + row] = rs; }
        }
    } else {
        const T* A = (const T*)g.A;
        const int cpr = K >> 3;                                       // 16-byte chunks per row: 32 | 64 | 96
        for (int i = tid; i < 64 * cpr; i += LIN_NT) {
            const int row = i / cpr, ch = i - row * cpr;
            *(uint4*)&sA[row * PA + ch * 8] = *(const uint4*)(A + (m0 + row) * g.lda + ch * 8);
        }
    }
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    __syncthreads();

    FfnDrop dr;
    dr.init(g.p, g.seed, g.salt);
    const int r = tid >> 5, ch = tid & 31, n = ch * 8;               // row pass: 32 threads x 8 columns = one 256-column row piece
    T* Yo = (T*)g.Y;

    for (int nc = 0; nc < NCH; ++nc) {
        f32x16 S[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) S[i][e] = 0.f;
        for (int kc = 0; kc < KCH; ++kc) {
            const uint16_t* a0 = sA + frow * PA + kc * 256 + fk;
            const uint16_t* a1 = sA + (32 + frow) * PA + kc * 256 + fk;
            bf16x8 fa[2][2];
            fa[0][0] = *(const bf16x8*)a0;
            fa[0][1] = *(const bf16x8*)a1;
            // refills: the next segment of the stream (past the end: a harmless re-read of the last chunk's first segment)
            int kcn = kc + 1, ncn = nc;
            if (kcn == KCH) { kcn = 0; ncn = nc + 1 < NCH ? nc + 1 : nc; }
            const uint4* nxt = segment(ncn, kcn);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int cur = ks & 1, nx = cur ^ 1;
                if (ks + 1 < 16) {
                    fa[nx][0] = *(const bf16x8*)(a0 + (ks + 1) * 16);
                    fa[nx][1] = *(const bf16x8*)(a1 + (ks + 1) * 16);
                }
                const bf16x8 wf = __builtin_bit_cast(bf16x8, q[ks]);
                S[0] = mfma16<T>(wf, fa[cur][0], S[0]);
                S[1] = mfma16<T>(wf, fa[cur][1], S[1]);
                q[ks] = nxt[ks << 6];
            }
        }
        // ---- accumulators -> f32 staging -> 8-wide row pieces
        __syncthreads();                  // (staging may alias the input tile, and the previous chunk's row pass must be done with it)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *(float4*)&sY[(i * 32 + frow) * PY + w * 32 + 8 * gq + 4 * (lane >> 5)] =
                    make_float4(S[i][4 * gq + 0], S[i][4 * gq + 1], S[i][4 * gq + 2], S[i][4 * gq + 3]);
        __syncthreads();
        const int col = nc * 256 + n;
        if (g.XS) {
            // LayerNorm backward on the tile's rows (N == 256): see ffn2.hip / layernorm_bwd_kernel
            const TX* Xs = (const TX*)g.XS;
            float gam8[8], ag[8], ab[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { gam8[e] = g.ln_g[n + e]; ag[e] = 0.f; ab[e] = 0.f; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = r + 16 * j;
                const long m = m0 + row;
                const float4 c0 = *(const float4*)&sY[row * PY + n], c1 = *(const float4*)&sY[row * PY + n + 4];
                const float av[8] = {round_as<T>(c0.x), round_as<T>(c0.y), round_as<T>(c0.z), round_as<T>(c0.w),
                                     round_as<T>(c1.x), round_as<T>(c1.y), round_as<T>(c1.z), round_as<T>(c1.w)};
                const f8 xx = ld8(Xs + m * g.ldxs + n);
                const float mu = g.ln_mean[m], rs = g.ln_rstd[m];
                float xh[8], gv[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[e] = (xx.v[e] - mu) * rs;
                    gv[e] = av[e] * gam8[e];
                    s1 += gv[e]; s2 += gv[e] * xh[e];
                    ag[e] += av[e] * xh[e]; ab[e] += av[e];
                }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                s1 /= (float)D; s2 /= (float)D;
                f8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] = rs * (gv[e] - s1 - xh[e] * s2);
                if (g.R) {
                    const f8 rr = ld8((const T*)g.R + m * g.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.v[e] += rr.v[e];
                }
                st8(Yo + m * g.ldy + n, v);
                if (g.Y2) {
                    if (dr.p > 0.f) {
                        float k0[4], k1[4];
                        const unsigned long long base = (unsigned long long)m * D + n;
                        dr.scale4(base, k0);
                        dr.scale4(base + 4, k1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v.v[e] *= k0[e]; v.v[4 + e] *= k1[e]; }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.v[e] *= g.out_scale;
                    st8((T*)g.Y2 + m * (long)D + n, v);
                }
            }
            if (g.ln_partial) {
                __syncthreads();
                float* sR = sY;                                        // [16][2][256]
#pragma unroll
                for (int e = 0; e < 8; ++e) { sR[(r * 2 + 0) * D + n + e] = ag[e]; sR[(r * 2 + 1) * D + n + e] = ab[e]; }
                __syncthreads();
                float* P = g.ln_partial + (long)blockIdx.x * 2 * D;
                {
                    const int c = tid;                                 // 2 * 256 = 512 sums, one per thread
                    float acc_ = 0.f;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) acc_ += sR[rr * 2 * D + c];
                    P[c] = acc_;
                }
            }
            return;
        }
        float bias8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
        if (g.bias) {
            const float4 b0 = *(const float4*)(g.bias + col), b1 = *(const float4*)(g.bias + col + 4);
            bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r + 16 * j;
            const long m = m0 + row;
            const float4 c0 = *(const float4*)&sY[row * PY + n], c1 = *(const float4*)&sY[row * PY + n + 4];
            f8 v;
            v.v[0] = c0.x + bias8[0]; v.v[1] = c0.y + bias8[1]; v.v[2] = c0.z + bias8[2]; v.v[3] = c0.w + bias8[3];
            v.v[4] = c1.x + bias8[4]; v.v[5] = c1.y + bias8[5]; v.v[6] = c1.z + bias8[6]; v.v[7] = c1.w + bias8[7];
            if (dr.p > 0.f) {
                float k0[4], k1[4];
                const unsigned long long base = (unsigned long long)m * N + col;
                dr.scale4(base, k0);
                dr.scale4(base + 4, k1);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v.v[e] *= k0[e]; v.v[4 + e] *= k1[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] *= g.out_scale;
            if (g.R) {
                const f8 rr = ld8((const T*)g.R + m * g.ldr + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] += rr.v[e];
            }
            st8(Yo + m * g.ldy + col, v);
        }
    }
}

extern "C" int sarssl_lin256_supported(long M, int N, int K) {
    return (M > 0 && M % 64 == 0 && N > 0 && N % 256 == 0 && K > 0 && K % 256 == 0 && K <= 768 && (K == 256 || N == 256)) ? 1 : 0;
}

template <typename T, typename TX>
static int lin256_launch(const LinArgs& g, hipStream_t st) {
    const int grid = g.M / 64;
    if (g.K == 256) lin256_kernel<T, TX, 256><<<grid, LIN_NT, 0, st>>>(g);
    else lin256_kernel<T, TX, 768><<<grid, LIN_NT, 0, st>>>(g);
    SARSSL_CHECK_LAUNCH("lin256_kernel");
    return 0;
}

// y[M][N] = resid + out_scale * drop(p, seed)( a W^T + bias ), wp = pack(W [N x K]) (sarssl_ffn_pack); M % 64 == 0, N and K multiples of 256,
// K <= 768, K == 256 or N == 256 (sarssl_lin256_supported).  x_ln != null (K == 256): a = LayerNorm(x_ln) is formed in the launch
// (written to ln_out [M][256] with ln_mean / ln_rstd [M]); `a` is ignored.  dtype: SARSSL_F16 | SARSSL_BF16 (every 16-bit tensor).
extern "C" int sarssl_lin256_fwd(const void* a, long lda, const void* wp, const float* bias, void* y, long ldy, const void* resid, long ldr,
                                 long M, int N, int K, float p, unsigned long long seed, float out_scale, const void* x_ln, long ldx,
                                 const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out, float* ln_mean, float* ln_rstd,
                                 int dtype, void* stream) {
    SARSSL_REQUIRE(sarssl_lin256_supported(M, N, K) && ldy % 8 == 0 && (!resid || ldr % 8 == 0), "sarssl_lin256_fwd");
    SARSSL_REQUIRE(x_ln ? (K == 256 && ldx % 8 == 0 && ln_gamma && ln_beta && ln_out && ln_mean && ln_rstd) : (a != nullptr && lda % 8 == 0),
                   "sarssl_lin256_fwd(layernorm)");
    LinArgs g;
    g.A = a; g.lda = lda; g.Wp = wp; g.bias = bias; g.Y = y; g.ldy = ldy; g.R = resid; g.ldr = ldr; g.p = p; g.seed = seed;
    g.salt = sarssl_dropout_salt(); g.out_scale = out_scale; g.M = (int)M; g.N = N; g.K = K;
    g.X = x_ln; g.ldx = ldx; g.ln_g = ln_gamma; g.ln_b = ln_beta; g.ln_eps = ln_eps; g.LNout = ln_out; g.ln_mean = ln_mean; g.ln_rstd = ln_rstd;
    g.XS = nullptr; g.ldxs = 0; g.Y2 = nullptr; g.ln_partial = nullptr;
    if (dtype == SARSSL_F16) return lin256_launch<f16, f16>(g, (hipStream_t)stream);
    if (dtype == SARSSL_BF16) return lin256_launch<bf16, bf16>(g, (hipStream_t)stream);
    sarssl_set_error("sarssl_lin256_fwd: dtype %d", dtype);
    return -1;
}

// data gradient dx[M][N] = dy[M][K] Wt^T with wtp = pack(W^T [N x K]) (N = the layer's input width, K = its output width), bf16.
// x_ln != null (N == 256; the layer's input came out of a LayerNorm whose saved input is x_ln, dtype bf16 | fp16 = SARSSL_MIX16): the
// epilogue runs that LayerNorm's backward - dx receives LN'(dy W) + resid, dx2 (optional) = dx * dropmask(p2, s2) * gscale2,
// ln_partial [M / 64][2][256] the dgamma | dbeta partials (sarssl_ln_param_reduce_multi, nparts = M / 64).
extern "C" int sarssl_lin256_bwd(const void* dy, long lddy, const void* wtp, void* dx, long lddx, long M, int N, int K, const void* x_ln,
                                 long ldx, const float* ln_gamma, const float* ln_mean, const float* ln_rstd, const void* resid, long ldr,
                                 void* dx2, float p2, unsigned long long s2, float gscale2, float* ln_partial, int dtype, void* stream) {
    SARSSL_REQUIRE(sarssl_lin256_supported(M, N, K) && lddy % 8 == 0 && lddx % 8 == 0, "sarssl_lin256_bwd");
    SARSSL_REQUIRE(!x_ln || (N == 256 && ldx % 8 == 0 && ln_gamma && ln_mean && ln_rstd && (!resid || ldr % 8 == 0)), "sarssl_lin256_bwd(layernorm)");
    LinArgs g;
    g.A = dy; g.lda = lddy; g.Wp = wtp; g.bias = nullptr; g.Y = dx; g.ldy = lddx; g.R = x_ln ? resid : nullptr; g.ldr = ldr;
    g.p = x_ln ? p2 : 0.f; g.seed = s2; g.salt = sarssl_dropout_salt(); g.out_scale = x_ln ? gscale2 : 1.f; g.M = (int)M; g.N = N; g.K = K;
    g.X = nullptr; g.ldx = 0; g.ln_g = ln_gamma; g.ln_b = nullptr; g.ln_eps = 0.f; g.LNout = nullptr; g.ln_mean = const_cast<float*>(ln_mean);
    g.ln_rstd = const_cast<float*>(ln_rstd); g.XS = x_ln; g.ldxs = ldx; g.Y2 = x_ln ? dx2 : nullptr; g.ln_partial = x_ln ? ln_partial : nullptr;
    if (dtype == SARSSL_BF16) return lin256_launch<bf16, bf16>(g, (hipStream_t)stream);
    if (dtype == SARSSL_MIX16) return lin256_launch<bf16, f16>(g, (hipStream_t)stream);
    sarssl_set_error("sarssl_lin256_bwd: dtype %d", dtype);
    return -1;
}
