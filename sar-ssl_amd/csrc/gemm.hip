// Batched bf16-MFMA GEMM with fused epilogue for gfx950.
//
//   C[z][m][n] = epilogue( alpha * sum_k opA(A)[z][m][k] * opB(B)[z][n][k] )
//
// Layout flags say which dimension of each operand is contiguous in memory:
//   a_kc = 1: A is [M][K] (K contiguous, row stride lda);   a_kc = 0: A is [K][M] (M contiguous)
//   b_kc = 1: B is [N][K] (nn.Linear weight layout);          b_kc = 0: B is [K][N]
// so Linear forward is (1,1), dX = dY*W is (1,0), dW = dY^T*X is (0,0) with A=dY, B=X.
//
// Storage types: f32 or bf16 per tensor.  MFMA inputs are always bf16 (v_mfma_f32_32x32x16_bf16,
// f32 accumulate).  For f32 storage the "precise" mode splits every operand x = hi + lo
// (hi = bf16(x), lo = bf16(x - hi)) at LDS-staging time and the host runs three passes
// (hi*hi, hi*lo, lo*hi) accumulating in an f32 workspace: ~2^-16 relative error per product,
// on the same matrix cores and the same code path as the fast bf16 mode.
//
// Tiling: 128x128x64 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 MFMA
// fragments), operands staged through registers into LDS as [row][k] with a 144-byte pitch
// (conflict-free ds_read_b128 fragment reads).  Operand roles are swapped in the MFMA
// (weights as the "A"/row operand) so each lane ends up with 4 consecutive n for one m and the
// epilogue stores 8/16-byte vectors.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define PITCH (BK + 8)

struct GemmArgs {
    const void* A; const void* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    int batch_inner;
    long sA0, sA1, sB0, sB1, sC0, sC1;
    float alpha, out_scale;
    const float* bias;          // [N] or null
    int act;                    // 0 none, 1 relu, 2 swish
    const void* resid; long ldr; long sR0, sR1; float res_scale;
    void* preact;               // optional (same dtype/ld as C): alpha*acc + bias before activation
    float* acc_ws; int acc_in, acc_out;   // f32 [nbatch][M][N] workspace for split passes
    int partA, partB;
    float p_drop; unsigned long long seed;
    int split_k, k_per_split;   // split_k > 0: blockIdx.z = z * split_k + s, epilogue = atomicAdd(C, alpha*acc) (f32 C only)
};

template <typename T, bool KC>
__device__ __forceinline__ void stage_tile(const T* __restrict__ src, long ld, int r0, int k0, int R, int K,
                                           uint16_t* __restrict__ s, int part, int tid) {
    if (KC) {
        uint4 regs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + i * 256;
            int row = c >> 3, kc = c & 7;
            int gr = r0 + row, gk = k0 + kc * 8;
            if (gr < R && gk < K) {
                const T* p = src + (long)gr * ld + gk;
                if (sizeof(T) == 2) regs[i] = *(const uint4*)p;
                else { f8 v = ld8(p); regs[i] = pack8_part(v, part); }
            } else regs[i] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + i * 256;
            int row = c >> 3, kc = c & 7;
            *(uint4*)&s[row * PITCH + kc * 8] = regs[i];
        }
    } else {
        // source is [K][R]: each thread takes a 4(k) x 8(r) unit and writes 8 k-quads (8 B each)
        int kq = tid >> 4, rq = tid & 15;
        int gc = r0 + rq * 8;
        uint32_t bits[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int gk = k0 + kq * 4 + j;
            if (gk < K && gc < R) {
                const T* p = src + (long)gk * ld + gc;
                if (sizeof(T) == 2) {
                    uint4 u = *(const uint4*)p;
                    bits[j][0] = u.x & 0xffffu; bits[j][1] = u.x >> 16; bits[j][2] = u.y & 0xffffu; bits[j][3] = u.y >> 16;
                    bits[j][4] = u.z & 0xffffu; bits[j][5] = u.z >> 16; bits[j][6] = u.w & 0xffffu; bits[j][7] = u.w >> 16;
                } else {
                    f8 v = ld8(p);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bits[j][e] = bf16_part_bits(v.v[e], part);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) bits[j][e] = 0;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint2 w;
            w.x = bits[0][e] | (bits[1][e] << 16);
            w.y = bits[2][e] | (bits[3][e] << 16);
            *(uint2*)&s[(rq * 8 + e) * PITCH + kq * 4] = w;
        }
    }
}

template <typename TA, typename TB, typename TC, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) uint16_t sA[BM * PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t sB[BN * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nsplit = g.split_k > 0 ? g.split_k : 1;
    const int z = blockIdx.z / nsplit, ks = blockIdx.z % nsplit;
    const int z0 = z / g.batch_inner, z1 = z % g.batch_inner;
    const int k_begin = g.split_k > 0 ? ks * g.k_per_split : 0;
    const int k_end = g.split_k > 0 ? min(g.K, k_begin + g.k_per_split) : g.K;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const TA* A = (const TA*)g.A + z0 * g.sA0 + z1 * g.sA1;
    const TB* B = (const TB*)g.B + z0 * g.sB0 + z1 * g.sB1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        stage_tile<TA, AKC>(A, g.lda, m0, k0, g.M, k_end, sA, g.partA, tid);
        stage_tile<TB, BKC>(B, g.ldb, n0, k0, g.N, k_end, sB, g.partB, tid);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int koff = kk * 16 + (lane >> 5) * 8;
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *(const bf16x8*)&sA[(wm * 64 + i * 32 + (lane & 31)) * PITCH + koff];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *(const bf16x8*)&sB[(wn * 64 + j * 32 + (lane & 31)) * PITCH + koff];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n..n+3] for m = base + (lane&31), n = base + 8g + 4(lane>>5)
    TC* C = (TC*)g.C + z0 * g.sC0 + z1 * g.sC1;
    const TC* Rz = g.resid ? (const TC*)g.resid + z0 * g.sR0 + z1 * g.sR1 : nullptr;
    TC* P = g.preact ? (TC*)g.preact + z0 * g.sC0 + z1 * g.sC1 : nullptr;
    float* W = g.acc_ws ? g.acc_ws + (long)z * g.M * g.N : nullptr;
    const bool vec_ok = ((g.N & 3) == 0) && ((g.ldc & 3) == 0) && (!g.resid || (g.ldr & 3) == 0);
    const float inv_keep = g.p_drop > 0.f ? 1.0f / (1.0f - g.p_drop) : 1.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + (lane & 31);
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + wn * 64 + j * 32 + 8 * gq + 4 * (lane >> 5);
                if (n >= g.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = g.alpha * acc[i][j][gq * 4 + e];
                const int nvalid = min(4, g.N - n);
                if (g.split_k > 0) {
                    if constexpr (sizeof(TC) == 4) {
                        for (int e = 0; e < nvalid; ++e) atomicAdd((float*)C + (long)m * g.ldc + n + e, v[e]);
                    }
                    continue;
                }
                if (g.acc_in) {
                    for (int e = 0; e < nvalid; ++e) v[e] += W[(long)m * g.N + n + e];
                }
                if (g.acc_out) {
                    for (int e = 0; e < nvalid; ++e) W[(long)m * g.N + n + e] = v[e];
                    continue;
                }
                if (g.bias) {
                    for (int e = 0; e < nvalid; ++e) v[e] += g.bias[n + e];
                }
                if (P) {
                    if (vec_ok) st4(P + (long)m * g.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
                    else for (int e = 0; e < nvalid; ++e) st_f(P + (long)m * g.ldc + n + e, v[e]);
                }
                if (g.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (g.act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * sigmoidf_(v[e]);
                }
                if (g.p_drop > 0.f) {
                    const unsigned long long base = ((unsigned long long)z * g.M + m) * (unsigned long long)g.N + n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= dropout_scale(g.seed, base + e, g.p_drop, inv_keep);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= g.out_scale;
                if (Rz) {
                    if (vec_ok) {
                        float4 r = ld4(Rz + (long)m * g.ldr + n);
                        v[0] += g.res_scale * r.x; v[1] += g.res_scale * r.y; v[2] += g.res_scale * r.z; v[3] += g.res_scale * r.w;
                    } else for (int e = 0; e < nvalid; ++e) v[e] += g.res_scale * ld_f(Rz + (long)m * g.ldr + n + e);
                }
                if (vec_ok) st4(C + (long)m * g.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
                else for (int e = 0; e < nvalid; ++e) st_f(C + (long)m * g.ldc + n + e, v[e]);
            }
        }
    }
}

template <typename TA, typename TB, typename TC>
static int launch_layout(const GemmArgs& g, int a_kc, int b_kc, dim3 grid, hipStream_t st) {
    if (a_kc && b_kc) gemm_kernel<TA, TB, TC, true, true><<<grid, 256, 0, st>>>(g);
    else if (a_kc && !b_kc) gemm_kernel<TA, TB, TC, true, false><<<grid, 256, 0, st>>>(g);
    else if (!a_kc && b_kc) gemm_kernel<TA, TB, TC, false, true><<<grid, 256, 0, st>>>(g);
    else gemm_kernel<TA, TB, TC, false, false><<<grid, 256, 0, st>>>(g);
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
                           void* preact, float p_drop, unsigned long long seed,
                           int precise, float* ws, int split_k, void* stream) {
    SARSSL_REQUIRE(M > 0 && N > 0 && K > 0 && nbatch > 0 && batch_inner > 0, "sarssl_gemm");
    SARSSL_REQUIRE(a_kc ? (K % 8 == 0 && lda % 8 == 0) : (M % 8 == 0 && lda % 8 == 0), "sarssl_gemm(A alignment)");
    SARSSL_REQUIRE(b_kc ? (K % 8 == 0 && ldb % 8 == 0) : (N % 8 == 0 && ldb % 8 == 0), "sarssl_gemm(B alignment)");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.batch_inner = batch_inner; g.sA0 = sA0; g.sA1 = sA1; g.sB0 = sB0; g.sB1 = sB1; g.sC0 = sC0; g.sC1 = sC1;
    g.alpha = alpha; g.out_scale = out_scale; g.bias = bias; g.act = act;
    g.resid = resid; g.ldr = ldr; g.sR0 = sR0; g.sR1 = sR1; g.res_scale = res_scale;
    g.preact = preact; g.acc_ws = nullptr; g.acc_in = 0; g.acc_out = 0; g.partA = 0; g.partB = 0;
    g.p_drop = p_drop; g.seed = seed;
    g.split_k = 0; g.k_per_split = K;
    if (split_k > 0) {
        // accumulate mode: C (f32) += alpha * A*B, no other epilogue; K split over split_k workgroups per tile
        SARSSL_REQUIRE(dtC == SARSSL_F32 && !bias && !resid && !preact && act == 0 && p_drop == 0.f, "sarssl_gemm(split_k epilogue)");
        int per = ((K + split_k - 1) / split_k + BK - 1) / BK * BK;
        g.split_k = (K + per - 1) / per; g.k_per_split = per;
    }
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, nbatch * (g.split_k > 0 ? g.split_k : 1));
    hipStream_t st = (hipStream_t)stream;
    if (dtA == SARSSL_BF16 && dtB == SARSSL_BF16 && dtC == SARSSL_BF16)
        return launch_layout<bf16, bf16, bf16>(g, a_kc, b_kc, grid, st);
    if (dtA == SARSSL_BF16 && dtB == SARSSL_BF16 && dtC == SARSSL_F32)
        return launch_layout<bf16, bf16, float>(g, a_kc, b_kc, grid, st);
    if (dtA == SARSSL_F32 && dtB == SARSSL_F32 && dtC == SARSSL_F32) {
        if (!precise) return launch_layout<float, float, float>(g, a_kc, b_kc, grid, st);
        if (g.split_k > 0) {          // the accumulate epilogue is linear: the three split-precision passes just add up
            GemmArgs p = g;
            p.partA = 0; p.partB = 1; int rc = launch_layout<float, float, float>(p, a_kc, b_kc, grid, st); if (rc) return rc;
            p.partA = 1; p.partB = 0; rc = launch_layout<float, float, float>(p, a_kc, b_kc, grid, st); if (rc) return rc;
            p.partA = 0; p.partB = 0; return launch_layout<float, float, float>(p, a_kc, b_kc, grid, st);
        }
        SARSSL_REQUIRE(ws != nullptr, "sarssl_gemm(precise needs workspace)");
        g.acc_ws = ws;
        GemmArgs p = g;
        p.partA = 0; p.partB = 1; p.acc_in = 0; p.acc_out = 1;          // hi*lo
        int rc = launch_layout<float, float, float>(p, a_kc, b_kc, grid, st); if (rc) return rc;
        p.partA = 1; p.partB = 0; p.acc_in = 1; p.acc_out = 1;          // + lo*hi
        rc = launch_layout<float, float, float>(p, a_kc, b_kc, grid, st); if (rc) return rc;
        p.partA = 0; p.partB = 0; p.acc_in = 1; p.acc_out = 0;          // + hi*hi, then epilogue
        return launch_layout<float, float, float>(p, a_kc, b_kc, grid, st);
    }
    sarssl_set_error("sarssl_gemm: unsupported dtype combination (%d,%d,%d)", dtA, dtB, dtC);
    return -1;
}
