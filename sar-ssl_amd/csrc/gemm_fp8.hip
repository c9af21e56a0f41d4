// OCP-fp8 (e4m3fn) GEMM path for gfx950: C = epilogue( sA*sB * A8 B8^T ), A8 [M][K], B8 [N][K] fp8 with per-tensor scales
// chosen on the device, f32 accumulation on the block-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 with unit block
// scales (E8M0 = 127) - the only fp8 MFMA on this chip that runs above the bf16 rate (K = 64 per instruction instead of 16 at the
// same issue interval; MI355X_MICROARCH.md).  Operand bytes per tile are half of bf16, which also relieves the L2 -> LDS path that
// bounds the bf16 GEMM at these shapes.  BASELINE.json config 5 ("fp8 MFMA attention/FFN path"); no reference counterpart (the
// reference is fp32 / fp16-AMP, code/learner.py:46-50): pinned against the bf16 path in tests/test_gpu_fp8.py.
//
//   sarssl_fp8_quantize   amax over the tensor (float-bit atomicMax) -> q = e4m3(x * 448 / amax), inv_scale = amax / 448;
//                         optional transposed output (weights: the input-gradient GEMM needs W^T with K contiguous)
//   sarssl_gemm_fp8       128 x 128 x 128(k) tiles, 4 waves, same register-staged LDS image as the bf16 kernel (144-byte row
//                         pitch: a row holds 128 fp8 = 2 k-blocks of 64), fragment = 32 contiguous bytes per lane per k-block,
//                         same fused epilogue (bias / Swish / ReLU / pre-activation output / activation backward / dropout /
//                         residual) through gemm_epilogue.h.
// hardware semantics checked by tools/probe_fp8.hip (conversion round trip, rounding, unit-scale MFMA == plain dot products).
#include "gemm_epilogue.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;

// ------------------------------------------------------------------------------------------------ quantisation
template <typename T>
__global__ void fp8_amax_kernel(const T* __restrict__ x, long rows, long cols, long ld, unsigned* __restrict__ amax_bits) {
    const long n8 = rows * (cols >> 3);
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long r = i / (cols >> 3), c = (i % (cols >> 3)) << 3;
        const f8 v = ld8(x + r * ld + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v.v[e]));
    }
    __shared__ float smax[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per workgroup (same-address atomics serialise at ~12 ns each); non-negative floats order like their bit patterns
    if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]))));
}

__device__ __forceinline__ uint2 cvt8_fp8(const f8& v, float s) {
    uint2 o;
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(v.v[0] * s, v.v[1] * s, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v.v[2] * s, v.v[3] * s, w, true);
    o.x = (unsigned)w;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v.v[4] * s, v.v[5] * s, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v.v[6] * s, v.v[7] * s, w, true);
    o.y = (unsigned)w;
    return o;
}

template <typename T>
__global__ void fp8_quant_kernel(const T* __restrict__ x, long rows, long cols, long ld, const unsigned* __restrict__ amax_bits,
                                 uint8_t* __restrict__ q, long ldq, float* __restrict__ inv_scale) {
    const float amax = __uint_as_float(*amax_bits);
    const float s = amax > 0.f ? 448.0f / amax : 1.0f;            // all-zero tensor: any scale
    if (blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = amax > 0.f ? amax / 448.0f : 1.0f;
    const long n8 = rows * (cols >> 3);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const long r = i / (cols >> 3), c = (i % (cols >> 3)) << 3;
        *(uint2*)(q + r * ldq + c) = cvt8_fp8(ld8(x + r * ld + c), s);
    }
}

// transposed: q[c][r] = e4m3(x[r][c] * s)   (weights only: a few MB, a simple 32 x 32 LDS transpose)
template <typename T>
__global__ void fp8_quant_t_kernel(const T* __restrict__ x, long rows, long cols, long ld, const unsigned* __restrict__ amax_bits,
                                   uint8_t* __restrict__ q, long ldq, float* __restrict__ inv_scale) {
    __shared__ float tile[32][33];
    const float amax = __uint_as_float(*amax_bits);
    const float s = amax > 0.f ? 448.0f / amax : 1.0f;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *inv_scale = amax > 0.f ? amax / 448.0f : 1.0f;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                 // 256 threads: 32 x 8
    const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 32;
    for (int k = ty; k < 32; k += 8) tile[k][tx] = (r0 + k < rows && c0 + tx < cols) ? ld_f(x + (r0 + k) * ld + c0 + tx) : 0.f;
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const long c = c0 + k, r = r0 + tx;
        if (c < cols && r < rows) {
            const int w = __builtin_amdgcn_cvt_pk_fp8_f32(tile[tx][k] * s, 0.f, 0, false);
            q[c * ldq + r] = (uint8_t)(w & 0xff);
        }
    }
}

// x: [rows][cols] (row stride ld) f32 | bf16 -> q: fp8 [rows][cols] (ldq) or, transpose != 0, [cols][rows] (ldq);
// amax_ws: one device word (zeroed here); inv_scale: device float = amax / 448.  cols % 8 == 0, ld % 8 == 0.
extern "C" int sarssl_fp8_quantize(const void* x, int dtype, long rows, long cols, long ld, void* q, long ldq, float* amax_ws,
                                   float* inv_scale, int transpose, void* stream) {
    SARSSL_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && (transpose || ldq % 8 == 0), "sarssl_fp8_quantize");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(amax_ws, 0, sizeof(unsigned), st) != hipSuccess) { sarssl_set_error("sarssl_fp8_quantize: memset"); return -2; }
    const long n8 = rows * (cols >> 3);
    int nblk = (int)((n8 + 255) / 256); if (nblk > 2048) nblk = 2048;
    const int nblk_amax = nblk > 512 ? 512 : nblk;
    if (dtype == SARSSL_BF16) fp8_amax_kernel<bf16><<<nblk_amax, 256, 0, st>>>((const bf16*)x, rows, cols, ld, (unsigned*)amax_ws);
    else if (dtype == SARSSL_F32) fp8_amax_kernel<float><<<nblk_amax, 256, 0, st>>>((const float*)x, rows, cols, ld, (unsigned*)amax_ws);
    else { sarssl_set_error("sarssl_fp8_quantize: unsupported dtype %d", dtype); return -1; }
    if (!transpose) {
        if (dtype == SARSSL_BF16) fp8_quant_kernel<bf16><<<nblk, 256, 0, st>>>((const bf16*)x, rows, cols, ld, (const unsigned*)amax_ws, (uint8_t*)q, ldq, inv_scale);
        else fp8_quant_kernel<float><<<nblk, 256, 0, st>>>((const float*)x, rows, cols, ld, (const unsigned*)amax_ws, (uint8_t*)q, ldq, inv_scale);
    } else {
        dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
        if (dtype == SARSSL_BF16) fp8_quant_t_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)x, rows, cols, ld, (const unsigned*)amax_ws, (uint8_t*)q, ldq, inv_scale);
        else fp8_quant_t_kernel<float><<<grid, 256, 0, st>>>((const float*)x, rows, cols, ld, (const unsigned*)amax_ws, (uint8_t*)q, ldq, inv_scale);
    }
    SARSSL_CHECK_LAUNCH("sarssl_fp8_quantize");
    return 0;
}

// ------------------------------------------------------------------------------------------------ GEMM
#define F8_BM 128
#define F8_BN 128
#define F8_BKB 128                    // bytes (= fp8 elements) of K per tile
#define F8_PITCH (F8_BKB + 16)        // 144-byte rows: conflict-free ds_read_b128 fragment reads (as the bf16 kernel)

// one 16-byte chunk of an operand tile: row (tid >> 3) + 32 i, bytes (tid & 7) * 16 .. + 16 of the current 128-byte K-slab
// (named registers instead of an array: hipcc 7.2 demoted the array form of this prefetch set to scratch memory)
template <bool EDGE>
__device__ __forceinline__ uint4 f8_chunk(const uint8_t* __restrict__ p, long ld, int i, int r0, int k0, int R, int Kend, int tid) {
    const uint8_t* q = p + (long)(32 * i) * ld;
    if constexpr (EDGE) {
        const int gr = r0 + (tid >> 3) + 32 * i, gk = k0 + (tid & 7) * 16;
        return (gr < R && gk < Kend) ? *(const uint4*)q : make_uint4(0, 0, 0, 0);
    } else return *(const uint4*)q;
}
#define F8_LOAD_TILE(K0)                                                                                                         \
    ra0 = f8_chunk<EDGE>(pa, g.lda, 0, m0, K0, g.M, g.K, tid); ra1 = f8_chunk<EDGE>(pa, g.lda, 1, m0, K0, g.M, g.K, tid);          \
    ra2 = f8_chunk<EDGE>(pa, g.lda, 2, m0, K0, g.M, g.K, tid); ra3 = f8_chunk<EDGE>(pa, g.lda, 3, m0, K0, g.M, g.K, tid);          \
    rb0 = f8_chunk<EDGE>(pb, g.ldb, 0, n0, K0, g.N, g.K, tid); rb1 = f8_chunk<EDGE>(pb, g.ldb, 1, n0, K0, g.N, g.K, tid);          \
    rb2 = f8_chunk<EDGE>(pb, g.ldb, 2, n0, K0, g.N, g.K, tid); rb3 = f8_chunk<EDGE>(pb, g.ldb, 3, n0, K0, g.N, g.K, tid);

template <typename TC, bool EDGE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void gemm_fp8_kernel(GemmArgs g, const float* __restrict__ sa,
                                                                                               const float* __restrict__ sb) {
    constexpr int PC = F8_BN + 4;
    constexpr int TILE_B = F8_BM * F8_PITCH;                     // 18432 bytes per operand tile
    constexpr int LDS_B = 2 * TILE_B > 64 * PC * 4 ? 2 * TILE_B : 64 * PC * 4;
    __shared__ __attribute__((aligned(16))) uint8_t smem[LDS_B];
    uint8_t* sA = smem;
    uint8_t* sB = smem + TILE_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.y & 7) == 0) {                                   // XCD-aware tile order (see gemm.hip)
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = lin & 7, j = lin >> 3;
        by = xcd + 8 * (j / (int)gridDim.x);
        bx = j % (int)gridDim.x;
    }
    const int m0 = by * F8_BM, n0 = bx * F8_BN;
    const uint8_t* pa = (const uint8_t*)g.A + (long)(m0 + (tid >> 3)) * g.lda + (tid & 7) * 16;
    const uint8_t* pb = (const uint8_t*)g.B + (long)(n0 + (tid >> 3)) * g.ldb + (tid & 7) * 16;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int ch = tid & 15, n = n0 + ch * 8;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (g.bias && n + e < g.N) ? g.bias[n + e] : 0.f;
    const float alpha_q = g.alpha * sa[0] * sb[0];                // per-tensor dequantisation scales (device scalars)

    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    F8_LOAD_TILE(0)
    uint8_t* wa = &sA[(tid >> 3) * F8_PITCH + (tid & 7) * 16];
    uint8_t* wb = &sB[(tid >> 3) * F8_PITCH + (tid & 7) * 16];
    for (int k0 = 0; k0 < g.K; k0 += F8_BKB) {
        *(uint4*)(wa) = ra0; *(uint4*)(wa + 32 * F8_PITCH) = ra1; *(uint4*)(wa + 64 * F8_PITCH) = ra2; *(uint4*)(wa + 96 * F8_PITCH) = ra3;
        *(uint4*)(wb) = rb0; *(uint4*)(wb + 32 * F8_PITCH) = rb1; *(uint4*)(wb + 64 * F8_PITCH) = rb2; *(uint4*)(wb + 96 * F8_PITCH) = rb3;
        __syncthreads();
        if (k0 + F8_BKB < g.K) {
            pa += F8_BKB; pb += F8_BKB;
            F8_LOAD_TILE(k0 + F8_BKB)
        }
        // both k-blocks' fragments are requested before the first MFMA (two register sets, order pinned as in gemm.hip)
        i32x8 fa0[2], fb0[2], fa1[2], fb1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa0[i] = *(const i32x8*)&sA[(wm * 64 + i * 32 + (lane & 31)) * F8_PITCH + half * 32];
            fb0[i] = *(const i32x8*)&sB[(wn * 64 + i * 32 + (lane & 31)) * F8_PITCH + half * 32];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa1[i] = *(const i32x8*)&sA[(wm * 64 + i * 32 + (lane & 31)) * F8_PITCH + 64 + half * 32];
            fb1[i] = *(const i32x8*)&sB[(wn * 64 + i * 32 + (lane & 31)) * F8_PITCH + 64 + half * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb0[j], fa0[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb1[j], fa1[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        __syncthreads();
    }
    TC* C = (TC*)g.C;
    const TC* Rz = (const TC*)g.resid;
    TC* P = (TC*)g.preact;
    const TC* Xa = (const TC*)g.aux;
    const bool vec_ok = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && (!g.resid || (g.ldr & 7) == 0);
    DropCtx dc;
    dc.init(g);
    float* sC = (float*)smem;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *(float4*)&sC[(wm * 32 + (lane & 31)) * PC + wn * 64 + j * 32 + 8 * gq + 4 * half] =
                    make_float4(acc[i][j][gq * 4 + 0], acc[i][j][gq * 4 + 1], acc[i][j][gq * 4 + 2], acc[i][j][gq * 4 + 3]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (tid >> 4) + 16 * k;
            const int m = m0 + (r >> 5) * 64 + i * 32 + (r & 31);
            if (EDGE && (m >= g.M || n >= g.N)) continue;
            f8 v;
            const float4 a0 = *(const float4*)&sC[r * PC + ch * 8], a1 = *(const float4*)&sC[r * PC + ch * 8 + 4];
            v.v[0] = alpha_q * a0.x; v.v[1] = alpha_q * a0.y; v.v[2] = alpha_q * a0.z; v.v[3] = alpha_q * a0.w;
            v.v[4] = alpha_q * a1.x; v.v[5] = alpha_q * a1.y; v.v[6] = alpha_q * a1.z; v.v[7] = alpha_q * a1.w;
            // (workspace pointers passed as runtime values: literal nullptrs here crash hipcc 7.2's SimplifyCFG at -O2 and above)
            epilogue8<TC, EDGE>(g, v, 0, m, n, C, Rz, P, Xa, g.acc_ws, g.acc_ws, bias8, vec_ok, dc, false, v);
        }
        if (i == 0) __syncthreads();
    }
}

// C[m][n] = epilogue(alpha * sa[0] * sb[0] * sum_k A8[m][k] B8[n][k]); A8 [M][lda], B8 [N][ldb] fp8 (e4m3fn), K % 16 == 0,
// lda / ldb % 16 == 0; C / resid / preact / aux: bf16 (dtC = 1) or f32 (dtC = 0); epilogue arguments as sarssl_gemm.
extern "C" int sarssl_gemm_fp8(const void* A8, const void* B8, const float* sa, const float* sb, void* C, int dtC, int M, int N, int K,
                               long lda, long ldb, long ldc, float alpha, float out_scale, const float* bias, int act,
                               const void* resid, long ldr, float res_scale, void* preact, const void* aux, int aux_act, float p_drop,
                               unsigned long long seed, void* stream) {
    SARSSL_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0 && lda % 16 == 0 && ldb % 16 == 0 && sa && sb, "sarssl_gemm_fp8");
    GemmArgs g = {};
    g.A = A8; g.B = B8; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.batch_inner = 1;
    g.alpha = alpha; g.out_scale = out_scale; g.bias = bias; g.act = act; g.resid = resid; g.ldr = ldr; g.res_scale = res_scale;
    g.preact = preact; g.aux = aux; g.aux_act = aux_act; g.aux_f16 = 0; g.p_drop = p_drop; g.seed = seed; g.salt = sarssl_dropout_salt(); g.prio = 0; g.k_per_split = K;
    const bool vec_ok = ((N & 7) == 0) && ((ldc & 7) == 0) && (!resid || (ldr & 7) == 0);
    const bool edge = (M % F8_BM) != 0 || (N % F8_BN) != 0 || (K % F8_BKB) != 0 || !vec_ok;
    dim3 grid((N + F8_BN - 1) / F8_BN, (M + F8_BM - 1) / F8_BM);
    hipStream_t st = (hipStream_t)stream;
    if (dtC == SARSSL_BF16) {
        if (edge) gemm_fp8_kernel<bf16, true><<<grid, 256, 0, st>>>(g, sa, sb);
        else gemm_fp8_kernel<bf16, false><<<grid, 256, 0, st>>>(g, sa, sb);
    } else if (dtC == SARSSL_F32) {
        if (edge) gemm_fp8_kernel<float, true><<<grid, 256, 0, st>>>(g, sa, sb);
        else gemm_fp8_kernel<float, false><<<grid, 256, 0, st>>>(g, sa, sb);
    } else { sarssl_set_error("sarssl_gemm_fp8: unsupported output dtype %d", dtC); return -1; }
    SARSSL_CHECK_LAUNCH("sarssl_gemm_fp8");
    return 0;
}
