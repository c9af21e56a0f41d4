// GEMM argument block and the fused 8-wide epilogue shared by the bf16 (gemm.hip) and fp8 (gemm_fp8.hip) kernels.
#pragma once
#include "common.h"

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
    const void* aux; int aux_act;   // optional (same dtype/ld as C): multiply by act'(aux) (1 relu, 2 swish) - fused activation backward
    int aux_f16;                    // aux is an fp16 tensor although C is bf16 (pre-activation saved by the fp16 forward pass)
    float* acc_ws; int acc_in, acc_out;   // f32 [nbatch][M][N] workspace for split passes
    int partA, partB;
    float p_drop; unsigned long long seed; const unsigned long long* salt;   // salt: device-resident seed addend (graph replay) or null
    int prio;                   // != 0: waves raise their issue priority during the MFMA phase of a K-tile (s_setprio)
    int split_k, k_per_split;   // split_k > 0: blockIdx.z = z * split_k + s; raw alpha*acc partial -> acc_ws[z][s][M][N], reduced into C afterwards
    float* csum_ws;             // grouped weight-gradient launch only: f32 [split_k][M] partial column sums of the A operand (= the bias
                                // gradient of the layer whose weight gradient this product is), written by the workgroups of column tile 0
    const void* A2; const void* B2; int nseg;     // K-segment products (sarssl_gemm_split, NT layout, same strides as A / B): the contraction
                                // runs over nseg segments of K - nseg = 2: A B^T + A B2^T; nseg = 3: A B^T + A2 B^T + A B2^T (x = hi + lo
                                // splits of an f32 operand in fp16 pairs: hi hi + lo hi + hi lo); 0 / 1: the plain product
    int row_shift;              // != 0 (= T, with M = N = ldc = T): row m of every batch matrix is stored m + 1 - T elements further
                                // (elements falling before the matrix are dropped): the relative-position shift of the reference
                                // (attention.py:105-113: pad one zero column, reinterpret (T, T+1) as (T+1, T), drop the first row)
#ifdef GEMM_STAMPS
    unsigned long long* stamps;     // probe build only (tools/gemm_stamps.py): s_memtime per K-loop phase of workgroup (0,0,0)
#endif
};

// fast path of sarssl_gemm for bf16 NT products without ragged edges (gemm_nt.hip): 0 = launched, 1 = not this kernel's shape
int sarssl_gemm_nt_try(const GemmArgs& g, int dtC, void* stream);

// per-thread dropout state of an epilogue: the salted seed, the hash key of index word 0 (dropout_key recomputed it - one of five hashes -
// for every 8-wide piece although it only changes every 2^33 elements), threshold and keep scale
struct DropCtx {
    unsigned long long seed; uint32_t key0, thr; float inv_keep;
    __device__ __forceinline__ void init(const GemmArgs& g) {
        inv_keep = g.p_drop > 0.f ? 1.0f / (1.0f - g.p_drop) : 1.0f;
        seed = 0; key0 = 0; thr = 0;
        if (g.p_drop > 0.f) { seed = salted_seed(g.seed, g.salt); key0 = dropout_key(seed, 0u); thr = dropout_thr16(g.p_drop); }
    }
};

// 8 elements at p decoded as fp16 (as_f16, 16-bit TC only) or as TC
template <typename TC>
__device__ __forceinline__ f8 ld8_as(const TC* p, bool as_f16) {
    if constexpr (sizeof(TC) == 2) {
        const uint4 u = *(const uint4*)p;
        return as_f16 ? unpack8<f16>(u) : unpack8<TC>(u);
    } else return ld8(p);
}
// ---- fused epilogue on one 8-wide piece of one output row (v = alpha * accumulator) -------------------------------------------------
template <typename TC, bool EDGE>
__device__ __forceinline__ void epilogue8(const GemmArgs& g, f8 v, int z, int m, int n, TC* __restrict__ C, const TC* __restrict__ Rz,
                                          TC* __restrict__ P, const TC* __restrict__ Xa, float* __restrict__ W, float* __restrict__ Wp,
                                          const float (&bias8)[8], bool vec_ok, const DropCtx& dc, const bool has_pre, const f8& pre) {
    const float inv_keep = dc.inv_keep;
    const int nvalid = EDGE ? min(8, g.N - n) : 8;
    const bool vec = EDGE ? (vec_ok && nvalid == 8) : true;
    if (g.split_k > 0) {                                 // raw partial for the split-K second stage
        float* q = Wp + (long)m * g.N + n;
        if (!EDGE || ((g.N & 3) == 0 && nvalid == 8)) {
            *(float4*)q = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]); *(float4*)(q + 4) = make_float4(v.v[4], v.v[5], v.v[6], v.v[7]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nvalid) q[e] = v.v[e];
        }
        return;
    }
    if (g.acc_in) {
        const float* q = W + (long)m * g.N + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) v.v[e] += q[e];
    }
    if (g.acc_out) {
        float* q = W + (long)m * g.N + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) q[e] = v.v[e];
        return;
    }
    if (g.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] += bias8[e];
    }
    const long co = (long)m * g.ldc + n;
    if (P) {
        if (vec) st8(P + co, v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nvalid) st_f(P + co + e, v.v[e]);
        }
    }
    if (g.act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] = fmaxf(v.v[e], 0.f);
    } else if (g.act == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] = v.v[e] * sigmoidf_(v.v[e]);
    }
    if (Xa) {                                            // dX epilogue: times act'(saved pre-activation)
        f8 h;
        if (has_pre) h = pre;
        else if (vec) h = ld8_as(Xa + co, g.aux_f16 != 0);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) h.v[e] = (e < nvalid) ? (g.aux_f16 ? ld_f((const f16*)Xa + co + e) : ld_f(Xa + co + e)) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (g.aux_act == 1) v.v[e] = h.v[e] > 0.f ? v.v[e] : 0.f;
            else { const float sg = sigmoidf_(h.v[e]); v.v[e] *= sg * (1.f + h.v[e] * (1.f - sg)); }
        }
    }
    if (g.p_drop > 0.f) {
        const unsigned long long base = ((unsigned long long)z * g.M + m) * (unsigned long long)g.N + n;
        const unsigned long long seed = dc.seed;
        if ((base & 1ull) == 0 && ((base + 7) >> 33) == 0) {          // the usual case: four pair hashes with the hoisted key
            const uint32_t pair = (uint32_t)(base >> 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t h = hash_u32((pair + q) ^ dc.key0);
                v.v[2 * q] *= (h & 0xffffu) >= dc.thr ? inv_keep : 0.0f;
                v.v[2 * q + 1] *= (h >> 16) >= dc.thr ? inv_keep : 0.0f;
            }
        } else if ((base & 1ull) == 0 && (((base + 7) >> 33) == (base >> 33))) dropout_apply8(v.v, seed, base, g.p_drop, inv_keep);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] *= dropout_scale(seed, base + e, g.p_drop, inv_keep);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v.v[e] *= g.out_scale;
    if (Rz) {
        const long ro = (long)m * g.ldr + n;
        if (vec) {
            const f8 rr = has_pre ? pre : ld8(Rz + ro);
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] += g.res_scale * rr.v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nvalid) v.v[e] += g.res_scale * ld_f(Rz + ro + e);
        }
    }
    if (EDGE && g.row_shift) {
        const long d0 = co + (m + 1 - g.row_shift);
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid && d0 + e >= 0) st_f(C + d0 + e, v.v[e]);
        return;
    }
    if (vec) st8(C + co, v);
    else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) st_f(C + co + e, v.v[e]);
    }
}

