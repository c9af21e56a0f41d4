// HBM-bound pieces of the CNN stem (code/model.py:50-64) and generic channels-last ([N rows][C ch])
// BatchNorm machinery shared with the Conformer conv module's BatchNorm1d:
//   mask_inputs   : builds the two masked 4-channel stem inputs from the STFT tensor (model.py:533-564)
//   stem_c1       : 1x1 conv 4 -> 64 (+ weight gradient)
//   stem_c4       : BN+ReLU prologue, 1x1 conv 64 -> 4 written in (B,T,F,4) order = patch-GEMM A operand
//   stem_c4_bwd   : dz3 = dy4*W4 masked by ReLU, dW4, and the BatchNorm-backward reductions in one pass
//   cl_stats / bn_finalize / bn_eval_affine / cl_affine_act / cl_bn_bwd_reduce / cl_bn_bwd_apply
// All reductions: per-thread partials -> LDS -> one f64 atomicAdd per (block, channel).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// x: (B, 2, F, T, 2) f32 [mic][f][t][reim];  mp: (B, T) u8 (1 = visible frame, 0 = masked);
// mch: (B) int32 masked channel.  spec/spat: (B, F, T, 4) with c = reim*2 + mic.
// mode 0: pretrain masks (model.py:541, :563);  mode 1: no masking, both outputs = x (model.py:676).
template <typename T>
__global__ void mask_inputs_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mp, const int* __restrict__ mch,
                                   int nb, int F, int Tn, int mode, T* __restrict__ spec, T* __restrict__ spat, int* __restrict__ ovf) {
    const long total = (long)nb * F * Tn;
    bool over = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % Tn);
        const long bf = i / Tn;
        const int f = (int)(bf % F), b = (int)(bf / F);
        const float2 m0 = *(const float2*)(x + ((((long)b * 2 + 0) * F + f) * Tn + t) * 2);
        const float2 m1 = *(const float2*)(x + ((((long)b * 2 + 1) * F + f) * Tn + t) * 2);
        float s0 = 1.f, s1 = 1.f, p = 1.f;
        if (mode == 0) {
            p = mp[(long)b * Tn + t] ? 1.f : 0.f;
            const int mc = mch[b];
            const float v0 = (mc == 0) ? 0.f : 1.f, v1 = (mc == 1) ? 0.f : 1.f;   // mask_ch_dense per mic
            s0 = (1.f - p) * v0 + p * (1.f - v0);
            s1 = (1.f - p) * v1 + p * (1.f - v1);
        }
        // this is where externally scaled data (the spectrum divided by mean|X_0| + eps, code/learner.py:539-542) is first encoded in the
        // forward dtype: a value outside fp16's range would become inf, and inf does NOT reach the loss (BatchNorm turns it into NaN,
        // the next ReLU's max(NaN, 0) into 0): flag it for the loss launch instead of training on silently clipped input
        if constexpr (__is_same(T, f16)) over = over || !(fmaxf(fmaxf(fabsf(m0.x), fabsf(m0.y)), fmaxf(fabsf(m1.x), fabsf(m1.y))) <= 65504.f);
        st4(spec + i * 4, make_float4(m0.x * s0, m1.x * s1, m0.y * s0, m1.y * s1));
        st4(spat + i * 4, make_float4(m0.x * p, m1.x * p, m0.y * p, m1.y * p));
    }
    if (over && ovf) atomicOr(ovf, 1);
}

// ------------------------------------------------------------------------------------------------
// 8 consecutive elements kept as loaded (bf16: one 16-byte register quad) until they are consumed
template <typename T> struct Raw8;
template <> struct Raw8<bf16> { uint4 u; };
template <> struct Raw8<float> { f8 v; };
template <> struct Raw8<f16> { uint4 u; };
__device__ __forceinline__ Raw8<f16> raw8_load(const f16* p) { Raw8<f16> r; r.u = *(const uint4*)p; return r; }
__device__ __forceinline__ f8 raw8_unpack(const Raw8<f16>& r) { return unpack8<f16>(r.u); }
__device__ __forceinline__ Raw8<bf16> raw8_load(const bf16* p) { Raw8<bf16> r; r.u = *(const uint4*)p; return r; }
__device__ __forceinline__ Raw8<float> raw8_load(const float* p) { Raw8<float> r; r.v = ld8(p); return r; }
__device__ __forceinline__ f8 raw8_unpack(const Raw8<float>& r) { return r.v; }
__device__ __forceinline__ f8 raw8_unpack(const Raw8<bf16>& r) {
    f8 o;
    o.v[0] = bf16_bits_to_f32(r.u.x & 0xffffu); o.v[1] = __uint_as_float(r.u.x & 0xffff0000u);
    o.v[2] = bf16_bits_to_f32(r.u.y & 0xffffu); o.v[3] = __uint_as_float(r.u.y & 0xffff0000u);
    o.v[4] = bf16_bits_to_f32(r.u.z & 0xffffu); o.v[5] = __uint_as_float(r.u.z & 0xffff0000u);
    o.v[6] = bf16_bits_to_f32(r.u.w & 0xffffu); o.v[7] = __uint_as_float(r.u.w & 0xffff0000u);
    return o;
}

// y1[p][co] = sum_c W1[co][c] a0[p][c]     a0: (N,4), y1: (N,64).  Thread = (pixel, 8-channel group).
// stats (optional f64[128]): per-channel sum / sum of squares of the stored y1 for the following BatchNorm.
template <typename T>
__global__ __launch_bounds__(256) void stem_c1_fwd_kernel(const T* __restrict__ a0, const float* __restrict__ W1, long npix,
                                                          T* __restrict__ y1, double* __restrict__ stats) {
    __shared__ float sred[4][8][16];
    const int cg = threadIdx.x & 7;
    float w[8][4];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) w[e][c] = W1[(cg * 8 + e) * 4 + c];
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const long nthreads = (long)gridDim.x * blockDim.x;
#ifndef SARSSL_C1F_U
#define SARSSL_C1F_U 4
#endif
    constexpr int U = SARSSL_C1F_U;                    // pixels in flight per thread: the pass is a pure 537 MB write stream
    // the input pixels of the NEXT round are requested before this round is computed and stored: the loads are tiny (8 bytes per
    // pixel, shared by 8 lanes) but their latency was the whole round trip of a wave (3.7 TB/s of stores at 4 waves / SIMD)
    float4 a[U], an[U];
    long g0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = ld4(a0 + (min(g0 + u * nthreads, npix * 8 - 1) >> 3) * 4);
    for (; g0 < npix * 8; g0 += nthreads * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long g = min(g0 + (U + u) * nthreads, npix * 8 - 1);     // clamped, unconditional (results of the overshoot are dropped)
            an[u] = ld4(a0 + (g >> 3) * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long g = g0 + u * nthreads;
            if (g >= npix * 8) continue;
            f8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] = w[e][0] * a[u].x + w[e][1] * a[u].y + w[e][2] * a[u].z + w[e][3] * a[u].w;
            st8(y1 + (g >> 3) * 64 + cg * 8, o);
            if (stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float r = round_as<T>(o.v[e]);   // what was stored
                    acc[e] += r; acc[8 + e] += r * r;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = an[u];
    }
    if (stats) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] += __shfl_xor(acc[i], 8, 64); acc[i] += __shfl_xor(acc[i], 16, 64); acc[i] += __shfl_xor(acc[i], 32, 64);
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane < 8) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sred[wave][lane][i] = acc[i];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
            const int g8 = c >> 3, e = c & 7;
            const float t = sred[0][g8][which * 8 + e] + sred[1][g8][which * 8 + e] + sred[2][g8][which * 8 + e] + sred[3][g8][which * 8 + e];
            atomicAdd(&stats[threadIdx.x], (double)t);
        }
    }
}

// dW1[co][c] = sum_p dy1[p][co] a0[p][c]   (double accumulators dW1d[64*4], zeroed by the caller)
// BNF = 1: dy1 is not materialised - the kernel reads dz1 (gradient w.r.t. relu(bn1(y1))) and y1 and applies the BatchNorm-backward
// normalisation dy1 = gamma*rstd*(g - s1/N - xhat*s2/N), g = dz1*relu'(bn1(y1)), on the fly (folded to A*g + B*y + C per channel).
// The first conv's input is data, so dy1 has no other consumer: this removes one 64-channel write and one read per encoder.
template <typename T, typename TA, int BNF>
__global__ void stem_c1_wgrad_kernel(const T* __restrict__ dy1, const TA* __restrict__ a0, long npix, double* __restrict__ dW1d,
                                     const TA* __restrict__ y1, const float* __restrict__ aff, const double* __restrict__ bnred, int use_stats) {
    __shared__ float red[256][33];
    const int cg = threadIdx.x & 7;
    float cA[8], cB[8], cC[8], thr[8];
    unsigned sgn = 0u;
    if (BNF) {
        const float invN = 1.0f / (float)npix;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = cg * 8 + e;
            const float sc = aff[ch], sh = aff[64 + ch], mu = aff[128 + ch], rs = aff[192 + ch];
            const float m1 = use_stats ? (float)bnred[ch] * invN : 0.f, m2 = use_stats ? (float)bnred[64 + ch] * invN : 0.f;
            cA[e] = sc; cB[e] = -sc * m2 * rs; cC[e] = -sc * m1 - cB[e] * mu;
            float t = sc != 0.f ? -sh / sc : (sh > 0.f ? -INFINITY : INFINITY);
            if (sc < 0.f) { t = -t; sgn |= 1u << e; }
            thr[e] = t;
        }
    }
    float acc[8][4];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[e][c] = 0.f;
    const long nthreads = (long)gridDim.x * blockDim.x;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < npix * 8; g += nthreads) {
        const long p = g >> 3;
        const float4 a = ld4(a0 + p * 4);
        f8 d = ld8(dy1 + p * 64 + cg * 8);
        if (BNF) {
            const f8 v = ld8(y1 + p * 64 + cg * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ys = __uint_as_float(__float_as_uint(v.v[e]) ^ (((sgn >> e) & 1u) << 31));
                const float g = ys > thr[e] ? d.v[e] : 0.f;
                d.v[e] = fmaf(cA[e], g, fmaf(cB[e], v.v[e], cC[e]));
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[e][0] += d.v[e] * a.x; acc[e][1] += d.v[e] * a.y; acc[e][2] += d.v[e] * a.z; acc[e][3] += d.v[e] * a.w; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) red[threadIdx.x][e * 4 + c] = acc[e][c];
    __syncthreads();
    // 256 outputs; thread o sums the 32 threads that share its channel group
    const int o = threadIdx.x;           // o = co*4 + c, co = cg*8 + e
    const int co = o >> 2, c = o & 3;
    const int ocg = co >> 3, oe = co & 7;
    float s = 0.f;
    for (int k = 0; k < 32; ++k) s += red[k * 8 + ocg][oe * 4 + c];
    atomicAdd(&dW1d[o], (double)s);
}

// ------------------------------------------------------------------------------------------------
// Whole backward of the first stem layer in ONE pass over (dz1, y1, a0).  With g = dz1 * relu'(bn1(y1)), xhat = (y1 - mean) * rstd:
//   BatchNorm sums   s1[co] = sum_p g,  s2[co] = sum_p g * xhat
//   dW1[co][c] = sum_p dy1[p][co] a0[p][c],  dy1 = gamma*rstd * (g - s1/N - xhat * s2/N)
//              = gamma*rstd * ( G[co][c] - (s1/N) * Sa[c] - (s2/N) * X[co][c] ),   G = sum g a0,  X = sum xhat a0,  Sa = sum a0
// so the gradient never has to be normalised per element: G, X, Sa, s1, s2 are accumulated together and a 256-thread finalize
// kernel combines them.  Replaces cl_bn_bwd_reduce + cl_bn_bwd_apply + stem_c1_wgrad (6 passes over 64-channel tensors -> 2).
// red: f64[644] = [G 64x4 | X 64x4 | s1 64 | s2 64 | Sa 4], zeroed by the caller.
template <typename T, typename TA>
__global__ __launch_bounds__(256) void stem_c1_bwd_kernel(const T* __restrict__ dz1, const TA* __restrict__ y1, const TA* __restrict__ a0,
                                                          long npix, const float* __restrict__ aff, double* __restrict__ red) {
    __shared__ float sred[4][8][84];
    const int cg = threadIdx.x & 7;
    float mu[8], rs[8], thr[8];
    unsigned sgn = 0u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = cg * 8 + e;
        const float sc = aff[ch], sh = aff[64 + ch];
        mu[e] = aff[128 + ch]; rs[e] = aff[192 + ch];
        float t = sc != 0.f ? -sh / sc : (sh > 0.f ? -INFINITY : INFINITY);
        if (sc < 0.f) { t = -t; sgn |= 1u << e; }
        thr[e] = t;
    }
    float acc[84];                               // [0,32) G[e][c], [32,64) X[e][c], [64,72) s1, [72,80) s2, [80,84) Sa
#pragma unroll
    for (int i = 0; i < 84; ++i) acc[i] = 0.f;
    const long nthreads = (long)gridDim.x * blockDim.x;
    for (long g0 = (long)blockIdx.x * blockDim.x + threadIdx.x; g0 < npix * 8; g0 += nthreads) {
        const long p = g0 >> 3;
        const float4 a = ld4(a0 + p * 4);
        const f8 d = ld8(dz1 + p * 64 + cg * 8);
        const f8 v = ld8(y1 + p * 64 + cg * 8);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ys = __uint_as_float(__float_as_uint(v.v[e]) ^ (((sgn >> e) & 1u) << 31));
            const float g = ys > thr[e] ? d.v[e] : 0.f;
            const float xh = (v.v[e] - mu[e]) * rs[e];
#pragma unroll
            for (int c = 0; c < 4; ++c) { acc[e * 4 + c] = fmaf(g, av[c], acc[e * 4 + c]); acc[32 + e * 4 + c] = fmaf(xh, av[c], acc[32 + e * 4 + c]); }
            acc[64 + e] += g;
            acc[72 + e] = fmaf(g, xh, acc[72 + e]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[80 + c] += av[c];
    }
#pragma unroll
    for (int i = 0; i < 84; ++i) {
        acc[i] += __shfl_xor(acc[i], 8, 64); acc[i] += __shfl_xor(acc[i], 16, 64); acc[i] += __shfl_xor(acc[i], 32, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 8) {
#pragma unroll
        for (int i = 0; i < 84; ++i) sred[wave][lane][i] = acc[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 644; o += 256) {
        int grp, slot;
        if (o < 512) { const int which = o >> 8, co = (o & 255) >> 2, c = o & 3; grp = co >> 3; slot = which * 32 + (co & 7) * 4 + c; }
        else if (o < 640) { const int which = (o - 512) >> 6, co = (o - 512) & 63; grp = co >> 3; slot = 64 + which * 8 + (co & 7); }
        else { grp = 0; slot = 80 + (o - 640); }               // Sa: every channel group saw every pixel once - take group 0's copy
        atomicAdd(&red[o], (double)(sred[0][grp][slot] + sred[1][grp][slot] + sred[2][grp][slot] + sred[3][grp][slot]));
    }
}
// The same pass for 16-bit activations and npix % 64 == 0, rebuilt like stem_c4_bwd_sums16_kernel: a thread owns 4 channels (44
// accumulators instead of 84), 4 waves / SIMD, four pixels (12 8-byte loads) requested before the first is used, packed f32 math, and
// the accumulators pinned after every pixel so that hipcc does not keep four pixels' worth of intermediates live.  The general kernel
// had one pixel (40 bytes) in flight per thread at 3 waves / SIMD: 253 us for 1.1 GB (4.4 TB/s) at B = 64.
// FROM_A0: y1 = W1 a0 is recomputed from the 4-channel input (4 packed FMAs per channel pair) instead of being read - the first
// stem layer's output is then never stored (engine.stem_fwd: the 3x3 convolution behind it forms it while staging, too).
template <bool FROM_A0, typename TA>     // TA: 16-bit type of the tensors saved by the forward pass (y1, a0); the gradient dz1 is bf16
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void stem_c1_bwd16_kernel(const bf16* __restrict__ dz1, const TA* __restrict__ y1, const TA* __restrict__ a0, long npix,
                          const float* __restrict__ aff, double* __restrict__ red, const float* __restrict__ W1) {
    typedef sarssl_f32x2 f2;
    __shared__ float sred[4][16][44];
    const int cq = threadIdx.x & 15, ps = threadIdx.x >> 4;          // channels 4cq..4cq+3; pixel slot 0..15
    f2 xa[2], xb[2];                                                  // xhat = y * xa + xb
    float thr[4];
    unsigned sgn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = cq * 4 + e;
        const float sc = aff[ch], sh = aff[64 + ch], mu = aff[128 + ch], rs = aff[192 + ch];
        float t = sc != 0.f ? -sh / sc : (sh > 0.f ? -INFINITY : INFINITY);          // relu'(y*sc + sh) as a threshold test on +-y
        sgn[e] = 0u;
        if (sc < 0.f) { t = -t; sgn[e] = 0x80000000u; }
        thr[e] = t;
        if (e & 1) { xa[e >> 1].y = rs; xb[e >> 1].y = -mu * rs; } else { xa[e >> 1].x = rs; xb[e >> 1].x = -mu * rs; }
    }
    f2 w1p[4][2];                                                     // FROM_A0: W1[channel pair][c]
    if (FROM_A0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 4; ++c) w1p[c][h] = f2{W1[(cq * 4 + 2 * h) * 4 + c], W1[(cq * 4 + 2 * h + 1) * 4 + c]};
    }
    f2 aG[4][2], aX[4][2], aS1[2], aS2[2], aSa[2];                    // G[c][pair], X[c][pair], s1, s2 [pair], Sa[c pair]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        aS1[h] = f2{0.f, 0.f}; aS2[h] = f2{0.f, 0.f}; aSa[h] = f2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) { aG[c][h] = f2{0.f, 0.f}; aX[c][h] = f2{0.f, 0.f}; }
    }
    constexpr int U = 4;
    const long nblk = npix / (16 * U);                               // blocks of 64 consecutive pixels
    const unsigned yoff = threadIdx.x * 8u, aoff = ps * 8u;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const bf16* dzb = dz1 + blk * (16 * U * 64);
        const TA* yb = y1 + blk * (16 * U * 64);
        const TA* ab = a0 + blk * (16 * U * 4);
        uint2 d[U], v[U], a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u] = *(const uint2*)((const char*)(dzb + u * 1024) + yoff);
            if (!FROM_A0) v[u] = *(const uint2*)((const char*)(yb + u * 1024) + yoff);
            a[u] = *(const uint2*)((const char*)(ab + u * 64) + aoff);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float av[4] = {H16<TA>::lo(a[u].x), H16<TA>::hi(a[u].x), H16<TA>::lo(a[u].y), H16<TA>::hi(a[u].y)};
            const unsigned yr[2] = {FROM_A0 ? 0u : v[u].x, FROM_A0 ? 0u : v[u].y}, dr[2] = {d[u].x, d[u].y};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 y = FROM_A0 ? w1p[0][h] * av[0] + w1p[1][h] * av[1] + w1p[2][h] * av[2] + w1p[3][h] * av[3]
                                     : f2{H16<TA>::lo(yr[h]), H16<TA>::hi(yr[h])};
                const f2 dd = f2{bf16_bits_to_f32(dr[h] & 0xffffu), __uint_as_float(dr[h] & 0xffff0000u)};
                const f2 g = f2{__uint_as_float(__float_as_uint(y.x) ^ sgn[2 * h]) > thr[2 * h] ? dd.x : 0.f,
                                __uint_as_float(__float_as_uint(y.y) ^ sgn[2 * h + 1]) > thr[2 * h + 1] ? dd.y : 0.f};
                const f2 xh = y * xa[h] + xb[h];
#pragma unroll
                for (int c = 0; c < 4; ++c) { aG[c][h] += g * av[c]; aX[c][h] += xh * av[c]; }
                aS1[h] += g;
                aS2[h] += g * xh;
            }
            aSa[0] += f2{av[0], av[1]}; aSa[1] += f2{av[2], av[3]};
            asm volatile("" : "+v"(aG[0][0]), "+v"(aG[0][1]), "+v"(aG[1][0]), "+v"(aG[1][1]), "+v"(aG[2][0]), "+v"(aG[2][1]),
                              "+v"(aG[3][0]), "+v"(aG[3][1]), "+v"(aX[0][0]), "+v"(aX[0][1]), "+v"(aX[1][0]), "+v"(aX[1][1]),
                              "+v"(aX[2][0]), "+v"(aX[2][1]), "+v"(aX[3][0]), "+v"(aX[3][1]), "+v"(aS1[0]), "+v"(aS1[1]),
                              "+v"(aS2[0]), "+v"(aS2[1]), "+v"(aSa[0]), "+v"(aSa[1]));
        }
    }
    float acc[44];                                       // [0,16) G[e][c], [16,32) X[e][c], [32,36) s1, [36,40) s2, [40,44) Sa
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[(2 * h) * 4 + c] = aG[c][h].x; acc[(2 * h + 1) * 4 + c] = aG[c][h].y;
            acc[16 + (2 * h) * 4 + c] = aX[c][h].x; acc[16 + (2 * h + 1) * 4 + c] = aX[c][h].y;
        }
        acc[32 + 2 * h] = aS1[h].x; acc[33 + 2 * h] = aS1[h].y;
        acc[36 + 2 * h] = aS2[h].x; acc[37 + 2 * h] = aS2[h].y;
        acc[40 + 2 * h] = aSa[h].x; acc[41 + 2 * h] = aSa[h].y;
    }
#pragma unroll
    for (int i = 0; i < 44; ++i) { acc[i] += __shfl_xor(acc[i], 16, 64); acc[i] += __shfl_xor(acc[i], 32, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 44; ++i) sred[wave][lane][i] = acc[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 644; o += 256) {       // [G 64x4 | X 64x4 | s1 64 | s2 64 | Sa 4]
        int grp, slot;
        if (o < 512) { const int which = o >> 8, co = (o & 255) >> 2, c = o & 3; grp = co >> 2; slot = which * 16 + (co & 3) * 4 + c; }
        else if (o < 640) { const int which = (o - 512) >> 6, co = (o - 512) & 63; grp = co >> 2; slot = 32 + which * 4 + (co & 3); }
        else { grp = 0; slot = 40 + (o - 640); }               // Sa: every channel quad saw every pixel once - take quad 0's copy
        atomicAdd(&red[o], (double)(sred[0][grp][slot] + sred[1][grp][slot] + sred[2][grp][slot] + sred[3][grp][slot]));
    }
}
// dW1[co][c] += gamma*rstd * (G - s1/N * Sa - s2/N * X)   (use_stats = 0, eval-mode BatchNorm: gamma*rstd * G);  dgamma += s2, dbeta += s1
__global__ void stem_c1_bwd_finalize_kernel(const double* __restrict__ red, long npix, const float* __restrict__ aff, int use_stats,
                                            float* __restrict__ dW1, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int o = threadIdx.x;                                 // co*4 + c
    const int co = o >> 2, c = o & 3;
    const double invN = 1.0 / (double)npix;
    const double m1 = use_stats ? red[512 + co] * invN : 0.0, m2 = use_stats ? red[576 + co] * invN : 0.0;
    dW1[o] += (float)((double)aff[co] * (red[o] - m1 * red[640 + c] - m2 * red[256 + o]));
    if (c == 0) { dbeta[co] += (float)red[512 + co]; dgamma[co] += (float)red[576 + co]; }
}

// ------------------------------------------------------------------------------------------------
// y4[b][t][f][c] = sum_ci W4[c][ci] * relu(y3[b][f][t][ci]*scale[ci] + shift[ci])
// 8 lanes per pixel (one 16-byte chunk each), shuffle-reduced.
template <typename T, int U>
__global__ void stem_c4_fwd_kernel(const T* __restrict__ y3, const float* __restrict__ W4, const float* __restrict__ scale,
                                   const float* __restrict__ shift, int nb, int F, int Tn, T* __restrict__ y4,
                                   double* __restrict__ stats = nullptr, T* __restrict__ y4lo = nullptr) {
    // y4lo (hybrid mode, sarssl_stem_c4_fwd_pair): the f32 result leaves as a pair - y4 = T(o), y4lo = T(o - y4); the statistics are
    // those of the pair's value
    const int cg = threadIdx.x & 7;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};      // stats: sum / sum of squares of the STORED y4 (BatchNorm(4) statistics)
    float w[4][8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = scale[cg * 8 + e]; sh[e] = shift[cg * 8 + e];
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c][e] = W4[c * 64 + cg * 8 + e];
    }
    // (measured and rejected in round 2, B = 64: a (bin, frame)-tiled variant writing the (B,T,F,4) output in 128-byte rows through
    //  LDS - 182 us, two barriers per tile cost more than the scattered 8-byte stores of this 33 MB tensor; four loads in flight per
    //  thread - 162 us; this one-chunk-per-iteration loop at 16 waves / CU: 146 us)
    //  Round 3: the output is (B,T,F,4) - bins contiguous - while y3 is (B,F,T,64): walking the pixels in y3's order made every 8-byte
    //  store land in a different 2 KB-strided line (4.2 M partial-line writes for a 33 MB tensor), and one 16-byte load in flight per
    //  thread at ~7 waves / SIMD is ~28 KB per CU - 189 us = 3.0 TB/s for the pass.  Now a workgroup takes an (8 bins x 16 frames) item:
    //  the 8 pixels of a wave are 8 consecutive BINS of one frame (one 64-byte store per wave), every thread has the U frames
    //  tl, tl+4, ... of its bin in flight at once (each bin row of the item is U * 512 contiguous bytes), reads stay whole 128-byte
    //  lines.  B = 64: 189 -> 147 us (U = 4) -> 121 us (U = 8, one resident round of workgroups) = 4.7 TB/s.
    const long npix = (long)nb * F * Tn;
    auto pixel = [&](const f8& v, int b, int f, int t) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float z = fmaxf(fmaf(v.v[e], sc[e], sh[e]), 0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] += w[c][e] * z;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            o[c] += __shfl_xor(o[c], 1, 64); o[c] += __shfl_xor(o[c], 2, 64); o[c] += __shfl_xor(o[c], 4, 64);
        }
        if (cg == 0) {
            st4(y4 + ((((long)b * Tn + t) * F + f) * 4), make_float4(o[0], o[1], o[2], o[3]));
            float lo[4] = {0.f, 0.f, 0.f, 0.f};
            if (y4lo) {
#pragma unroll
                for (int c = 0; c < 4; ++c) lo[c] = o[c] - round_as<T>(o[c]);
                st4(y4lo + ((((long)b * Tn + t) * F + f) * 4), make_float4(lo[0], lo[1], lo[2], lo[3]));
            }
            if (stats) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float r = y4lo ? round_as<T>(o[c]) + round_as<T>(lo[c]) : round_as<T>(o[c]);
                    ssum[c] += r; ssq[c] = fmaf(r, r, ssq[c]);
                }
            }
        }
    };
    if constexpr (U > 0) {                              // host: F % 8 == 0 and Tn % (4 U) == 0
        const int fblocks = F >> 3, tblocks = Tn / (4 * U);
        const int nitem = nb * fblocks * tblocks;
        const int fi = (threadIdx.x >> 3) & 7, tl = threadIdx.x >> 6;
        for (int item = blockIdx.x; item < nitem; item += gridDim.x) {
            const int tb = item % tblocks, r = item / tblocks;
            const int f = (r % fblocks) * 8 + fi, b = r / fblocks, t0 = tb * (4 * U) + tl;
            const T* src = y3 + (((long)b * F + f) * Tn + t0) * 64 + cg * 8;
            Raw8<T> v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = raw8_load(src + (long)u * 4 * 64);
#pragma unroll
            for (int u = 0; u < U; ++u) pixel(raw8_unpack(v[u]), b, f, t0 + 4 * u);
        }
    } else {
        const long nthreads = (long)gridDim.x * blockDim.x;
        for (long g0 = (long)blockIdx.x * blockDim.x + threadIdx.x; g0 < npix * 8; g0 += nthreads) {
            const long p = g0 >> 3;
            const int t = (int)(p % Tn);
            const long bf = p / Tn;
            pixel(ld8(y3 + p * 64 + cg * 8), (int)(bf / F), (int)(bf % F), t);
        }
    }
    if (stats) {                    // lanes 0, 8, 16, ... hold the pixels: fold over lane bits 3..5, then the 4 waves through LDS, 8 atomics per workgroup
        __shared__ float sfold[4][8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ssum[c] += __shfl_xor(ssum[c], 8, 64); ssum[c] += __shfl_xor(ssum[c], 16, 64); ssum[c] += __shfl_xor(ssum[c], 32, 64);
            ssq[c] += __shfl_xor(ssq[c], 8, 64); ssq[c] += __shfl_xor(ssq[c], 16, 64); ssq[c] += __shfl_xor(ssq[c], 32, 64);
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { sfold[wave][c] = ssum[c]; sfold[wave][4 + c] = ssq[c]; }
        }
        __syncthreads();
        if (threadIdx.x < 8) atomicAdd(&stats[threadIdx.x], (double)(sfold[0][threadIdx.x] + sfold[1][threadIdx.x] + sfold[2][threadIdx.x] + sfold[3][threadIdx.x]));
    }
}

// Backward of the 64->4 conv + the ReLU mask + BatchNorm(3)-backward reductions, one pass over y3:
//   g3[p][ci]  = (sum_c dy4[p][c] W4[c][ci]) * [bn3(y3) > 0]           (written, same layout as y3)
//   dW4[c][ci] = sum_p dy4[p][c] * relu(bn3(y3))[p][ci]
//   s1[ci] = sum_p g3 ; s2[ci] = sum_p g3 * xhat3                         (BatchNorm backward sums)
// dy4 is (B,T,F,4); y3/g3 are (B,F,T,64).  red: double [4*64 + 64 + 64], zeroed by the caller.
// MODE 0: g3 + sums in one pass (g3 is then normalised in place by cl_bn_bwd_apply: 3 more passes over 64-channel tensors).
// MODE 1: sums only (no store).  MODE 2: second phase - recomputes the masked gradient and writes the finished BatchNorm input
// gradient dy3 = gamma*rstd*(g - s1/N - xhat*s2/N) directly (use_stats = 0: eval-mode BatchNorm, dy3 = gamma*rstd*g).
// MODE 1 + MODE 2 move 1.7 GB per encoder instead of 2.7 GB (the 64->4 contraction is recomputed, 4 FMAs per element).
template <typename T, typename TA, int MODE>
__global__ __launch_bounds__(256) void stem_c4_bwd_kernel(const TA* __restrict__ y3, const T* __restrict__ dy4, const float* __restrict__ W4,
                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                   int nb, int F, int Tn, T* __restrict__ g3, double* __restrict__ red, int use_stats,
                                   float* __restrict__ gW4 = nullptr, float* __restrict__ dgamma = nullptr, float* __restrict__ dbeta = nullptr) {
    if (MODE == 2 && gW4 && blockIdx.x == 0) {   // parameter gradients from the finished sums (were two launches of their own)
        for (int o = threadIdx.x; o < 256; o += blockDim.x) gW4[o] += (float)red[o];
        for (int c = threadIdx.x; c < 64; c += blockDim.x) { dbeta[c] += (float)red[256 + c]; dgamma[c] += (float)red[320 + c]; }
    }
    __shared__ float sred[4][8][48];
    const int cg = threadIdx.x & 7;
    float w[4][8], sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = cg * 8 + e;
        sc[e] = scale[ci]; sh[e] = shift[ci]; mu[e] = mean[ci]; rs[e] = rstd[ci];
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c][e] = W4[c * 64 + ci];
    }
    float acc[48];                               // [0,32) dW4[c][e], [32,40) s1[e], [40,48) s2[e]
#pragma unroll
    for (int i = 0; i < 48; ++i) acc[i] = 0.f;
    const long npix = (long)nb * F * Tn;
    float cA[8], cB[8], cC[8];                   // MODE 2: dy3 = cA*g + cB*y + cC
    if (MODE == 2) {
        const float invN = 1.0f / (float)npix;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = cg * 8 + e;
            const float m1 = use_stats ? (float)red[256 + ci] * invN : 0.f, m2 = use_stats ? (float)red[320 + ci] * invN : 0.f;
            cA[e] = sc[e]; cB[e] = -sc[e] * m2 * rs[e]; cC[e] = -sc[e] * m1 - cB[e] * mu[e];
        }
    }
    // Pixel order.  y3 / g3 are (B,F,T,64) - frames contiguous - while dy4 is (B,T,F,4) - bins contiguous: walking pixels in y3's
    // order reads dy4 with a 2 KB stride (one useful 8-byte piece per 128-byte line: the L2 -> L1 traffic of the 33 MB dy4 tensor
    // then equals that of the 537 MB y3 tensor; round-2 counters: 2.3 TB/s for this pass).  Workgroups therefore take 16 x 16
    // (bin, frame) tiles: the dy4 tile is fetched in 128-byte rows into LDS, the y3 tile in 2 KB rows straight into registers.
    __shared__ float4 sD[16][17];
    const int ftiles = (F + 15) >> 4, ttiles = (Tn + 15) >> 4;
    const long ntile = (long)nb * ftiles * ttiles;
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int tt0 = (int)(tile % ttiles) << 4;
        const long r = tile / ttiles;
        const int f0 = (int)(r % ftiles) << 4, b = (int)(r / ftiles);
        __syncthreads();                                             // previous tile's readers are done with sD
        if (threadIdx.x < 128) {                                     // 16 frames x 8 pieces of 16 bytes (2 bins x 4 channels)
            const int tt = threadIdx.x >> 3, fc = (threadIdx.x & 7) * 2;
            float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
            if (tt0 + tt < Tn) {
                const T* q = dy4 + (((long)b * Tn + tt0 + tt) * F + f0 + fc) * 4;
                if (f0 + fc + 1 < F) { const f8 v = ld8(q); lo = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]); hi = make_float4(v.v[4], v.v[5], v.v[6], v.v[7]); }
                else if (f0 + fc < F) lo = ld4(q);
            }
            sD[tt][fc] = lo; sD[tt][fc + 1] = hi;
        }
        __syncthreads();
        // all 8 pixel chunks of the thread are requested before the first is used and stay packed (4 VGPRs each for bf16) until
        // then: the accumulators keep this kernel at 2 waves / SIMD, so bytes in flight per wave is what sets its bandwidth
        constexpr int U = 8;
        Raw8<TA> v[U]; bool ok[U]; long pix[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pl = (threadIdx.x >> 3) + 32 * u;                 // 0..255: bin pl >> 4, frame pl & 15
            const int f = f0 + (pl >> 4), t = tt0 + (pl & 15);
            ok[u] = f < F && t < Tn;
            pix[u] = ((long)b * F + min(f, F - 1)) * Tn + min(t, Tn - 1);      // clamped: the load is unconditional (see stem_c4_fwd)
            v[u] = raw8_load(y3 + pix[u] * 64 + cg * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const int pl = (threadIdx.x >> 3) + 32 * u;
            const float4 d = sD[pl & 15][pl >> 4];
            const f8 vv = raw8_unpack(v[u]);
            f8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float uu = fmaf(vv.v[e], sc[e], sh[e]);
                float gi = d.x * w[0][e] + d.y * w[1][e] + d.z * w[2][e] + d.w * w[3][e];
                gi = (uu > 0.f) ? gi : 0.f;
                if (MODE == 2) o.v[e] = fmaf(cA[e], gi, fmaf(cB[e], vv.v[e], cC[e]));
                else {
                    const float z = fmaxf(uu, 0.f);
                    o.v[e] = gi;
                    acc[0 * 8 + e] += d.x * z; acc[1 * 8 + e] += d.y * z; acc[2 * 8 + e] += d.z * z; acc[3 * 8 + e] += d.w * z;
                    acc[32 + e] += gi;
                    acc[40 + e] += gi * (vv.v[e] - mu[e]) * rs[e];
                }
            }
            if (MODE != 1) st8(g3 + pix[u] * 64 + cg * 8, o);
        }
    }
    if (MODE == 2) return;
    // lanes sharing a channel group are 8 apart: butterfly over lane bits 3..5, then across the 4 waves through LDS
#pragma unroll
    for (int i = 0; i < 48; ++i) {
        acc[i] += __shfl_xor(acc[i], 8, 64); acc[i] += __shfl_xor(acc[i], 16, 64); acc[i] += __shfl_xor(acc[i], 32, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 8) {
#pragma unroll
        for (int i = 0; i < 48; ++i) sred[wave][lane][i] = acc[i];
    }
    __syncthreads();
    // 384 outputs: [0,256) dW4[c][ci], [256,320) s1[ci], [320,384) s2[ci]
    for (int o = threadIdx.x; o < 384; o += 256) {
        int ci, slot;
        if (o < 256) { const int c = o >> 6; ci = o & 63; slot = c * 8 + (ci & 7); }
        else if (o < 320) { ci = o - 256; slot = 32 + (ci & 7); }
        else { ci = o - 320; slot = 40 + (ci & 7); }
        const int ocg = ci >> 3;
        const float sum = sred[0][ocg][slot] + sred[1][ocg][slot] + sred[2][ocg][slot] + sred[3][ocg][slot];
        atomicAdd(&red[o], (double)sum);
    }
}

// Phase 1 of the two-phase 64->4 backward (stem_c4_bwd_kernel<T, 1>) for 16-bit activations and full 16 x 16 tiles, rebuilt around
// occupancy and bytes in flight.  The general kernel keeps 48 accumulators + 32 weights per thread (8 channels each): 228 VGPRs,
// 2 waves / SIMD, and its load - wait - reduce loop left HBM idle during the ~800 VALU instructions of a tile: 238 us for 570 MB
// (2.4 TB/s) at B = 64 while its VALU work alone is ~100 us.  Here a thread owns 4 channels (24 accumulators, 16 weights):
// <= 128 VGPRs = 4 waves / SIMD, and the 8-byte chunks of the next half tile (8 bins x 16 frames; plus the thread's piece of the
// next dy4 tile) are requested before the current half is reduced.  Addresses are a scalar base (tile, bin) plus a per-thread
// byte offset that never changes (8 * threadIdx.x: one bin row of 16 frames is 2 KB contiguous).
// sum g*xhat is accumulated as the raw moment sum g*y and converted per thread at the end (frees the mean / rstd registers).
template <typename TA>                   // TA: 16-bit type of the saved forward tensor y3; the gradient dy4 is bf16
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void stem_c4_bwd_sums16_kernel(const TA* __restrict__ y3, const bf16* __restrict__ dy4, const float* __restrict__ W4,
                               const float* __restrict__ scale, const float* __restrict__ shift,
                               const float* __restrict__ mean, const float* __restrict__ rstd,
                               int nb, int F, int Tn, double* __restrict__ red, int nstream) {
    __shared__ float4 sD[2][16][17];                     // dy4 tile [frame][bin] (f32), two stages: one barrier per tile
    __shared__ float sred[4][16][24];
    const int cq = threadIdx.x & 15, ps = threadIdx.x >> 4;          // channels 4cq..4cq+3; frame ps of every bin row
    // everything per channel pair (packed f32 math: v_pk_fma_f32 on register pairs, dy4 values broadcast through op_sel)
    typedef sarssl_f32x2 f2;
    f2 w[4][2], sc[2], sh[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ci = cq * 4 + 2 * h;
        sc[h] = f2{scale[ci], scale[ci + 1]}; sh[h] = f2{shift[ci], shift[ci + 1]};
#pragma unroll
        for (int c = 0; c < 4; ++c) w[c][h] = f2{W4[c * 64 + ci], W4[c * 64 + ci + 1]};
    }
    f2 aW[4][2], aS1[2], aS2[2];                         // dW4[c][pair], s1[pair], sum g*y [pair]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        aS1[h] = f2{0.f, 0.f}; aS2[h] = f2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) aW[c][h] = f2{0.f, 0.f};
    }
    const int ftiles = F >> 4, ttiles = Tn >> 4;
    const int ntile = nb * ftiles * ttiles;
    const long ystride = (long)Tn * 64;                  // one bin
    const int ttl = threadIdx.x >> 3 & 15, fcl = (threadIdx.x & 7) * 2;     // dy4 piece: frame ttl, bins fcl, fcl+1 (threads 0..127 stage it)
    const unsigned yoff = threadIdx.x * 8u, doff = (unsigned)((ttl * F + fcl) * 4) * 2u;
    constexpr int H = 8;
    const int tper = (nstream > 1 && ntile % nstream == 0) ? ntile / nstream : 0;      // tiles dealt round-robin to nstream address ranges
    auto tile_base = [&](int tile, long& ybase, long& dbase) {
        if (tper) tile = (tile % nstream) * tper + tile / nstream;
        // (hipcc divides on the vector unit and would keep everything derived from the quotients in vector registers)
        const int tt0 = __builtin_amdgcn_readfirstlane((tile % ttiles) << 4);
        const int r = tile / ttiles;
        const int f0 = __builtin_amdgcn_readfirstlane((r % ftiles) << 4), b = __builtin_amdgcn_readfirstlane(r / ftiles);
        ybase = (((long)b * F + f0) * Tn + tt0) * 64;
        dbase = (((long)b * Tn + tt0) * F + f0) * 4;
    };
    auto issue_y = [&](int tile, int half, uint2 (&v)[H]) {             // past the last tile: clamped to it, never used
        long ybase, dbase; tile_base(tile < ntile ? tile : ntile - 1, ybase, dbase);
        unsigned yo = yoff;
        asm volatile("" : "+v"(yo));                     // opaque: nothing per-thread and 64-bit for hipcc to hoist out of the loop (and spill)
#pragma unroll
        for (int u = 0; u < H; ++u) v[u] = *(const uint2*)((const char*)(y3 + ybase + (half * H + u) * ystride) + yo);
    };
    auto issue_d = [&](int tile, uint4& dp) {
        long ybase, dbase; tile_base(tile < ntile ? tile : ntile - 1, ybase, dbase);
        unsigned dof = doff;
        asm volatile("" : "+v"(dof));
        dp = *(const uint4*)((const char*)(dy4 + dbase) + dof);
    };
    auto reduce = [&](int half, int buf, const uint2 (&v)[H]) {
#pragma unroll
        for (int u = 0; u < H; ++u) {
            const float4 d = sD[buf][ps][half * H + u];
            const f2 y[2] = {f2{H16<TA>::lo(v[u].x), H16<TA>::hi(v[u].x)}, f2{H16<TA>::lo(v[u].y), H16<TA>::hi(v[u].y)}};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 uu = y[h] * sc[h] + sh[h];
                f2 gi = w[0][h] * d.x + w[1][h] * d.y + w[2][h] * d.z + w[3][h] * d.w;
                gi = f2{uu.x > 0.f ? gi.x : 0.f, uu.y > 0.f ? gi.y : 0.f};
                const f2 z = f2{fmaxf(uu.x, 0.f), fmaxf(uu.y, 0.f)};
                aW[0][h] += z * d.x; aW[1][h] += z * d.y; aW[2][h] += z * d.z; aW[3][h] += z * d.w;
                aS1[h] += gi;
                aS2[h] += gi * y[h];
            }
            // pin the chunk order: hipcc otherwise computes the masked gradients of all 8 chunks first and keeps them (and the 8
            // dy4 rows, and the unpacked y) live until the accumulation - ~100 registers, spilled and reloaded through vmcnt
            asm volatile("" : "+v"(aW[0][0]), "+v"(aW[0][1]), "+v"(aW[1][0]), "+v"(aW[1][1]), "+v"(aW[2][0]), "+v"(aW[2][1]),
                              "+v"(aW[3][0]), "+v"(aW[3][1]), "+v"(aS1[0]), "+v"(aS1[1]), "+v"(aS2[0]), "+v"(aS2[1]));
        }
    };
    uint2 vA[H], vB[H];
    uint4 dp;
    int tile = blockIdx.x;
    if (tile < ntile) {
        issue_d(tile, dp);
        issue_y(tile, 0, vA);
        for (int buf = 0;; buf ^= 1) {
            if (threadIdx.x < 128) {
                sD[buf][ttl][fcl] = make_float4(bf16_bits_to_f32(dp.x & 0xffffu), __uint_as_float(dp.x & 0xffff0000u),
                                                bf16_bits_to_f32(dp.y & 0xffffu), __uint_as_float(dp.y & 0xffff0000u));
                sD[buf][ttl][fcl + 1] = make_float4(bf16_bits_to_f32(dp.z & 0xffffu), __uint_as_float(dp.z & 0xffff0000u),
                                                    bf16_bits_to_f32(dp.w & 0xffffu), __uint_as_float(dp.w & 0xffff0000u));
            }
            __syncthreads();
            issue_y(tile, 1, vB);
            __builtin_amdgcn_sched_barrier(0);
            reduce(0, buf, vA);
            issue_d(tile + gridDim.x, dp);
            issue_y(tile + gridDim.x, 0, vA);
            __builtin_amdgcn_sched_barrier(0);
            reduce(1, buf, vB);
            tile += gridDim.x;
            if (tile >= ntile) break;
        }
    }
    float acc[24];                                       // [0,16) dW4[c][e], [16,20) s1[e], [20,24) s2[e]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc[c * 4 + 2 * h] = aW[c][h].x; acc[c * 4 + 2 * h + 1] = aW[c][h].y; }
        acc[16 + 2 * h] = aS1[h].x; acc[17 + 2 * h] = aS1[h].y;
        acc[20 + 2 * h] = aS2[h].x; acc[21 + 2 * h] = aS2[h].y;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)        // sum g*xhat = rstd * (sum g*y - mean * sum g) over the thread's own pixels
        acc[20 + e] = (acc[20 + e] - mean[cq * 4 + e] * acc[16 + e]) * rstd[cq * 4 + e];
    // lanes sharing a channel quad are 16 apart: butterfly over lane bits 4, 5, then across the 4 waves through LDS
#pragma unroll
    for (int i = 0; i < 24; ++i) { acc[i] += __shfl_xor(acc[i], 16, 64); acc[i] += __shfl_xor(acc[i], 32, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 24; ++i) sred[wave][lane][i] = acc[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 384; o += 256) {       // [0,256) dW4[c][ci], [256,320) s1[ci], [320,384) s2[ci]
        int ci, slot;
        if (o < 256) { ci = o & 63; slot = (o >> 6) * 4 + (ci & 3); }
        else if (o < 320) { ci = o - 256; slot = 16 + (ci & 3); }
        else { ci = o - 320; slot = 20 + (ci & 3); }
        const int q = ci >> 2;
        atomicAdd(&red[o], (double)(sred[0][q][slot] + sred[1][q][slot] + sred[2][q][slot] + sred[3][q][slot]));
    }
}

// ------------------------------------------------------------------------------------------------
// Generic channels-last helpers on X[N][C], C = 4 * 2^k <= 1024.  Thread = (row slot, 4-channel group).

// Tail of the per-channel reductions: 32 row slots -> one value per (quantity, virtual column); when C < 64 the 64/C
// virtual columns of a channel are folded in LDS first so a workgroup issues 2C (not 128) same-address f64 atomics.
__device__ __forceinline__ void cl_fold_atomic(float (&sred)[256][17], int col0, int L, int C, double* __restrict__ out) {
    __shared__ float sfold[2][64];
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        float acc = 0.f;
        if (col0 + c < L)
            for (int r = 0; r < 32; ++r) acc += sred[r * 8 + (c >> 3)][which * 8 + (c & 7)];
        if (C >= 64) { if (col0 + c < L) atomicAdd(&out[which * C + col0 + c], (double)acc); }
        else sfold[which][c] = acc;
    }
    if (C < 64) {
        __syncthreads();
        if (threadIdx.x < 2 * C) {
            const int which = threadIdx.x / C, ch = threadIdx.x % C;
            float a = 0.f;
            for (int c = ch; c < 64; c += C) a += sfold[which][c];
            atomicAdd(&out[which * C + ch], (double)a);
        }
    }
}

// Column tiles of 64 channels (8 threads x 8 ch, 16-byte loads) x 32 row slots.  For C < 64 (64 % C == 0) the tensor is
// viewed as [N*C/64][64] and the 64 virtual columns fold back onto channel (col % C).
// sums[c] += sum_n x ; sums[C + c] += sum_n x^2
template <typename T>
__global__ void cl_stats_kernel(const T* __restrict__ x, long rows, int L, int C, double* __restrict__ sums) {
    __shared__ float sred[256][17];
    const int cgp = threadIdx.x & 7, rslot = threadIdx.x >> 3;
    const int col0 = blockIdx.x * 64, col = col0 + cgp * 8;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (col < L) {
        for (long n = (long)blockIdx.y * 32 + rslot; n < rows; n += (long)gridDim.y * 32) {
            const f8 v = ld8(x + n * L + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += v.v[e]; q[e] += v.v[e] * v.v[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sred[threadIdx.x][e] = s[e]; sred[threadIdx.x][8 + e] = q[e]; }
    __syncthreads();
    cl_fold_atomic(sred, col0, L, C, sums);
}

// training-mode BatchNorm statistics -> affine; updates running stats (nn.BatchNorm defaults: momentum 0.1, unbiased)
__global__ void bn_finalize_kernel(const double* __restrict__ sums, long N, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, long* __restrict__ nbt,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean,
                                   float* __restrict__ rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    const double m = sums[c] / (double)N;
    double var = sums[C + c] / (double)N - m * m;
    if (var < 0.0) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m; rstd[c] = r;
    scale[c] = gamma[c] * r;
    shift[c] = beta[c] - (float)m * gamma[c] * r;
    if (running_mean && isfinite(m) && isfinite(var)) {      // (an overflowed fp16 forward must not poison the running statistics)
        const double unb = (N > 1) ? var * (double)N / (double)(N - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

__global__ void bn_eval_affine_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean,
                                      float* __restrict__ rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float r = rsqrtf(running_var[c] + eps);
    mean[c] = running_mean[c]; rstd[c] = r;
    scale[c] = gamma[c] * r;
    shift[c] = beta[c] - running_mean[c] * gamma[c] * r;
}

__device__ __forceinline__ float act_fwd(float u, int act) {
    return act == 1 ? fmaxf(u, 0.f) : (act == 2 ? u * sigmoidf_(u) : u);
}
__device__ __forceinline__ float act_bwd(float u, int act) {   // d act / d u
    if (act == 1) return u > 0.f ? 1.f : 0.f;
    if (act == 2) { const float s = sigmoidf_(u); return s * (1.f + u * (1.f - s)); }
    return 1.f;
}

// z = act(x*scale[c] + shift[c]);  x viewed as [rows][L], thread = fixed 8-channel group
template <typename T>
__global__ void cl_affine_act_kernel(const T* __restrict__ x, long rows, int L, int C, const float* __restrict__ scale,
                                     const float* __restrict__ shift, int act, T* __restrict__ z) {
    const int gpr = L >> 3, cg = threadIdx.x % gpr, rslot = threadIdx.x / gpr, rpb = 256 / gpr;
    const int col = cg * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int ch = (col + e) % C; sc[e] = scale[ch]; sh[e] = shift[ch]; }
    // four rows in flight per thread (the launch gives every thread four rows: one 16-byte request per thread at a time made these passes
    // request-latency bound at ~1 TB/s)
    const long stride = (long)gridDim.x * rpb;
    long n = (long)blockIdx.x * rpb + rslot;
    for (; n + 3 * stride < rows; n += 4 * stride) {
        f8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld8(x + (n + u * stride) * L + col);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[u].v[e] = act_fwd(fmaf(v[u].v[e], sc[e], sh[e]), act);
            st8(z + (n + u * stride) * L + col, v[u]);
        }
    }
    for (; n < rows; n += stride) {
        f8 v = ld8(x + n * L + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] = act_fwd(fmaf(v.v[e], sc[e], sh[e]), act);
        st8(z + n * L + col, v);
    }
}

// The same with training-mode BatchNorm's affine formed from the finished sums by every thread for its own 8 channels (bn_finalize_kernel's
// arithmetic: f64 mean / variance, one rounding to f32 - bit-identical) instead of by a 5-us launch of its own in front of this one;
// workgroup 0 also writes the affine / statistics rows (the backward pass reads them) and moves the running statistics.  C >= 64 (L == C).
template <typename T>
__global__ void cl_bn_train_act_kernel(const T* __restrict__ x, long rows, int C, const double* __restrict__ sums,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                       float* __restrict__ running_mean, float* __restrict__ running_var, long* __restrict__ nbt,
                                       float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean,
                                       float* __restrict__ rstd, int act, T* __restrict__ z) {
    const int gpr = C >> 3, cg = threadIdx.x % gpr, rslot = threadIdx.x / gpr, rpb = 256 / gpr;
    const int col = cg * 8;
    const bool writer = blockIdx.x == 0 && rslot == 0;
    if (writer && cg == 0 && nbt) *nbt += 1;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = col + e;
        const double m = sums[ch] / (double)rows;
        double var = sums[C + ch] / (double)rows - m * m;
        if (var < 0.0) var = 0.0;
        const float r = (float)(1.0 / sqrt(var + (double)eps));
        sc[e] = gamma[ch] * r;
        sh[e] = beta[ch] - (float)m * gamma[ch] * r;
        if (writer) {
            mean[ch] = (float)m; rstd[ch] = r; scale[ch] = sc[e]; shift[ch] = sh[e];
            if (running_mean && isfinite(m) && isfinite(var)) {
                const double unb = (rows > 1) ? var * (double)rows / (double)(rows - 1) : var;
                running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
                running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unb;
            }
        }
    }
    const long stride = (long)gridDim.x * rpb;          // four rows in flight per thread, as in cl_affine_act_kernel
    long n = (long)blockIdx.x * rpb + rslot;
    for (; n + 3 * stride < rows; n += 4 * stride) {
        f8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld8(x + (n + u * stride) * C + col);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[u].v[e] = act_fwd(fmaf(v[u].v[e], sc[e], sh[e]), act);
            st8(z + (n + u * stride) * C + col, v[u]);
        }
    }
    for (; n < rows; n += stride) {
        f8 v = ld8(x + n * C + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] = act_fwd(fmaf(v.v[e], sc[e], sh[e]), act);
        st8(z + n * C + col, v);
    }
}

// BatchNorm(+act) backward, pass 1: g = dz * act'(u), u = y*scale+shift; red[c] += sum g, red[C+c] += sum g*xhat
template <typename T, typename TA>
__global__ void cl_bn_bwd_reduce_kernel(const T* __restrict__ dz, const TA* __restrict__ y, long rows, int L, int C,
                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                        const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                        double* __restrict__ red) {
    __shared__ float sred[256][17];
    const int cgp = threadIdx.x & 7, rslot = threadIdx.x >> 3;
    const int col0 = blockIdx.x * 64, col = col0 + cgp * 8;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (col < L) {
        float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int ch = (col + e) % C; sc[e] = scale[ch]; sh[e] = shift[ch]; mu[e] = mean[ch]; rs[e] = rstd[ch]; }
        if (act == 1) {
            // ReLU: threshold test on y, and sum g*xhat = rs * (sum g*y - mu * sum g) finalised per thread (4 ops / element)
            float thr[8];
            unsigned sgn = 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = sc[e] != 0.f ? -sh[e] / sc[e] : (sh[e] > 0.f ? -INFINITY : INFINITY);
                if (sc[e] < 0.f) { t = -t; sgn |= 1u << e; }
                thr[e] = t;
            }
            for (long n = (long)blockIdx.y * 32 + rslot; n < rows; n += (long)gridDim.y * 32) {
                const f8 d = ld8(dz + n * L + col);
                const f8 v = ld8(y + n * L + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ys = __uint_as_float(__float_as_uint(v.v[e]) ^ (((sgn >> e) & 1u) << 31));
                    const float g = ys > thr[e] ? d.v[e] : 0.f;
                    s[e] += g;
                    q[e] = fmaf(g, v.v[e] - mu[e], q[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] *= rs[e];
        } else
        for (long n = (long)blockIdx.y * 32 + rslot; n < rows; n += (long)gridDim.y * 32) {
            const f8 d = ld8(dz + n * L + col);
            const f8 v = ld8(y + n * L + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = d.v[e] * act_bwd(fmaf(v.v[e], sc[e], sh[e]), act);
                s[e] += g;
                q[e] += g * (v.v[e] - mu[e]) * rs[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sred[threadIdx.x][e] = s[e]; sred[threadIdx.x][8 + e] = q[e]; }
    __syncthreads();
    cl_fold_atomic(sred, col0, L, C, red);
}

// pass 2: dy = gamma*rstd * (g - s1/N - xhat*s2/N)      (train)   or   dy = gamma*rstd * g   (eval: use_stats = 0)
// g_is_masked = 1 when dz already contains g (act' applied upstream).  Tensors viewed as [rows][L]; thread = fixed
// 8-channel group so all per-channel constants live in registers.
template <typename T, typename TA>
__global__ void cl_bn_bwd_apply_kernel(const T* __restrict__ dz, const TA* __restrict__ y, long rows, int L, int C, long N,
                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                       const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                       int g_is_masked, int use_stats, const double* __restrict__ red, T* __restrict__ dy,
                                       float* __restrict__ dgamma = nullptr, float* __restrict__ dbeta = nullptr) {
    if (dgamma && blockIdx.x == 0) {        // BatchNorm parameter gradients from the same sums: dbeta += s1, dgamma += s2 (was a launch of its own)
        for (int c = threadIdx.x; c < C; c += blockDim.x) { dbeta[c] += (float)red[c]; dgamma[c] += (float)red[C + c]; }
    }
    const int gpr = L >> 3, cg = threadIdx.x % gpr, rslot = threadIdx.x / gpr, rpb = 256 / gpr;
    const int col = cg * 8;
    const float invN = 1.0f / (float)N;
    float sc[8], sh[8], mu[8], rs[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = (col + e) % C;
        sc[e] = scale[ch]; sh[e] = shift[ch]; mu[e] = mean[ch]; rs[e] = rstd[ch];
        m1[e] = use_stats ? (float)red[ch] * invN : 0.f;
        m2[e] = use_stats ? (float)red[C + ch] * invN : 0.f;
    }
    if (act == 1 || g_is_masked) {
        // ReLU (or pre-masked) fast path, 6 VALU ops per element instead of 11 (the pass is otherwise VALU- as much as HBM-bound):
        //   relu'(y*sc + sh) is a threshold test on y;  sc*(g - m1 - (y-mu)*rs*m2) = A*g + (B*y + C) with per-channel A, B, C
        float A[8], Bc[8], Cc[8], thr[8];
        unsigned sgn = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            A[e] = sc[e];
            Bc[e] = -sc[e] * m2[e] * rs[e];
            Cc[e] = -sc[e] * m1[e] - Bc[e] * mu[e];
            float t = sc[e] != 0.f ? -sh[e] / sc[e] : (sh[e] > 0.f ? -INFINITY : INFINITY);
            if (g_is_masked) t = -INFINITY;
            else if (sc[e] < 0.f) { t = -t; sgn |= 1u << e; }
            thr[e] = t;
        }
        const long stride = (long)gridDim.x * rpb;
        long n = (long)blockIdx.x * rpb + rslot;
        for (; n + stride < rows; n += 2 * stride) {          // two rows in flight per thread (four 16-byte loads before the first use)
            const f8 d0 = ld8(dz + n * L + col), v0 = ld8(y + n * L + col);
            const f8 d1 = ld8(dz + (n + stride) * L + col), v1 = ld8(y + (n + stride) * L + col);
            f8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned sb = ((sgn >> e) & 1u) << 31;
                const float g0 = __uint_as_float(__float_as_uint(v0.v[e]) ^ sb) > thr[e] ? d0.v[e] : 0.f;
                const float g1 = __uint_as_float(__float_as_uint(v1.v[e]) ^ sb) > thr[e] ? d1.v[e] : 0.f;
                o0.v[e] = fmaf(A[e], g0, fmaf(Bc[e], v0.v[e], Cc[e]));
                o1.v[e] = fmaf(A[e], g1, fmaf(Bc[e], v1.v[e], Cc[e]));
            }
            st8(dy + n * L + col, o0);
            st8(dy + (n + stride) * L + col, o1);
        }
        for (; n < rows; n += stride) {
            const f8 d = ld8(dz + n * L + col);
            const f8 v = ld8(y + n * L + col);
            f8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ys = __uint_as_float(__float_as_uint(v.v[e]) ^ (((sgn >> e) & 1u) << 31));
                const float g = ys > thr[e] ? d.v[e] : 0.f;
                o.v[e] = fmaf(A[e], g, fmaf(Bc[e], v.v[e], Cc[e]));
            }
            st8(dy + n * L + col, o);
        }
        return;
    }
    for (long n = (long)blockIdx.x * rpb + rslot; n < rows; n += (long)gridDim.x * rpb) {
        const f8 d = ld8(dz + n * L + col);
        const f8 v = ld8(y + n * L + col);
        f8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = d.v[e] * act_bwd(fmaf(v.v[e], sc[e], sh[e]), act);
            const float xh = (v.v[e] - mu[e]) * rs[e];
            o.v[e] = sc[e] * (g - m1[e] - xh * m2[e]);          // sc = gamma * rstd
        }
        st8(dy + n * L + col, o);
    }
}

// ================================================================================================ C ABI
// grid caps of the stem passes: tuning knobs (A/B runs), e.g. SARSSL_GRID_C1F=2048
#ifdef SARSSL_PROBE_ENV      // tools/bench_stem.py probe build: caps / read-stream counts from the environment
static inline int grid_cap(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline int grid_cap(const char*, int dflt) { return dflt; }      // (workgroup caps measured in round 2: fixed; the name documents which pass)
#endif
static inline int nblocks_for(long work, int per_block, int cap = 2048) {
    long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}
#define ST ((hipStream_t)stream)
#define DISPATCH_T(dtype, CALL)                                                             \
    if (dtype == SARSSL_BF16) { typedef bf16 T; CALL; }                                     \
    else if (dtype == SARSSL_F32) { typedef float T; CALL; }                                \
    else if (dtype == SARSSL_F16) { typedef f16 T; CALL; }                                  \
    else { sarssl_set_error("unsupported dtype %d", dtype); return -1; }
// backward passes: gradients of type T next to tensors saved by the forward pass of type TA (SARSSL_MIX16: bf16 / fp16, common.h)
#define DISPATCH_GA(dtype, CALL)                                                            \
    if (dtype == SARSSL_BF16) { typedef bf16 T; typedef bf16 TA; CALL; }                    \
    else if (dtype == SARSSL_F32) { typedef float T; typedef float TA; CALL; }              \
    else if (dtype == SARSSL_MIX16) { typedef bf16 T; typedef f16 TA; CALL; }               \
    else { sarssl_set_error("unsupported dtype %d", dtype); return -1; }

extern "C" int sarssl_mask_inputs(const float* x, const unsigned char* mp, const int* mch, int nb, int F, int Tn, int mode,
                                  void* spec, void* spat, int dtype, void* stream) {
    SARSSL_REQUIRE(nb > 0 && F > 0 && Tn > 0, "sarssl_mask_inputs");
    const int nblk = nblocks_for((long)nb * F * Tn, 256, 8192);
    DISPATCH_T(dtype, (mask_inputs_kernel<T><<<nblk, 256, 0, ST>>>(x, mp, mch, nb, F, Tn, mode, (T*)spec, (T*)spat, sarssl_overflow_flag())));
    SARSSL_CHECK_LAUNCH("mask_inputs_kernel");
    return 0;
}

extern "C" int sarssl_stem_c1_fwd(const void* a0, const float* W1, long npix, void* y1, double* stats, int dtype, void* stream) {
    if (stats && SARSSL_ZERO(stats, 128 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    static const int cap_stats = grid_cap("SARSSL_GRID_C1F", 1024);
    const int nblk = nblocks_for(npix * 8, 256, stats ? cap_stats : 4096);
    DISPATCH_T(dtype, (stem_c1_fwd_kernel<T><<<nblk, 256, 0, ST>>>((const T*)a0, W1, npix, (T*)y1, stats)));
    SARSSL_CHECK_LAUNCH("stem_c1_fwd_kernel");
    return 0;
}

// dW1d: f64[256] workspace (zeroed here); result must be read as double
extern "C" int sarssl_stem_c1_wgrad(const void* dy1, const void* a0, long npix, double* dW1d, int dtype, void* stream) {
    if (hipMemsetAsync(dW1d, 0, 256 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    const int nblk = nblocks_for(npix * 8, 256, 1024);
    DISPATCH_GA(dtype, (stem_c1_wgrad_kernel<T, TA, 0><<<nblk, 256, 0, ST>>>((const T*)dy1, (const TA*)a0, npix, dW1d, (const TA*)nullptr, nullptr, nullptr, 0)));
    SARSSL_CHECK_LAUNCH("stem_c1_wgrad_kernel");
    return 0;
}
// dW1 from (dz1, y1) with the first BatchNorm's backward normalisation applied in registers; aff = [scale|shift|mean|rstd] (4 x 64),
// bnred = [s1 | s2] f64[128] (cl_bn_bwd_reduce / conv epilogue), use_stats = 0 for eval-mode BatchNorm.
extern "C" int sarssl_stem_c1_wgrad_bn(const void* dz1, const void* y1, const void* a0, long npix, const float* aff,
                                       const double* bnred, int use_stats, double* dW1d, int dtype, void* stream) {
    if (hipMemsetAsync(dW1d, 0, 256 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    const int nblk = nblocks_for(npix * 8, 256, 1024);
    DISPATCH_GA(dtype, (stem_c1_wgrad_kernel<T, TA, 1><<<nblk, 256, 0, ST>>>((const T*)dz1, (const TA*)a0, npix, dW1d, (const TA*)y1, aff, bnred, use_stats)));
    SARSSL_CHECK_LAUNCH("stem_c1_wgrad_kernel<bn>");
    return 0;
}

// One-pass backward of the first stem layer (see stem_c1_bwd_kernel).  red: f64[644] workspace (zeroed here); dW1 (64x4), dgamma, dbeta
// (64) are f32 gradient buffers that are accumulated into.  aff = [scale|shift|mean|rstd] (4 x 64).
extern "C" int sarssl_stem_c1_bwd(const void* dz1, const void* y1, const void* a0, long npix, const float* aff, int use_stats,
                                  double* red, float* dW1, float* dgamma, float* dbeta, int dtype, void* stream) {
    SARSSL_REQUIRE(npix > 0 && red && dW1 && dgamma && dbeta, "sarssl_stem_c1_bwd");
    if (SARSSL_ZERO(red, 644 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    constexpr int fast = 1;      // 4-channel-per-thread kernel wherever its shape constraints hold (the general kernel covers the rest)
    if (fast && (dtype == SARSSL_BF16 || dtype == SARSSL_MIX16) && (npix & 63) == 0) {
        const long nb64 = npix >> 6;
        static const int cap = grid_cap("SARSSL_GRID_C1B", 1024);
        const int nblk = (int)(nb64 < cap ? nb64 : cap);
        if (dtype == SARSSL_MIX16) stem_c1_bwd16_kernel<false, f16><<<nblk, 256, 0, ST>>>((const bf16*)dz1, (const f16*)y1, (const f16*)a0, npix, aff, red, nullptr);
        else stem_c1_bwd16_kernel<false, bf16><<<nblk, 256, 0, ST>>>((const bf16*)dz1, (const bf16*)y1, (const bf16*)a0, npix, aff, red, nullptr);
    } else {
        const int nblk = nblocks_for(npix * 8, 256, 1024);
        DISPATCH_GA(dtype, (stem_c1_bwd_kernel<T, TA><<<nblk, 256, 0, ST>>>((const T*)dz1, (const TA*)y1, (const TA*)a0, npix, aff, red)));
    }
    stem_c1_bwd_finalize_kernel<<<1, 256, 0, ST>>>(red, npix, aff, use_stats, dW1, dgamma, dbeta);
    SARSSL_CHECK_LAUNCH("stem_c1_bwd_kernel");
    return 0;
}

// Finalize of the first layer's backward when only G and s1 were reduced from the data (sarssl_conv3x3_dgrad_c1red): with
// y = W1 a0, xhat = (y - mean) * rstd everything else is a function of the input's moments mom = [Sa (4) | Saa (10, upper triangle)]:
//   X[co][c] = sum_p xhat a0[c] = rstd * (sum_c' W1[co][c'] Saa[c'][c] - mean * Sa[c])
//   s2[co]   = sum_p g xhat     = rstd * (sum_c W1[co][c] G[co][c] - mean * s1[co])
__global__ void stem_c1_bwd_finalize_mom_kernel(const double* __restrict__ red, const double* __restrict__ mom, const float* __restrict__ W1,
                                                long npix, const float* __restrict__ aff, int use_stats,
                                                float* __restrict__ dW1, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int o = threadIdx.x;                                 // co*4 + c
    const int co = o >> 2, c = o & 3;
    const double mean = (double)aff[128 + co], rstd = (double)aff[192 + co];
    double saa[4][4];
    {
        int k = 4;
        for (int i = 0; i < 4; ++i)
            for (int j = i; j < 4; ++j) { saa[i][j] = mom[k]; saa[j][i] = mom[k]; ++k; }
    }
    double wg = 0.0, wsaa = 0.0;
    for (int cc = 0; cc < 4; ++cc) { wg += (double)W1[co * 4 + cc] * red[co * 4 + cc]; wsaa += (double)W1[co * 4 + cc] * saa[cc][c]; }
    const double s1 = red[512 + co];
    const double s2 = rstd * (wg - mean * s1);
    const double X = rstd * (wsaa - mean * mom[c]);
    const double invN = 1.0 / (double)npix;
    const double m1 = use_stats ? s1 * invN : 0.0, m2 = use_stats ? s2 * invN : 0.0;
    dW1[o] += (float)((double)aff[co] * (red[o] - m1 * mom[c] - m2 * X));
    if (c == 0) { dbeta[co] += (float)s1; dgamma[co] += (float)s2; }
}
extern "C" int sarssl_stem_c1_bwd_finalize_mom(const double* red, const double* mom14, const float* W1, long npix, const float* aff,
                                               int use_stats, float* dW1, float* dgamma, float* dbeta, void* stream) {
    SARSSL_REQUIRE(red && mom14 && W1 && aff && dW1 && dgamma && dbeta && npix > 0, "sarssl_stem_c1_bwd_finalize_mom");
    stem_c1_bwd_finalize_mom_kernel<<<1, 256, 0, ST>>>(red, mom14, W1, npix, aff, use_stats, dW1, dgamma, dbeta);
    SARSSL_CHECK_LAUNCH("stem_c1_bwd_finalize_mom_kernel");
    return 0;
}

// The same pass without y1: y1 = W1 a0 is recomputed (bf16 activations, npix % 64 == 0; W1 = the 64 x 4 first-layer weight).
extern "C" int sarssl_stem_c1_bwd_a0(const void* dz1, const void* a0, const float* W1, long npix, const float* aff, int use_stats,
                                     double* red, float* dW1, float* dgamma, float* dbeta, int dtype, void* stream) {
    SARSSL_REQUIRE(dtype == SARSSL_BF16 || dtype == SARSSL_MIX16, "sarssl_stem_c1_bwd_a0(dtype)");
    SARSSL_REQUIRE(npix > 0 && (npix & 63) == 0 && red && dW1 && dgamma && dbeta && W1, "sarssl_stem_c1_bwd_a0");
    if (SARSSL_ZERO(red, 644 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    static const int cap = grid_cap("SARSSL_GRID_C1B", 1024);
    const long nb64 = npix >> 6;
    const int nblk = (int)(nb64 < cap ? nb64 : cap);
    if (dtype == SARSSL_MIX16) stem_c1_bwd16_kernel<true, f16><<<nblk, 256, 0, ST>>>((const bf16*)dz1, nullptr, (const f16*)a0, npix, aff, red, W1);
    else stem_c1_bwd16_kernel<true, bf16><<<nblk, 256, 0, ST>>>((const bf16*)dz1, nullptr, (const bf16*)a0, npix, aff, red, W1);
    stem_c1_bwd_finalize_kernel<<<1, 256, 0, ST>>>(red, npix, aff, use_stats, dW1, dgamma, dbeta);
    SARSSL_CHECK_LAUNCH("stem_c1_bwd16_kernel<a0>");
    return 0;
}

// BatchNorm(1) statistics of y1 = W1 a0 without forming y1: sum_p y1[co] = W1[co] . Sa and sum_p y1[co]^2 = W1[co]^T Saa W1[co] with
// Sa = sum_p a0[p] (4) and Saa = sum_p a0[p] a0[p]^T (10 unique entries) - one pass over the 4-channel input (33 MB at B = 64).
// mom: f64[14] = [Sa | Saa upper triangle, row-major], zeroed by the caller.
template <typename T>
__global__ __launch_bounds__(256) void stem_a0_moments_kernel(const T* __restrict__ a0, long npix, double* __restrict__ mom) {
    __shared__ float sred[4][14];
    float acc[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) acc[i] = 0.f;
    const long nthreads = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += nthreads) {
        const float4 a = ld4(a0 + p * 4);
        const float v[4] = {a.x, a.y, a.z, a.w};
        int k = 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] += v[i];
#pragma unroll
            for (int j = i; j < 4; ++j) { acc[k] = fmaf(v[i], v[j], acc[k]); ++k; }
        }
    }
#pragma unroll
    for (int i = 0; i < 14; ++i) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) acc[i] += __shfl_xor(acc[i], o, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 14; ++i) sred[wave][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < 14) atomicAdd(&mom[threadIdx.x], (double)(sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x]));
}
// sums[co] = W1[co] . Sa, sums[64 + co] = W1[co]^T Saa W1[co]   (f64; the layout bn_finalize / sarssl_bn_train_affine take)
__global__ void stem_c1_stats_from_moments_kernel(const double* __restrict__ mom, const float* __restrict__ W1, double* __restrict__ sums) {
    const int co = threadIdx.x;
    if (co >= 64) return;
    double w[4];
    for (int c = 0; c < 4; ++c) w[c] = (double)W1[co * 4 + c];
    double s = 0.0, q = 0.0;
    int k = 4;
    for (int i = 0; i < 4; ++i) {
        s += w[i] * mom[i];
        for (int j = i; j < 4; ++j) { q += (i == j ? 1.0 : 2.0) * w[i] * w[j] * mom[k]; ++k; }
    }
    sums[co] = s; sums[64 + co] = q;
}
extern "C" int sarssl_stem_c1_stats(const void* a0, long npix, const float* W1, double* mom14, double* sums128, int dtype, void* stream) {
    SARSSL_REQUIRE(npix > 0 && mom14 && sums128 && W1, "sarssl_stem_c1_stats");
    if (SARSSL_ZERO(mom14, 14 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    const int nblk = nblocks_for(npix, 256 * 4, 512);
    DISPATCH_T(dtype, (stem_a0_moments_kernel<T><<<nblk, 256, 0, ST>>>((const T*)a0, npix, mom14)));
    stem_c1_stats_from_moments_kernel<<<1, 64, 0, ST>>>(mom14, W1, sums128);
    SARSSL_CHECK_LAUNCH("stem_a0_moments_kernel");
    return 0;
}
// sarssl_stem_c1_stats + sarssl_bn_finalize (C = 64, N = npix) with the sums -> affine step done by the same one-workgroup kernel that
// forms the sums from the moments: aff = [scale | shift | mean | rstd] (4 x 64), running statistics / batch counter updated.
__global__ void stem_c1_affine_from_moments_kernel(const double* __restrict__ mom, const float* __restrict__ W1, long N,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                                   float* __restrict__ running_mean, float* __restrict__ running_var, long* __restrict__ nbt,
                                                   float* __restrict__ aff) {
    const int co = threadIdx.x;
    if (co == 0 && nbt) *nbt += 1;
    if (co >= 64) return;
    double w[4];
    for (int c = 0; c < 4; ++c) w[c] = (double)W1[co * 4 + c];
    double s = 0.0, q = 0.0;
    int k = 4;
    for (int i = 0; i < 4; ++i) {
        s += w[i] * mom[i];
        for (int j = i; j < 4; ++j) { q += (i == j ? 1.0 : 2.0) * w[i] * w[j] * mom[k]; ++k; }
    }
    const double m = s / (double)N;
    double var = q / (double)N - m * m;
    if (var < 0.0) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    aff[co] = gamma[co] * r; aff[64 + co] = beta[co] - (float)m * gamma[co] * r; aff[128 + co] = (float)m; aff[192 + co] = r;
    if (running_mean && isfinite(m) && isfinite(var)) {      // (an overflowed fp16 forward must not poison the running statistics)
        const double unb = (N > 1) ? var * (double)N / (double)(N - 1) : var;
        running_mean[co] = (1.f - momentum) * running_mean[co] + momentum * (float)m;
        running_var[co] = (1.f - momentum) * running_var[co] + momentum * (float)unb;
    }
}
extern "C" int sarssl_stem_c1_stats_affine(const void* a0, long npix, const float* W1, double* mom14, const float* gamma, const float* beta,
                                           float eps, float momentum, float* running_mean, float* running_var, long* nbt, float* aff,
                                           int dtype, void* stream) {
    SARSSL_REQUIRE(npix > 0 && mom14 && W1 && gamma && beta && aff, "sarssl_stem_c1_stats_affine");
    if (SARSSL_ZERO(mom14, 14 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    const int nblk = nblocks_for(npix, 256 * 4, 512);
    DISPATCH_T(dtype, (stem_a0_moments_kernel<T><<<nblk, 256, 0, ST>>>((const T*)a0, npix, mom14)));
    stem_c1_affine_from_moments_kernel<<<1, 64, 0, ST>>>(mom14, W1, npix, gamma, beta, eps, momentum, running_mean, running_var, nbt, aff);
    SARSSL_CHECK_LAUNCH("stem_c1_affine_from_moments_kernel");
    return 0;
}

// (8 bins x 4U frames) items, U loads in flight per thread; 2048 workgroups = one resident round at B = 64 (sweep 1024 / 2048 / 4096 /
// 8192 with U = 8: 125 / 121 / 126 / 141 us); frame counts that are no multiple of 16 (or bin counts of 8) take the pixel-order loop
static int stem_c4_fwd_launch(const void* y3, const float* W4, const float* scale, const float* shift, int nb, int F, int Tn, void* y4,
                              double* stats8, int dtype, void* stream, void* y4lo = nullptr) {
    SARSSL_REQUIRE(nb > 0 && F > 0 && Tn > 0 && (long)nb * F * Tn < (1L << 31), "sarssl_stem_c4_fwd");
    static const int cap = grid_cap("SARSSL_GRID_C4F", 2048);
    const int u = (F % 8 == 0) ? (Tn % 32 == 0 ? 8 : (Tn % 16 == 0 ? 4 : 0)) : 0;
    if (u == 0) {
        const int nblk = nblocks_for((long)nb * F * Tn * 8, 256, 8192);
        DISPATCH_T(dtype, (stem_c4_fwd_kernel<T, 0><<<nblk, 256, 0, ST>>>((const T*)y3, W4, scale, shift, nb, F, Tn, (T*)y4, stats8, (T*)y4lo)));
    } else {
        const int nblk = nblocks_for((long)nb * (F / 8) * (Tn / (4 * u)), 1, cap);
        if (u == 8) { DISPATCH_T(dtype, (stem_c4_fwd_kernel<T, 8><<<nblk, 256, 0, ST>>>((const T*)y3, W4, scale, shift, nb, F, Tn, (T*)y4, stats8, (T*)y4lo))); }
        else { DISPATCH_T(dtype, (stem_c4_fwd_kernel<T, 4><<<nblk, 256, 0, ST>>>((const T*)y3, W4, scale, shift, nb, F, Tn, (T*)y4, stats8, (T*)y4lo))); }
    }
    SARSSL_CHECK_LAUNCH("stem_c4_fwd_kernel");
    return 0;
}
extern "C" int sarssl_stem_c4_fwd(const void* y3, const float* W4, const float* scale, const float* shift, int nb, int F,
                                  int Tn, void* y4, int dtype, void* stream) {
    return stem_c4_fwd_launch(y3, W4, scale, shift, nb, F, Tn, y4, nullptr, dtype, stream);
}
// The same, also returning the BatchNorm(4) sums of the stored output: stats8 f64[8] = [sum (4) | sum of squares (4)] (zeroed here) - the
// separate statistics pass over y4 is not needed.
extern "C" int sarssl_stem_c4_fwd_stats(const void* y3, const float* W4, const float* scale, const float* shift, int nb, int F, int Tn,
                                        void* y4, double* stats8, int dtype, void* stream) {
    SARSSL_REQUIRE(stats8 != nullptr, "sarssl_stem_c4_fwd_stats");
    if (SARSSL_ZERO(stats8, 8 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    return stem_c4_fwd_launch(y3, W4, scale, shift, nb, F, Tn, y4, stats8, dtype, stream);
}

// Hybrid mode: the f32 result as a pair of `dtype` tensors (y4_hi = T(o), y4_lo = T(o - y4_hi)); stats8 (may be null) = the sums of the
// pair's value.  y4_hi is exactly what sarssl_stem_c4_fwd stores - the backward kernels read it unchanged.
extern "C" int sarssl_stem_c4_fwd_pair(const void* y3, const float* W4, const float* scale, const float* shift, int nb, int F, int Tn,
                                       void* y4_hi, void* y4_lo, double* stats8, int dtype, void* stream) {
    SARSSL_REQUIRE(y4_lo != nullptr && (dtype == SARSSL_F16 || dtype == SARSSL_BF16), "sarssl_stem_c4_fwd_pair");
    if (stats8 && SARSSL_ZERO(stats8, 8 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    return stem_c4_fwd_launch(y3, W4, scale, shift, nb, F, Tn, y4_hi, stats8, dtype, stream, y4_lo);
}

// red: f64[384] (zeroed here): [0,256) dW4[c][ci], [256,320) s1, [320,384) s2
extern "C" int sarssl_stem_c4_bwd(const void* y3, const void* dy4, const float* W4, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, int nb, int F, int Tn, void* g3, double* red,
                                  int dtype, void* stream) {
    if (SARSSL_ZERO(red, 384 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    const int nblk = nblocks_for((long)nb * F * Tn * 8, 256, 1024);
    DISPATCH_GA(dtype, (stem_c4_bwd_kernel<T, TA, 0><<<nblk, 256, 0, ST>>>((const TA*)y3, (const T*)dy4, W4, scale, shift, mean, rstd,
                                                                          nb, F, Tn, (T*)g3, red, 1)));
    SARSSL_CHECK_LAUNCH("stem_c4_bwd_kernel");
    return 0;
}
// Two-phase form of the same backward step.  phase 1: red (zeroed here) = [dW4 | s1 | s2], nothing stored.
// phase 2: dy3 = BatchNorm(3)-input gradient written directly from (y3, dy4, red); use_stats = 0 for eval-mode BatchNorm.
extern "C" int sarssl_stem_c4_bwd_sums(const void* y3, const void* dy4, const float* W4, const float* scale, const float* shift,
                                       const float* mean, const float* rstd, int nb, int F, int Tn, double* red, int dtype,
                                       void* stream) {
    if (SARSSL_ZERO(red, 384 * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    constexpr int fast = 1;      // 4-channel-per-thread kernel wherever its shape constraints hold (the general kernel covers the rest)
    if (fast && (dtype == SARSSL_BF16 || dtype == SARSSL_MIX16) && (F & 15) == 0 && (Tn & 15) == 0 && nb > 0) {
        const long ntile = (long)nb * (F >> 4) * (Tn >> 4);
        static const int cap = grid_cap("SARSSL_GRID_C4S", 1024);         // 4 workgroups per CU
        const int nblk = (int)(ntile < cap ? ntile : cap);
        static const int nstream = grid_cap("SARSSL_C4S_STREAMS", 1);
        if (dtype == SARSSL_MIX16) stem_c4_bwd_sums16_kernel<f16><<<nblk, 256, 0, ST>>>((const f16*)y3, (const bf16*)dy4, W4, scale, shift, mean, rstd, nb, F, Tn, red, nstream);
        else stem_c4_bwd_sums16_kernel<bf16><<<nblk, 256, 0, ST>>>((const bf16*)y3, (const bf16*)dy4, W4, scale, shift, mean, rstd, nb, F, Tn, red, nstream);
        SARSSL_CHECK_LAUNCH("stem_c4_bwd_sums16_kernel");
        return 0;
    }
    const int nblk = nblocks_for((long)nb * F * Tn * 8, 256, 1024);
    DISPATCH_GA(dtype, (stem_c4_bwd_kernel<T, TA, 1><<<nblk, 256, 0, ST>>>((const TA*)y3, (const T*)dy4, W4, scale, shift, mean, rstd,
                                                                          nb, F, Tn, (T*)nullptr, red, 1)));
    SARSSL_CHECK_LAUNCH("stem_c4_bwd_kernel<sums>");
    return 0;
}
extern "C" int sarssl_stem_c4_bwd_apply_pg(const void* y3, const void* dy4, const float* W4, const float* scale, const float* shift,
                                           const float* mean, const float* rstd, int nb, int F, int Tn, const double* red,
                                           int use_stats, void* dy3, float* gW4, float* dgamma, float* dbeta, int dtype, void* stream);
extern "C" int sarssl_stem_c4_bwd_apply(const void* y3, const void* dy4, const float* W4, const float* scale, const float* shift,
                                        const float* mean, const float* rstd, int nb, int F, int Tn, const double* red,
                                        int use_stats, void* dy3, int dtype, void* stream) {
    return sarssl_stem_c4_bwd_apply_pg(y3, dy4, W4, scale, shift, mean, rstd, nb, F, Tn, red, use_stats, dy3, nullptr, nullptr, nullptr, dtype, stream);
}
// The same pass; workgroup 0 also adds gW4 (4 x 64) += red[0:256], dbeta += red[256:320], dgamma += red[320:384] (all three or none).
extern "C" int sarssl_stem_c4_bwd_apply_pg(const void* y3, const void* dy4, const float* W4, const float* scale, const float* shift,
                                           const float* mean, const float* rstd, int nb, int F, int Tn, const double* red,
                                           int use_stats, void* dy3, float* gW4, float* dgamma, float* dbeta, int dtype, void* stream) {
    SARSSL_REQUIRE((gW4 == nullptr) == (dgamma == nullptr) && (gW4 == nullptr) == (dbeta == nullptr), "sarssl_stem_c4_bwd_apply_pg");
    static const int cap = grid_cap("SARSSL_GRID_C4A", 4096);
    const int nblk = nblocks_for((long)nb * F * Tn * 8, 256, cap);
    DISPATCH_GA(dtype, (stem_c4_bwd_kernel<T, TA, 2><<<nblk, 256, 0, ST>>>((const TA*)y3, (const T*)dy4, W4, scale, shift, mean, rstd,
                                                                          nb, F, Tn, (T*)dy3, (double*)red, use_stats, gW4, dgamma, dbeta)));
    SARSSL_CHECK_LAUNCH("stem_c4_bwd_kernel<apply>");
    return 0;
}

// view [N][C] as [rows][L] with L = max(C, 64) (C < 64 needs 64 % C == 0 and N*C % 64 == 0)
static inline bool cl_view(long N, int C, long* rows, int* L) {
    if (C <= 0 || (C & 3) || N <= 0) return false;
    if (C >= 64) { if (C & 7) return false; *rows = N; *L = C; return true; }
    if (64 % C || (N * C) % 64) return false;
    *rows = N * C / 64; *L = 64; return true;
}
static inline bool cl_rowthreads_ok(int L) { return (L & 7) == 0 && (L >> 3) <= 256 && 256 % (L >> 3) == 0; }
static inline int cl_rowgrid(long rows, int L) {
    const int rpb = 256 / (L >> 3);
    long b = (rows + (long)rpb * 4 - 1) / ((long)rpb * 4); if (b < 1) b = 1;
    return (int)(b > 4096 ? 4096 : b);
}
static inline dim3 cl_grid(long rows, int L, int C) {
    int gx = (L + 63) / 64;
    long gy = (rows + 255) / 256; if (gy < 1) gy = 1;
    long cap = (C < 64 ? 512 : 2048) / gx; if (cap < 1) cap = 1;   // C < 64: every workgroup hits the same 2C addresses
    if (gy > cap) gy = cap;
    return dim3(gx, (unsigned)gy, 1);
}

// sums: f64[2C] (zeroed here)
extern "C" int sarssl_cl_stats(const void* x, long N, int C, double* sums, int dtype, void* stream) {
    long rows; int L;
    SARSSL_REQUIRE(cl_view(N, C, &rows, &L), "sarssl_cl_stats(C % 8 == 0 or C | 64; C < 64 needs N*C % 64 == 0)");
    if (SARSSL_ZERO(sums, 2 * C * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    const dim3 grid = cl_grid(rows, L, C);
    DISPATCH_T(dtype, (cl_stats_kernel<T><<<grid, 256, 0, ST>>>((const T*)x, rows, L, C, sums)));
    SARSSL_CHECK_LAUNCH("cl_stats_kernel");
    return 0;
}

extern "C" int sarssl_bn_finalize(const double* sums, long N, int C, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, long* nbt, float* scale,
                                  float* shift, float* mean, float* rstd, void* stream) {
    bn_finalize_kernel<<<(C + 63) / 64, 64, 0, ST>>>(sums, N, C, gamma, beta, eps, momentum, running_mean, running_var, nbt,
                                                     scale, shift, mean, rstd);
    SARSSL_CHECK_LAUNCH("bn_finalize_kernel");
    return 0;
}

extern "C" int sarssl_bn_eval_affine(int C, const float* gamma, const float* beta, float eps, const float* running_mean,
                                     const float* running_var, float* scale, float* shift, float* mean, float* rstd,
                                     void* stream) {
    bn_eval_affine_kernel<<<(C + 63) / 64, 64, 0, ST>>>(C, gamma, beta, eps, running_mean, running_var, scale, shift, mean, rstd);
    SARSSL_CHECK_LAUNCH("bn_eval_affine_kernel");
    return 0;
}

extern "C" int sarssl_cl_affine_act(const void* x, long N, int C, const float* scale, const float* shift, int act, void* z,
                                    int dtype, void* stream) {
    long rows; int L;
    SARSSL_REQUIRE(cl_view(N, C, &rows, &L) && cl_rowthreads_ok(L), "sarssl_cl_affine_act");
    DISPATCH_T(dtype, (cl_affine_act_kernel<T><<<cl_rowgrid(rows, L), 256, 0, ST>>>((const T*)x, rows, L, C, scale, shift, act, (T*)z)));
    SARSSL_CHECK_LAUNCH("cl_affine_act_kernel");
    return 0;
}

// z = act(BatchNorm_train(x)) from the finished sums (sum | sum of squares, f64[2C]) in ONE launch: sarssl_bn_finalize followed by
// sarssl_cl_affine_act, bit-identical (scale / shift / mean / rstd rows and the running statistics are written by workgroup 0).
// C % 8 == 0, 64 <= C <= 2048, 256 % (C / 8) == 0.
extern "C" int sarssl_cl_bn_train_act(const void* x, long N, int C, const double* sums, const float* gamma, const float* beta, float eps,
                                      float momentum, float* running_mean, float* running_var, long* nbt, float* scale, float* shift,
                                      float* mean, float* rstd, int act, void* z, int dtype, void* stream) {
    SARSSL_REQUIRE(N > 0 && C >= 64 && cl_rowthreads_ok(C) && sums && gamma && beta && scale && shift && mean && rstd, "sarssl_cl_bn_train_act");
    DISPATCH_T(dtype, (cl_bn_train_act_kernel<T><<<cl_rowgrid(N, C), 256, 0, ST>>>((const T*)x, N, C, sums, gamma, beta, eps, momentum, running_mean,
                                                                                  running_var, nbt, scale, shift, mean, rstd, act, (T*)z)));
    SARSSL_CHECK_LAUNCH("cl_bn_train_act_kernel");
    return 0;
}

// red: f64[2C] (zeroed here)
extern "C" int sarssl_cl_bn_bwd_reduce(const void* dz, const void* y, long N, int C, const float* scale, const float* shift,
                                       const float* mean, const float* rstd, int act, double* red, int dtype, void* stream) {
    long rows; int L;
    SARSSL_REQUIRE(cl_view(N, C, &rows, &L), "sarssl_cl_bn_bwd_reduce");
    if (SARSSL_ZERO(red, 2 * C * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    const dim3 grid = cl_grid(rows, L, C);
    DISPATCH_GA(dtype, (cl_bn_bwd_reduce_kernel<T, TA><<<grid, 256, 0, ST>>>((const T*)dz, (const TA*)y, rows, L, C, scale, shift,
                                                                            mean, rstd, act, red)));
    SARSSL_CHECK_LAUNCH("cl_bn_bwd_reduce_kernel");
    return 0;
}

extern "C" int sarssl_cl_bn_bwd_apply_pg(const void* dz, const void* y, long N, int C, const float* scale, const float* shift,
                                         const float* mean, const float* rstd, int act, int g_is_masked, int use_stats,
                                         const double* red, void* dy, float* dgamma, float* dbeta, int dtype, void* stream);
extern "C" int sarssl_cl_bn_bwd_apply(const void* dz, const void* y, long N, int C, const float* scale, const float* shift,
                                      const float* mean, const float* rstd, int act, int g_is_masked, int use_stats,
                                      const double* red, void* dy, int dtype, void* stream) {
    return sarssl_cl_bn_bwd_apply_pg(dz, y, N, C, scale, shift, mean, rstd, act, g_is_masked, use_stats, red, dy, nullptr, nullptr, dtype, stream);
}
// The same pass; workgroup 0 also adds the BatchNorm parameter gradients dbeta += red[0:C], dgamma += red[C:2C] (f32 buffers; both or neither).
extern "C" int sarssl_cl_bn_bwd_apply_pg(const void* dz, const void* y, long N, int C, const float* scale, const float* shift,
                                         const float* mean, const float* rstd, int act, int g_is_masked, int use_stats,
                                         const double* red, void* dy, float* dgamma, float* dbeta, int dtype, void* stream) {
    SARSSL_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "sarssl_cl_bn_bwd_apply_pg");
    long rows; int L;
    SARSSL_REQUIRE(cl_view(N, C, &rows, &L) && cl_rowthreads_ok(L), "sarssl_cl_bn_bwd_apply");
    static const int cap = grid_cap("SARSSL_GRID_BNA", 1024);       // one resident round (4 workgroups per CU): every further round repeats the per-channel prologue and adds a tail (4096: 378 us, 1024: 321 us at B = 64)
    const int rpb_ = 256 / (L >> 3);
    long nb_ = (rows + (long)rpb_ * 2 - 1) / ((long)rpb_ * 2); if (nb_ < 1) nb_ = 1;          // two rows per thread and round
    const int grid_ = (int)(nb_ > cap ? cap : nb_);
    DISPATCH_GA(dtype, (cl_bn_bwd_apply_kernel<T, TA><<<grid_, 256, 0, ST>>>((const T*)dz, (const TA*)y, rows, L, C, N, scale,
                                                                                     shift, mean, rstd, act, g_is_masked,
                                                                                     use_stats, red, (T*)dy, dgamma, dbeta)));
    SARSSL_CHECK_LAUNCH("cl_bn_bwd_apply_kernel");
    return 0;
}
