// Depthwise-convolution part of the Conformer convolution module as LDS-staged tiles
// (code/common/conformer/convolution.py:136-149: ... PointwiseConv1d(d -> 2d) -> GLU -> DepthwiseConv1d(k = 31, pad 15, no bias) ->
// BatchNorm1d -> Swish -> ...; activations are [B*T][channels], channels contiguous).
//
//   dwglu_fwd    c[t][ch] = sum_k w[ch][k] * g[t + k - 15][ch],  g = a * sigmoid(b) taken straight from the pointwise conv's [.., 2d]
//                output (GLU fused into the tile load), plus the BatchNorm batch sums (sum, sum of squares per channel) in the epilogue:
//                replaces glu_fwd + dwconv_fwd + cl_stats (3 passes, 2-byte loads per lane) by one pass of 16-byte loads / stores
//   dwglu_bwd    dh = GLU backward of dg[t][ch] = sum_k w[ch][30 - k] * dc[t + k - 15][ch]  (data gradient + glu_bwd in one pass)
//   dwglu_wgrad  dw[ch][k] = sum_{b,t} dc[t][ch] * g[t + k - 15][ch], g recomputed from h (it is never stored)
// Tile = 64 frames x 64 channels (+ 15 halo frames each side) per 256-thread workgroup: 16-byte global accesses (8 channels of one
// frame), LDS columns read with lane = channel (conflict-free), 16 outputs per thread from a 46-deep register window.
#include "common.h"

#define DWK 31
#define DTT 64                       // frames per tile
#define DHR (DTT + DWK - 1)          // 94 tile rows incl. halo
#define DTC 64                       // channels per tile

// value as the consumer will read it back from storage (bf16 rounding for bf16 tensors): keeps the fused and the unfused paths, and
// the forward and the recomputing weight-gradient kernel, on identical numbers
template <typename T> __device__ __forceinline__ float as_stored(float v) { return round_as<T>(v); }

// GLU tile: sG[r][c] = a * sigmoid(b) for frames t0 - 15 + r (zero outside [0, T)), channels c0 .. c0 + 63 (zero beyond d)
template <typename T>
__device__ __forceinline__ void load_glu_tile(const T* __restrict__ h, long row0, int Tn, int d, int t0, int c0, float (*sG)[DTC], int tid) {
    const int ch8 = (tid & 7) * 8;
    for (int r = tid >> 3; r < DHR; r += 32) {
        const int t = t0 - 15 + r;
        f8 g;
#pragma unroll
        for (int e = 0; e < 8; ++e) g.v[e] = 0.f;
        if (t >= 0 && t < Tn && c0 + ch8 < d) {
            const T* p = h + (row0 + t) * (2L * d) + c0 + ch8;
            const f8 a = ld8(p), b = ld8(p + d);
#pragma unroll
            for (int e = 0; e < 8; ++e) g.v[e] = as_stored<T>(a.v[e] * sigmoidf_(b.v[e]));
        }
        *(float4*)&sG[r][ch8] = make_float4(g.v[0], g.v[1], g.v[2], g.v[3]);
        *(float4*)&sG[r][ch8 + 4] = make_float4(g.v[4], g.v[5], g.v[6], g.v[7]);
    }
}
template <typename T>
__device__ __forceinline__ void load_plain_tile(const T* __restrict__ x, long row0, int Tn, int d, int t0, int c0, int nrow, int roff,
                                                float (*sX)[DTC], int tid) {
    const int ch8 = (tid & 7) * 8;
    for (int r = tid >> 3; r < nrow; r += 32) {
        const int t = t0 + roff + r;
        f8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] = 0.f;
        if (t >= 0 && t < Tn && c0 + ch8 < d) v = ld8(x + (row0 + t) * (long)d + c0 + ch8);
        *(float4*)&sX[r][ch8] = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]);
        *(float4*)&sX[r][ch8 + 4] = make_float4(v.v[4], v.v[5], v.v[6], v.v[7]);
    }
}

// 16 consecutive outputs of channel column c from the 46-row window starting at row r0 of the tile
__device__ __forceinline__ void conv16(const float (*sX)[DTC], int r0, int c, const float (&wk)[DWK], float (&out)[16]) {
    float win[16 + DWK - 1];
#pragma unroll
    for (int i = 0; i < 16 + DWK - 1; ++i) win[i] = sX[r0 + i][c];
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < DWK; ++k) acc = fmaf(wk[k], win[o + k], acc);
        out[o] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dwglu_fwd_kernel(const T* __restrict__ h, const float* __restrict__ w, int Tn, int d,
                                                        T* __restrict__ c, double* __restrict__ sums) {
    __shared__ __attribute__((aligned(16))) float sG[DHR][DTC];
    __shared__ float sRed[4][DTC][2];
    const int tid = threadIdx.x, ch = tid & 63, q = tid >> 6;
    const int c0 = blockIdx.x * DTC, t0 = blockIdx.y * DTT;
    const long row0 = (long)blockIdx.z * Tn;
    const bool live = c0 + ch < d;
    float wk[DWK];
#pragma unroll
    for (int k = 0; k < DWK; ++k) wk[k] = live ? w[(long)(c0 + ch) * DWK + k] : 0.f;
    load_glu_tile<T>(h, row0, Tn, d, t0, c0, sG, tid);
    __syncthreads();
    float out[16];
    conv16(sG, q * 16, ch, wk, out);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        out[o] = as_stored<T>(out[o]);
        if (t0 + q * 16 + o < Tn) { s1 += out[o]; s2 += out[o] * out[o]; }
    }
    __syncthreads();                                     // everyone is done reading the input tile: reuse it for the output tile
#pragma unroll
    for (int o = 0; o < 16; ++o) sG[q * 16 + o][ch] = out[o];
    sRed[q][ch][0] = s1; sRed[q][ch][1] = s2;
    __syncthreads();
    {
        const int ch8 = (tid & 7) * 8;
        for (int r = tid >> 3; r < DTT; r += 32) {
            if (t0 + r < Tn && c0 + ch8 < d) {
                f8 v;
                const float4 a = *(const float4*)&sG[r][ch8], b = *(const float4*)&sG[r][ch8 + 4];
                v.v[0] = a.x; v.v[1] = a.y; v.v[2] = a.z; v.v[3] = a.w; v.v[4] = b.x; v.v[5] = b.y; v.v[6] = b.z; v.v[7] = b.w;
                st8(c + (row0 + t0 + r) * (long)d + c0 + ch8, v);
            }
        }
    }
    if (sums && tid < 128) {
        const int cc = tid & 63, which = tid >> 6;
        if (c0 + cc < d) {
            const float s = (sRed[0][cc][which] + sRed[1][cc][which]) + (sRed[2][cc][which] + sRed[3][cc][which]);
            atomicAdd(&sums[which * d + c0 + cc], (double)s);
        }
    }
}

// dh[.., 0:d] = dg * sigmoid(b),  dh[.., d:2d] = dg * a * sigmoid(b) * (1 - sigmoid(b)),  dg = flipped-tap convolution of dc
template <typename T, typename TA>      // T: gradients, TA: the pointwise-conv output h saved by the forward pass
__global__ __launch_bounds__(256) void dwglu_bwd_kernel(const T* __restrict__ dc, const TA* __restrict__ h, const float* __restrict__ w,
                                                        int Tn, int d, T* __restrict__ dh) {
    __shared__ __attribute__((aligned(16))) float sX[DHR][DTC];
    const int tid = threadIdx.x, ch = tid & 63, q = tid >> 6;
    const int c0 = blockIdx.x * DTC, t0 = blockIdx.y * DTT;
    const long row0 = (long)blockIdx.z * Tn;
    const bool live = c0 + ch < d;
    float wk[DWK];
#pragma unroll
    for (int k = 0; k < DWK; ++k) wk[k] = live ? w[(long)(c0 + ch) * DWK + (DWK - 1 - k)] : 0.f;
    load_plain_tile<T>(dc, row0, Tn, d, t0, c0, DHR, -15, sX, tid);
    __syncthreads();
    float out[16];
    conv16(sX, q * 16, ch, wk, out);
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 16; ++o) sX[q * 16 + o][ch] = out[o];
    __syncthreads();
    const int ch8 = (tid & 7) * 8;
    for (int r = tid >> 3; r < DTT; r += 32) {
        if (t0 + r < Tn && c0 + ch8 < d) {
            const TA* p = h + (row0 + t0 + r) * (2L * d) + c0 + ch8;
            const f8 a = ld8(p), b = ld8(p + d);
            const float4 g0 = *(const float4*)&sX[r][ch8], g1 = *(const float4*)&sX[r][ch8 + 4];
            const float dg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            f8 da, db;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sg = sigmoidf_(b.v[e]);
                da.v[e] = dg[e] * sg;
                db.v[e] = dg[e] * a.v[e] * sg * (1.f - sg);
            }
            T* o = dh + (row0 + t0 + r) * (2L * d) + c0 + ch8;
            st8(o, da);
            st8(o + d, db);
        }
    }
}

// partial[part][ch][k] = sum over this workgroup's share of (batch, frame tile)s; grid (channel tiles, parts)
template <typename T, typename TA>
__global__ __launch_bounds__(256) void dwglu_wgrad_kernel(const T* __restrict__ dc, const TA* __restrict__ h, int nb, int Tn, int d,
                                                          float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float sG[DHR][DTC];
    __shared__ __attribute__((aligned(16))) float sD[DTT][DTC];
    const int tid = threadIdx.x, ch = tid & 63, q = tid >> 6;
    const int c0 = blockIdx.x * DTC;
    const int ttiles = (Tn + DTT - 1) / DTT, ntile = nb * ttiles;
    float acc[DWK];
#pragma unroll
    for (int k = 0; k < DWK; ++k) acc[k] = 0.f;
    for (int tile = blockIdx.y; tile < ntile; tile += gridDim.y) {
        const int b = tile / ttiles, t0 = (tile % ttiles) * DTT;
        const long row0 = (long)b * Tn;
        __syncthreads();
        load_glu_tile<TA>(h, row0, Tn, d, t0, c0, sG, tid);
        load_plain_tile<T>(dc, row0, Tn, d, t0, c0, DTT, 0, sD, tid);
        __syncthreads();
        float win[16 + DWK - 1];
#pragma unroll
        for (int i = 0; i < 16 + DWK - 1; ++i) win[i] = sG[q * 16 + i][ch];
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            const float g = sD[q * 16 + o][ch];
#pragma unroll
            for (int k = 0; k < DWK; ++k) acc[k] = fmaf(g, win[o + k], acc[k]);
        }
    }
    __syncthreads();
    // fold the four frame groups: groups 1..3 park their 31 sums in the (now idle) 94-row input tile: 3 x 31 = 93 rows
    float (*sAcc)[DTC] = sG;
    if (q > 0) {
#pragma unroll
        for (int k = 0; k < DWK; ++k) sAcc[(q - 1) * DWK + k][ch] = acc[k];
    }
    __syncthreads();
    if (q == 0 && c0 + ch < d) {
        float* P = partial + ((long)blockIdx.y * d + c0 + ch) * DWK;
#pragma unroll
        for (int k = 0; k < DWK; ++k) P[k] = acc[k] + sAcc[k][ch] + sAcc[DWK + k][ch] + sAcc[2 * DWK + k][ch];
    }
}

// out[i] += sum_p partial[p][i]
__global__ __launch_bounds__(256) void dw_partial_reduce_kernel(const float* __restrict__ partial, int nparts, long n, float* __restrict__ out) {
    __shared__ float sred[4][64];
    const int col = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + col;
    float s = 0.f;
    if (i < n)
        for (int p = slot; p < nparts; p += 4) s += partial[(long)p * n + i];
    sred[slot][col] = s;
    __syncthreads();
    if (slot == 0 && i < n) out[i] += (sred[0][col] + sred[1][col]) + (sred[2][col] + sred[3][col]);
}

#define ST ((hipStream_t)stream)
#define DW_DISPATCH(dtype, stmt)                                                                          \
    do {                                                                                                  \
        if ((dtype) == SARSSL_BF16) { typedef bf16 T; stmt; }                                             \
        else if ((dtype) == SARSSL_F32) { typedef float T; stmt; }                                        \
        else if ((dtype) == SARSSL_F16) { typedef f16 T; stmt; }                                          \
        else { sarssl_set_error("dwconv: unsupported dtype %d", (int)(dtype)); return -1; }              \
    } while (0)
// backward kernels: gradients T next to the saved forward tensor TA (SARSSL_MIX16 = bf16 gradients, fp16 saved activations)
#define DW_DISPATCH_GA(dtype, stmt)                                                                       \
    do {                                                                                                  \
        if ((dtype) == SARSSL_BF16) { typedef bf16 T; typedef bf16 TA; stmt; }                            \
        else if ((dtype) == SARSSL_F32) { typedef float T; typedef float TA; stmt; }                      \
        else if ((dtype) == SARSSL_MIX16) { typedef bf16 T; typedef f16 TA; stmt; }                       \
        else { sarssl_set_error("dwconv: unsupported dtype %d", (int)(dtype)); return -1; }              \
    } while (0)

// h: [nb*Tn][2d] pointwise-conv output; w: f32 [d][31]; c: [nb*Tn][d] (pre-BatchNorm); sums: optional f64[2d] = per-channel
// sum | sum of squares of c over all nb*Tn frames (zeroed here).  d % 8 == 0.
extern "C" int sarssl_dwglu_fwd(const void* h, const float* w, int nb, int Tn, int d, int ksize, void* c, double* sums, int dtype,
                                void* stream) {
    SARSSL_REQUIRE(ksize == DWK && nb > 0 && Tn > 0 && d > 0 && d % 8 == 0, "sarssl_dwglu_fwd(kernel size 31, d % 8 == 0)");
    if (sums && SARSSL_ZERO(sums, 2L * d * sizeof(double), ST) != hipSuccess) { sarssl_set_error("sarssl_dwglu_fwd: memset"); return -2; }
    dim3 grid((d + DTC - 1) / DTC, (Tn + DTT - 1) / DTT, nb);
    DW_DISPATCH(dtype, (dwglu_fwd_kernel<T><<<grid, 256, 0, ST>>>((const T*)h, w, Tn, d, (T*)c, sums)));
    SARSSL_CHECK_LAUNCH("dwglu_fwd_kernel");
    return 0;
}
// dc: [nb*Tn][d] gradient w.r.t. the depthwise-conv output; dh: [nb*Tn][2d] gradient w.r.t. the pointwise-conv output
extern "C" int sarssl_dwglu_bwd(const void* dc, const void* h, const float* w, int nb, int Tn, int d, int ksize, void* dh, int dtype,
                                void* stream) {
    SARSSL_REQUIRE(ksize == DWK && nb > 0 && Tn > 0 && d > 0 && d % 8 == 0, "sarssl_dwglu_bwd(kernel size 31, d % 8 == 0)");
    dim3 grid((d + DTC - 1) / DTC, (Tn + DTT - 1) / DTT, nb);
    DW_DISPATCH_GA(dtype, (dwglu_bwd_kernel<T, TA><<<grid, 256, 0, ST>>>((const T*)dc, (const TA*)h, w, Tn, d, (T*)dh)));
    SARSSL_CHECK_LAUNCH("dwglu_bwd_kernel");
    return 0;
}
static inline int dwglu_parts(int nb, int Tn) {
    const int ntile = nb * ((Tn + DTT - 1) / DTT);
    return ntile > 64 ? 64 : ntile;
}
extern "C" long sarssl_dwglu_wgrad_workspace_bytes(int nb, int Tn, int d) { return (long)dwglu_parts(nb, Tn) * d * DWK * sizeof(float); }
// dw: f32 [d][31] += sum dc * g  (g = GLU(h) recomputed); partial: workspace of sarssl_dwglu_wgrad_workspace_bytes
extern "C" int sarssl_dwglu_wgrad(const void* dc, const void* h, int nb, int Tn, int d, int ksize, float* dw, float* partial, int dtype,
                                  void* stream) {
    SARSSL_REQUIRE(ksize == DWK && partial && d % 8 == 0, "sarssl_dwglu_wgrad(kernel size 31, d % 8 == 0)");
    const int parts = dwglu_parts(nb, Tn);
    dim3 grid((d + DTC - 1) / DTC, parts);
    DW_DISPATCH_GA(dtype, (dwglu_wgrad_kernel<T, TA><<<grid, 256, 0, ST>>>((const T*)dc, (const TA*)h, nb, Tn, d, partial)));
    const long n = (long)d * DWK;
    if (dw) dw_partial_reduce_kernel<<<(int)((n + 63) / 64), 256, 0, ST>>>(partial, parts, n, dw);    // (dw == null: the caller folds the partials)
    SARSSL_CHECK_LAUNCH("dwglu_wgrad_kernel");
    return 0;
}
