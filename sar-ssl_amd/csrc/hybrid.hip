// Kernels of the "hybrid" numeric mode: fp16 CNN stem, f32 residual stream in the Conformer blocks and the decoder.
//
// In this mode the tensors that CARRY VALUES FROM LAYER TO LAYER - the residual stream x [M, d], the LayerNorm outputs that feed the
// Linear layers, the prediction - are f32, while everything that is only ever a matrix-core operand or a module-internal tensor (feed-forward
// hidden activations, q / k / v, the convolution module's inner tensors, the decoder's hidden layer) stays 16-bit as in the fp16-forward
// mode.  An f32 tensor enters a matrix-core product as an fp16 PAIR (hi = fp16(x), lo = fp16(x - hi): 22 significant bits), written by
// the kernel that produces it: sarssl_gemm_split (csrc/gemm.hip) contracts hi hi + lo hi + hi lo as a three-segment K loop.  Why:
// oracle/operand_rounding_study.py - the fp16-forward mode's per-bin deviation from the reference (1.2e-3 of range in train mode) is the
// rounding of stored activations and of single-pass operands in the Conformer / decoder; this layout brings it to ~6e-4.
//
//   layernorm fwd -> fp16 pair                  code/common/conformer/*.py LayerNorm uses (feed_forward.py:48, attention.py:146, ...)
//   layernorm fwd x 2 (block boundary)          code/common/Conformer.py:88-90 + feed_forward.py:48
//   layernorm bwd, f32 stream gradient          the same layers' backward; branch gradient (dy) bf16 or f32, dropped copy bf16
//   f32 -> fp16 pair / lo part                  weights (runtime.FlatParams.wl16), stem outputs
#include "common.h"

#define ST ((hipStream_t)stream)
#define LN_MAXV 4
static inline int nblocks_for(long work, int per_block, int cap = 4096) {
    long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

// hi / lo halves of four values, two 8-byte stores
__device__ __forceinline__ void st4_pair(f16* __restrict__ hi, f16* __restrict__ lo, const float4 o) {
    uint2 h, l;
    h.x = pack2_f16(o.x, o.y); h.y = pack2_f16(o.z, o.w);
    const sarssl_f32x2 a = unpack2_f16(h.x), b = unpack2_f16(h.y);
    l.x = pack2_f16(o.x - a.x, o.y - a.y); l.y = pack2_f16(o.z - b.x, o.w - b.y);
    *(uint2*)hi = h;
    if (lo) *(uint2*)lo = l;                // (lo == null: the consumer contracts the hi half only - a Linear layer whose OUTPUT is an fp16 tensor)
}

// ---- LayerNorm forward, f32 rows -> fp16 pair (the arithmetic of layernorm_fwd_kernel<float>, csrc/elementwise.hip; the f32 result is
// split instead of stored).  y32 (optional): the f32 result as well.
__global__ __launch_bounds__(256) void layernorm_fwd_pair_kernel(const float* __restrict__ x, long ldx, long M, int d,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                 f16* __restrict__ yhi, f16* __restrict__ ylo, long ldy,
                                                                 float* __restrict__ y32, long ldy32,
                                                                 float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < nv) { v[i] = *(const float4*)(x + row * ldx + c4 * 4); s += v[i].x + v[i].y + v[i].z + v[i].w; }
        }
        const float mu = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < nv) {
                const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, e = v[i].w - mu;
                q += a * a + b * b + c * c + e * e;
            }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < nv) {
                const float4 g = *(const float4*)(gamma + c4 * 4), bb = *(const float4*)(beta + c4 * 4);
                const float4 o = make_float4((v[i].x - mu) * rs * g.x + bb.x, (v[i].y - mu) * rs * g.y + bb.y,
                                             (v[i].z - mu) * rs * g.z + bb.z, (v[i].w - mu) * rs * g.w + bb.w);
                st4_pair(yhi + row * ldy + c4 * 4, ylo ? ylo + row * ldy + c4 * 4 : nullptr, o);
                if (y32) *(float4*)(y32 + row * ldy32 + c4 * 4) = o;
            }
        }
        if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    }
}

// Several rows per wave for d = 256 / 512 (four rows of 16 lanes / two rows of 32 lanes: 2 KB in flight per wave instead of 512 B-1 KB -
// the layout of layernorm_fwd_rows_kernel, csrc/elementwise.hip, with the same summation order as the one-row kernel above)
template <int D>
__global__ __launch_bounds__(256) void layernorm_fwd_pair_rows_kernel(const float* __restrict__ x, long ldx, long M, int d,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                      f16* __restrict__ yhi, f16* __restrict__ ylo, long ldy,
                                                                      float* __restrict__ mean, float* __restrict__ rstd) {
    constexpr int NI = D / 256, RPW = 4 / NI, LPR = 64 / RPW, J = RPW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane % LPR, slot = lane / LPR;
    for (long row0 = ((long)blockIdx.x * 4 + wave) * RPW; row0 < M; row0 += (long)gridDim.x * 4 * RPW) {
        const long row = row0 + slot;
        const bool ok = row < M;
        float4 v[J][NI];
        float p[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c4 = lr + LPR * j + 64 * i;
                v[j][i] = ok ? *(const float4*)(x + row * ldx + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                s += v[j][i].x + v[j][i].y + v[j][i].z + v[j][i].w;
            }
            p[j] = s;
        }
        auto fold = [&](float (&q)[J]) -> float {
#pragma unroll
            for (int m = J / 2; m > 0; m >>= 1) {
                float t[J];
#pragma unroll
                for (int j = 0; j < J; ++j) t[j] = q[j] + q[j ^ m];
#pragma unroll
                for (int j = 0; j < J; ++j) q[j] = t[j];
            }
            float r = q[0];
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
            return r;
        };
        const float mu = fold(p) / (float)d;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float a = v[j][i].x - mu, b = v[j][i].y - mu, c = v[j][i].z - mu, e = v[j][i].w - mu;
                // four products, three additions, no fused multiply-add - the arithmetic hipcc makes of the one-row kernel's sum of
                // squares (layernorm_fwd_rows_kernel, csrc/elementwise.hip): the pair is the split of sarssl_layernorm_fwd's f32 result
                float a2 = a * a, b2 = b * b, c2 = c * c, e2 = e * e;
                asm volatile("" : "+v"(a2), "+v"(b2), "+v"(c2), "+v"(e2));
                q = q + (((a2 + b2) + c2) + e2);
            }
            p[j] = q;
        }
        const float rs = rsqrtf(fold(p) / (float)d + eps);
        if (ok) {
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int c4 = lr + LPR * j + 64 * i;
                    const float4 g = *(const float4*)(gamma + c4 * 4), bb = *(const float4*)(beta + c4 * 4);
                    st4_pair(yhi + row * ldy + c4 * 4, ylo ? ylo + row * ldy + c4 * 4 : nullptr,
                             make_float4((v[j][i].x - mu) * rs * g.x + bb.x, (v[j][i].y - mu) * rs * g.y + bb.y,
                                         (v[j][i].z - mu) * rs * g.z + bb.z, (v[j][i].w - mu) * rs * g.w + bb.w));
                }
            if (lr == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
        }
    }
}

// y = LN_a(x) stored in f32 (the residual stream leaving a Conformer block), z = LN_b(y) as a pair (the next block's first LayerNorm)
__global__ __launch_bounds__(256) void layernorm_fwd2_pair_kernel(const float* __restrict__ x, long ldx, long M, int d,
                                                                  const float* __restrict__ ga, const float* __restrict__ ba, float epsa,
                                                                  float* __restrict__ y, long ldy, float* __restrict__ meana, float* __restrict__ rstda,
                                                                  const float* __restrict__ gb, const float* __restrict__ bb, float epsb,
                                                                  f16* __restrict__ zhi, f16* __restrict__ zlo, long ldz,
                                                                  float* __restrict__ meanb, float* __restrict__ rstdb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
        float4 v[LN_MAXV];
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < nv) v[i] = *(const float4*)(x + row * ldx + c4 * 4);
        }
#pragma unroll
        for (int stage = 0; stage < 2; ++stage) {
            const float* gamma = stage ? gb : ga; const float* beta = stage ? bb : ba;
            const float eps = stage ? epsb : epsa;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) { const int c4 = lane + i * 64; if (c4 < nv) s += v[i].x + v[i].y + v[i].z + v[i].w; }
            const float mu = wave_sum(s) / (float)d;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) {
                const int c4 = lane + i * 64;
                if (c4 < nv) {
                    const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, e = v[i].w - mu;
                    q += a * a + b * b + c * c + e * e;
                }
            }
            const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) {
                const int c4 = lane + i * 64;
                if (c4 < nv) {
                    const float4 g = *(const float4*)(gamma + c4 * 4), be = *(const float4*)(beta + c4 * 4);
                    const float4 o = make_float4((v[i].x - mu) * rs * g.x + be.x, (v[i].y - mu) * rs * g.y + be.y,
                                                 (v[i].z - mu) * rs * g.z + be.z, (v[i].w - mu) * rs * g.w + be.w);
                    if (stage == 0) { *(float4*)(y + row * ldy + c4 * 4) = o; v[i] = o; }
                    else st4_pair(zhi + row * ldz + c4 * 4, zlo ? zlo + row * ldz + c4 * 4 : nullptr, o);
                }
            }
            float* mean = stage ? meanb : meana; float* rstd = stage ? rstdb : rstda;
            if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
        }
    }
}

// ---- LayerNorm backward on the f32 stream: dx = rstd (dy g - mean(dy g) - xhat mean(dy g xhat)) + resid, x / resid / dx f32; dy (the
// gradient arriving from the branch: a GEMM result) bf16 or f32; dx2 (optional, contiguous [M][d], bf16) = dx * dropout_mask(seed) * gscale
// - the matrix-core operand of the next module of the backward chain (its dropout backward applied; p = 0: a plain bf16 copy).
// The arithmetic of layernorm_bwd_kernel (csrc/elementwise.hip); partial [gridDim.x][2][d] = per-workgroup dgamma | dbeta sums.
template <typename TD, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_stream_kernel(const TD* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                                   long M, int d, const float* __restrict__ gamma,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ resid, long ldr, float* __restrict__ dx, long lddx,
                                                                   float* __restrict__ partial, bf16* __restrict__ dx2, float p_drop,
                                                                   unsigned long long seed0, const unsigned long long* __restrict__ salt, float gscale) {
    const unsigned long long seed = salted_seed(seed0, salt);
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    __shared__ float4 sg[4][64 * NV];
    __shared__ float4 sb[4][64 * NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;
    constexpr int R = 2;
    float4 ag[NV], ab[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0);
        const int c4 = lane + i * 64;
        gm[i] = (c4 < nv) ? *(const float4*)(gamma + c4 * 4) : make_float4(0, 0, 0, 0);
    }
    const long rstride = (long)gridDim.x * 4;
    for (long row0 = (long)blockIdx.x * 4 + wave; row0 < M; row0 += rstride * R) {
        float4 dyv[R][NV], xv[R][NV], rv[R][NV];
        float mu[R], rs[R];
        bool ok[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const long row = row0 + u * rstride;
            ok[u] = row < M;
            if (ok[u]) {
                mu[u] = mean[row]; rs[u] = rstd[row];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c4 = lane + i * 64;
                    if (c4 < nv) {
                        dyv[u][i] = ld4(dy + row * lddy + c4 * 4);
                        xv[u][i] = *(const float4*)(x + row * ldx + c4 * 4);
                        if (resid) rv[u][i] = *(const float4*)(resid + row * ldr + c4 * 4);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (!ok[u]) continue;
            const long row = row0 + u * rstride;
            float4 g[NV], xh[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c4 = lane + i * 64;
                if (c4 < nv) {
                    const float4 a = dyv[u][i], xx = xv[u][i];
                    xh[i] = make_float4((xx.x - mu[u]) * rs[u], (xx.y - mu[u]) * rs[u], (xx.z - mu[u]) * rs[u], (xx.w - mu[u]) * rs[u]);
                    g[i] = make_float4(a.x * gm[i].x, a.y * gm[i].y, a.z * gm[i].z, a.w * gm[i].w);
                    s1 += g[i].x + g[i].y + g[i].z + g[i].w;
                    s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
                    ag[i].x += a.x * xh[i].x; ag[i].y += a.y * xh[i].y; ag[i].z += a.z * xh[i].z; ag[i].w += a.w * xh[i].w;
                    ab[i].x += a.x; ab[i].y += a.y; ab[i].z += a.z; ab[i].w += a.w;
                }
            }
            s1 = wave_sum(s1) / (float)d; s2 = wave_sum(s2) / (float)d;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c4 = lane + i * 64;
                if (c4 < nv) {
                    float4 o = make_float4(rs[u] * (g[i].x - s1 - xh[i].x * s2), rs[u] * (g[i].y - s1 - xh[i].y * s2),
                                           rs[u] * (g[i].z - s1 - xh[i].z * s2), rs[u] * (g[i].w - s1 - xh[i].w * s2));
                    if (resid) { o.x += rv[u][i].x; o.y += rv[u][i].y; o.z += rv[u][i].z; o.w += rv[u][i].w; }
                    *(float4*)(dx + row * lddx + c4 * 4) = o;
                    if (dx2) {
                        const unsigned long long idx = (unsigned long long)row * d + c4 * 4;
                        float4 q = o;
                        if (p_drop > 0.f) {
                            float kp[4];
                            dropout_scale4(seed, idx, p_drop, inv_keep, kp);
                            q.x *= kp[0]; q.y *= kp[1]; q.z *= kp[2]; q.w *= kp[3];
                        }
                        q.x *= gscale; q.y *= gscale; q.z *= gscale; q.w *= gscale;
                        st4(dx2 + row * d + c4 * 4, q);
                    }
                }
            }
        }
    }
    if (partial) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { sg[wave][lane + i * 64] = ag[i]; sb[wave][lane + i * 64] = ab[i]; }
        __syncthreads();
        float* P = partial + (long)blockIdx.x * 2 * d;
        for (int c4 = threadIdx.x; c4 < nv; c4 += 256) {
            float4 a = sg[0][c4], b = sb[0][c4];
            for (int w = 1; w < 4; ++w) {
                const float4 a2 = sg[w][c4], b2 = sb[w][c4];
                a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w; b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
            }
            *(float4*)(P + c4 * 4) = a;
            *(float4*)(P + d + c4 * 4) = b;
        }
    }
}

// ---- f32 -> fp16 pair (hi may be null: the lo part only - the weights' second shadow next to the fp16 one the Adam kernel writes);
// TS = float, or f16 / bf16 sources (lo = exact remainder of the fp16 encoding: zero for fp16 sources)
template <typename TS>
__global__ void split_pair_kernel(const TS* __restrict__ s, long n4, f16* __restrict__ hi, f16* __restrict__ lo) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = ld4(s + i * 4);
        uint2 h, l;
        h.x = pack2_f16(v.x, v.y); h.y = pack2_f16(v.z, v.w);
        const sarssl_f32x2 a = unpack2_f16(h.x), b = unpack2_f16(h.y);
        l.x = pack2_f16(v.x - a.x, v.y - a.y); l.y = pack2_f16(v.z - b.x, v.w - b.y);
        if (hi) *(uint2*)(hi + i * 4) = h;
        *(uint2*)(lo + i * 4) = l;
    }
}

// ---- channels-last per-channel affine + activation on a pair: z = act(scale[c] * (hi + lo) + shift[c]) -> pair (the stem's BatchNorm(4)
// + ReLU in front of the frame-patch product, code/model.py:60-62; act 1 = relu, 2 = swish).  C % 4 == 0 or C == 4.
__global__ void cl_affine_act_pair_kernel(const f16* __restrict__ xhi, const f16* __restrict__ xlo, long n4, int C,
                                          const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                          f16* __restrict__ zhi, f16* __restrict__ zlo) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % C);
        const float4 a = ld4(xhi + i * 4), b = ld4(xlo + i * 4);
        const float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
        float4 z = make_float4(fmaf(a.x + b.x, sc.x, sh.x), fmaf(a.y + b.y, sc.y, sh.y), fmaf(a.z + b.z, sc.z, sh.z), fmaf(a.w + b.w, sc.w, sh.w));
        if (act == 1) z = make_float4(fmaxf(z.x, 0.f), fmaxf(z.y, 0.f), fmaxf(z.z, 0.f), fmaxf(z.w, 0.f));
        else if (act == 2) z = make_float4(z.x * sigmoidf_(z.x), z.y * sigmoidf_(z.y), z.z * sigmoidf_(z.z), z.w * sigmoidf_(z.w));
        st4_pair(zhi + i * 4, zlo + i * 4, z);
    }
}

// ================================================================================================ C ABI
extern "C" int sarssl_cl_affine_act_pair(const void* x_hi, const void* x_lo, long n, int C, const float* scale, const float* shift, int act,
                                         void* z_hi, void* z_lo, void* stream) {
    SARSSL_REQUIRE(n > 0 && C > 0 && (C & 3) == 0 && n % C == 0 && x_hi && x_lo && z_hi && z_lo, "sarssl_cl_affine_act_pair");
    cl_affine_act_pair_kernel<<<nblocks_for(n / 4, 256, 8192), 256, 0, ST>>>((const f16*)x_hi, (const f16*)x_lo, n / 4, C, scale, shift, act,
                                                                             (f16*)z_hi, (f16*)z_lo);
    SARSSL_CHECK_LAUNCH("cl_affine_act_pair_kernel");
    return 0;
}
extern "C" int sarssl_layernorm_fwd_pair(const float* x, long ldx, long M, int d, const float* gamma, const float* beta, float eps,
                                         void* y_hi, void* y_lo, long ldy, float* y32, long ldy32, float* mean, float* rstd, void* stream) {
    SARSSL_REQUIRE(M > 0 && d > 0 && (d & 3) == 0 && d <= 1024 && (ldx & 3) == 0 && (ldy & 3) == 0 && (ldy32 & 3) == 0 && y_hi,
                   "sarssl_layernorm_fwd_pair");          // y_lo may be null: the hi half only
    if (!y32 && (d == 256 || d == 512) && M >= 4096) {
        const int nblk = nblocks_for(M, d == 256 ? 16 : 8, 4096);
        if (d == 256) layernorm_fwd_pair_rows_kernel<256><<<nblk, 256, 0, ST>>>(x, ldx, M, d, gamma, beta, eps, (f16*)y_hi, (f16*)y_lo, ldy, mean, rstd);
        else layernorm_fwd_pair_rows_kernel<512><<<nblk, 256, 0, ST>>>(x, ldx, M, d, gamma, beta, eps, (f16*)y_hi, (f16*)y_lo, ldy, mean, rstd);
        SARSSL_CHECK_LAUNCH("layernorm_fwd_pair_rows_kernel");
        return 0;
    }
    layernorm_fwd_pair_kernel<<<nblocks_for(M, 4, 4096), 256, 0, ST>>>(x, ldx, M, d, gamma, beta, eps, (f16*)y_hi, (f16*)y_lo, ldy, y32, ldy32, mean, rstd);
    SARSSL_CHECK_LAUNCH("layernorm_fwd_pair_kernel");
    return 0;
}
extern "C" int sarssl_layernorm_fwd2_pair(const float* x, long ldx, long M, int d, const float* gamma_a, const float* beta_a, float eps_a,
                                          float* y, long ldy, float* mean_a, float* rstd_a, const float* gamma_b, const float* beta_b,
                                          float eps_b, void* z_hi, void* z_lo, long ldz, float* mean_b, float* rstd_b, void* stream) {
    SARSSL_REQUIRE(M > 0 && d > 0 && (d & 3) == 0 && d <= 1024 && (ldx & 3) == 0 && (ldy & 3) == 0 && (ldz & 3) == 0 && y && z_hi,
                   "sarssl_layernorm_fwd2_pair");         // z_lo may be null
    layernorm_fwd2_pair_kernel<<<nblocks_for(M, 4, 4096), 256, 0, ST>>>(x, ldx, M, d, gamma_a, beta_a, eps_a, y, ldy, mean_a, rstd_a, gamma_b, beta_b,
                                                                     eps_b, (f16*)z_hi, (f16*)z_lo, ldz, mean_b, rstd_b);
    SARSSL_CHECK_LAUNCH("layernorm_fwd2_pair_kernel");
    return 0;
}
// partial: sarssl_layernorm_bwd_workspace_bytes(M, d) bytes ([sarssl_layernorm_bwd_nparts(M)][2][d]); dgamma != null: folded here, else
// the caller folds (sarssl_ln_param_reduce_multi).  dy_dtype: SARSSL_BF16 | SARSSL_F32.  dx2 may be null.
__global__ __launch_bounds__(1024) void ln_param_fold_kernel(const float* __restrict__ partial, int nparts, int d, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
    __shared__ float sred[16][64];
    const int col = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + col;
    float s = 0.f;
    if (c < 2 * d) for (int p = slot; p < nparts; p += 16) s += partial[(long)p * 2 * d + c];
    sred[slot][col] = s;
    __syncthreads();
    if (slot == 0 && c < 2 * d) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += sred[w][col];
        if (c < d) dgamma[c] += t; else dbeta[c - d] += t;
    }
}
extern "C" int sarssl_layernorm_bwd_stream(const void* dy, int dy_dtype, long lddy, const float* x, long ldx, long M, int d, const float* gamma,
                                           const float* mean, const float* rstd, const float* resid, long ldr, float* dx, long lddx,
                                           float* dgamma, float* dbeta, float* partial, void* dx2, float p_drop, unsigned long long seed,
                                           float gscale, void* stream) {
    SARSSL_REQUIRE(M > 0 && d > 0 && (d & 3) == 0 && d <= 1024 && (!dgamma || partial) && (dy_dtype == SARSSL_BF16 || dy_dtype == SARSSL_F32),
                   "sarssl_layernorm_bwd_stream");
    long b = (M + 31) / 32; if (b < 1) b = 1; if (b > 512) b = 512;       // = sarssl_layernorm_bwd_nparts(M)
    const int nblk = (int)b;
    const unsigned long long* salt = sarssl_dropout_salt();
#define LNS(TD, NVv) layernorm_bwd_stream_kernel<TD, NVv><<<nblk, 256, 0, ST>>>((const TD*)dy, lddy, x, ldx, M, d, gamma, mean, rstd, resid, ldr, dx, lddx, \
                                                                              partial, (bf16*)dx2, p_drop, seed, salt, gscale)
    if (dy_dtype == SARSSL_BF16) { if (d <= 256) LNS(bf16, 1); else if (d <= 512) LNS(bf16, 2); else LNS(bf16, 4); }
    else { if (d <= 256) LNS(float, 1); else if (d <= 512) LNS(float, 2); else LNS(float, 4); }
#undef LNS
    if (dgamma) ln_param_fold_kernel<<<(2 * d + 63) / 64, 1024, 0, ST>>>(partial, nblk, d, dgamma, dbeta);
    SARSSL_CHECK_LAUNCH("layernorm_bwd_stream_kernel");
    return 0;
}
// src (f32 | fp16 | bf16, n elements, n % 4 == 0, 16-byte aligned) -> hi = fp16(src) (may be null), lo = fp16(src - hi)
extern "C" int sarssl_split_pair(const void* src, int src_dtype, long n, void* hi, void* lo, void* stream) {
    SARSSL_REQUIRE(n > 0 && (n & 3) == 0 && lo, "sarssl_split_pair");
    const int nblk = nblocks_for(n / 4, 256, 4096);
    if (src_dtype == SARSSL_F32) split_pair_kernel<float><<<nblk, 256, 0, ST>>>((const float*)src, n / 4, (f16*)hi, (f16*)lo);
    else if (src_dtype == SARSSL_F16) split_pair_kernel<f16><<<nblk, 256, 0, ST>>>((const f16*)src, n / 4, (f16*)hi, (f16*)lo);
    else if (src_dtype == SARSSL_BF16) split_pair_kernel<bf16><<<nblk, 256, 0, ST>>>((const bf16*)src, n / 4, (f16*)hi, (f16*)lo);
    else { sarssl_set_error("sarssl_split_pair: unsupported dtype %d", src_dtype); return -1; }
    SARSSL_CHECK_LAUNCH("split_pair_kernel");
    return 0;
}
