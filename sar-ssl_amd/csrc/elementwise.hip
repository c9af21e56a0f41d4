// HBM-bound row / elementwise kernels of the Conformer blocks, decoder loss and optimiser.
//   layernorm fwd/bwd (wave per row, f32 statistics)            code/common/conformer/*.py LayerNorm uses
//   glu fwd/bwd                                                  conformer/activation.py:31-42
//   depthwise conv k=31 fwd / dgrad (flipped) / wgrad            conformer/convolution.py:140
//   relative-shift softmax fwd/bwd + shift-gather                conformer/attention.py:87-113
//   bias2 add (q+u, q+v), axpby, column sums (bias grads), act-backward with dropout mask
//   masked-channel MSE loss fwd/bwd                              code/model.py:585-592, 721-747
//   fused Adam over a flat parameter buffer (+ bf16 shadow)      code/learner.py:83
#include "common.h"

#define ST ((hipStream_t)stream)
#define DISPATCH_T(dtype, CALL)                                                             \
    if (dtype == SARSSL_BF16) { typedef bf16 T; CALL; }                                     \
    else if (dtype == SARSSL_F32) { typedef float T; CALL; }                                \
    else if (dtype == SARSSL_F16) { typedef f16 T; CALL; }                                  \
    else { sarssl_set_error("unsupported dtype %d", dtype); return -1; }
// kernels of the backward pass that read a tensor SAVED BY THE FORWARD PASS (type TA) next to gradients (type T):
// SARSSL_MIX16 = bf16 gradients, fp16 saved activations (common.h)
#define DISPATCH_GA(dtype, CALL)                                                            \
    if (dtype == SARSSL_BF16) { typedef bf16 T; typedef bf16 TA; CALL; }                    \
    else if (dtype == SARSSL_F32) { typedef float T; typedef float TA; CALL; }              \
    else if (dtype == SARSSL_MIX16) { typedef bf16 T; typedef f16 TA; CALL; }               \
    else if (dtype == SARSSL_MIXF32) { typedef bf16 T; typedef float TA; CALL; }            \
    else { sarssl_set_error("unsupported dtype %d", dtype); return -1; }
static inline int nblocks_for(long work, int per_block, int cap = 4096) {
    long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

// ------------------------------------------------------------------------------------ LayerNorm
#define LN_MAXV 4      // up to 4 float4 per lane -> d <= 1024
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, long ldx, long M, int d,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, T* __restrict__ y, long ldy, float* __restrict__ mean,
                                                            float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;                                  // float4 per row
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < nv) { v[i] = ld4(x + row * ldx + c4 * 4); s += v[i].x + v[i].y + v[i].z + v[i].w; }
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
                st4(y + row * ldy + c4 * 4, make_float4((v[i].x - mu) * rs * g.x + bb.x, (v[i].y - mu) * rs * g.y + bb.y,
                                                        (v[i].z - mu) * rs * g.z + bb.z, (v[i].w - mu) * rs * g.w + bb.w));
            }
        }
        if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
    }
}

// The same arithmetic with SEVERAL rows per wave (d = 256: four rows of 16 lanes, d = 512: two rows of 32 lanes).  The kernel above keeps
// one 512-byte row per wave in flight, so a [16384, 256] pass is bound by request round trips (2 TB/s); here a wave has 2 KB in flight and
// the four rows' reductions are ONE set of lane exchanges.  Bit-identical by construction: a lane holds the float4 columns of the virtual
// lanes v = l + LPR * j of the 64-lane layout (column c4 = v + 64 i), forms each virtual lane's partial exactly as above, combines them in
// the order of wave_sum's butterfly (exchange distances 32 .. LPR are additions of its own partials - a + b == b + a - the rest are lane
// exchanges inside the row's LPR lanes).  (Round 4 measured a 16-lane layout with 32-byte requests 9.0 -> 6.8 us but with another summation
// order, which moved the model-level per-bin maxima - noise, but not kept; this one changes no bit.)
template <typename T, int D>
__global__ __launch_bounds__(256) void layernorm_fwd_rows_kernel(const T* __restrict__ x, long ldx, long M, int d /* == D: a run-time
                                                                 divisor keeps the divisions of the one-row kernel (no reciprocal / FMA folding) */,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, T* __restrict__ y, long ldy, float* __restrict__ mean,
                                                                 float* __restrict__ rstd) {
    constexpr int NI = D / 256;                             // float4 columns per virtual lane (c4 = v + 64 i)
    constexpr int RPW = 4 / NI;                             // rows per wave: 4 (d = 256), 2 (d = 512)
    constexpr int LPR = 64 / RPW;                           // lanes per row
    constexpr int J = RPW;                                  // virtual lanes per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane % LPR, slot = lane / LPR;
    float4 g[J][NI], bb[J][NI];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c4 = lr + LPR * j + 64 * i;
            g[j][i] = *(const float4*)(gamma + c4 * 4); bb[j][i] = *(const float4*)(beta + c4 * 4);
        }
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
                v[j][i] = ok ? ld4(x + row * ldx + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                s += v[j][i].x + v[j][i].y + v[j][i].z + v[j][i].w;
            }
            p[j] = s;
        }
        // wave_sum's butterfly: distances 32 .. LPR among the lane's own partials, LPR/2 .. 1 between lanes
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
                // four products, three additions, no fused multiply-add: what hipcc makes of the one-row kernel's `a * a + b * b + c * c + e * e`
                // (packed multiplies); left to -ffp-contract=fast it picks FMAs here and rstd moves by an ulp on 6 % of the rows
                float a2 = a * a, b2 = b * b, c2 = c * c, e2 = e * e;
                asm volatile("" : "+v"(a2), "+v"(b2), "+v"(c2), "+v"(e2));       // (the products as values: nothing to contract with)
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
                    st4(y + row * ldy + c4 * 4, make_float4((v[j][i].x - mu) * rs * g[j][i].x + bb[j][i].x, (v[j][i].y - mu) * rs * g[j][i].y + bb[j][i].y,
                                                            (v[j][i].z - mu) * rs * g[j][i].z + bb[j][i].z, (v[j][i].w - mu) * rs * g[j][i].w + bb[j][i].w));
                }
            if (lr == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
        }
    }
}

// Two LayerNorms in a row on the same data - a Conformer block's closing LayerNorm and the first one of the next block's feed-forward
// module (Conformer.py:88-90, feed_forward.py:48): y = LN_a(x) is stored (rounded to T), z = LN_b(y as stored) - the arithmetic of two
// launches of the kernel above, bit for bit, in one pass over the row.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd2_kernel(const T* __restrict__ x, long ldx, long M, int d,
                                                             const float* __restrict__ ga, const float* __restrict__ ba, float epsa,
                                                             T* __restrict__ y, long ldy, float* __restrict__ meana, float* __restrict__ rstda,
                                                             const float* __restrict__ gb, const float* __restrict__ bb, float epsb,
                                                             T* __restrict__ z, long ldz, float* __restrict__ meanb, float* __restrict__ rstdb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
        float4 v[LN_MAXV];
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c4 = lane + i * 64;
            if (c4 < nv) v[i] = ld4(x + row * ldx + c4 * 4);
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
            T* out = stage ? z : y; const long ldo = stage ? ldz : ldy;
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) {
                const int c4 = lane + i * 64;
                if (c4 < nv) {
                    const float4 g = *(const float4*)(gamma + c4 * 4), be = *(const float4*)(beta + c4 * 4);
                    const float4 o = make_float4((v[i].x - mu) * rs * g.x + be.x, (v[i].y - mu) * rs * g.y + be.y,
                                                 (v[i].z - mu) * rs * g.z + be.z, (v[i].w - mu) * rs * g.w + be.w);
                    st4(out + row * ldo + c4 * 4, o);
                    // the next stage reads the row as it was stored
                    v[i] = make_float4(round_as<T>(o.x), round_as<T>(o.y), round_as<T>(o.z), round_as<T>(o.w));
                }
            }
            float* mean = stage ? meanb : meana; float* rstd = stage ? rstdb : rstda;
            if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
        }
    }
}

// dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) (+ resid);  dgamma += sum dy*xhat, dbeta += sum dy
// Two rows per wave are in flight at a time (loads of both issued before either reduction) to hide HBM latency.
template <typename T, typename TA, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, long lddy, const TA* __restrict__ x, long ldx,
                                                            long M, int d, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const T* __restrict__ resid, long ldr, T* __restrict__ dx, long lddx,
                                                            float* __restrict__ partial /* [gridDim.x][2][d] or null */,
                                                            T* __restrict__ dx2 = nullptr, float p_drop = 0.f, unsigned long long seed0 = 0,
                                                            const unsigned long long* __restrict__ salt = nullptr, float gscale = 1.f) {
    // dx2 (optional, contiguous [M][d]): dx * dropout_mask(seed, m*d + c) * gscale - the dropout backward the NEXT module of the
    // backward chain applies to its incoming gradient (act_bwd), produced here while dx is still in registers
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
                        xv[u][i] = ld4(x + row * ldx + c4 * 4);
                        if (resid) rv[u][i] = ld4(resid + row * ldr + c4 * 4);
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
                    st4(dx + row * lddx + c4 * 4, o);
                    if (dx2) {
                        const unsigned long long idx = (unsigned long long)row * d + c4 * 4;
                        float4 q = o;
                        if (p_drop > 0.f) {             // (d % 4 == 0: idx is a multiple of 4)
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
// out[i] += sum_p partial[p][i], i < n.  Workgroup = 64 columns x 4 part-slots (4-way unrolled loads), LDS combine.
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ partial, int nparts, long n,
                                                             float* __restrict__ out) {
    __shared__ float sred[4][64];
    const int col = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int p = slot;
        for (; p + 12 < nparts; p += 16) {
            s0 += partial[(long)p * n + i]; s1 += partial[(long)(p + 4) * n + i];
            s2 += partial[(long)(p + 8) * n + i]; s3 += partial[(long)(p + 12) * n + i];
        }
        for (; p < nparts; p += 4) s0 += partial[(long)p * n + i];
    }
    sred[slot][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slot == 0 && i < n) out[i] += (sred[0][col] + sred[1][col]) + (sred[2][col] + sred[3][col]);
}

// same with a row pitch: out[i] += sum_p partial[p*pitch + i], i < n
__global__ __launch_bounds__(256) void partial_reduce_strided_kernel(const float* __restrict__ partial, int nparts, long pitch, long n,
                                                                     float* __restrict__ out) {
    __shared__ float sred[4][64];
    const int col = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int p = slot;
        for (; p + 12 < nparts; p += 16) {
            s0 += partial[(long)p * pitch + i]; s1 += partial[(long)(p + 4) * pitch + i];
            s2 += partial[(long)(p + 8) * pitch + i]; s3 += partial[(long)(p + 12) * pitch + i];
        }
        for (; p < nparts; p += 4) s0 += partial[(long)p * pitch + i];
    }
    sred[slot][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slot == 0 && i < n) out[i] += (sred[0][col] + sred[1][col]) + (sred[2][col] + sred[3][col]);
}

// LayerNorm parameter gradients: dgamma[c] += sum_p partial[p][c], dbeta[c] += sum_p partial[p][d + c] in ONE launch
// (partial is [nparts][2d]).  Workgroup = 64 columns x 16 part-slots: with 512 partial rows a thread issues 32 loads (4 in flight)
// instead of the 128 dependent rounds of the 4-slot kernel (12 us per call, 40 calls per step in round 1).
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ partial, int nparts, int d,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sred[16][64];
    const int col = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;                 // column of the [nparts][2d] matrix
    const long pitch = 2L * d;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < 2 * d) {
        int p = slot;
        for (; p + 48 < nparts; p += 64) {
            s0 += partial[(long)p * pitch + i]; s1 += partial[(long)(p + 16) * pitch + i];
            s2 += partial[(long)(p + 32) * pitch + i]; s3 += partial[(long)(p + 48) * pitch + i];
        }
        for (; p < nparts; p += 16) s0 += partial[(long)p * pitch + i];
    }
    sred[slot][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slot == 0 && i < 2 * d) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sred[k][col];
        if (i < d) dgamma[i] += t; else dbeta[i - d] += t;
    }
}

// The same fold for the LayerNorm modules of one Conformer block in ONE launch (5 per block, 20 launches of ~5 us per step otherwise;
// each launch is 8-16 workgroups, so together they also fill more of the chip).
#define LNR_MAXP 8
struct LnReduceMulti {
    const float* partial[LNR_MAXP]; float* dgamma[LNR_MAXP]; float* dbeta[LNR_MAXP];
    int nparts[LNR_MAXP], d[LNR_MAXP], first[LNR_MAXP + 1];
    int n;
};
__global__ __launch_bounds__(1024) void ln_param_reduce_multi_kernel(LnReduceMulti a) {
    __shared__ float sred[16][64];
    int q = 0;
    while (q + 1 < a.n && (int)blockIdx.x >= a.first[q + 1]) ++q;
    const float* __restrict__ partial = a.partial[q];
    const int nparts = a.nparts[q], d = a.d[q];
    const int col = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int i = (blockIdx.x - a.first[q]) * 64 + col;
    const long pitch = 2L * d;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < 2 * d) {
        int p = slot;
        for (; p + 48 < nparts; p += 64) {
            s0 += partial[(long)p * pitch + i]; s1 += partial[(long)(p + 16) * pitch + i];
            s2 += partial[(long)(p + 32) * pitch + i]; s3 += partial[(long)(p + 48) * pitch + i];
        }
        for (; p < nparts; p += 16) s0 += partial[(long)p * pitch + i];
    }
    sred[slot][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slot == 0 && i < 2 * d) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sred[k][col];
        if (i < d) a.dgamma[q][i] += t; else a.dbeta[q][i - d] += t;
    }
}

// ------------------------------------------------------------------------------------ GLU
template <typename T>
__global__ void glu_fwd_kernel(const T* __restrict__ h, long M, int d, T* __restrict__ g) {
    const long total4 = M * (d >> 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / (d >> 2); const int c = (int)(i % (d >> 2)) * 4;
        const float4 a = ld4(h + row * 2 * d + c), b = ld4(h + row * 2 * d + d + c);
        st4(g + row * d + c, make_float4(a.x * sigmoidf_(b.x), a.y * sigmoidf_(b.y), a.z * sigmoidf_(b.z), a.w * sigmoidf_(b.w)));
    }
}
template <typename T, typename TA>
__global__ void glu_bwd_kernel(const T* __restrict__ dg, const TA* __restrict__ h, long M, int d, T* __restrict__ dh) {
    const long total4 = M * (d >> 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / (d >> 2); const int c = (int)(i % (d >> 2)) * 4;
        const float4 a = ld4(h + row * 2 * d + c), b = ld4(h + row * 2 * d + d + c), g = ld4(dg + row * d + c);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float da[4], db[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float s = sigmoidf_(bv[e]); da[e] = gv[e] * s; db[e] = gv[e] * av[e] * s * (1.f - s); }
        st4(dh + row * 2 * d + c, make_float4(da[0], da[1], da[2], da[3]));
        st4(dh + row * 2 * d + d + c, make_float4(db[0], db[1], db[2], db[3]));
    }
}

// ------------------------------------------------------------------------------------ depthwise conv, K = 31
#define DWK 31
#define DWT 32      // outputs per thread per tile
// y[b][t][c] = sum_k w[c][flip ? K-1-k : k] * x[b][t + k - 15][c];  thread = channel, lanes = consecutive channels
template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, int nb, int Tn, int d,
                                                         int flip, T* __restrict__ y) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    float wk[DWK];
#pragma unroll
    for (int k = 0; k < DWK; ++k) wk[k] = w[(long)c * DWK + (flip ? DWK - 1 - k : k)];
    const int ntile = (Tn + DWT - 1) / DWT;
    for (int tile = blockIdx.y; tile < nb * ntile; tile += gridDim.y) {
        const int b = tile / ntile, t0 = (tile % ntile) * DWT;
        float win[DWT + DWK - 1];
#pragma unroll
        for (int i = 0; i < DWT + DWK - 1; ++i) {
            const int t = t0 - 15 + i;
            win[i] = (t >= 0 && t < Tn) ? ld_f(x + ((long)b * Tn + t) * d + c) : 0.f;
        }
#pragma unroll
        for (int o = 0; o < DWT; ++o) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < DWK; ++k) acc += wk[k] * win[o + k];
            if (t0 + o < Tn) st_f(y + ((long)b * Tn + t0 + o) * d + c, acc);
        }
    }
}
// partial[by][c][k] = sum over this workgroup's (b, t-tile) share of dy[b][t][c] * x[b][t + k - 15][c]
// Workgroup = 64 channels x 4 waves; the waves take different (b, t-tile)s (4x the loads in flight of the former one-wave
// workgroups - the kernel is latency-bound on its strided row loads) and are folded through LDS before the partial is written.
template <typename T, typename TA>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ dy, const TA* __restrict__ x, int nb, int Tn, int d,
                                                           float* __restrict__ partial) {
    __shared__ float sacc[3][64][DWK + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < d;
    float acc[DWK];
#pragma unroll
    for (int k = 0; k < DWK; ++k) acc[k] = 0.f;
    const int ntile = (Tn + DWT - 1) / DWT;
    if (live) {
        for (int tile = blockIdx.y * 4 + wv; tile < nb * ntile; tile += gridDim.y * 4) {
            const int b = tile / ntile, t0 = (tile % ntile) * DWT;
            float win[DWT + DWK - 1];
#pragma unroll
            for (int i = 0; i < DWT + DWK - 1; ++i) {
                const int t = t0 - 15 + i;
                win[i] = (t >= 0 && t < Tn) ? ld_f(x + ((long)b * Tn + t) * d + c) : 0.f;
            }
#pragma unroll
            for (int o = 0; o < DWT; ++o) {
                const float g = (t0 + o < Tn) ? ld_f(dy + ((long)b * Tn + t0 + o) * d + c) : 0.f;
#pragma unroll
                for (int k = 0; k < DWK; ++k) acc[k] += g * win[o + k];
            }
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int k = 0; k < DWK; ++k) sacc[wv - 1][lane][k] = acc[k];
    }
    __syncthreads();
    if (wv == 0 && live) {
        float* P = partial + ((long)blockIdx.y * d + c) * DWK;
#pragma unroll
        for (int k = 0; k < DWK; ++k) P[k] = acc[k] + sacc[0][lane][k] + sacc[1][lane][k] + sacc[2][lane][k];
    }
}

// ------------------------------------------------------------------------------------ attention softmax
// score[i][j] = (content[i][j] + shift(pos)[i][j]) * scale,  shift per conformer/attention.py:105-113:
//   j <= i   : pos[i][T-1-i+j]      j == i+1 : 0      j >= i+2 : pos[i+1][j-i-2]
// p = softmax_j(score) (saved for backward);  pd = p * dropout mask (GEMM operand), pd == p when p_drop == 0.
#define SM_MAXV 16      // T <= 1024
template <typename T>
__global__ __launch_bounds__(256) void softmax_relshift_fwd_kernel(const float* __restrict__ content, const float* __restrict__ pos,
                                                                   long nrows_total, int Tn, float scale, T* __restrict__ p,
                                                                   T* __restrict__ pd, float p_drop, unsigned long long seed0,
                                                                   const unsigned long long* __restrict__ salt) {
    const unsigned long long seed = salted_seed(seed0, salt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (long row = (long)blockIdx.x * 4 + wave; row < nrows_total; row += (long)gridDim.x * 4) {
        const int i = (int)(row % Tn);
        const long mat = row / Tn;                       // (b*H + h)
        const float* cr = content + row * Tn;
        const float* pm = pos + mat * Tn * Tn;
        float v[SM_MAXV];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < SM_MAXV; ++k) {
            const int j = lane + k * 64;
            if (j < Tn) {
                float ps;
                if (j <= i) ps = pm[(long)i * Tn + (Tn - 1 - i + j)];
                else if (j == i + 1) ps = 0.f;
                else ps = pm[(long)(i + 1) * Tn + (j - i - 2)];
                v[k] = (cr[j] + ps) * scale;
                mx = fmaxf(mx, v[k]);
            }
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < SM_MAXV; ++k) {
            const int j = lane + k * 64;
            if (j < Tn) { v[k] = __expf(v[k] - mx); s += v[k]; }
        }
        const float inv = 1.f / wave_sum(s);
#pragma unroll
        for (int k = 0; k < SM_MAXV; ++k) {
            const int j = lane + k * 64;
            if (j < Tn) {
                const float pr = v[k] * inv;
                st_f(p + row * Tn + j, pr);
                if (pd != p) st_f(pd + row * Tn + j, pr * dropout_scale(seed, (unsigned long long)(row * Tn + j), p_drop, inv_keep));
            }
        }
    }
}
// dscore = scale * p * (dp - sum_j dp*p),  dp = dpd * dropout mask
template <typename T, typename TA>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dpd, const TA* __restrict__ p, long nrows_total,
                                                          int Tn, float scale, float p_drop, unsigned long long seed0,
                                                          T* __restrict__ dscore, const unsigned long long* __restrict__ salt) {
    const unsigned long long seed = salted_seed(seed0, salt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (long row = (long)blockIdx.x * 4 + wave; row < nrows_total; row += (long)gridDim.x * 4) {
        float dp[SM_MAXV], pr[SM_MAXV];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < SM_MAXV; ++k) {
            const int j = lane + k * 64;
            if (j < Tn) {
                pr[k] = ld_f(p + row * Tn + j);
                dp[k] = dpd[row * Tn + j];
                if (p_drop > 0.f) dp[k] *= dropout_scale(seed, (unsigned long long)(row * Tn + j), p_drop, inv_keep);
                s += dp[k] * pr[k];
            }
        }
        s = wave_sum(s);
#pragma unroll
        for (int k = 0; k < SM_MAXV; ++k) {
            const int j = lane + k * 64;
            if (j < Tn) st_f(dscore + row * Tn + j, scale * pr[k] * (dp[k] - s));
        }
    }
}
// dpos[r][m] gathered from dscore through the inverse of the relative shift.  In the flat (T x T) layout the shift is a per-row offset:
// dpos_flat[r T + m] = dscore_flat[r T + m - (T - 1 - r)] (both branches of attention.py:105-113's inverse), zero where that falls before
// the matrix (row 0, m < T - 1).
template <typename T>
__global__ void relshift_bwd_kernel(const T* __restrict__ dscore, long nmat, int Tn, T* __restrict__ dpos) {
    const long total = nmat * Tn * Tn;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx % Tn);
        const long rr = idx / Tn;
        const int r = (int)(rr % Tn);
        const long mat = rr / Tn;
        float v = 0.f;
        if (m >= Tn - 1 - r) v = ld_f(dscore + (mat * Tn + r) * Tn + (m - Tn + 1 + r));
        else if (r >= 1 && m <= Tn - 2 - r) v = ld_f(dscore + (mat * Tn + r - 1) * Tn + (m + r + 1));
        st_f(dpos + idx, v);
    }
}
// bf16, T % 8 == 0: a thread owns 8 consecutive outputs of one row (one 16-byte store; the 8 source elements are 2-byte loads at the
// row's offset - neighbouring lanes share their cache lines).  The element-per-thread kernel above spent its time on three 64-bit
// divisions per element: 33 us per layer at (B, H, T) = (64, 4, 256).
__global__ __launch_bounds__(256) void relshift_bwd8_kernel(const uint16_t* __restrict__ dscore, long nrows, int Tn, uint16_t* __restrict__ dpos) {
    const unsigned cpr = (unsigned)Tn >> 3;
    const long nchunk = nrows * cpr;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nchunk; q += (long)gridDim.x * blockDim.x) {
        const long row = q / cpr;                              // mat * T + r
        const int c8 = (int)(q - row * cpr) << 3;
        const int r = (int)(row % Tn);
        const long mat0 = (row - r) * Tn;                      // first element of this matrix
        const long src = row * Tn + c8 - (Tn - 1 - r);         // flat source index of output element c8
        const uint16_t* sp = dscore + src;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t lo = (src + 2 * e >= mat0) ? sp[2 * e] : 0u, hi = (src + 2 * e + 1 >= mat0) ? sp[2 * e + 1] : 0u;
            w[e] = lo | (hi << 16);
        }
        *(uint4*)(dpos + row * Tn + c8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ------------------------------------------------------------------------------------ small elementwise
// qu = q + u[col], qv = q + v[col]   (q has row stride ldq; outputs dense [M][d])
template <typename T>
__global__ void bias2_kernel(const T* __restrict__ q, long ldq, long M, int d, const float* __restrict__ u,
                             const float* __restrict__ v, T* __restrict__ qu, T* __restrict__ qv) {
    const long total4 = M * (d >> 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / (d >> 2); const int c = (int)(i % (d >> 2)) * 4;
        const float4 a = ld4(q + row * ldq + c);
        const float4 uu = *(const float4*)(u + c), vv = *(const float4*)(v + c);
        st4(qu + i * 4, make_float4(a.x + uu.x, a.y + uu.y, a.z + uu.z, a.w + uu.w));
        st4(qv + i * 4, make_float4(a.x + vv.x, a.y + vv.y, a.z + vv.z, a.w + vv.w));
    }
}
// out = a*x + b*y  (y may be null)
template <typename T>
__global__ void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y, float a, float b, long n4, T* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 xv = ld4(x + i * 4);
        float4 o = make_float4(a * xv.x, a * xv.y, a * xv.z, a * xv.w);
        if (y) { const float4 yv = ld4(y + i * 4); o.x += b * yv.x; o.y += b * yv.y; o.z += b * yv.z; o.w += b * yv.w; }
        st4(out + i * 4, o);
    }
}
// out[m][n] = a*x[m][n] + b*y[m][n] on row-strided [M][N] views
template <typename T>
__global__ void axpby2d_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ y, long ldy, float a, float b, long M, int N,
                               T* __restrict__ out, long ldo) {
    const long total4 = M * (N >> 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / (N >> 2); const int c = (int)(i % (N >> 2)) * 4;
        const float4 xv = ld4(x + row * ldx + c), yv = ld4(y + row * ldy + c);
        st4(out + row * ldo + c, make_float4(a * xv.x + b * yv.x, a * xv.y + b * yv.y, a * xv.z + b * yv.z, a * xv.w + b * yv.w));
    }
}
// out[n] = sum_m x[m][n] written in the activation dtype (few rows, many columns: the batch sum of the positional-score gradient that
// feeds the positional projection's weight-gradient GEMM - was memset + colsum_kernel + cast_kernel).  One workgroup per 64 columns.
template <typename T>
__global__ void colsum_store_kernel(const T* __restrict__ x, long ldx, long M, int N, T* __restrict__ out) {
    __shared__ float sred[256][5];
    const int cgp = threadIdx.x & 15, rslot = threadIdx.x >> 4;
    const int col0 = blockIdx.x * 64, col = col0 + cgp * 4;
    float s[4] = {0, 0, 0, 0};
    if (col < N) {
        for (long m = rslot; m < M; m += 16) {
            const float4 v = ld4(x + m * ldx + col);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sred[threadIdx.x][e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64 && col0 + threadIdx.x < N) {
        const int c = threadIdx.x;
        float acc = 0.f;
        for (int r = 0; r < 16; ++r) acc += sred[r * 16 + (c >> 2)][c & 3];
        st_f(out + col0 + c, acc);
    }
}
// Several independent column sums in ONE launch (the ~46 bias-gradient reductions of a backward pass were 8 us launches each):
// problem q: part_q[row slice][n] = sum over the slice's rows of x_q[m][n].  Workgroups are numbered through the problems'
// (column tile, row slice) grids.  Round 3: the slices' sums are WRITTEN (round 2 added them to the gradient with f32 atomics, in
// arrival order - bias gradients differed in the last bits from run to run); sarssl_splitk_reduce_multi folds them in slice order.
#define COLSUM_MAXP 24
struct ColsumMulti {
    const void* x[COLSUM_MAXP]; long ld[COLSUM_MAXP]; long M[COLSUM_MAXP]; int N[COLSUM_MAXP]; float* out[COLSUM_MAXP];
    int gx[COLSUM_MAXP], gy[COLSUM_MAXP], first[COLSUM_MAXP + 1];
    int n;
};
template <typename T>
__global__ void colsum_multi_kernel(ColsumMulti a) {
    __shared__ float sred[256][5];
    int q = 0;
    while (q + 1 < a.n && (int)blockIdx.x >= a.first[q + 1]) ++q;
    const int local = blockIdx.x - a.first[q];
    const int bx = local % a.gx[q], by = local / a.gx[q];
    const T* x = (const T*)a.x[q];
    const long ldx = a.ld[q], M = a.M[q];
    const int N = a.N[q];
    const int cgp = threadIdx.x & 15, rslot = threadIdx.x >> 4;
    const int col0 = bx * 64, col = col0 + cgp * 4;
    float s[4] = {0, 0, 0, 0};
    if (col < N) {
        for (long m = (long)by * 16 + rslot; m < M; m += (long)a.gy[q] * 16) {
            const float4 v = ld4(x + m * ldx + col);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sred[threadIdx.x][e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64 && col0 + threadIdx.x < N) {
        const int c = threadIdx.x;
        float acc = 0.f;
        for (int r = 0; r < 16; ++r) acc += sred[r * 16 + (c >> 2)][c & 3];
        a.out[q][(long)by * N + col0 + c] = acc;
    }
}
// dh = dz * act'(h) * dropout_mask(seed, idx) * gscale
//   act 1: relu, h_is_post = 1 means h holds relu output;  act 2: swish with h = pre-activation;  act 0: none
template <typename T, typename TA>
__global__ void act_bwd_kernel(const T* __restrict__ dz, const TA* __restrict__ h, long n, int act, float p_drop,
                               unsigned long long seed0, float gscale, T* __restrict__ dh, const unsigned long long* __restrict__ salt) {
    const unsigned long long seed = salted_seed(seed0, salt);
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += (long)gridDim.x * blockDim.x) {
        const float4 d = ld4(dz + i * 4);
        float dv[4] = {d.x, d.y, d.z, d.w}, hv[4] = {0, 0, 0, 0};
        if (act) { const float4 hh = ld4(h + i * 4); hv[0] = hh.x; hv[1] = hh.y; hv[2] = hh.z; hv[3] = hh.w; }
        float kp[4] = {1.f, 1.f, 1.f, 1.f};
        if (p_drop > 0.f) dropout_scale4(seed, (unsigned long long)(i * 4), p_drop, inv_keep, kp);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float g = dv[e] * gscale * kp[e];
            if (act == 1) g = hv[e] > 0.f ? g : 0.f;
            else if (act == 2) { const float s = sigmoidf_(hv[e]); g *= s * (1.f + hv[e] * (1.f - s)); }
            dv[e] = g;
        }
        st4(dh + i * 4, make_float4(dv[0], dv[1], dv[2], dv[3]));
    }
}
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, long n, TD* __restrict__ d) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) st_f(d + i, ld_f(s + i));
}
__global__ void f64_accum_kernel(const double* __restrict__ s, float* __restrict__ d, int n, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] += (float)(s[i]) * scale;
}
// two destinations in one launch: d1[i] += s[i], d2[i] += s[n + i]  (BatchNorm dbeta | dgamma from the backward sums)
__global__ void f64_accum2_kernel(const double* __restrict__ s, float* __restrict__ d1, float* __restrict__ d2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d1[i] += (float)s[i];
    else if (i < 2 * n) d2[i - n] += (float)s[i];
}

// ------------------------------------------------------------------------------------ loss
// pred: (B, T, F, 2, 2) [f][reim][mic];  x: (B, 2, F, T, 2) f32;  idx: (B, nm) int32 masked frames; mch: (B) int32
// sums[0] += sum (pred_mch - x_mch)^2 ; sums[1] += sum (x_mch - x_other)^2   over masked frames
// Workgroup = (signal b, 8 consecutive frames).  pred is frame-major ([b][t][f][re/im][mic]) and x is frequency-major
// ([b][mic][f][t][re/im]): the masked pred rows of the group are staged in LDS (contiguous 16-byte reads), then x is walked in
// its own order (64 contiguous bytes per f) - both sides coalesced.  Partial sums go to 64 slots x 2 quantities.
#define MSE_TT 8
#define MSE_SLOTS 64
// dpred (optional; TG = gradient dtype): the loss gradient 2 (pred - tar) / count of the same entries, zero elsewhere, written for all
// eight frames of the workgroup's group - what masked_mse_bwd_kernel computes with an incoming gradient of 1 (the captured step's
// backward starts right behind this launch; a separate pass re-read pred and x from a stretch of the step nothing overlaps)
// COMPACT: pred / dpred hold the MASKED frames only, [nb][nm][F * 4] with an item's rows in ascending frame order (the decoder ran on
// the gathered rows, sarssl_gather_rows): frame t of item b lives in row (number of masked frames of b below t).
template <typename T, typename TG, bool COMPACT = false>
__global__ __launch_bounds__(256) void masked_mse_fwd_kernel(const T* __restrict__ pred, const float* __restrict__ x,
                                                             const int* __restrict__ idx, const int* __restrict__ mch, int nb,
                                                             int F, int Tn, int nm, double* __restrict__ sums,
                                                             TG* __restrict__ dpred = nullptr, float coef = 0.f) {
    extern __shared__ float sp[];                       // [MSE_TT][F * 4]
    __shared__ unsigned smask;
    __shared__ int sbelow;
    __shared__ float red[2][4];
    const int groups = (Tn + MSE_TT - 1) / MSE_TT;
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int t0 = g * MSE_TT, mc = mch[b];
    if (threadIdx.x == 0) { smask = 0u; sbelow = 0; }
    __syncthreads();
    for (int k = threadIdx.x; k < nm; k += 256) {
        const int t = idx[(long)b * nm + k];
        if (t >= t0 && t < t0 + MSE_TT) atomicOr(&smask, 1u << (t - t0));
        if (COMPACT && t < t0) atomicAdd(&sbelow, 1);
    }
    __syncthreads();
    const unsigned mask = smask;
    const int row = F * 4;
    // row of frame t0 + tl in pred / dpred
    auto prow = [&](int tl) -> long {
        return COMPACT ? (long)b * nm + sbelow + __builtin_popcount(mask & ((1u << tl) - 1u)) : (long)b * Tn + t0 + tl;
    };
    if (dpred && !COMPACT) {                            // frames of the group that are not masked: zero gradient rows
        for (int tl = 0; tl < MSE_TT; ++tl) {
            if (((mask >> tl) & 1u) || t0 + tl >= Tn) continue;
            TG* dst = dpred + ((long)b * Tn + t0 + tl) * row;
            for (int e = threadIdx.x; e < F; e += 256) st4(dst + e * 4, make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    if (mask == 0u) return;
    for (int tl = 0; tl < MSE_TT; ++tl) {
        if (!((mask >> tl) & 1u)) continue;
        const T* src = pred + prow(tl) * row;
        for (int e = threadIdx.x; e < F; e += 256) {
            *(float4*)&sp[tl * row + e * 4] = ld4(src + e * 4);
        }
    }
    __syncthreads();
    float l = 0.f, dsum = 0.f;
    const float* xt = x + (((long)b * 2 + mc) * F) * Tn * 2;
    const float* xo = x + (((long)b * 2 + (1 - mc)) * F) * Tn * 2;
    for (int i = threadIdx.x; i < F * MSE_TT; i += 256) {
        const int tl = i & (MSE_TT - 1), f = i >> 3;
        if (!((mask >> tl) & 1u)) continue;
        const long off = ((long)f * Tn + t0 + tl) * 2;
        const float2 tar = *(const float2*)(xt + off), oth = *(const float2*)(xo + off);
        const float p0 = sp[tl * row + f * 4 + mc], p1 = sp[tl * row + f * 4 + 2 + mc];
        l += (p0 - tar.x) * (p0 - tar.x) + (p1 - tar.y) * (p1 - tar.y);
        dsum += (tar.x - oth.x) * (tar.x - oth.x) + (tar.y - oth.y) * (tar.y - oth.y);
        if (dpred) {                                    // the staged row becomes the gradient row (each thread owns its four entries)
            float* q = &sp[tl * row + f * 4];
            q[mc] = coef * (p0 - tar.x); q[2 + mc] = coef * (p1 - tar.y);
            q[1 - mc] = 0.f; q[3 - mc] = 0.f;
        }
    }
    if (dpred) {
        __syncthreads();
        for (int tl = 0; tl < MSE_TT; ++tl) {
            if (!((mask >> tl) & 1u)) continue;
            TG* dst = dpred + prow(tl) * row;
            for (int e = threadIdx.x; e < F; e += 256) st4(dst + e * 4, *(const float4*)&sp[tl * row + e * 4]);
        }
    }
    l = wave_sum(l); dsum = wave_sum(dsum);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = l; red[1][wave] = dsum; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int q = threadIdx.x;
        atomicAdd(&sums[q * MSE_SLOTS + (blockIdx.x & (MSE_SLOTS - 1))], (double)(red[q][0] + red[q][1] + red[q][2] + red[q][3]));
    }
}
// out_keep (optional f32[2]) receives a copy, acc (optional f64[2]) accumulates the two values (the captured step's running sums: three
// copy / cast / add launches of ~4 us each sat between the loss and its backward, in a stretch of the step nothing overlaps)
// ovf (optional): the context's fp16-overflow word (common.h) - set means the forward pass encoded a value outside fp16's range: the loss
// is reported as NaN (what the step's optimizer guard and the caller see) and the word is cleared for the next step.
__global__ void loss_finalize_kernel(const double* __restrict__ sums, double count, float* __restrict__ out, float* __restrict__ out_keep,
                                     double* __restrict__ acc, int* __restrict__ ovf) {
    const int lane = threadIdx.x;                        // 64 threads
    double a = sums[lane], c = sums[MSE_SLOTS + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
    if (lane == 0) {
        float l = (float)(a / count);
        const float d = (float)(c / count);
        if (ovf && ovf[0]) { l = __builtin_nanf(""); ovf[0] = 0; }
        out[0] = l; out[1] = d;
        if (out_keep) { out_keep[0] = l; out_keep[1] = d; }
        if (acc && isfinite(l)) { acc[0] += (double)l; acc[1] += (double)d; }      // (a skipped step - non-finite loss - is not part of the epoch mean)
    }
}
// dpred = gscale * 2 (pred - tar) / count on (masked frame, masked channel) entries, 0 elsewhere
template <typename T, typename TA>
__global__ void masked_mse_bwd_kernel(const TA* __restrict__ pred, const float* __restrict__ x, const uint8_t* __restrict__ mp,
                                      const int* __restrict__ mch, int nb, int F, int Tn, float coef0,
                                      const float* __restrict__ gs_dev, T* __restrict__ dpred) {
    const float coef = gs_dev ? coef0 * gs_dev[0] : coef0;        // upstream d(loss) kept on the device: no host sync
    const long total = (long)nb * Tn * F * 2;          // one thread per (b,t,f,reim): 2 mics = 2 consecutive outputs
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i & 1);
        long q = i >> 1;
        const int f = (int)(q % F); q /= F;
        const int t = (int)(q % Tn);
        const int b = (int)(q / Tn);
        float g0 = 0.f, g1 = 0.f;
        if (!mp[(long)b * Tn + t]) {
            const int mc = mch[b];
            const float pv = ld_f(pred + i * 2 + mc);
            const float tar = x[((((long)b * 2 + mc) * F + f) * Tn + t) * 2 + r];
            const float g = coef * (pv - tar);
            if (mc == 0) g0 = g; else g1 = g;
        }
        st_f(dpred + i * 2, g0); st_f(dpred + i * 2 + 1, g1);
    }
}

// ------------------------------------------------------------------------------------ Adam
// torch.optim.Adam (no amsgrad, no weight decay) on a flat f32 buffer; also refreshes the bf16 shadow copy.
// guard (optional): the step's loss on the device.  Not finite -> the whole update is skipped (parameters, moments and shadow copies keep
// their values, *nskipped counts it): what torch.cuda.amp.GradScaler.step does for the reference's fp16 autocast path when the
// forward overflowed (code/learner.py:105-108) - decided on the device, no host synchronisation.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            bf16* __restrict__ p16, f16* __restrict__ ph16, long n, float gscale, float beta1, float beta2, float step_size,
                            float inv_bc2_sqrt, float eps, const float* __restrict__ guard, int* __restrict__ nskipped,
                            f16* __restrict__ pl16 = nullptr /* hybrid mode: fp16(p - fp16(p)), the lo shadow */) {
    if (guard && !isfinite(guard[0])) {
        if (nskipped && blockIdx.x == 0 && threadIdx.x == 0) nskipped[0] += 1;
        return;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        const float pi = p[i] - step_size * mi / denom;
        p[i] = pi;
        if (p16) st_f(p16 + i, pi);
        if (ph16) st_f(ph16 + i, pi);
        if (pl16) st_f(pl16 + i, pi - (float)(f16)pi);
    }
}

// Same update with the step-dependent factors read from the device-resident step state (graph replay), optionally clearing the
// gradient buffer in the same pass (the reference's optimizer.zero_grad() right after optimizer.step(), code/learner.py:113-115).
__global__ void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                bf16* __restrict__ p16, f16* __restrict__ ph16, long n, float gscale, SarsslStepState* __restrict__ st, float eps,
                                int zero_g, const float* __restrict__ guard, f16* __restrict__ pl16 = nullptr) {
    const float beta1 = st->beta1, beta2 = st->beta2, step_size = st->step_size, inv_bc2_sqrt = st->inv_bc2_sqrt;
    if (guard && !isfinite(guard[0])) {
        // skipped step: nothing moves, the gradient buffer is still cleared (zero_grad follows the step), the Adam step count is taken
        // back so that the next tick recomputes this step's bias corrections (GradScaler: a skipped step is not an optimizer step)
        if (zero_g)
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) g[i] = 0.f;
        if (blockIdx.x == 0 && threadIdx.x == 0) { st->step -= 1; st->nskipped += 1; }
        return;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        const float pi = p[i] - step_size * mi / denom;
        p[i] = pi;
        if (p16) st_f(p16 + i, pi);
        if (ph16) st_f(ph16 + i, pi);
        if (pl16) st_f(pl16 + i, pi - (float)(f16)pi);
        if (zero_g) g[i] = 0.f;
    }
}

// ================================================================================================ C ABI
extern "C" int sarssl_layernorm_fwd(const void* x, long ldx, long M, int d, const float* gamma, const float* beta, float eps,
                                    void* y, long ldy, float* mean, float* rstd, int dtype, void* stream) {
    SARSSL_REQUIRE(M > 0 && d > 0 && (d & 3) == 0 && d <= 1024 && (ldx & 3) == 0 && (ldy & 3) == 0, "sarssl_layernorm_fwd");
    static const int rows_kernel = [] { const char* e = getenv("SARSSL_LN_ROWS"); return (e && atoi(e) == 0) ? 0 : 1; }();   // 0: one row per wave (A/B runs)
    if (rows_kernel && (d == 256 || d == 512) && M >= 4096) {
        const int nblk = nblocks_for(M, d == 256 ? 16 : 8, 4096);
        if (d == 256) { DISPATCH_T(dtype, (layernorm_fwd_rows_kernel<T, 256><<<nblk, 256, 0, ST>>>((const T*)x, ldx, M, d, gamma, beta, eps, (T*)y, ldy, mean, rstd))); }
        else { DISPATCH_T(dtype, (layernorm_fwd_rows_kernel<T, 512><<<nblk, 256, 0, ST>>>((const T*)x, ldx, M, d, gamma, beta, eps, (T*)y, ldy, mean, rstd))); }
        SARSSL_CHECK_LAUNCH("layernorm_fwd_rows_kernel");
        return 0;
    }
    const int nblk = nblocks_for(M, 4, 4096);
    DISPATCH_T(dtype, (layernorm_fwd_kernel<T><<<nblk, 256, 0, ST>>>((const T*)x, ldx, M, d, gamma, beta, eps, (T*)y, ldy, mean, rstd)));
    SARSSL_CHECK_LAUNCH("layernorm_fwd_kernel");
    return 0;
}
// y = LN_a(x), z = LN_b(y) in one launch (bit-identical to two sarssl_layernorm_fwd launches); statistics of both are saved
extern "C" int sarssl_layernorm_fwd2(const void* x, long ldx, long M, int d, const float* gamma_a, const float* beta_a, float eps_a,
                                     void* y, long ldy, float* mean_a, float* rstd_a, const float* gamma_b, const float* beta_b,
                                     float eps_b, void* z, long ldz, float* mean_b, float* rstd_b, int dtype, void* stream) {
    SARSSL_REQUIRE(M > 0 && d > 0 && (d & 3) == 0 && d <= 1024 && (ldx & 3) == 0 && (ldy & 3) == 0 && (ldz & 3) == 0, "sarssl_layernorm_fwd2");
    const int nblk = nblocks_for(M, 4, 4096);
    DISPATCH_T(dtype, (layernorm_fwd2_kernel<T><<<nblk, 256, 0, ST>>>((const T*)x, ldx, M, d, gamma_a, beta_a, eps_a, (T*)y, ldy, mean_a, rstd_a,
                                                                     gamma_b, beta_b, eps_b, (T*)z, ldz, mean_b, rstd_b)));
    SARSSL_CHECK_LAUNCH("layernorm_fwd2_kernel");
    return 0;
}
static inline int ln_bwd_blocks(long M) { return nblocks_for(M, 4 * 8, 512); }
extern "C" long sarssl_layernorm_bwd_workspace_bytes(long M, int d) {
    return (long)ln_bwd_blocks(M) * 2 * d * sizeof(float);
}
// dgamma/dbeta accumulate (+=) through per-workgroup partials in `partial` (sarssl_layernorm_bwd_workspace_bytes) and a second
// reduce kernel; pass dgamma = null to skip the parameter gradients.
extern "C" int sarssl_layernorm_bwd(const void* dy, long lddy, const void* x, long ldx, long M, int d, const float* gamma,
                                    const float* mean, const float* rstd, const void* resid, long ldr, void* dx, long lddx,
                                    float* dgamma, float* dbeta, float* partial, int dtype, void* stream) {
    SARSSL_REQUIRE(M > 0 && d > 0 && (d & 3) == 0 && d <= 1024 && (!dgamma || partial), "sarssl_layernorm_bwd");
    const int nblk = ln_bwd_blocks(M);
    float* part = partial;          // dgamma == null && partial != null: partials only, folded later (sarssl_ln_param_reduce_multi)
#define LN_BWD_LAUNCH(NVv) layernorm_bwd_kernel<T, TA, NVv><<<nblk, 256, 0, ST>>>((const T*)dy, lddy, (const TA*)x, ldx, M, d, gamma, mean, rstd, \
                                                                           (const T*)resid, ldr, (T*)dx, lddx, part)
    DISPATCH_GA(dtype, (d <= 256 ? LN_BWD_LAUNCH(1) : (d <= 512 ? LN_BWD_LAUNCH(2) : LN_BWD_LAUNCH(4))));
#undef LN_BWD_LAUNCH
    if (dgamma) {       // partial is [nblk][2][d]: viewed as nblk rows of 2d, column halves go to dgamma / dbeta
        ln_param_reduce_kernel<<<(2 * d + 63) / 64, 1024, 0, ST>>>(part, nblk, d, dgamma, dbeta);
    }
    SARSSL_CHECK_LAUNCH("layernorm_bwd_kernel");
    return 0;
}
// The same with a second output dx2 [M][d] (contiguous) = dx * dropout_mask(seed, m*d + c) * gscale: the dropout backward (sarssl_act_bwd,
// act = 0) the next module of the backward chain would run on dx as its first step.
extern "C" int sarssl_layernorm_bwd_drop(const void* dy, long lddy, const void* x, long ldx, long M, int d, const float* gamma,
                                         const float* mean, const float* rstd, const void* resid, long ldr, void* dx, long lddx,
                                         float* dgamma, float* dbeta, float* partial, void* dx2, float p_drop, unsigned long long seed,
                                         float gscale, int dtype, void* stream) {
    SARSSL_REQUIRE(M > 0 && d > 0 && (d & 3) == 0 && d <= 1024 && (!dgamma || partial) && dx2, "sarssl_layernorm_bwd_drop");
    const int nblk = ln_bwd_blocks(M);
    float* part = partial;
    const unsigned long long* salt = sarssl_dropout_salt();
#define LN_BWD_LAUNCH(NVv) layernorm_bwd_kernel<T, TA, NVv><<<nblk, 256, 0, ST>>>((const T*)dy, lddy, (const TA*)x, ldx, M, d, gamma, mean, rstd, \
                                                                           (const T*)resid, ldr, (T*)dx, lddx, part, (T*)dx2, p_drop, seed, salt, gscale)
    DISPATCH_GA(dtype, (d <= 256 ? LN_BWD_LAUNCH(1) : (d <= 512 ? LN_BWD_LAUNCH(2) : LN_BWD_LAUNCH(4))));
#undef LN_BWD_LAUNCH
    if (dgamma) ln_param_reduce_kernel<<<(2 * d + 63) / 64, 1024, 0, ST>>>(part, nblk, d, dgamma, dbeta);
    SARSSL_CHECK_LAUNCH("layernorm_bwd_kernel(drop)");
    return 0;
}
extern "C" int sarssl_layernorm_bwd_nparts(long M) { return ln_bwd_blocks(M); }
// dgamma_q / dbeta_q += column halves of partial_q [nparts_q][2 d_q] for n_prob <= 8 LayerNorm backward launches that wrote partials only
extern "C" int sarssl_ln_param_reduce_multi(const float* const* partial, const int* nparts, const int* d, float* const* dgamma,
                                            float* const* dbeta, int n_prob, void* stream) {
    SARSSL_REQUIRE(n_prob > 0 && n_prob <= LNR_MAXP, "sarssl_ln_param_reduce_multi");
    LnReduceMulti a;
    a.n = n_prob;
    int total = 0;
    for (int q = 0; q < n_prob; ++q) {
        a.partial[q] = partial[q]; a.dgamma[q] = dgamma[q]; a.dbeta[q] = dbeta[q]; a.nparts[q] = nparts[q]; a.d[q] = d[q];
        a.first[q] = total; total += (2 * d[q] + 63) / 64;
    }
    a.first[n_prob] = total;
    ln_param_reduce_multi_kernel<<<total, 1024, 0, ST>>>(a);
    SARSSL_CHECK_LAUNCH("ln_param_reduce_multi_kernel");
    return 0;
}
extern "C" int sarssl_glu_fwd(const void* h, long M, int d, void* g, int dtype, void* stream) {
    SARSSL_REQUIRE((d & 3) == 0, "sarssl_glu_fwd");
    DISPATCH_T(dtype, (glu_fwd_kernel<T><<<nblocks_for(M * (d >> 2), 256), 256, 0, ST>>>((const T*)h, M, d, (T*)g)));
    SARSSL_CHECK_LAUNCH("glu_fwd_kernel");
    return 0;
}
extern "C" int sarssl_glu_bwd(const void* dg, const void* h, long M, int d, void* dh, int dtype, void* stream) {
    SARSSL_REQUIRE((d & 3) == 0, "sarssl_glu_bwd");
    DISPATCH_GA(dtype, (glu_bwd_kernel<T, TA><<<nblocks_for(M * (d >> 2), 256), 256, 0, ST>>>((const T*)dg, (const TA*)h, M, d, (T*)dh)));
    SARSSL_CHECK_LAUNCH("glu_bwd_kernel");
    return 0;
}
// w: f32 [d][31].  flip = 0 forward, flip = 1 data gradient.
extern "C" int sarssl_dwconv_fwd(const void* x, const float* w, int nb, int Tn, int d, int ksize, int flip, void* y, int dtype,
                                 void* stream) {
    SARSSL_REQUIRE(ksize == DWK, "sarssl_dwconv_fwd(kernel size must be 31)");
    const int ntile = nb * ((Tn + DWT - 1) / DWT);
    dim3 grid((d + 255) / 256, ntile > 4096 ? 4096 : ntile);
    DISPATCH_T(dtype, (dwconv_fwd_kernel<T><<<grid, 256, 0, ST>>>((const T*)x, w, nb, Tn, d, flip, (T*)y)));
    SARSSL_CHECK_LAUNCH("dwconv_fwd_kernel");
    return 0;
}
static inline int dwconv_wgrad_parts(int nb, int Tn) {
    const int ntile = nb * ((Tn + DWT - 1) / DWT);
    return ntile > 128 ? 128 : ntile;
}
extern "C" long sarssl_dwconv_wgrad_workspace_bytes(int nb, int Tn, int d) {
    return (long)dwconv_wgrad_parts(nb, Tn) * d * DWK * sizeof(float);
}
// dw: f32 [d][31], accumulated (+=) from per-workgroup partials (workspace of sarssl_dwconv_wgrad_workspace_bytes)
extern "C" int sarssl_dwconv_wgrad(const void* dy, const void* x, int nb, int Tn, int d, int ksize, float* dw, float* partial,
                                   int dtype, void* stream) {
    SARSSL_REQUIRE(ksize == DWK && partial, "sarssl_dwconv_wgrad(kernel size must be 31)");
    const int parts = dwconv_wgrad_parts(nb, Tn);
    dim3 grid((d + 63) / 64, parts);
    DISPATCH_GA(dtype, (dwconv_wgrad_kernel<T, TA><<<grid, 256, 0, ST>>>((const T*)dy, (const TA*)x, nb, Tn, d, partial)));
    const long n = (long)d * DWK;
    partial_reduce_kernel<<<(int)((n + 63) / 64), 256, 0, ST>>>(partial, parts, n, dw);
    SARSSL_CHECK_LAUNCH("dwconv_wgrad_kernel");
    return 0;
}
// content/pos: f32 (nmat, T, T).  p, pd: dtype (nmat, T, T); pass pd == p when p_drop == 0.
extern "C" int sarssl_softmax_relshift_fwd(const float* content, const float* pos, long nmat, int Tn, float scale, void* p,
                                           void* pd, float p_drop, unsigned long long seed, int dtype, void* stream) {
    SARSSL_REQUIRE(Tn > 0 && Tn <= 64 * SM_MAXV, "sarssl_softmax_relshift_fwd(T <= 1024)");
    const int nblk = nblocks_for(nmat * Tn, 4, 8192);
    DISPATCH_T(dtype, (softmax_relshift_fwd_kernel<T><<<nblk, 256, 0, ST>>>(content, pos, nmat * Tn, Tn, scale, (T*)p, (T*)pd, p_drop, seed, sarssl_dropout_salt())));
    SARSSL_CHECK_LAUNCH("softmax_relshift_fwd_kernel");
    return 0;
}
extern "C" int sarssl_softmax_bwd(const float* dpd, const void* p, long nmat, int Tn, float scale, float p_drop,
                                  unsigned long long seed, void* dscore, int dtype, void* stream) {
    SARSSL_REQUIRE(Tn > 0 && Tn <= 64 * SM_MAXV, "sarssl_softmax_bwd(T <= 1024)");
    const int nblk = nblocks_for(nmat * Tn, 4, 8192);
    DISPATCH_GA(dtype, (softmax_bwd_kernel<T, TA><<<nblk, 256, 0, ST>>>(dpd, (const TA*)p, nmat * Tn, Tn, scale, p_drop, seed, (T*)dscore, sarssl_dropout_salt())));
    SARSSL_CHECK_LAUNCH("softmax_bwd_kernel");
    return 0;
}
extern "C" int sarssl_relshift_bwd(const void* dscore, long nmat, int Tn, void* dpos, int dtype, void* stream) {
    if ((dtype == SARSSL_BF16 || dtype == SARSSL_F16) && (Tn & 7) == 0) {      // (16-bit elements are moved, not decoded)
        const long nrows = nmat * Tn;
        relshift_bwd8_kernel<<<nblocks_for(nrows * (Tn >> 3), 256, 8192), 256, 0, ST>>>((const uint16_t*)dscore, nrows, Tn, (uint16_t*)dpos);
        SARSSL_CHECK_LAUNCH("relshift_bwd8_kernel");
        return 0;
    }
    DISPATCH_T(dtype, (relshift_bwd_kernel<T><<<nblocks_for(nmat * Tn * Tn, 256, 8192), 256, 0, ST>>>((const T*)dscore, nmat, Tn, (T*)dpos)));
    SARSSL_CHECK_LAUNCH("relshift_bwd_kernel");
    return 0;
}
extern "C" int sarssl_bias2(const void* q, long ldq, long M, int d, const float* u, const float* v, void* qu, void* qv, int dtype,
                            void* stream) {
    SARSSL_REQUIRE((d & 3) == 0 && (ldq & 3) == 0, "sarssl_bias2");
    DISPATCH_T(dtype, (bias2_kernel<T><<<nblocks_for(M * (d >> 2), 256), 256, 0, ST>>>((const T*)q, ldq, M, d, u, v, (T*)qu, (T*)qv)));
    SARSSL_CHECK_LAUNCH("bias2_kernel");
    return 0;
}
extern "C" int sarssl_axpby(const void* x, const void* y, float a, float b, long n, void* out, int dtype, void* stream) {
    SARSSL_REQUIRE((n & 3) == 0, "sarssl_axpby(n % 4)");
    DISPATCH_T(dtype, (axpby_kernel<T><<<nblocks_for(n >> 2, 256), 256, 0, ST>>>((const T*)x, (const T*)y, a, b, n >> 2, (T*)out)));
    SARSSL_CHECK_LAUNCH("axpby_kernel");
    return 0;
}
extern "C" int sarssl_axpby2d(const void* x, long ldx, const void* y, long ldy, float a, float b, long M, int N, void* out, long ldo,
                              int dtype, void* stream) {
    SARSSL_REQUIRE((N & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && (ldo & 3) == 0, "sarssl_axpby2d");
    DISPATCH_T(dtype, (axpby2d_kernel<T><<<nblocks_for(M * (N >> 2), 256), 256, 0, ST>>>((const T*)x, ldx, (const T*)y, ldy, a, b, M, N, (T*)out, ldo)));
    SARSSL_CHECK_LAUNCH("axpby2d_kernel");
    return 0;
}
// out[n] (same dtype as x) = sum_m x[m][n]; N % 4 == 0
extern "C" int sarssl_colsum_store(const void* x, long ldx, long M, int N, void* out, int dtype, void* stream) {
    SARSSL_REQUIRE((N & 3) == 0 && (ldx & 3) == 0 && M > 0, "sarssl_colsum_store(shape)");
    DISPATCH_T(dtype, (colsum_store_kernel<T><<<(N + 63) / 64, 256, 0, ST>>>((const T*)x, ldx, M, N, (T*)out)));
    SARSSL_CHECK_LAUNCH("colsum_store_kernel");
    return 0;
}
// Row slices the column sums of an M x N problem are split into (the caller sizes the partial buffer [slices][N] with it)
static int colsum_slices(long M, int N) {
    const int gx = (N + 63) / 64;
    long gy = (M + 255) / 256; if (gy < 1) gy = 1;
    long cap = 512 / gx; if (cap < 1) cap = 1;
    if (gy > cap) gy = cap;
    return (int)gy;
}
extern "C" int sarssl_colsum_slices(long M, int N) { return colsum_slices(M, N); }
// n <= 24 problems: xs[q] (dtype, row stride ldxs[q], Ms[q] x Ns[q], Ns[q] % 4 == 0) -> parts[q][sarssl_colsum_slices(Ms[q], Ns[q])][Ns[q]]
// per-slice column sums (f32, every element written); fold them with sarssl_splitk_reduce_multi (M = 1, nsplit = slices)
extern "C" int sarssl_colsum_multi_partials(const void* const* xs, const long* ldxs, const long* Ms, const int* Ns, float* const* parts, int n,
                                            int dtype, void* stream) {
    SARSSL_REQUIRE(n > 0 && n <= COLSUM_MAXP, "sarssl_colsum_multi_partials");
    ColsumMulti a;
    a.n = n;
    int total = 0;
    for (int q = 0; q < n; ++q) {
        SARSSL_REQUIRE((Ns[q] & 3) == 0 && (ldxs[q] & 3) == 0 && Ms[q] > 0, "sarssl_colsum_multi_partials(shape)");
        a.x[q] = xs[q]; a.ld[q] = ldxs[q]; a.M[q] = Ms[q]; a.N[q] = Ns[q]; a.out[q] = parts[q];
        const int gx = (Ns[q] + 63) / 64;
        a.gx[q] = gx; a.gy[q] = colsum_slices(Ms[q], Ns[q]); a.first[q] = total;
        total += gx * a.gy[q];
    }
    a.first[n] = total;
    DISPATCH_T(dtype, (colsum_multi_kernel<T><<<total, 256, 0, ST>>>(a)));
    SARSSL_CHECK_LAUNCH("colsum_multi_kernel");
    return 0;
}
extern "C" int sarssl_act_bwd(const void* dz, const void* h, long n, int act, float p_drop, unsigned long long seed, float gscale,
                              void* dh, int dtype, void* stream) {
    SARSSL_REQUIRE((n & 3) == 0, "sarssl_act_bwd(n % 4)");
    DISPATCH_GA(dtype, (act_bwd_kernel<T, TA><<<nblocks_for(n >> 2, 256), 256, 0, ST>>>((const T*)dz, (const TA*)h, n, act, p_drop, seed, gscale, (T*)dh, sarssl_dropout_salt())));
    SARSSL_CHECK_LAUNCH("act_bwd_kernel");
    return 0;
}
extern "C" int sarssl_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream) {
    const int nblk = nblocks_for(n, 256);
    if (src_dtype == SARSSL_F32 && dst_dtype == SARSSL_BF16) cast_kernel<float, bf16><<<nblk, 256, 0, ST>>>((const float*)src, n, (bf16*)dst);
    else if (src_dtype == SARSSL_BF16 && dst_dtype == SARSSL_F32) cast_kernel<bf16, float><<<nblk, 256, 0, ST>>>((const bf16*)src, n, (float*)dst);
    else if (src_dtype == SARSSL_F32 && dst_dtype == SARSSL_F32) cast_kernel<float, float><<<nblk, 256, 0, ST>>>((const float*)src, n, (float*)dst);
    else if (src_dtype == SARSSL_F32 && dst_dtype == SARSSL_F16) cast_kernel<float, f16><<<nblk, 256, 0, ST>>>((const float*)src, n, (f16*)dst);
    else if (src_dtype == SARSSL_F16 && dst_dtype == SARSSL_F32) cast_kernel<f16, float><<<nblk, 256, 0, ST>>>((const f16*)src, n, (float*)dst);
    else if (src_dtype == SARSSL_F16 && dst_dtype == SARSSL_BF16) cast_kernel<f16, bf16><<<nblk, 256, 0, ST>>>((const f16*)src, n, (bf16*)dst);
    else if (src_dtype == SARSSL_BF16 && dst_dtype == SARSSL_F16) cast_kernel<bf16, f16><<<nblk, 256, 0, ST>>>((const bf16*)src, n, (f16*)dst);
    else { sarssl_set_error("sarssl_cast: unsupported (%d -> %d)", src_dtype, dst_dtype); return -1; }
    SARSSL_CHECK_LAUNCH("cast_kernel");
    return 0;
}
extern "C" int sarssl_f64_accum2(const double* src, float* dst1, float* dst2, int n, void* stream) {
    f64_accum2_kernel<<<(2 * n + 255) / 256, 256, 0, ST>>>(src, dst1, dst2, n);
    SARSSL_CHECK_LAUNCH("f64_accum2_kernel");
    return 0;
}
extern "C" int sarssl_f64_accum(const double* src, float* dst, int n, float scale, void* stream) {
    f64_accum_kernel<<<(n + 255) / 256, 256, 0, ST>>>(src, dst, n, scale);
    SARSSL_CHECK_LAUNCH("f64_accum_kernel");
    return 0;
}
// out: f32[2] = (loss, diff).  sums: f64[128] workspace (zeroed here).
static int masked_mse_fwd_impl(const void* pred, const float* x, const int* idx, const int* mch, int nb, int F, int Tn,
                               int nm, double* sums, float* out, float* out_keep, double* acc, int dtype, void* stream,
                               void* dpred = nullptr, bool compact = false) {
    const size_t lds = (size_t)MSE_TT * F * 4 * sizeof(float);
    SARSSL_REQUIRE(nb > 0 && nm > 0 && lds <= 60 * 1024, "sarssl_masked_mse_fwd (F <= 480)");
    const int groups = (Tn + MSE_TT - 1) / MSE_TT;
    if (SARSSL_ZERO(sums, 2 * MSE_SLOTS * sizeof(double), ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    if (compact) {
        const float coef = 2.0f / (float)((double)nb * nm * F * 2);
        if (dpred) { DISPATCH_GA(dtype, (masked_mse_fwd_kernel<TA, T, true><<<nb * groups, 256, lds, ST>>>((const TA*)pred, x, idx, mch, nb, F, Tn, nm, sums, (T*)dpred, coef))); }
        else { DISPATCH_T(dtype, (masked_mse_fwd_kernel<T, T, true><<<nb * groups, 256, lds, ST>>>((const T*)pred, x, idx, mch, nb, F, Tn, nm, sums))); }
    } else if (dpred) {
        const float coef = 2.0f / (float)((double)nb * nm * F * 2);
        DISPATCH_GA(dtype, (masked_mse_fwd_kernel<TA, T><<<nb * groups, 256, lds, ST>>>((const TA*)pred, x, idx, mch, nb, F, Tn, nm, sums, (T*)dpred, coef)));
    } else {
        DISPATCH_T(dtype, (masked_mse_fwd_kernel<T, T><<<nb * groups, 256, lds, ST>>>((const T*)pred, x, idx, mch, nb, F, Tn, nm, sums)));
    }
    loss_finalize_kernel<<<1, 64, 0, ST>>>(sums, (double)nb * nm * F * 2, out, out_keep, acc, sarssl_overflow_flag());
    SARSSL_CHECK_LAUNCH("masked_mse_fwd_kernel");
    return 0;
}
extern "C" int sarssl_masked_mse_fwd(const void* pred, const float* x, const int* idx, const int* mch, int nb, int F, int Tn,
                                     int nm, double* sums, float* out, int dtype, void* stream) {
    return masked_mse_fwd_impl(pred, x, idx, mch, nb, F, Tn, nm, sums, out, nullptr, nullptr, dtype, stream);
}
// The same; the finalize launch also copies (loss, diff) to out_keep (f32[2]) and adds them to acc (f64[2]) - either may be null.
extern "C" int sarssl_masked_mse_fwd_acc(const void* pred, const float* x, const int* idx, const int* mch, int nb, int F, int Tn,
                                         int nm, double* sums, float* out, float* out_keep, double* acc, int dtype, void* stream) {
    return masked_mse_fwd_impl(pred, x, idx, mch, nb, F, Tn, nm, sums, out, out_keep, acc, dtype, stream);
}
// Loss forward AND its gradient w.r.t. pred for an incoming gradient of 1 in one pass over pred / x (gen_loss, code/model.py:585-592, and
// its backward): dpred (nb, Tn, F*4) of the gradient dtype = what sarssl_masked_mse_bwd(gscale = 1) writes.  dtype: bf16 | f32 | mixed 16.
extern "C" int sarssl_masked_mse_fwd_bwd(const void* pred, const float* x, const int* idx, const int* mch, int nb, int F, int Tn,
                                         int nm, double* sums, float* out, float* out_keep, double* acc, void* dpred, int dtype,
                                         void* stream) {
    SARSSL_REQUIRE(dpred != nullptr, "sarssl_masked_mse_fwd_bwd");
    return masked_mse_fwd_impl(pred, x, idx, mch, nb, F, Tn, nm, sums, out, out_keep, acc, dtype, stream, dpred);
}
// ---- the decoder on the masked frames only.  gen_loss (code/model.py:721-747) reads the prediction at the masked frames of the masked
// channel and nowhere else, and the decoder (Linear - ReLU - Linear, code/model.py:296-301) acts on every frame independently: rows of
// unmasked frames receive a zero gradient and contribute nothing to any parameter gradient.  So the training step gathers the nm masked
// frames of every item (ascending frame order = ascending idx), runs the decoder forward / backward on [nb * nm] rows - half of them -
// and scatters the input gradient back (zeros elsewhere).  Exact, not an approximation; the full prediction (vis) is formed on request.
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, long lds_, const int* __restrict__ idx, int nb, int Tn, int nm, int d,
                                   T* __restrict__ dst) {
    const int cpr = d >> 3;                                   // 16-byte chunks per row
    const long total = (long)nb * nm * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        const long r = i / cpr;                               // b * nm + j
        const int b = (int)(r / nm);
        const int t = idx[r];
        *(uint4*)(dst + r * d + c * 8) = *(const uint4*)(src + ((long)b * Tn + t) * lds_ + c * 8);
    }
}
template <typename T>
__global__ void scatter_rows_kernel(const T* __restrict__ src, const int* __restrict__ idx, int nb, int Tn, int nm, int d,
                                    T* __restrict__ dst, long ldd) {
    const int cpr = d >> 3;
    const long total = (long)nb * nm * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        const long r = i / cpr;
        const int b = (int)(r / nm);
        const int t = idx[r];
        *(uint4*)(dst + ((long)b * Tn + t) * ldd + c * 8) = *(const uint4*)(src + r * d + c * 8);
    }
}
// dst [nb * nm][d] (contiguous) <- rows (b, idx[b][j]) of src [nb * Tn][d] (row stride lds); 16-bit or f32 elements, d % 8 == 0
extern "C" int sarssl_gather_rows(const void* src, long ld_src, const int* idx, int nb, int Tn, int nm, int d, void* dst, int dtype,
                                  void* stream) {
    SARSSL_REQUIRE(nb > 0 && nm > 0 && d > 0 && d % 8 == 0 && ld_src % 8 == 0, "sarssl_gather_rows");
    const int es = dtype == SARSSL_F32 ? 2 : 1;              // f32 rows move as twice as many 16-bit elements
    const int nblk = nblocks_for((long)nb * nm * (d * es / 8), 256, 8192);
    gather_rows_kernel<uint16_t><<<nblk, 256, 0, ST>>>((const uint16_t*)src, ld_src * es, idx, nb, Tn, nm, d * es, (uint16_t*)dst);
    SARSSL_CHECK_LAUNCH("gather_rows_kernel");
    return 0;
}
// dst [nb * Tn][d] (row stride ld_dst; the whole [nb * Tn][d] block is zeroed first) <- src rows (b, j) at rows (b, idx[b][j])
extern "C" int sarssl_scatter_rows(const void* src, const int* idx, int nb, int Tn, int nm, int d, void* dst, long ld_dst, int dtype,
                                   void* stream) {
    SARSSL_REQUIRE(nb > 0 && nm > 0 && d > 0 && d % 8 == 0 && ld_dst == d, "sarssl_scatter_rows (contiguous destination)");
    const int es = dtype == SARSSL_F32 ? 2 : 1;
    if (hipMemsetAsync(dst, 0, (size_t)nb * Tn * d * es * 2, ST) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    const int nblk = nblocks_for((long)nb * nm * (d * es / 8), 256, 8192);
    scatter_rows_kernel<uint16_t><<<nblk, 256, 0, ST>>>((const uint16_t*)src, idx, nb, Tn, nm, d * es, (uint16_t*)dst, ld_dst * es);
    SARSSL_CHECK_LAUNCH("scatter_rows_kernel");
    return 0;
}
// Hybrid mode, block tails on the gathered rows: the f32 stream gradient src [nb * nm][d] goes back to its frames (dst32, zeros elsewhere)
// AND leaves as the bf16 operand of the next module of the backward chain with that module's dropout backward applied (dst16 =
// bf16(bf16(v) * gscale * keep(seed, full-tensor index)), zeros elsewhere) - what sarssl_scatter_rows + sarssl_cast + sarssl_act_bwd
// computed in three passes over the FULL tensor (bit-identical: same roundings in the same order).
__global__ void scatter_rows_drop16_kernel(const float* __restrict__ src, const int* __restrict__ idx, int nb, int Tn, int nm, int d,
                                           float* __restrict__ dst32, bf16* __restrict__ dst16, float p_drop, unsigned long long seed0,
                                           float gscale, const unsigned long long* __restrict__ salt) {
    const unsigned long long seed = salted_seed(seed0, salt);
    const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    const int cpr = d >> 2;
    const long total = (long)nb * nm * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        const long r = i / cpr;
        const int b = (int)(r / nm);
        const long o = ((long)b * Tn + idx[r]) * d + c * 4;
        const float4 v = *(const float4*)(src + r * d + c * 4);
        *(float4*)(dst32 + o) = v;
        float kp[4] = {1.f, 1.f, 1.f, 1.f};
        if (p_drop > 0.f) dropout_scale4(seed, (unsigned long long)o, p_drop, inv_keep, kp);
        st4(dst16 + o, make_float4(round_as<bf16>(v.x) * gscale * kp[0], round_as<bf16>(v.y) * gscale * kp[1],
                                   round_as<bf16>(v.z) * gscale * kp[2], round_as<bf16>(v.w) * gscale * kp[3]));
    }
}
extern "C" int sarssl_scatter_rows_drop16(const float* src, const int* idx, int nb, int Tn, int nm, int d, float* dst32, void* dst16,
                                          float p_drop, unsigned long long seed, float gscale, void* stream) {
    SARSSL_REQUIRE(nb > 0 && nm > 0 && d > 0 && d % 8 == 0 && src && dst32 && dst16, "sarssl_scatter_rows_drop16");
    if (hipMemsetAsync(dst32, 0, (size_t)nb * Tn * d * 4, ST) != hipSuccess || hipMemsetAsync(dst16, 0, (size_t)nb * Tn * d * 2, ST) != hipSuccess) {
        sarssl_set_error("memset");
        return -2;
    }
    scatter_rows_drop16_kernel<<<nblocks_for((long)nb * nm * (d / 4), 256, 8192), 256, 0, ST>>>(src, idx, nb, Tn, nm, d, dst32, (bf16*)dst16, p_drop, seed,
                                                                                            gscale, sarssl_dropout_salt());
    SARSSL_CHECK_LAUNCH("scatter_rows_drop16_kernel");
    return 0;
}
// loss (and, dpred != null, its gradient for an incoming gradient of 1) on the compact prediction pred_c [nb][nm][F * 4]; idx must be
// ascending per item.  dtype as sarssl_masked_mse_fwd (dpred == null) / sarssl_masked_mse_fwd_bwd (dpred != null).
extern "C" int sarssl_masked_mse_compact(const void* pred_c, const float* x, const int* idx, const int* mch, int nb, int F, int Tn, int nm,
                                         double* sums, float* out, float* out_keep, double* acc, void* dpred_c, int dtype, void* stream) {
    return masked_mse_fwd_impl(pred_c, x, idx, mch, nb, F, Tn, nm, sums, out, out_keep, acc, dtype, stream, dpred_c, true);
}
// dpred_c = gscale * (*gscale_dev) * dLoss/dpred_c for a compact prediction (the stand-alone backward of the launch above)
template <typename T, typename TA>
__global__ void masked_mse_bwd_compact_kernel(const TA* __restrict__ pred, const float* __restrict__ x, const int* __restrict__ idx,
                                              const int* __restrict__ mch, int nb, int F, int Tn, int nm, float coef0,
                                              const float* __restrict__ gs_dev, T* __restrict__ dpred) {
    const float coef = gs_dev ? coef0 * gs_dev[0] : coef0;
    const long total = (long)nb * nm * F * 2;                 // one thread per (b, j, f, reim): 2 mics = 2 consecutive outputs
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i & 1);
        long q = i >> 1;
        const int f = (int)(q % F); q /= F;                   // q = b * nm + j
        const int b = (int)(q / nm);
        const int t = idx[q], mc = mch[b];
        const float pv = ld_f(pred + i * 2 + mc);
        const float tar = x[((((long)b * 2 + mc) * F + f) * Tn + t) * 2 + r];
        const float gv = coef * (pv - tar);
        st_f(dpred + i * 2, mc == 0 ? gv : 0.f); st_f(dpred + i * 2 + 1, mc == 0 ? 0.f : gv);
    }
}
extern "C" int sarssl_masked_mse_bwd_compact(const void* pred_c, const float* x, const int* idx, const int* mch, int nb, int F, int Tn,
                                             int nm, float gscale, const float* gscale_dev, void* dpred_c, int dtype, void* stream) {
    const float coef = gscale * 2.0f / (float)((double)nb * nm * F * 2);
    DISPATCH_GA(dtype, (masked_mse_bwd_compact_kernel<T, TA><<<nblocks_for((long)nb * nm * F * 2, 256, 8192), 256, 0, ST>>>((const TA*)pred_c, x, idx, mch, nb, F, Tn, nm, coef, gscale_dev, (T*)dpred_c)));
    SARSSL_CHECK_LAUNCH("masked_mse_bwd_compact_kernel");
    return 0;
}
// dpred = gscale * dLoss/dpred, loss = mean over nb*nm*F*2 entries
extern "C" int sarssl_masked_mse_bwd(const void* pred, const float* x, const unsigned char* mp, const int* mch, int nb, int F,
                                     int Tn, int nm, float gscale, const float* gscale_dev, void* dpred, int dtype,
                                     void* stream) {
    const float coef = gscale * 2.0f / (float)((double)nb * nm * F * 2);
    DISPATCH_GA(dtype, (masked_mse_bwd_kernel<T, TA><<<nblocks_for((long)nb * Tn * F * 2, 256, 8192), 256, 0, ST>>>((const TA*)pred, x, mp, mch, nb, F, Tn, coef, gscale_dev, (T*)dpred)));
    SARSSL_CHECK_LAUNCH("masked_mse_bwd_kernel");
    return 0;
}
// p16 / ph16 (either may be null): bf16 and fp16 shadow copies of the updated parameters (the GEMM / convolution operands)
extern "C" int sarssl_adam_step_guard(float* p, const float* g, float* m, float* v, void* p16, void* ph16, long n, float gscale, float lr,
                                      float beta1, float beta2, float eps, int step, const float* guard, int* nskipped, void* stream) {
    SARSSL_REQUIRE(n > 0 && step >= 1, "sarssl_adam_step");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    adam_kernel<<<nblocks_for(n, 256, 8192), 256, 0, ST>>>(p, g, m, v, (bf16*)p16, (f16*)ph16, n, gscale, beta1, beta2, (float)(lr / bc1),
                                                           (float)(1.0 / sqrt(bc2)), eps, guard, nskipped);
    SARSSL_CHECK_LAUNCH("adam_kernel");
    return 0;
}
// hybrid mode: the same pass also rewrites the fp16 lo shadow pl16 = fp16(p - fp16(p)) (with ph16 the pair a forward product contracts against)
extern "C" int sarssl_adam_step_guard_lo(float* p, const float* g, float* m, float* v, void* p16, void* ph16, void* pl16, long n, float gscale,
                                         float lr, float beta1, float beta2, float eps, int step, const float* guard, int* nskipped, void* stream) {
    SARSSL_REQUIRE(n > 0 && step >= 1, "sarssl_adam_step");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    adam_kernel<<<nblocks_for(n, 256, 8192), 256, 0, ST>>>(p, g, m, v, (bf16*)p16, (f16*)ph16, n, gscale, beta1, beta2, (float)(lr / bc1),
                                                           (float)(1.0 / sqrt(bc2)), eps, guard, nskipped, (f16*)pl16);
    SARSSL_CHECK_LAUNCH("adam_kernel");
    return 0;
}
extern "C" int sarssl_adam_step_dev_guard_lo(float* p, float* g, float* m, float* v, void* p16, void* ph16, void* pl16, long n, float gscale,
                                             void* state, float eps, int zero_grad, const float* guard, void* stream) {
    SARSSL_REQUIRE(n > 0 && state, "sarssl_adam_step_dev");
    adam_dev_kernel<<<nblocks_for(n, 256, 8192), 256, 0, ST>>>(p, g, m, v, (bf16*)p16, (f16*)ph16, n, gscale, (SarsslStepState*)state, eps,
                                                               zero_grad, guard, (f16*)pl16);
    SARSSL_CHECK_LAUNCH("adam_dev_kernel");
    return 0;
}
extern "C" int sarssl_adam_step(float* p, const float* g, float* m, float* v, void* p16, void* ph16, long n, float gscale, float lr,
                                float beta1, float beta2, float eps, int step, void* stream) {
    return sarssl_adam_step_guard(p, g, m, v, p16, ph16, n, gscale, lr, beta1, beta2, eps, step, nullptr, nullptr, stream);
}
extern "C" int sarssl_adam_step_dev_guard(float* p, float* g, float* m, float* v, void* p16, void* ph16, long n, float gscale, void* state,
                                          float eps, int zero_grad, const float* guard, void* stream) {
    SARSSL_REQUIRE(n > 0 && state, "sarssl_adam_step_dev");
    adam_dev_kernel<<<nblocks_for(n, 256, 8192), 256, 0, ST>>>(p, g, m, v, (bf16*)p16, (f16*)ph16, n, gscale, (SarsslStepState*)state, eps,
                                                               zero_grad, guard);
    SARSSL_CHECK_LAUNCH("adam_dev_kernel");
    return 0;
}
extern "C" int sarssl_adam_step_dev(float* p, float* g, float* m, float* v, void* p16, void* ph16, long n, float gscale, const void* state,
                                    float eps, int zero_grad, void* stream) {
    return sarssl_adam_step_dev_guard(p, g, m, v, p16, ph16, n, gscale, const_cast<void*>(state), eps, zero_grad, nullptr, stream);
}
