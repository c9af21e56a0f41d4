// Downstream heads (code/model.py:667-719, 793-821): mean over frames + LayerNorm + one or two small Linear layers on [B, d]
// tensors, B <= a few hundred rows.  Tiny f32 kernels - the point is that no device arithmetic of the downstream path leaves the
// C-ABI library (round-5 verdict), not speed: one wave per output element / one thread per gradient element, f32 throughout.
//   mean over frames fwd / bwd                         embed.mean(dim = 1)
//   small Linear fwd (+ ReLU) / dX / dW, db            nn.Linear(d, n) with n as small as 1 (the MFMA GEMM needs n % 8 == 0)
#include "common.h"
#define ST ((hipStream_t)stream)

template <typename T>
__global__ void mean_rows_kernel(const T* __restrict__ x, int B, int Tn, int d, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // (b, c)
    if (i >= (long)B * d) return;
    const int b = (int)(i / d), c = (int)(i - (long)b * d);
    float s = 0.f;
    for (int t = 0; t < Tn; ++t) s += ld_f(x + ((long)b * Tn + t) * d + c);
    out[i] = s / (float)Tn;
}
template <typename T>
__global__ void mean_rows_bwd_kernel(const float* __restrict__ dy, int B, int Tn, int d, T* __restrict__ dx) {
    const long n = (long)B * Tn * d;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % d);
        const long b = i / ((long)Tn * d);
        st_f(dx + i, dy[b * d + c] / (float)Tn);
    }
}
// y[m][n] = act(sum_k x[m][k] W[n][k] + b[n]); one wave per (m, n)
__global__ void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias, int M, int N, int K,
                                        int act, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long o = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (o >= (long)M * N) return;
    const int m = (int)(o / N), n = (int)(o - (long)m * N);
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += x[(long)m * K + k] * W[(long)n * K + k];
    s = wave_sum(s);
    if (lane == 0) {
        s += bias ? bias[n] : 0.f;
        y[o] = act == 1 ? fmaxf(s, 0.f) : s;
    }
}
// dz = dy * act'(y) (in a temporary the caller provides, or dy itself for act 0); dx[m][k] = sum_n dz[m][n] W[n][k]
__global__ void small_linear_dx_kernel(const float* __restrict__ dz, const float* __restrict__ W, int M, int N, int K, float* __restrict__ dx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * K) return;
    const int m = (int)(i / K), k = (int)(i - (long)m * K);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dz[(long)m * N + n] * W[(long)n * K + k];
    dx[i] = s;
}
// dW[n][k] += sum_m dz[m][n] x[m][k];  db[n] += sum_m dz[m][n]  (thread k == 0 of row n)
__global__ void small_linear_dw_kernel(const float* __restrict__ dz, const float* __restrict__ x, int M, int N, int K, float* __restrict__ dW,
                                       float* __restrict__ db) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * K) return;
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    float s = 0.f, sb = 0.f;
    for (int m = 0; m < M; ++m) { const float g = dz[(long)m * N + n]; s += g * x[(long)m * K + k]; sb += g; }
    dW[i] += s;
    if (k == 0 && db) db[n] += sb;
}
__global__ void relu_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y, long n, float* __restrict__ dz) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dz[i] = y[i] > 0.f ? dy[i] : 0.f;
}

extern "C" int sarssl_mean_rows(const void* x, int B, int Tn, int d, float* out, int dtype, void* stream) {
    SARSSL_REQUIRE(B > 0 && Tn > 0 && d > 0 && x && out, "sarssl_mean_rows");
    const int nb = (int)(((long)B * d + 255) / 256);
    if (dtype == SARSSL_F32) mean_rows_kernel<float><<<nb, 256, 0, ST>>>((const float*)x, B, Tn, d, out);
    else if (dtype == SARSSL_F16) mean_rows_kernel<f16><<<nb, 256, 0, ST>>>((const f16*)x, B, Tn, d, out);
    else if (dtype == SARSSL_BF16) mean_rows_kernel<bf16><<<nb, 256, 0, ST>>>((const bf16*)x, B, Tn, d, out);
    else { sarssl_set_error("sarssl_mean_rows: dtype %d", dtype); return -1; }
    SARSSL_CHECK_LAUNCH("mean_rows_kernel");
    return 0;
}
extern "C" int sarssl_mean_rows_bwd(const float* dy, int B, int Tn, int d, void* dx, int dtype, void* stream) {
    SARSSL_REQUIRE(B > 0 && Tn > 0 && d > 0 && dy && dx, "sarssl_mean_rows_bwd");
    long n = (long)B * Tn * d;
    int nb = (int)((n + 255) / 256); if (nb > 8192) nb = 8192;
    if (dtype == SARSSL_F32) mean_rows_bwd_kernel<float><<<nb, 256, 0, ST>>>(dy, B, Tn, d, (float*)dx);
    else if (dtype == SARSSL_BF16) mean_rows_bwd_kernel<bf16><<<nb, 256, 0, ST>>>(dy, B, Tn, d, (bf16*)dx);
    else { sarssl_set_error("sarssl_mean_rows_bwd: dtype %d", dtype); return -1; }
    SARSSL_CHECK_LAUNCH("mean_rows_bwd_kernel");
    return 0;
}
// act: 0 none, 1 relu.  All f32, row-major contiguous: x [M][K], W [N][K], y [M][N].
extern "C" int sarssl_small_linear_fwd(const float* x, const float* W, const float* bias, int M, int N, int K, int act, float* y, void* stream) {
    SARSSL_REQUIRE(M > 0 && N > 0 && K > 0 && x && W && y, "sarssl_small_linear_fwd");
    const long outs = (long)M * N;
    small_linear_fwd_kernel<<<(int)((outs + 3) / 4), 256, 0, ST>>>(x, W, bias, M, N, K, act, y);
    SARSSL_CHECK_LAUNCH("small_linear_fwd_kernel");
    return 0;
}
// dx (may be null) = dz W, dW += dz^T x, db (may be null) += column sums of dz, with dz = dy (act 0) or dy * [y > 0] (act 1; dz_ws [M][N]
// receives it).
extern "C" int sarssl_small_linear_bwd(const float* dy, const float* y, const float* x, const float* W, int M, int N, int K, int act, float* dz_ws,
                                       float* dx, float* dW, float* db, void* stream) {
    SARSSL_REQUIRE(M > 0 && N > 0 && K > 0 && dy && x && W && dW && (act == 0 || (y && dz_ws)), "sarssl_small_linear_bwd");
    const float* dz = dy;
    if (act == 1) {
        relu_mask_kernel<<<(int)(((long)M * N + 255) / 256), 256, 0, ST>>>(dy, y, (long)M * N, dz_ws);
        dz = dz_ws;
    }
    if (dx) small_linear_dx_kernel<<<(int)(((long)M * K + 255) / 256), 256, 0, ST>>>(dz, W, M, N, K, dx);
    small_linear_dw_kernel<<<(int)(((long)N * K + 255) / 256), 256, 0, ST>>>(dz, x, M, N, K, dW, db);
    SARSSL_CHECK_LAUNCH("small_linear_bwd");
    return 0;
}
