// Fused 2-channel real STFT front-end for gfx950.
//
// Replaces STFT.forward (code/common/utils_module.py:49-72) and the normalise / mic-pair /
// drop-DC steps of STFTLearner.data_preprocess (code/learner.py:525-553).
//
// Two real channels are transformed by ONE complex FFT-512 (z = ch_a + i*ch_b, periodic Hann,
// hop 256, center=False) and separated with the conjugate-symmetry identities.  One wave (64
// lanes x 8 points) owns one frame: three radix-8 passes with two in-LDS exchanges
// (n = 64 n1 + 8 n2 + n3, k = k1 + 8 k2 + 64 k3).  A 1024-thread workgroup = 16 consecutive
// frames, so every (channel, bin) row is written as one 128-byte line.  HBM-bound: reads each
// sample ~2x (50 % frame overlap, served from L2), writes 257 bins x 8 B per frame per channel.
#include "common.h"

#define NFFT 512
#define HOP 256
#define NBIN 257
#define FR_PER_BLOCK 16
#define ZSTRIDE 513   // float2 elements per wave buffer (padded)

struct cpx { float x, y; };
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cpx mul_mi(cpx a) { return {a.y, -a.x}; }   // a * (-i)

__device__ __forceinline__ void dft8(cpx* x) {
    cpx a0 = cadd(x[0], x[4]), a4 = csub(x[0], x[4]), a2 = cadd(x[2], x[6]), a6 = mul_mi(csub(x[2], x[6]));
    cpx a1 = cadd(x[1], x[5]), a5 = csub(x[1], x[5]), a3 = cadd(x[3], x[7]), a7 = mul_mi(csub(x[3], x[7]));
    cpx b0 = cadd(a0, a2), b2 = csub(a0, a2), b4 = cadd(a4, a6), b6 = csub(a4, a6);
    cpx b1 = cadd(a1, a3), b3 = mul_mi(csub(a1, a3)), b5 = cadd(a5, a7), b7 = csub(a5, a7);
    const float s = 0.70710678118654752440f;
    b5 = cmul(b5, cpx{s, -s});
    b7 = cmul(b7, cpx{-s, -s});
    x[0] = cadd(b0, b1); x[1] = cadd(b4, b5); x[2] = cadd(b2, b3); x[3] = cadd(b6, b7);
    x[4] = csub(b0, b1); x[5] = csub(b4, b5); x[6] = csub(b2, b3); x[7] = csub(b6, b7);
}
// exp(-2*pi*i * num / den), den a power of two
__device__ __forceinline__ cpx twiddle(int num, int den) {
    float s, c;
    sincospif(-2.0f * (float)(num & (den - 1)) / (float)den, &s, &c);
    return {c, s};
}
__device__ __forceinline__ float ld_sample(const float* p) { return *p; }
__device__ __forceinline__ float ld_sample(const int16_t* p) { return (float)(*p) * (1.0f / 32768.0f); }

// One wave transforms one 512-point frame: in v[j] = z[lane + 64 j], out zb[k] in natural order.  Three radix-8 passes with
// two in-LDS exchanges (n = 64 n1 + 8 n2 + n3, k = k1 + 8 k2 + 64 k3).  Every thread of the workgroup must call it (barriers).
__device__ __forceinline__ void fft512_wave(cpx (&v)[8], float2* zb, int lane, bool live) {
    if (live) {
        dft8(v);                                                                   // over n1
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            cpx y = cmul(v[k1], twiddle(lane * k1, 512));
            zb[k1 * 64 + lane] = make_float2(y.x, y.y);
        }
    }
    __syncthreads();
    if (live) {
        const int k1 = lane >> 3, n3 = lane & 7;
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) { float2 q = zb[k1 * 64 + n2 * 8 + n3]; v[n2] = {q.x, q.y}; }
        dft8(v);                                                                   // over n2
    }
    __syncthreads();
    if (live) {
        const int k1 = lane >> 3, n3 = lane & 7;
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            cpx y = cmul(v[k2], twiddle(n3 * k2, 64));
            zb[k1 * 64 + k2 * 8 + n3] = make_float2(y.x, y.y);
        }
    }
    __syncthreads();
    if (live) {
        const int k2 = lane >> 3, k1 = lane & 7;
#pragma unroll
        for (int n3 = 0; n3 < 8; ++n3) { float2 q = zb[k1 * 64 + k2 * 8 + n3]; v[n3] = {q.x, q.y}; }
        dft8(v);                                                                   // over n3
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) zb[lane + 64 * k3] = make_float2(v[k3].x, v[k3].y);
    }
    __syncthreads();
}

// U: (B, nch, 257, nt, 2) f32 unnormalised spectrum; magsum[b] += sum |X_ch0| over 257 bins x nt frames
template <typename TIn>
__global__ __launch_bounds__(1024) void stft_pair_kernel(const TIn* __restrict__ sig, long nsample, int nch, int nt,
                                                         float* __restrict__ U, double* __restrict__ magsum) {
    __shared__ float2 Z[FR_PER_BLOCK * ZSTRIDE];
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x * FR_PER_BLOCK + wave;
    const int pair = blockIdx.y, b = blockIdx.z;
    const int c0 = pair * 2, c1 = pair * 2 + 1;
    float2* zb = Z + wave * ZSTRIDE;
    const bool live = t < nt;

    cpx v[8];
    // A frame whose samples are ALL zero in one channel (a dead / muted microphone) has an exactly zero spectrum in the reference
    // (torch.stft per channel).  Two channels share one complex FFT here, and the separation below leaves ~1e-7 |X_other| of rounding
    // residue in the silent one - enough to move the normaliser mean|X_0| + 1e-6 by 8 % when the reference microphone is silent
    // (fixture F15).  Such frames are written as exact zeros.
    __shared__ unsigned char nonzero[FR_PER_BLOCK][2];
    bool nz0 = false, nz1 = false;
    if (live) {
        const TIn* base = sig + ((long)b * nsample + (long)t * HOP) * nch;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = lane + 64 * j;
            const float w = 0.5f - 0.5f * cospif((float)n * (1.0f / 256.0f));     // periodic Hann, N = 512
            const float re = ld_sample(base + (long)n * nch + c0);
            const float im = (c1 < nch) ? ld_sample(base + (long)n * nch + c1) : 0.f;
            nz0 = nz0 || re != 0.f; nz1 = nz1 || im != 0.f;
            v[j] = {re * w, im * w};
        }
    }
    {
        const bool a0 = __ballot(nz0) != 0ull, a1 = __ballot(nz1) != 0ull;
        if (lane == 0) { nonzero[wave][0] = a0; nonzero[wave][1] = a1; }
    }
    fft512_wave(v, zb, lane, live);

    // separate the two real channels and write (c, f, t, reim); 16 frames = one 128-byte line per (c, f)
    const int tl = tid & 15, fo = tid >> 4;
    const int tt = blockIdx.x * FR_PER_BLOCK + tl;
    float local = 0.f;
    for (int f = fo; f < NBIN; f += 64) {
        if (tt < nt) {
            const float2 p = Z[tl * ZSTRIDE + f];
            const float2 q = Z[tl * ZSTRIDE + ((NFFT - f) & (NFFT - 1))];
            // X_a = (Z[f] + conj(Z[N-f]))/2 ; X_b = (Z[f] - conj(Z[N-f]))/(2i)
            float ar = 0.5f * (p.x + q.x), ai = 0.5f * (p.y - q.y);
            float br = 0.5f * (p.y + q.y), bi = -0.5f * (p.x - q.x);
            if (!nonzero[tl][0]) { ar = 0.f; ai = 0.f; }
            if (!nonzero[tl][1]) { br = 0.f; bi = 0.f; }
            float* oa = U + ((((long)b * nch + c0) * NBIN + f) * nt + tt) * 2;
            *(float2*)oa = make_float2(ar, ai);
            if (c1 < nch) {
                float* ob = U + ((((long)b * nch + c1) * NBIN + f) * nt + tt) * 2;
                *(float2*)ob = make_float2(br, bi);
            }
            if (pair == 0) local += sqrtf(ar * ar + ai * ai);
        }
    }
    if (pair == 0) {
        local = wave_sum(local);
        if (lane == 0) red[wave] = local;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < 16; ++i) s += (double)red[i];
            atomicAdd(&magsum[b], s);
        }
    }
}

// out[(b*npair+p), mic, f(256), t, reim] = U[b, ch(p, mic), f+1, t, reim] / (mean|X_ch0| + eps)
// pair_mode 0 ('M'): pair p = (0, p+1); pair_mode 1 ('MM'): all i<j pairs in the order (0,1),(0,2),..,(1,2),..
__global__ void frontend_pack_kernel(const float* __restrict__ U, const double* __restrict__ magsum, int nb, int nch,
                                     int nt, float eps, int pair_mode, int npair, float* __restrict__ out) {
    const long per_ch = (long)256 * nt;           // float2 elements per (pair, mic)
    const long total = (long)nb * npair * 2 * per_ch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long e = i % per_ch;
        const long q = i / per_ch;
        const int mic = (int)(q & 1);
        const long bp = q >> 1;
        const int p = (int)(bp % npair);
        const int b = (int)(bp / npair);
        int c;
        if (pair_mode == 0) c = mic ? p + 1 : 0;
        else {
            int i0 = 0, rem = p;
            while (rem >= nch - 1 - i0) { rem -= nch - 1 - i0; ++i0; }
            c = mic ? i0 + 1 + rem : i0;
        }
        const float scale = 1.0f / ((float)(magsum[b] / ((double)NBIN * nt)) + eps);
        const float2 x = *(const float2*)(U + ((((long)b * nch + c) * NBIN) * nt + nt + e) * 2);   // skip DC row
        *(float2*)(out + i * 2) = make_float2(x.x * scale, x.y * scale);
    }
}

// The same with the pretraining masks applied in the same pass (model.py:541, :563 - what sarssl_mask_inputs does to `out`): one thread
// per (pair, bin, frame) handles both microphones, writes their two `out` entries and the (B, F, T, 4) 16-bit inputs of the two encoders:
//   spec = x * mask_patch_ch (frame mask on the masked channel, its complement on the other), spat = x * frame mask
// - bit-identical to the two-launch sequence (the masks multiply the very f32 values that are stored to `out`), one 67 MB read less.
template <typename T>
__global__ void frontend_pack_masked_kernel(const float* __restrict__ U, const double* __restrict__ magsum, int nb, int nch, int nt,
                                            float eps, int pair_mode, int npair, float* __restrict__ out,
                                            const uint8_t* __restrict__ mp, const int* __restrict__ mch, T* __restrict__ spec,
                                            T* __restrict__ spat, int* __restrict__ ovf) {
    const long per = (long)256 * nt;
    const long total = (long)nb * npair * per;
    bool over = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long e = i % per;
        const long bp = i / per;
        const int p = (int)(bp % npair);
        const int b = (int)(bp / npair);
        int c0 = 0, c1 = p + 1;
        if (pair_mode != 0) {
            int i0 = 0, rem = p;
            while (rem >= nch - 1 - i0) { rem -= nch - 1 - i0; ++i0; }
            c0 = i0; c1 = i0 + 1 + rem;
        }
        const float scale = 1.0f / ((float)(magsum[b] / ((double)NBIN * nt)) + eps);
        const float2 u0 = *(const float2*)(U + ((((long)b * nch + c0) * NBIN) * nt + nt + e) * 2);   // skip DC row
        const float2 u1 = *(const float2*)(U + ((((long)b * nch + c1) * NBIN) * nt + nt + e) * 2);
        const float2 m0 = make_float2(u0.x * scale, u0.y * scale), m1 = make_float2(u1.x * scale, u1.y * scale);
        *(float2*)(out + ((bp * 2 + 0) * per + e) * 2) = m0;
        *(float2*)(out + ((bp * 2 + 1) * per + e) * 2) = m1;
        const int t = (int)(e % nt);
        const float pm = mp[bp * nt + t] ? 1.f : 0.f;
        const int mc = mch[bp];
        const float v0 = (mc == 0) ? 0.f : 1.f, v1 = (mc == 1) ? 0.f : 1.f;       // mask_ch_dense per mic
        const float s0 = (1.f - pm) * v0 + pm * (1.f - v0), s1 = (1.f - pm) * v1 + pm * (1.f - v1);
        // (values outside fp16's range are flagged for the loss launch: see mask_inputs_kernel, csrc/stem.hip)
        if constexpr (__is_same(T, f16)) over = over || !(fmaxf(fmaxf(fabsf(m0.x), fabsf(m0.y)), fmaxf(fabsf(m1.x), fabsf(m1.y))) <= 65504.f);
        st4(spec + i * 4, make_float4(m0.x * s0, m1.x * s1, m0.y * s0, m1.y * s1));
        st4(spat + i * 4, make_float4(m0.x * pm, m1.x * pm, m0.y * pm, m1.y * pm));
    }
    if (over && ovf) atomicOr(ovf, 1);
}

// complex64 (B, 257, nt, nch) view of U for the STFT.forward drop-in
__global__ void stft_permute_kernel(const float* __restrict__ U, int nb, int nch, int nt, float* __restrict__ out) {
    const long total = (long)nb * NBIN * nt * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        long r = i / nch;
        const int t = (int)(r % nt); r /= nt;
        const int f = (int)(r % NBIN);
        const int b = (int)(r / NBIN);
        const float2 x = *(const float2*)(U + ((((long)b * nch + c) * NBIN + f) * nt + t) * 2);
        *(float2*)(out + i * 2) = x;
    }
}

static int launch_stft(const void* sig, int sig_dtype, int nb, long nsample, int nch, int nt, float* U, double* magsum,
                       hipStream_t st) {
    if (hipMemsetAsync(magsum, 0, sizeof(double) * nb, st) != hipSuccess) { sarssl_set_error("stft: memset failed"); return -2; }
    dim3 grid((nt + FR_PER_BLOCK - 1) / FR_PER_BLOCK, (nch + 1) / 2, nb);
    if (sig_dtype == SARSSL_F32) stft_pair_kernel<float><<<grid, 1024, 0, st>>>((const float*)sig, nsample, nch, nt, U, magsum);
    else if (sig_dtype == SARSSL_I16) stft_pair_kernel<int16_t><<<grid, 1024, 0, st>>>((const int16_t*)sig, nsample, nch, nt, U, magsum);
    else { sarssl_set_error("stft: unsupported signal dtype %d", sig_dtype); return -1; }
    SARSSL_CHECK_LAUNCH("stft_pair_kernel");
    return 0;
}

// sig: (B, nsample, nch) f32 or int16 PCM.  U: workspace (B, nch, 257, nt, 2) f32.  magsum: (B) f64 workspace.
// out: (B*npair, 2, 256, nt, 2) f32 = data_preprocess output; pair_mode 0 = ch_mode 'M' (npair = nch-1: mic 0 with every
// other mic), 1 = ch_mode 'MM' (npair = nch(nch-1)/2: every mic pair).
extern "C" int sarssl_stft_frontend_pairs(const void* sig, int sig_dtype, int nb, long nsample, int nch, int win_len, int hop,
                                          int nfft, int nt, float eps, int pair_mode, float* U, double* magsum, float* out,
                                          void* stream) {
    SARSSL_REQUIRE(win_len == NFFT && nfft == NFFT && hop == HOP, "sarssl_stft_frontend(only win=nfft=512, hop=256)");
    SARSSL_REQUIRE(nch >= 2 && nb > 0 && nt > 0 && (long)(nt - 1) * HOP + NFFT <= nsample, "sarssl_stft_frontend");
    SARSSL_REQUIRE(pair_mode == 0 || pair_mode == 1, "sarssl_stft_frontend(pair_mode 0 'M' | 1 'MM')");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_stft(sig, sig_dtype, nb, nsample, nch, nt, U, magsum, st);
    if (rc) return rc;
    const int npair = pair_mode == 0 ? nch - 1 : nch * (nch - 1) / 2;
    const long total = (long)nb * npair * 2 * 256 * nt;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    frontend_pack_kernel<<<blocks, 256, 0, st>>>(U, magsum, nb, nch, nt, eps, pair_mode, npair, out);
    SARSSL_CHECK_LAUNCH("frontend_pack_kernel");
    return 0;
}
// data_preprocess + the pretraining input masks in one pass over the spectrum (sarssl_stft_frontend_pairs followed by sarssl_mask_inputs
// mode 0 on its result): mp (B*npair, nt) u8 frame mask (0 = masked), mch (B*npair) i32 masked channel; spec / spat (B*npair, 256, nt, 4)
// of `dtype`.
extern "C" int sarssl_stft_frontend_pairs_masked(const void* sig, int sig_dtype, int nb, long nsample, int nch, int win_len, int hop,
                                                 int nfft, int nt, float eps, int pair_mode, float* U, double* magsum, float* out,
                                                 const unsigned char* mp, const int* mch, void* spec, void* spat, int dtype, void* stream) {
    SARSSL_REQUIRE(win_len == NFFT && nfft == NFFT && hop == HOP, "sarssl_stft_frontend(only win=nfft=512, hop=256)");
    SARSSL_REQUIRE(nch >= 2 && nb > 0 && nt > 0 && (long)(nt - 1) * HOP + NFFT <= nsample, "sarssl_stft_frontend_pairs_masked");
    SARSSL_REQUIRE((pair_mode == 0 || pair_mode == 1) && mp && mch && spec && spat, "sarssl_stft_frontend_pairs_masked");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_stft(sig, sig_dtype, nb, nsample, nch, nt, U, magsum, st);
    if (rc) return rc;
    const int npair = pair_mode == 0 ? nch - 1 : nch * (nch - 1) / 2;
    const long total = (long)nb * npair * 256 * nt;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    if (dtype == SARSSL_F16) frontend_pack_masked_kernel<f16><<<blocks, 256, 0, st>>>(U, magsum, nb, nch, nt, eps, pair_mode, npair, out, mp, mch, (f16*)spec, (f16*)spat, sarssl_overflow_flag());
    else if (dtype == SARSSL_BF16) frontend_pack_masked_kernel<bf16><<<blocks, 256, 0, st>>>(U, magsum, nb, nch, nt, eps, pair_mode, npair, out, mp, mch, (bf16*)spec, (bf16*)spat, nullptr);
    else if (dtype == SARSSL_F32) frontend_pack_masked_kernel<float><<<blocks, 256, 0, st>>>(U, magsum, nb, nch, nt, eps, pair_mode, npair, out, mp, mch, (float*)spec, (float*)spat, nullptr);
    else { sarssl_set_error("sarssl_stft_frontend_pairs_masked: dtype"); return -1; }
    SARSSL_CHECK_LAUNCH("frontend_pack_masked_kernel");
    return 0;
}
extern "C" int sarssl_stft_frontend(const void* sig, int sig_dtype, int nb, long nsample, int nch, int win_len, int hop,
                                    int nfft, int nt, float eps, float* U, double* magsum, float* out, void* stream) {
    return sarssl_stft_frontend_pairs(sig, sig_dtype, nb, nsample, nch, win_len, hop, nfft, nt, eps, 0, U, magsum, out, stream);
}

// out: complex64 (B, 257, nt, nch) as interleaved f32 pairs.
extern "C" int sarssl_stft_raw(const void* sig, int sig_dtype, int nb, long nsample, int nch, int win_len, int hop,
                               int nfft, int nt, float* U, double* magsum, float* out, void* stream) {
    SARSSL_REQUIRE(win_len == NFFT && nfft == NFFT && hop == HOP, "sarssl_stft_raw(only win=nfft=512, hop=256)");
    SARSSL_REQUIRE(nch >= 1 && nb > 0 && nt > 0 && (long)(nt - 1) * HOP + NFFT <= nsample, "sarssl_stft_raw");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_stft(sig, sig_dtype, nb, nsample, nch, nt, U, magsum, st);
    if (rc) return rc;
    const long total = (long)nb * NBIN * nt * nch;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    stft_permute_kernel<<<blocks, 256, 0, st>>>(U, nb, nch, nt, out);
    SARSSL_CHECK_LAUNCH("stft_permute_kernel");
    return 0;
}

// ------------------------------------------------------------------------------------------------ inverse STFT
// ISTFT.forward (code/common/utils_module.py:91-113) = torch.istft(n_fft=512, hop=256, win_length=512, window=None (ones),
// center=False | True, onesided): per frame a real inverse FFT-512 (imaginary parts of DC / Nyquist ignored), overlap-add,
// division by the window envelope (= number of frames covering the sample).  Two channels share one complex FFT:
// Z = X_a + i X_b (Hermitian-extended), z = IFFT(Z) = conj(FFT(conj(Z)))/N, x_a = Re z, x_b = Im z.
// spec: (B, 257, nt, nch) complex64 interleaved; frames: workspace (B, nch, nt, 512) f32.
__global__ __launch_bounds__(1024) void istft_frames_kernel(const float* __restrict__ spec, int nch, int nt,
                                                            float* __restrict__ frames) {
    __shared__ float2 Z[FR_PER_BLOCK * ZSTRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x * FR_PER_BLOCK + wave;
    const int pair = blockIdx.y, b = blockIdx.z;
    const int c0 = pair * 2, c1 = pair * 2 + 1;
    float2* zb = Z + wave * ZSTRIDE;
    const bool live = t < nt;
    cpx v[8];
    if (live) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j;
            const int kk = k <= 256 ? k : NFFT - k;
            const float sgn = k <= 256 ? 1.f : -1.f;                               // X[N-k] = conj(X[k])
            const float* q = spec + ((((long)b * NBIN + kk) * nt + t) * nch) * 2;
            float2 a = *(const float2*)(q + c0 * 2);
            float2 c = (c1 < nch) ? *(const float2*)(q + c1 * 2) : make_float2(0.f, 0.f);
            if (kk == 0 || kk == 256) { a.y = 0.f; c.y = 0.f; }
            a.y *= sgn; c.y *= sgn;
            // Z = a + i c = (a.x - c.y) + i (a.y + c.x); feed conj(Z)
            v[j] = {a.x - c.y, -(a.y + c.x)};
        }
    }
    fft512_wave(v, zb, lane, live);
    if (live) {
        float* o0 = frames + (((long)b * nch + c0) * nt + t) * NFFT;
        float* o1 = frames + (((long)b * nch + c1) * nt + t) * NFFT;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = lane + 64 * j;
            const float2 y = zb[n];
            o0[n] = y.x * (1.0f / NFFT);
            if (c1 < nch) o1[n] = -y.y * (1.0f / NFFT);
        }
    }
}

// sig[b][n][c] = (sum of the <= 2 frames covering sample n) / (their number); center: drop NFFT/2 samples at both ends
__global__ void istft_ola_kernel(const float* __restrict__ frames, int nb, int nch, int nt, int center, long nsample,
                                 float* __restrict__ sig) {
    const long total = (long)nb * nsample * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const long r = i / nch;
        const long n = r % nsample;
        const int b = (int)(r / nsample);
        const long m = n + (center ? NFFT / 2 : 0);
        const int t1 = (int)(m / HOP), t0 = t1 - 1;
        const float* fb = frames + ((long)b * nch + c) * nt * NFFT;
        float acc = 0.f, cnt = 0.f;
        if (t1 < nt) { acc += fb[(long)t1 * NFFT + (m - (long)t1 * HOP)]; cnt += 1.f; }
        if (t0 >= 0 && t0 < nt) { acc += fb[(long)t0 * NFFT + (m - (long)t0 * HOP)]; cnt += 1.f; }
        sig[i] = cnt > 0.f ? acc / cnt : 0.f;
    }
}

extern "C" long sarssl_istft_workspace_bytes(int nb, int nch, int nt) { return (long)nb * nch * nt * NFFT * sizeof(float); }

// spec: (B, 257, nt, nch) complex64; sig: (B, nsample, nch) f32 with nsample = (nt+1)*256 (center = 0) or (nt-1)*256 (center = 1)
extern "C" int sarssl_istft(const float* spec, int nb, int nch, int nt, int win_len, int hop, int nfft, int center,
                            float* frames_ws, float* sig, void* stream) {
    SARSSL_REQUIRE(win_len == NFFT && nfft == NFFT && hop == HOP, "sarssl_istft(only win=nfft=512, hop=256)");
    SARSSL_REQUIRE(nb > 0 && nch >= 1 && nt >= (center ? 2 : 1), "sarssl_istft");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((nt + FR_PER_BLOCK - 1) / FR_PER_BLOCK, (nch + 1) / 2, nb);
    istft_frames_kernel<<<grid, 1024, 0, st>>>(spec, nch, nt, frames_ws);
    SARSSL_CHECK_LAUNCH("istft_frames_kernel");
    const long nsample = center ? (long)(nt - 1) * HOP : (long)(nt + 1) * HOP;
    const long total = (long)nb * nsample * nch;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    istft_ola_kernel<<<blocks, 256, 0, st>>>(frames_ws, nb, nch, nt, center, nsample, sig);
    SARSSL_CHECK_LAUNCH("istft_ola_kernel");
    return 0;
}
