// Hardware probe (not part of the product library): semantics of the gfx950 OCP-fp8 conversion and of the block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit (E8M0 = 127) block scales, as used by csrc/gemm_fp8.hip.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_fp8 tools/probe_fp8.hip && tools/probe_fp8
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// OCP e4m3fn decode (bias 7, no inf, 0x7f / 0xff = NaN)
static float e4m3_decode(uint8_t b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 0) v = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) v = NAN;
    else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

__global__ void cvt_kernel(const float* x, uint8_t* q, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {
        const int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
        q[2 * i] = (uint8_t)(w & 0xff);
        q[2 * i + 1] = (uint8_t)((w >> 8) & 0xff);
    }
}

// A [32][64], B [32][64] fp8 row-major (k contiguous); D[n][m] = sum_k B[n][k] A[m][k] written as D[row = n][col = m]
__global__ void mfma_kernel(const uint8_t* A, const uint8_t* B, float* D) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    const int* pa = (const int*)(A + (lane & 31) * 64 + (lane >> 5) * 32);
    const int* pb = (const int*)(B + (lane & 31) * 64 + (lane >> 5) * 32);
    for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        D[row * 32 + col] = c[r];
    }
}

int main() {
    // 1. conversion: every finite e4m3 value round-trips; rounding is to nearest even; 448 is the largest magnitude
    const int n = 512;
    float hx[n]; uint8_t hq[n];
    for (int i = 0; i < 256; ++i) { float v = e4m3_decode((uint8_t)i); hx[i] = isnan(v) ? 0.f : v; }
    for (int i = 256; i < n; ++i) hx[i] = ((rand() % 20001) - 10000) * (448.0f / 10000.0f);
    float* dx; uint8_t* dq;
    hipMalloc(&dx, n * 4); hipMalloc(&dq, n);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    cvt_kernel<<<1, 256>>>(dx, dq, n);
    hipMemcpy(hq, dq, n, hipMemcpyDeviceToHost);
    int bad_rt = 0; double worst = 0;
    for (int i = 0; i < 256; ++i) { float v = e4m3_decode((uint8_t)i); if (!isnan(v) && e4m3_decode(hq[i]) != v) ++bad_rt; }
    for (int i = 256; i < n; ++i) {
        const float d = e4m3_decode(hq[i]);
        double rel = fabs(d - hx[i]) / fmax(fabs(hx[i]), 1e-3);
        if (rel > worst) worst = rel;
    }
    printf("cvt: roundtrip mismatches %d / 254, worst relative rounding error %.4f (<= 1/16 expected), 448 -> 0x%02x, -448 -> 0x%02x\n",
           bad_rt, worst, hq[0x7e], hq[0xfe]);
    // 2. MFMA with unit block scales == plain fp8 dot products
    uint8_t hA[32 * 64], hB[32 * 64];
    for (int i = 0; i < 32 * 64; ++i) { hA[i] = (uint8_t)(rand() & 0xff); hB[i] = (uint8_t)(rand() & 0xff); if ((hA[i] & 0x7f) == 0x7f) hA[i] = 0x38; if ((hB[i] & 0x7f) == 0x7f) hB[i] = 0x38;
        hA[i] = (hA[i] & 0x80) | ((hA[i] & 0x7f) % 0x50); hB[i] = (hB[i] & 0x80) | ((hB[i] & 0x7f) % 0x50); }   // |v| <= 8: exact f32 sums
    uint8_t *dA, *dB; float* dD; float hD[1024];
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int nn = 0; nn < 32; ++nn)
        for (int m = 0; m < 32; ++m) {
            double ref = 0;
            for (int k = 0; k < 64; ++k) ref += (double)e4m3_decode(hB[nn * 64 + k]) * e4m3_decode(hA[m * 64 + k]);
            maxerr = fmax(maxerr, fabs(ref - hD[nn * 32 + m])); maxref = fmax(maxref, fabs(ref));
        }
    printf("mfma_scale_f32_32x32x64 (fp8 x fp8, E8M0 scales 127): max |err| %.3e vs max |ref| %.3e -> %s\n", maxerr, maxref,
           maxerr <= 1e-4 * maxref ? "OK" : "MISMATCH");
    return (bad_rt == 0 && maxerr <= 1e-4 * maxref) ? 0 : 1;
}
