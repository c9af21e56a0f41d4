// Sustained bf16 MFMA rate of the whole chip as a function of waves per SIMD and independent accumulators per wave
// (v_mfma_f32_32x32x16_bf16 back to back, no memory traffic).  Build: hipcc --offload-arch=gfx950 -O3 tools/probe_mfma.hip -o tools/probe_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) out[0] = s;
}

template <int NACC>
static void run(int waves_per_simd, int ncu) {
    const int threads = 256;                       // 4 waves = one per SIMD
    const int blocks = ncu * waves_per_simd;       // waves_per_simd workgroups per CU
    const int iters = 20000 / NACC * 4;
    float* out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mfma_loop<NACC><<<blocks, threads>>>(out, iters / 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, threads>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * NACC * 32.0 * 32 * 16 * 2;
    printf("waves/SIMD %d  accumulators/wave %d : %7.3f ms  %7.1f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", waves_per_simd, NACC, ms,
           flop / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * NACC));
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", p.name, ncu, p.clockRate / 1000);
    for (int w = 1; w <= 4; w *= 2) { run<1>(w, ncu); run<2>(w, ncu); run<4>(w, ncu); run<8>(w, ncu); }
    return 0;
}
