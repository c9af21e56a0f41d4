// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (prints, for each lane and element, which LDS u16 index it received).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(int* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int lane = threadIdx.x;
    // mode 0: lane supplies address lane*8 bytes (contiguous)
    // mode 1: lane supplies (lane>>2 & 3)*pitch(64B) + (lane&3)*8  + (lane>>4)*1024
    uint32_t addr;
    if (mode == 0) addr = lane * 8;
    else addr = ((lane >> 2) & 3) * 64 + (lane & 3) * 8 + (lane >> 4) * 1024;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr / 2));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (int)(uint16_t)v[j];
}
int main() {
    int* d; hipMalloc(&d, 64 * 4 * sizeof(int));
    int h[256];
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
