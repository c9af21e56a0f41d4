// Helpers shared by the row-tile-resident fused launches (ffn2.hip, lin256.hip).
#pragma once
#include <stdlib.h>
#include "common.h"

// SARSSL_FFN_ROT=0: every workgroup of the fused feed-forward launches walks the hidden chunks from chunk 0 (the order up to round 6)
static inline int ffn_rot() {
    static const int rot = [] { const char* e = getenv("SARSSL_FFN_ROT"); return e ? atoi(e) : 1; }();
    return rot;
}

struct FfnDrop {
    unsigned long long seed; uint32_t key0, thr; float inv_keep; float p;
    __device__ __forceinline__ void init(float p_drop, unsigned long long s, const unsigned long long* salt) {
        p = p_drop;
        inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
        seed = 0; key0 = 0; thr = 0;
        if (p_drop > 0.f) { seed = salted_seed(s, salt); key0 = dropout_key(seed, 0u); thr = dropout_thr16(p_drop); }
    }
    // keep-scales of 4 consecutive elements from idx (a multiple of 4)
    __device__ __forceinline__ void scale4(unsigned long long idx, float (&k)[4]) const {
        const uint32_t key = (idx >> 33) == 0 ? key0 : dropout_key(seed, (uint32_t)(idx >> 33));
        const uint32_t pair = (uint32_t)(idx >> 1);
        const uint32_t h0 = hash_u32(pair ^ key), h1 = hash_u32((pair + 1u) ^ key);
        k[0] = (h0 & 0xffffu) >= thr ? inv_keep : 0.0f; k[1] = (h0 >> 16) >= thr ? inv_keep : 0.0f;
        k[2] = (h1 & 0xffffu) >= thr ? inv_keep : 0.0f; k[3] = (h1 >> 16) >= thr ? inv_keep : 0.0f;
    }
};

template <typename T>
__device__ __forceinline__ uint2 pack4(const float (&v)[4]) {
    uint2 u;
    u.x = H16<T>::pack(v[0], v[1]);
    u.y = H16<T>::pack(v[2], v[3]);
    return u;
}

