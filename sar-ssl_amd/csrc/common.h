// Shared device helpers for the SAR-SSL gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// dtype codes of the C ABI (include/sarssl_hip.h).  SARSSL_F16: IEEE half.  SARSSL_MIX16 (kernels with both gradient-side and
// saved-activation-side 16-bit tensors): gradients in / out are bf16, tensors SAVED BY THE FORWARD PASS are fp16 - the "fp16 forward /
// bf16 backward" numeric mode (fp16 carries 3 more mantissa bits than bf16 at the same MFMA rate: per-bin deviation from the
// reference's f32 path 8e-3 -> 1e-3 of range; gradients keep bf16's exponent range, so there is no loss scaling)
// SARSSL_MIXF32 (hybrid mode): bf16 gradients next to an f32 tensor saved by the forward pass (the f32 prediction / residual stream)
enum { SARSSL_F32 = 0, SARSSL_BF16 = 1, SARSSL_I16 = 2, SARSSL_F16 = 3, SARSSL_MIX16 = 4, SARSSL_MIXF32 = 5 };

// error plumbing (api.cpp owns the storage)
extern "C" const char* sarssl_last_error();
void sarssl_set_error(const char* fmt, ...);
int sarssl_cu_count();          // CUs of the current device (api.hip): grid size of the persistent kernels
// Per-device context (api.hip; sarssl_create / sarssl_destroy / sarssl_make_current in include/sarssl_hip.h): everything a caller can
// configure lives in one of these - there is no process-global mutable state in the library.  The launch wrappers read the context
// that is CURRENT ON THE CALLING THREAD (null: library defaults).
struct sarssl_ctx {
    int device;
    int conv_cus_bwd;                       // workgroups of the 3x3 gradient launches (0: the default 7/8 rule)
    unsigned long long* conv_clk;           // clock-probe buffer of the 3x3 forward / data-gradient launches, or null
    const unsigned long long* salt;         // dropout-seed addend: &SarsslStepState::salt of the attached step state, or null
    const char* zero_lo; const char* zero_hi;   // host-zeroed accumulator arena: memsets of pointers inside it are skipped
    int* ovf_flag;                          // device word: set by the kernels that encode external-scale data as fp16 when a value does not
                                            // fit (|v| > 65 504); the loss launch reads it, poisons the loss with NaN and clears it
};
int* sarssl_overflow_flag();                // the current context's flag (null without a context)
sarssl_ctx* sarssl_current();   // may be null
// Device-resident step state (api.hip, sarssl_step_state_*): lets a step captured in a hipGraph vary per replay.  `salt` is added to
// every dropout seed by the kernels (null / 0 outside graph capture); the Adam fields are advanced by sarssl_step_tick.
struct SarsslStepState {
    unsigned long long salt;
    int step;                    // Adam step count (1-based after the first tick)
    float lr, beta1, beta2;
    float step_size;             // lr / (1 - beta1^step)
    float inv_bc2_sqrt;          // 1 / sqrt(1 - beta2^step)
    int nskipped;                // optimizer steps skipped because the step's loss was not finite (sarssl_adam_step_dev_guard)
};
int sarssl_mfma_prio();         // where waves raise their issue priority (s_setprio) during MFMA phases: 2 = the ping-pong convolution only
bool sarssl_prezeroed(const void* p);               // pointer inside the host-zeroed arena (api.hip): its memset can be skipped
// zero `bytes` at p on `st` unless p is a slice of the pre-zeroed arena
#define SARSSL_ZERO(p, bytes, st) (sarssl_prezeroed(p) ? hipSuccess : hipMemsetAsync((p), 0, (bytes), (st)))
const unsigned long long* sarssl_dropout_salt();   // pointer the launch wrappers hand to kernels that draw dropout masks (may be null)
#define SARSSL_CHECK_LAUNCH(name)                                          \
    do {                                                                   \
        hipError_t e__ = hipGetLastError();                                \
        if (e__ != hipSuccess) {                                           \
            sarssl_set_error("%s: %s", name, hipGetErrorString(e__));      \
            return -2;                                                     \
        }                                                                  \
    } while (0)
#define SARSSL_REQUIRE(cond, name)                                         \
    do {                                                                   \
        if (!(cond)) {                                                     \
            sarssl_set_error("%s: requirement failed: %s", name, #cond);   \
            return -1;                                                     \
        }                                                                  \
    } while (0)

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
// f32 -> bf16, round-to-nearest-even, two values per instruction: gfx950's v_cvt_pk_bf16_f32 (a software RNE costs ~6 VALU ops
// per element, which made the BatchNorm prologue and the bf16 epilogues of the convolution / GEMM kernels VALU-bound)
typedef float sarssl_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sarssl_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    sarssl_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sarssl_bf16x2));
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) { return pack2_bf16(f, 0.f) & 0xffffu; }
// fp16 <-> f32: v_cvt_f32_f16 (the high half through SDWA: one VALU op per element, like the bf16 shift / mask), v_cvt_pk_f16_f32 (RNE)
typedef _Float16 sarssl_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_f16(float lo, float hi) {
    sarssl_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sarssl_f16x2));
}
__device__ __forceinline__ sarssl_f32x2 unpack2_f16(uint32_t w) { return __builtin_convertvector(__builtin_bit_cast(sarssl_f16x2, w), sarssl_f32x2); }
__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) { return unpack2_f16(b).x; }
// 16-bit storage traits: two elements per 32-bit word
template <typename T> struct H16;
template <> struct H16<bf16> {
    static __device__ __forceinline__ float lo(uint32_t w) { return bf16_bits_to_f32(w & 0xffffu); }
    static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
    static __device__ __forceinline__ uint32_t pack(float a, float b) { return pack2_bf16(a, b); }
};
template <> struct H16<f16> {
    static __device__ __forceinline__ float lo(uint32_t w) { return unpack2_f16(w).x; }
    static __device__ __forceinline__ float hi(uint32_t w) { return unpack2_f16(w).y; }
    static __device__ __forceinline__ uint32_t pack(float a, float b) { return pack2_f16(a, b); }
};
__device__ __forceinline__ float ld_f(const float* p) { return *p; }
__device__ __forceinline__ float ld_f(const f16* p) { return (float)*p; }
__device__ __forceinline__ void st_f(f16* p, float v) { *p = (f16)v; }
__device__ __forceinline__ float ld_f(const bf16* p) { return bf16_bits_to_f32(*(const uint16_t*)p); }
__device__ __forceinline__ void st_f(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_f(bf16* p, float v) { *(uint16_t*)p = (uint16_t)f32_to_bf16_bits(v); }

// 4-wide vector access (address must be 4-element aligned)
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ float4 ld4(const bf16* p) {
    uint2 u = *(const uint2*)p;
    return make_float4(bf16_bits_to_f32(u.x & 0xffffu), __uint_as_float(u.x & 0xffff0000u),
                       bf16_bits_to_f32(u.y & 0xffffu), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float4 ld4(const f16* p) {
    uint2 u = *(const uint2*)p;
    const sarssl_f32x2 a = unpack2_f16(u.x), b = unpack2_f16(u.y);
    return make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void st4(f16* p, float4 v) {
    uint2 u;
    u.x = pack2_f16(v.x, v.y);
    u.y = pack2_f16(v.z, v.w);
    *(uint2*)p = u;
}
__device__ __forceinline__ void st4(float* p, float4 v) { *(float4*)p = v; }
__device__ __forceinline__ void st4(bf16* p, float4 v) {
    uint2 u;
    u.x = pack2_bf16(v.x, v.y);
    u.y = pack2_bf16(v.z, v.w);
    *(uint2*)p = u;
}
// 8-wide
struct f8 { float v[8]; };
__device__ __forceinline__ f8 ld8(const float* p) {
    f8 r; float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ f8 ld8(const bf16* p) {
    uint4 u = *(const uint4*)p; f8 r;
    r.v[0] = bf16_bits_to_f32(u.x & 0xffffu); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = bf16_bits_to_f32(u.y & 0xffffu); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = bf16_bits_to_f32(u.z & 0xffffu); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = bf16_bits_to_f32(u.w & 0xffffu); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
// 8 packed 16-bit elements <-> 8 floats
template <typename T>
__device__ __forceinline__ f8 unpack8(const uint4& u) {
    f8 r;
    r.v[0] = H16<T>::lo(u.x); r.v[1] = H16<T>::hi(u.x); r.v[2] = H16<T>::lo(u.y); r.v[3] = H16<T>::hi(u.y);
    r.v[4] = H16<T>::lo(u.z); r.v[5] = H16<T>::hi(u.z); r.v[6] = H16<T>::lo(u.w); r.v[7] = H16<T>::hi(u.w);
    return r;
}
template <typename T>
__device__ __forceinline__ uint4 pack8(const f8& r) {
    uint4 u;
    u.x = H16<T>::pack(r.v[0], r.v[1]); u.y = H16<T>::pack(r.v[2], r.v[3]);
    u.z = H16<T>::pack(r.v[4], r.v[5]); u.w = H16<T>::pack(r.v[6], r.v[7]);
    return u;
}
// x as it reads back after being stored as T (statistics of STORED tensors must see the rounded values)
template <typename T> __device__ __forceinline__ float round_as(float x) {
    if constexpr (sizeof(T) == 4) return x;
    else if constexpr (__is_same(T, f16)) return (float)(f16)x;
    else return bf16_bits_to_f32(f32_to_bf16_bits(x));
}
// 16-byte chunk of TS elements re-encoded as TD (identity when the types agree): saved fp16 activations entering a bf16 contraction
template <typename TS, typename TD>
__device__ __forceinline__ uint4 recode8(const uint4& u) {
    if constexpr (sizeof(TS) == sizeof(TD) && __is_same(TS, TD)) return u;
    else return pack8<TD>(unpack8<TS>(u));
}
__device__ __forceinline__ f8 ld8(const f16* p) { return unpack8<f16>(*(const uint4*)p); }
__device__ __forceinline__ void st8(f16* p, const f8& r) { *(uint4*)p = pack8<f16>(r); }
__device__ __forceinline__ void st8(float* p, const f8& r) {
    *(float4*)p = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    *(float4*)(p + 4) = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
__device__ __forceinline__ void st8(bf16* p, const f8& r) {
    uint4 u;
    u.x = pack2_bf16(r.v[0], r.v[1]);
    u.y = pack2_bf16(r.v[2], r.v[3]);
    u.z = pack2_bf16(r.v[4], r.v[5]);
    u.w = pack2_bf16(r.v[6], r.v[7]);
    *(uint4*)p = u;
}

// 32x32x16 MFMA on 16-bit operands of type TM (bf16x8 is the 128-bit fragment carrier for both encodings; same rate)
template <typename TM>
__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (__is_same(TM, f16)) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Split-bf16 ("precise") operand parts: x ~= hi + lo with hi = bf16(x), lo = bf16(x - hi).
// part 0 -> hi, part 1 -> lo.  For bf16 sources lo is exactly 0 and never requested.
__device__ __forceinline__ uint32_t bf16_part_bits(float x, int part) {
    uint32_t hi = f32_to_bf16_bits(x);
    if (part == 0) return hi;
    return f32_to_bf16_bits(x - bf16_bits_to_f32(hi));
}

// pack 8 floats -> 8 bf16 (uint4), selecting the hi or lo part
__device__ __forceinline__ uint4 pack8_part(const f8& r, int part) {
    uint4 u;
    if (part == 0) {
        u.x = pack2_bf16(r.v[0], r.v[1]); u.y = pack2_bf16(r.v[2], r.v[3]);
        u.z = pack2_bf16(r.v[4], r.v[5]); u.w = pack2_bf16(r.v[6], r.v[7]);
        return u;
    }
    u.x = bf16_part_bits(r.v[0], part) | (bf16_part_bits(r.v[1], part) << 16);
    u.y = bf16_part_bits(r.v[2], part) | (bf16_part_bits(r.v[3], part) << 16);
    u.z = bf16_part_bits(r.v[4], part) | (bf16_part_bits(r.v[5], part) << 16);
    u.w = bf16_part_bits(r.v[6], part) | (bf16_part_bits(r.v[7], part) << 16);
    return u;
}

// v_exp_f32 + v_rcp_f32 (1 ulp each).  `1.0f / (...)` is an IEEE division here (no fast-math: two v_div_scale, v_rcp, four FMAs,
// v_div_fmas, v_div_fixup): the Swish / Swish' / GLU epilogues evaluate ~400 M sigmoids per step, 8 400 division sequences in gemm.o
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// counter-based dropout RNG: keep-decision for element `idx` of stream `seed`
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// One 32-bit hash decides TWO neighbouring elements (16 bits each: keep iff bits >= p * 65536), so an 8-wide epilogue pays
// four hashes instead of sixteen - the dropout epilogues of the FFN GEMMs were VALU-bound on the per-element double hash.
// The decision is a pure function of (seed, idx): forward and backward recompute the same mask.
__device__ __forceinline__ uint32_t dropout_key(uint64_t seed, uint32_t idx_hi) {
    return hash_u32(idx_hi + (uint32_t)seed) ^ (uint32_t)(seed >> 32) * 0x9e3779b9u;
}
__device__ __forceinline__ uint32_t dropout_thr16(float p_drop) { return (uint32_t)(p_drop * 65536.0f + 0.5f); }
__device__ __forceinline__ uint64_t salted_seed(uint64_t seed, const unsigned long long* salt) { return salt ? seed + *salt : seed; }
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float p_drop, float inv_keep) {
    const uint32_t h = hash_u32((uint32_t)(idx >> 1) ^ dropout_key(seed, (uint32_t)(idx >> 33)));
    const uint32_t bits = (idx & 1) ? (h >> 16) : (h & 0xffffu);
    return bits >= dropout_thr16(p_drop) ? inv_keep : 0.0f;
}
// keep-scales of the 4 consecutive elements from idx (a multiple of 4): one key + two pair hashes (dropout_scale: two hashes per element)
__device__ __forceinline__ void dropout_scale4(uint64_t seed, uint64_t idx, float p_drop, float inv_keep, float (&k)[4]) {
    const uint32_t thr = dropout_thr16(p_drop), key = dropout_key(seed, (uint32_t)(idx >> 33)), pair = (uint32_t)(idx >> 1);
    const uint32_t h0 = hash_u32(pair ^ key), h1 = hash_u32((pair + 1u) ^ key);
    k[0] = (h0 & 0xffffu) >= thr ? inv_keep : 0.0f; k[1] = (h0 >> 16) >= thr ? inv_keep : 0.0f;
    k[2] = (h1 & 0xffffu) >= thr ? inv_keep : 0.0f; k[3] = (h1 >> 16) >= thr ? inv_keep : 0.0f;
}
// v[e] *= keep(seed, base + e) * inv_keep for e = 0..7; base must be even
__device__ __forceinline__ void dropout_apply8(float (&v)[8], uint64_t seed, uint64_t base, float p_drop, float inv_keep) {
    const uint32_t thr = dropout_thr16(p_drop);
    const uint32_t key = dropout_key(seed, (uint32_t)(base >> 33));
    const uint32_t pair = (uint32_t)(base >> 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t h = hash_u32((pair + q) ^ key);
        v[2 * q] *= (h & 0xffffu) >= thr ? inv_keep : 0.0f;
        v[2 * q + 1] *= (h >> 16) >= thr ? inv_keep : 0.0f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
