// Fused relative-position multi-head self-attention for gfx950 (bf16 storage, f32 softmax / accumulation).
//
// Replaces, per (batch, head), the reference's  score = ((q+u) k^T + shift((q+v) p^T)) / sqrt(d_model);  softmax;  dropout;  @ v
// (code/common/conformer/attention.py:87-101) without ever writing a (B, H, T, T) score / probability tensor:
//   * the shifted positional score `bias` (B,H,T,T) is produced by the positional-score GEMM, whose epilogue writes every raw row r
//     at the constant offset r + 1 - T - the reference's pad-and-reshape trick (attention.py:105-113) is exactly that row-dependent
//     shift of the flat layout; the element (i, i+1) it leaves unwritten is the zero of the padding column and is masked here;
//   * forward: flash-style online softmax over 64-key tiles, one workgroup per (batch, head, 128 query rows), one wave per 32
//     rows.  S^T = K Q^T is computed with the operands swapped so that a lane owns ONE query row (row max / sum are lane-local plus
//     one exchange with the partner half-wave); P goes back to the matrix cores through v_cvt_pk_bf16_f32 + v_permlane32_swap
//     (no LDS round trip); V is consumed with the transpose read ds_read_b64_tr_b16;
//   * backward recomputes the probabilities from the saved log-sum-exp: one kernel per query tile for dQ and d(bias), one per key
//     tile for dK and dV (no atomics, deterministic).
// Dropout uses the library's counter hash on the element index ((b*H + h)*T + i)*T + j, recomputed identically in backward.
#include "common.h"

// 16-bit tensors are passed as element pointers; the kernels' template parameter TA says how the FORWARD tensors (qu, k, v, bias,
// ctx) are encoded: bf16, or fp16 in the fp16-forward mode - there the forward kernel contracts fp16 operands
// (v_mfma_f32_32x32x16_f16), the backward kernels recompute the scores on the same fp16 operands and re-encode k / v / q to bf16
// while staging them for the products with the (always bf16) gradients.
typedef uint16_t h16;
struct AttnArgs {
    const h16* qu; long ldq;           // [B*T][ldq]: q + u_bias, head h at column h*DH
    const h16* k; const h16* v; long ldk;
    const h16* bias;                   // (B,H,T,T) shifted positional score (unscaled)
    h16* ctx; long ldc;                // [B*T][ldc]
    h16* ctx_lo;                       // (hybrid mode, optional; rows as ctx) fp16(ctx_f32 - fp16(ctx_f32)): ctx + ctx_lo is the pair operand of the
                                       // output projection - written here instead of a separate pass over ctx32
    float* ctx32;                      // [B*T][H*DH] f32 copy of ctx before rounding (forward out, backward in): D_i = dctx_i . ctx_i
                                       // enters dS as a small difference (flat softmax: 1/sqrt(d_model) scaling), so it must not
                                       // carry the bf16 rounding of ctx
    float* lse;                        // (B,H,T): log2-domain log-sum-exp of the scaled scores
    const h16* dctx; long lddc;        // backward: gradient of ctx (bf16)
    h16* dqu; long lddq;               // backward outputs (bf16)
    h16* dk; h16* dv; long lddk;
    h16* dbias;                        // (B,H,T,T): gradient of the shifted positional score (bf16)
    float* dsum;                       // (B,H,T): D_i = sum_c dctx * ctx, written by the dQ kernel, read by the dK / dV kernel
    // positional score computed in the kernel (relpos_attn_fwd_kernel<.., POS>): qv = q + v_bias [B*T][ldq], pos = the positional
    // projection [T][ldp] (head h at column h*DH, shared by the batch); bias_out (optional): the (B,H,T,T) shifted score the kernel
    // formed, for backward kernels that still read it
    const h16* qv; const h16* pos; long ldp; h16* bias_out;
    // backward with the positional-score gradients formed in the dQ kernel (relpos_attn_bwd_q_kernel<.., POS>): dqv [B*T][lddqv] =
    // gradient of q + v_bias through the positional score; dpos_part (B, ntile, T, H*DH) bf16: per (batch item, query tile) partial
    // gradient of the positional projection; dqv_fix f32 (B*H, ntile, 2, DH): the two halves of the one row per tile boundary whose
    // gradient comes from two workgroups (summed and rounded once by the dK / dV kernel that follows)
    h16* dqv; long lddqv; h16* dpos_part; float* dqv_fix;
    // ub / vb (optional, f32 [H*DH]): the u / v biases of attention.py:54-55.  When given, `qu` and `qv` both point at the plain query
    // projection q and every kernel forms q + u / q + v while it loads its rows - in f32, rounded to the forward encoding, i.e. the values
    // a separate bias pass (sarssl_bias2) would have stored
    const float* ub; const float* vb;
    // dq_sum (optional, [B*T][lddq_sum]): dqu + dqv, the gradient of the query projection, written by the dK / dV kernel for its 128 rows
    // after the boundary row is final (bf16 + bf16 in f32, rounded: what sarssl_axpby2d makes of the two tensors)
    h16* dq_sum; long lddq_sum;
    int B, H, T;
    float scale, p_drop; unsigned long long seed;
    const unsigned long long* salt;    // device-resident addend of the seed (graph replay), or null
};

#define LOG2E 1.4426950408889634f

// 8 consecutive 16-bit elements + 8 f32 biases -> the 16-bit encoding of the sums
template <typename T>
__device__ __forceinline__ uint4 addb8(const uint4& q, const float* __restrict__ b) {
    f8 v = unpack8<T>(q);
    const float4 b0 = *(const float4*)b, b1 = *(const float4*)(b + 4);
    v.v[0] += b0.x; v.v[1] += b0.y; v.v[2] += b0.z; v.v[3] += b0.w;
    v.v[4] += b1.x; v.v[5] += b1.y; v.v[6] += b1.z; v.v[7] += b1.w;
    return pack8<T>(v);
}

// v_exp_f32 as is: exp2f() wraps it in a denormal-range rescue (compare, two selects, add, ldexp: 7 instructions per probability);
// a probability below 2^-126 is zero for every purpose here
__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }

template <int PT>
__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* s, int kbase, int r0, int lane) {      // as gemm.hip: [k][row] tile
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const int col = r0 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
    union { s16x4 v[2]; bf16x8 b; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = kbase + (lane >> 5) * 8 + h * 4 + ((lane & 15) >> 2);
        u.v[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + k * PT + col));
    }
    return u.b;
}

// Accumulator fragment (lane = row, 16 values = columns 8g + 4*half + e of a 32-column block) -> the two MFMA operand fragments
// (8 consecutive columns at offset half*8 of each 16-column step) of the same 32 columns, encoded as T.
template <typename T>
__device__ __forceinline__ void acc_to_operand(const f32x16& p, bf16x8 (&out)[2]) {
    uint32_t w[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        w[g][0] = H16<T>::pack(p[4 * g + 0], p[4 * g + 1]);
        w[g][1] = H16<T>::pack(p[4 * g + 2], p[4 * g + 3]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        union { uint32_t u[4]; bf16x8 b; } f;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            // upper half-wave of w[2s] <-> lower half-wave of w[2s+1]
            auto r = __builtin_amdgcn_permlane32_swap(w[2 * s][x], w[2 * s + 1][x], false, false);
            f.u[x] = r[0];
            f.u[2 + x] = r[1];
        }
        out[s] = f.b;
    }
}

__device__ __forceinline__ float half_swap_f(float v) { return __shfl_xor(v, 32, 64); }

// keep-scale of attention probability (bh, i, j)
__device__ __forceinline__ float attn_keep(const AttnArgs& a, unsigned long long rowbase, int j, float inv_keep) {
    return dropout_scale(salted_seed(a.seed, a.salt), rowbase + (unsigned long long)j, a.p_drop, inv_keep);
}
// The same decisions (dropout_scale's hash, common.h) for the way these kernels walk the score matrix.  dropout_scale pays two hashes
// per element - the key of the index's high word and the pair hash - although one pair hash decides two neighbouring elements and
// the key only changes every 2^33 elements: with dropout on, the three attention kernels spent 170 us per step on it (tools/bench_attn.py).
struct AttnDrop {
    uint64_t seed; uint32_t thr, hi0, key0; float inv_keep;
    __device__ __forceinline__ void init(const AttnArgs& a, uint64_t idx0, float ik) {
        seed = salted_seed(a.seed, a.salt); thr = dropout_thr16(a.p_drop); inv_keep = ik;
        hi0 = (uint32_t)(idx0 >> 33); key0 = dropout_key(seed, hi0);
    }
    __device__ __forceinline__ uint32_t key(uint64_t idx) const {
        const uint32_t hi = (uint32_t)(idx >> 33);
        return hi == hi0 ? key0 : dropout_key(seed, hi);
    }
    // 4 consecutive elements from idx (a multiple of 4): two pair hashes
    __device__ __forceinline__ void keep4(uint64_t idx, float (&k)[4]) const {
        const uint32_t ky = key(idx), pair = (uint32_t)(idx >> 1);
        const uint32_t h0 = hash_u32(pair ^ ky), h1 = hash_u32((pair + 1u) ^ ky);
        k[0] = (h0 & 0xffffu) >= thr ? inv_keep : 0.f; k[1] = (h0 >> 16) >= thr ? inv_keep : 0.f;
        k[2] = (h1 & 0xffffu) >= thr ? inv_keep : 0.f; k[3] = (h1 >> 16) >= thr ? inv_keep : 0.f;
    }
    // one element: one hash
    __device__ __forceinline__ float keep1(uint64_t idx) const {
        const uint32_t h = hash_u32((uint32_t)(idx >> 1) ^ key(idx));
        return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= thr ? inv_keep : 0.f;
    }
};

// ------------------------------------------------------------------------------------------------------------------- forward
// (d_head <= 64: capped at 256 registers - 243 used, no spills - so that TWO workgroups share a CU: the kernel is a chain of
//  latencies (K / V / bias tiles, softmax exchanges) and ran one wave per SIMD at 263 registers)
// POS (round 4; T <= 256): the shifted positional score is not read from memory but formed here, on the matrix cores, before the key
// loop.  The reference's pad-and-reshape shift (attention.py:105-113) is a bijection of the unshifted product R[r][m] = (q_r + v) . p_m:
//     bias[i][j] = R[i][T-1-i+j]   for j <= i,      0 for j = i+1,      R[i+1][j-i-2]   for j >= i+2
// so a query tile needs R of its rows (the "low" part) and of its rows + 1 (the "up" part) against all T positions: the position
// tiles stream through the K buffer like key tiles, S'^T = P Q^T puts the query row on the lane as for the content score, and every
// value is scattered to its shifted column of a [128][T] LDS tile (2-byte stores) that the key loop then reads in place of the staged
// bias tile.  Per wave and position tile only the part that can be valid for its 32 rows is computed.  What this removes per layer: the
// batched positional-score GEMM launch (K = d_head: 53 us for a 33 MB output written element by element in the shifted layout) and
// the kernel's own 33 MB bias read.
// NW = 8 (d_head 64, POS): ONE workgroup of eight waves takes all 256 query rows of a (batch, head) - the [256][T] slab + one K / V tile
// are 157 KB of LDS, two waves share a SIMD (one on the matrix cores while the other is in its softmax / scatter / hash stretch) and
// every K / V tile is staged once per (batch, head) instead of once per 128 rows.
// LONGT (round 6; POS, T > 256, four waves): the slab holds the shifted score of ONE 256-key block at a time - per block the position
// tiles that can contribute to it (a diagonal band of the (row, position) plane) stream from memory through the K / V buffers, the block
// is written out for the backward kernels (bias_out) and its four key tiles run; the online softmax carries across blocks.  Replaces the
// batched positional-score GEMM (150 MB written in the shifted layout at B' = 48, T = 624) and this kernel's read of it.
template <int DH, typename TA, bool POS = false, int NW = 4, bool LONGT = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(((DH <= 64 && !POS) || NW == 8) ? 2 : 1))) void relpos_attn_fwd_kernel(AttnArgs a) {
    static_assert(!LONGT || (POS && NW == 4), "long-sequence form: in-kernel positional score, four waves");
    constexpr int TQ = 32 * NW, TK = 64, NT = 64 * NW;
#ifdef ATTN_EXP_1WG
    constexpr int PK = DH + 8, PV = DH + 32, PB = 256 + 8;
#else
    constexpr int PK = DH + 8, PV = DH + 32, PB = POS ? (256 + 8) : (TK + 8);
#endif
    constexpr int CPR = DH / 8;                               // 16-byte chunks per K / V row
    __shared__ __attribute__((aligned(16))) uint16_t smem[TK * PK + TK * PV + TQ * PB];
    uint16_t* sK = smem;
    uint16_t* sV = sK + TK * PK;
    uint16_t* sB = sV + TK * PV;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H, T = a.T;
    const int i0 = blockIdx.x * TQ;
    const int i = i0 + wave * 32 + (lane & 31);               // this lane's query row
    const bool row_ok = i < T;
    const h16* K = a.k + (long)b * T * a.ldk + h * DH;
    const h16* V = a.v + (long)b * T * a.ldk + h * DH;
    const h16* Bi = a.bias + (long)bh * T * T;
    bf16x8 fq[DH / 16];
    {
        const h16* q = a.qu + ((long)b * T + (row_ok ? i : 0)) * a.ldq + h * DH + half * 8;
#pragma unroll
        for (int s = 0; s < DH / 16; ++s) {
            uint4 u = row_ok ? *(const uint4*)(q + s * 16) : make_uint4(0, 0, 0, 0);
            if (a.ub && row_ok) u = addb8<TA>(u, a.ub + h * DH + s * 16 + half * 8);
            fq[s] = __builtin_bit_cast(bf16x8, u);
        }
    }
    f32x16 o[DH / 32];
#pragma unroll
    for (int c = 0; c < DH / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = a.scale * LOG2E;
    const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const unsigned long long rowbase = ((unsigned long long)bh * T + (unsigned long long)(row_ok ? i : 0)) * (unsigned long long)T;
    AttnDrop drop;
    drop.init(a, rowbase, inv_keep);

    // K, V (64 keys x DH) and the bias tile (128 rows x 64 keys) go through registers: the next tile's loads are in flight while
    // this one is on the matrix cores
    constexpr int NKV = TK * CPR / NT, NBI = TQ * (TK / 8) / NT, NRP = 256 * CPR / NT;
    static_assert(NKV >= 1 && NRP >= 2, "tile loads per thread");
    uint4 rk[NKV], rv[NKV], rbi[NBI];
    auto load_tile = [&](int j0) {
#pragma unroll
        for (int c = 0; c < NKV; ++c) {
            const int cid = tid + NT * c, row = cid / CPR, c8 = cid % CPR;
            rk[c] = make_uint4(0, 0, 0, 0); rv[c] = rk[c];
            if (j0 + row < T) {
                rk[c] = *(const uint4*)(K + (long)(j0 + row) * a.ldk + c8 * 8);
                rv[c] = *(const uint4*)(V + (long)(j0 + row) * a.ldk + c8 * 8);
            }
        }
        if constexpr (!POS) {
#pragma unroll
            for (int c = 0; c < NBI; ++c) {
                const int cid = tid + NT * c, row = cid >> 3, c8 = cid & 7;
                rbi[c] = make_uint4(0, 0, 0, 0);
                if (i0 + row < T && j0 + c8 * 8 < T) rbi[c] = *(const uint4*)(Bi + (long)(i0 + row) * T + j0 + c8 * 8);
            }
        }
    };
    uint4 rp[(POS && !LONGT) ? NRP : 1];                      // POS: the head's whole positional projection (T <= 256 rows), one request burst
    if constexpr (POS && !LONGT) {
        const h16* Pm = a.pos + h * DH;
#pragma unroll
        for (int c = 0; c < NRP; ++c) {
            const int row = tid / CPR + c * (NT / CPR);
            rp[c] = row < T ? *(const uint4*)(Pm + (long)row * a.ldp + (tid % CPR) * 8) : make_uint4(0, 0, 0, 0);
        }
    }
    load_tile(0);
    bf16x8 fqv[POS ? DH / 16 : 1], fqvn[POS ? DH / 16 : 1];
    if constexpr (POS) {
        const bool nx = row_ok && (i + 1 < T);
        const h16* q0 = a.qv + ((long)b * T + (row_ok ? i : 0)) * a.ldq + h * DH + half * 8;
#pragma unroll
        for (int s = 0; s < DH / 16; ++s) {
            uint4 w0 = row_ok ? *(const uint4*)(q0 + s * 16) : make_uint4(0, 0, 0, 0);
            uint4 w1 = nx ? *(const uint4*)(q0 + a.ldq + s * 16) : make_uint4(0, 0, 0, 0);
            if (a.vb) {
                if (row_ok) w0 = addb8<TA>(w0, a.vb + h * DH + s * 16 + half * 8);
                if (nx) w1 = addb8<TA>(w1, a.vb + h * DH + s * 16 + half * 8);
            }
            fqv[s] = __builtin_bit_cast(bf16x8, w0);
            fqvn[s] = __builtin_bit_cast(bf16x8, w1);
        }
    }
    if constexpr (POS && !LONGT) {
        const int il = wave * 32 + (lane & 31), r0 = i0 + wave * 32;
        if (half == 0 && row_ok && i + 1 < T) sB[il * PB + i + 1] = 0;          // the zero of the padding column (masked in the key loop anyway)
        // lane-constant parts of the shifted columns: low  R[i][m]   -> column m - (T-1) + i   (valid when >= 0)
        //                                             up   R[i+1][m] -> column m + i + 2       (valid when <= T-1)
        uint16_t* rowp = sB + il * PB;
        const int cl = i - (T - 1), cu = i + 2;
        uint16_t* sP = sK;                                                      // [128 positions][PK]: the K and V buffers together
        static_assert(TK * PK + TK * PV >= 128 * PK, "position stage must fit the K + V buffers");
#pragma unroll
        for (int stage = 0; stage < 2; ++stage) {
            if (stage * 128 < T) {
#pragma unroll
                for (int c = 0; c < NRP / 2; ++c) {
                    const int row = tid / CPR + c * (NT / CPR);
                    *(uint4*)&sP[row * PK + (tid % CPR) * 8] = rp[stage * (NRP / 2) + c];
                }
                __syncthreads();
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int m0 = stage * 128 + f * 32;
                    // wave-uniform: can any of this wave's rows r0 .. r0+31 have a valid entry among positions m0 .. m0+31 ?
                    const bool need[2] = {m0 < T && (m0 + 31) >= (T - 1 - (r0 + 31)), m0 <= (T - 3 - r0)};
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
                        if (!need[part]) continue;
                        f32x16 s;
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
                        for (int st = 0; st < DH / 16; ++st) {
                            const bf16x8 pf = *(const bf16x8*)&sP[(f * 32 + (lane & 31)) * PK + st * 16 + half * 8];
                            s = mfma16<TA>(pf, part == 0 ? fqv[st] : fqvn[st], s);
                        }
                        if (row_ok) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int e = 0; e < 4; e += 2) {
                                    const int m = m0 + 8 * g + 4 * half + e;
                                    const uint32_t w = H16<TA>::pack(s[4 * g + e], s[4 * g + e + 1]);
                                    if (part == 0) {
                                        if (m + cl >= 0 && m < T) rowp[m + cl] = (uint16_t)w;
                                        if (m + 1 + cl >= 0 && m + 1 < T) rowp[m + 1 + cl] = (uint16_t)(w >> 16);
                                    } else {
                                        if (m + cu < T) rowp[m + cu] = (uint16_t)w;
                                        if (m + 1 + cu < T) rowp[m + 1 + cu] = (uint16_t)(w >> 16);
                                    }
                                }
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (a.bias_out) {                                                       // (T % 8 == 0, T <= 256)
            h16* Bo = a.bias_out + (long)bh * T * T;
            const int cpr = T >> 3;
            for (int cid = tid; cid < TQ * cpr; cid += NT) {
                const int row = cid / cpr, c8 = cid - row * cpr;
                if (i0 + row < T) *(uint4*)(Bo + (long)(i0 + row) * T + c8 * 8) = *(const uint4*)&sB[row * PB + c8 * 8];
            }
        }
    }
    // one 64-key tile: jc = column of key j0 in the slab (POS: j0, LONGT: j0 - first key of the block; staged bias tile: 0)
    auto key_tile = [&](const int j0, const int jc) {
#pragma unroll
        for (int c = 0; c < NKV; ++c) {
            const int cid = tid + NT * c, row = cid / CPR, c8 = cid % CPR;
            *(uint4*)&sK[row * PK + c8 * 8] = rk[c];
            *(uint4*)&sV[row * PV + c8 * 8] = rv[c];
        }
        if constexpr (!POS) {
#pragma unroll
            for (int c = 0; c < NBI; ++c) {
                const int cid = tid + NT * c, row = cid >> 3, c8 = cid & 7;
                *(uint4*)&sB[row * PB + c8 * 8] = rbi[c];
            }
        }
        __syncthreads();
        if (j0 + TK < T) load_tile(j0 + TK);
        // ---- S^T = K Q^T: lane = query row, registers = keys
        f32x16 s[2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[f][r] = 0.f;
#pragma unroll
        for (int st = 0; st < DH / 16; ++st)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 kf = *(const bf16x8*)&sK[(f * 32 + (lane & 31)) * PK + st * 16 + half * 8];
                s[f] = mfma16<TA>(kf, fq[st], s[f]);
            }
        // ---- scores in the log2 domain, masks, running max
        float mx = -INFINITY;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int jl = f * 32 + 8 * g + 4 * half;
                const uint2 bu = *(const uint2*)&sB[(wave * 32 + (lane & 31)) * PB + jc + jl];
                const float bv[4] = {H16<TA>::lo(bu.x), H16<TA>::hi(bu.x), H16<TA>::lo(bu.y), H16<TA>::hi(bu.y)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = j0 + jl + e;
                    float x = (s[f][4 * g + e] + (j == i + 1 ? 0.f : bv[e])) * sc2;
                    x = j < T ? x : -INFINITY;
                    s[f][4 * g + e] = x;
                    mx = fmaxf(mx, x);
                }
            }
        mx = fmaxf(mx, half_swap_f(mx));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2_raw(m_run - m_new);
        m_run = m_new;
        float rs = 0.f;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp2_raw(s[f][r] - m_new);
                rs += p;
                s[f][r] = p;
            }
        l_run = l_run * alpha + rs;
#pragma unroll
        for (int c = 0; c < DH / 32; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c][r] *= alpha;
        if (a.p_drop > 0.f) {
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float kp[4];
                    drop.keep4(rowbase + (unsigned long long)(j0 + f * 32 + 8 * g + 4 * half), kp);     // (T % 8 == 0: rowbase is a multiple of 8)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[f][4 * g + e] *= kp[e];
                }
        }
        // ---- O^T += V^T P^T: lane = query row, registers = head channels
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            bf16x8 pf[2];
            acc_to_operand<TA>(s[f], pf);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int c = 0; c < DH / 32; ++c) {
                    const bf16x8 vf = tr_frag<PV>(sV, f * 32 + st * 16, c * 32, lane);
                    o[c] = mfma16<TA>(vf, pf[st], o[c]);
                }
        }
        __syncthreads();
    };
    if constexpr (!LONGT) {
        for (int j0 = 0; j0 < T; j0 += TK) key_tile(j0, POS ? j0 : 0);
    } else {
        const int il = wave * 32 + (lane & 31), r0 = i0 + wave * 32;
        uint16_t* rowp = sB + il * PB;
        uint16_t* sP = sK;                                                      // [128 positions][PK]: the K and V buffers together
        static_assert(TK * PK + TK * PV >= 128 * PK, "position stage must fit the K + V buffers");
        const h16* Pm = a.pos + h * DH;
        for (int jb = 0; jb < T; jb += 256) {
            const int jend = min(jb + 256, T);                                  // keys jb .. jend - 1
            // shifted columns relative to the block: low  R[i][m] -> column m - (T-1) + i - jb,   up  R[i+1][m] -> column m + i + 2 - jb
            const int cl = i - (T - 1) - jb, cu = i + 2 - jb;
            if (half == 0 && row_ok && i + 1 >= jb && i + 1 < jend) rowp[i + 1 - jb] = 0;     // the zero of the padding column
            for (int ms = 0; ms < T; ms += 128) {
                // workgroup-uniform: can positions ms .. ms+127 reach columns of this block for any of the tile's rows i0 .. i0+TQ-1 ?
                //   low: m + i in [jb + T-1, jend-1 + T-1];   up: m + i + 2 in [jb, jend-1]
                const int slo = ms + i0, shi = ms + 127 + i0 + TQ - 1;
                const bool wl_ = shi >= jb + T - 1 && slo <= jend - 1 + T - 1, wu_ = shi + 2 >= jb && slo + 2 <= jend - 1;
                if (!(wl_ || wu_)) continue;
#pragma unroll
                for (int c = 0; c < NRP / 2; ++c) {
                    const int row = tid / CPR + c * (NT / CPR);
                    *(uint4*)&sP[row * PK + (tid % CPR) * 8] = (ms + row < T) ? *(const uint4*)(Pm + (long)(ms + row) * a.ldp + (tid % CPR) * 8)
                                                                               : make_uint4(0, 0, 0, 0);
                }
                __syncthreads();
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int m0 = ms + f * 32;
                    const int wlo = m0 + r0, whi = m0 + 31 + r0 + 31;             // range of m + i over the wave's rows and this sub-tile
                    const bool need[2] = {m0 < T && whi >= jb + T - 1 && wlo <= jend - 1 + T - 1, m0 < T && whi + 2 >= jb && wlo + 2 <= jend - 1};
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
                        if (!need[part]) continue;
                        f32x16 s;
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
                        for (int st = 0; st < DH / 16; ++st) {
                            const bf16x8 pf = *(const bf16x8*)&sP[(f * 32 + (lane & 31)) * PK + st * 16 + half * 8];
                            s = mfma16<TA>(pf, part == 0 ? fqv[st] : fqvn[st], s);
                        }
                        if (row_ok) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int e = 0; e < 4; e += 2) {
                                    const int m = m0 + 8 * g + 4 * half + e;
                                    const uint32_t w = H16<TA>::pack(s[4 * g + e], s[4 * g + e + 1]);
                                    const int c0 = m + (part == 0 ? cl : cu), c1 = c0 + 1;
                                    if (part == 0) {          // valid: m <= T-1 (then j <= i) and j = c + jb >= 0
                                        if (c0 >= 0 && c0 + jb < jend && m < T) rowp[c0] = (uint16_t)w;
                                        if (c1 >= 0 && c1 + jb < jend && m + 1 < T) rowp[c1] = (uint16_t)(w >> 16);
                                    } else {                  // valid: j = c + jb <= T-1 (j >= i + 2 > 0)
                                        if (c0 >= 0 && c0 + jb < jend) rowp[c0] = (uint16_t)w;
                                        if (c1 >= 0 && c1 + jb < jend) rowp[c1] = (uint16_t)(w >> 16);
                                    }
                                }
                        }
                    }
                }
                __syncthreads();
            }
            if (a.bias_out) {                                                   // (T % 8 == 0)
                h16* Bo = a.bias_out + (long)bh * T * T;
                const int cpr = (jend - jb) >> 3;
                for (int cid = tid; cid < TQ * cpr; cid += NT) {
                    const int row = cid / cpr, c8 = cid - row * cpr;
                    if (i0 + row < T) *(uint4*)(Bo + (long)(i0 + row) * T + jb + c8 * 8) = *(const uint4*)&sB[row * PB + c8 * 8];
                }
            }
            for (int j0 = jb; j0 < jend; j0 += TK) key_tile(j0, j0 - jb);
        }
    }
    const float l_tot = l_run + half_swap_f(l_run);
    const float inv_l = 1.0f / l_tot;
    if (row_ok) {
        if (half == 0 && a.lse) a.lse[(long)bh * T + i] = m_run + log2f(l_tot);
        h16* out = a.ctx + ((long)b * T + i) * a.ldc + h * DH;
        float* out32 = a.ctx32 ? a.ctx32 + ((long)b * T + i) * ((long)a.H * DH) + h * DH : nullptr;
#pragma unroll
        for (int c = 0; c < DH / 32; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = make_float4(o[c][4 * g + 0] * inv_l, o[c][4 * g + 1] * inv_l, o[c][4 * g + 2] * inv_l, o[c][4 * g + 3] * inv_l);
                uint2 u;
                u.x = H16<TA>::pack(v.x, v.y);
                u.y = H16<TA>::pack(v.z, v.w);
                *(uint2*)(out + c * 32 + 8 * g + 4 * half) = u;
                if (out32) *(float4*)(out32 + c * 32 + 8 * g + 4 * half) = v;
                if (a.ctx_lo) {
                    uint2 l;
                    l.x = H16<TA>::pack(v.x - H16<TA>::lo(u.x), v.y - H16<TA>::hi(u.x));
                    l.y = H16<TA>::pack(v.z - H16<TA>::lo(u.y), v.w - H16<TA>::hi(u.y));
                    *(uint2*)(a.ctx_lo + ((long)b * T + i) * a.ldc + h * DH + c * 32 + 8 * g + 4 * half) = l;
                }
            }
    }
}

// ---------------------------------------------------------------------------- backward, part 1: D, dQ and d(bias), per query tile
// POS (round 4, T <= 256): the tile's whole [128][T] slab of the shifted positional score is loaded into LDS once, every key tile's dS
// overwrites the 64 columns it has just read, and after the key loop the slab IS the tile's gradient of the shifted score - so the two
// products that used to follow as batched GEMM launches over a (B,H,T,T) d(bias) tensor (after an un-shift pass over it) run here, on
// operands gathered from the slab through the inverse of the shift:
//     dR[r][m] = dS[r][m-(T-1)+r]  (m >= T-1-r),   dS[r-1][m+r+1]  (m <= T-2-r)          (R[r][m] = (q_r + v) . p_m, unshifted)
//     dqv[r]  = sum_m dR[r][m] p_m      (lane = query row, 8 consecutive m per operand register quad: 2-byte LDS gathers)
//     dpos[m] = sum_r dR[r][m] qv_r     (lane = position, 8 consecutive rows: gathers at the constant stride pitch + 1)
// A tile's last row feeds row i0+128 of the next tile (its "upper" part): that one row's two halves go to dqv_fix in f32.
template <int DH, typename TA, bool POS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((DH <= 64 && !POS) ? 2 : 1))) void relpos_attn_bwd_q_kernel(AttnArgs a) {
    constexpr int TQ = 128, TK = 64;
    constexpr int PK = DH + 8, PT = DH + 32, PB = POS ? (256 + 8) : (TK + 8);
    constexpr int CPR = DH / 8;
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * TK * PK + TK * PT + (TQ + (POS ? 1 : 0)) * PB];
    uint16_t* sK = smem;                 // [key][c]  (b128 fragment reads: S)
    uint16_t* sV = sK + TK * PK;         // [key][c]  (dP)
    uint16_t* sKt = sV + TK * PK;        // [key][c] with the transpose-read pitch (dQ += dS K)
    uint16_t* sB = sKt + TK * PT + (POS ? PB : 0);   // bias tile in, d(bias) tile out (POS: the whole slab, behind one row of zeros = "row -1")
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H, T = a.T;
    const int i0 = blockIdx.x * TQ;
    const int i = i0 + wave * 32 + (lane & 31);
    const bool row_ok = i < T;
    const h16* K = a.k + (long)b * T * a.ldk + h * DH;
    const h16* V = a.v + (long)b * T * a.ldk + h * DH;
    const h16* Bi = a.bias + (long)bh * T * T;
    h16* dBi = POS ? nullptr : a.dbias + (long)bh * T * T;
    bf16x8 fq[DH / 16], fdo[DH / 16];
    float dpart = 0.f;
    {
        const h16* q = a.qu + ((long)b * T + (row_ok ? i : 0)) * a.ldq + h * DH + half * 8;
        const h16* d = a.dctx + ((long)b * T + (row_ok ? i : 0)) * a.lddc + h * DH + half * 8;
        // D_i = dctx_i . ctx_i (the unrounded f32 context): this lane holds half of the row's dctx chunks anyway - the stand-alone
        // pass over dctx / ctx32 (17-19 us per layer) is gone; the row's value is stored for the dK / dV kernel that follows
        const float* c32 = a.ctx32 + ((long)b * T + (row_ok ? i : 0)) * ((long)a.H * DH) + h * DH + half * 8;
#pragma unroll
        for (int s = 0; s < DH / 16; ++s) {
            uint4 qw = row_ok ? *(const uint4*)(q + s * 16) : make_uint4(0, 0, 0, 0);
            if (a.ub && row_ok) qw = addb8<TA>(qw, a.ub + h * DH + s * 16 + half * 8);
            fq[s] = __builtin_bit_cast(bf16x8, qw);
            const uint4 du = row_ok ? *(const uint4*)(d + s * 16) : make_uint4(0, 0, 0, 0);
            fdo[s] = __builtin_bit_cast(bf16x8, du);
            const float4 c0 = *(const float4*)(c32 + s * 16), c1 = *(const float4*)(c32 + s * 16 + 4);
            dpart += bf16_bits_to_f32(du.x & 0xffffu) * c0.x + __uint_as_float(du.x & 0xffff0000u) * c0.y +
                     bf16_bits_to_f32(du.y & 0xffffu) * c0.z + __uint_as_float(du.y & 0xffff0000u) * c0.w +
                     bf16_bits_to_f32(du.z & 0xffffu) * c1.x + __uint_as_float(du.z & 0xffff0000u) * c1.y +
                     bf16_bits_to_f32(du.w & 0xffffu) * c1.z + __uint_as_float(du.w & 0xffff0000u) * c1.w;
        }
    }
    const float lse = row_ok ? a.lse[(long)bh * T + i] : 0.f;
    const float dsum = dpart + half_swap_f(dpart);
    if (row_ok && half == 0) a.dsum[(long)bh * T + i] = dsum;
    f32x16 dq[DH / 32];
#pragma unroll
    for (int c = 0; c < DH / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[c][r] = 0.f;
    const float sc2 = a.scale * LOG2E;
    const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const unsigned long long rowbase = ((unsigned long long)bh * T + (unsigned long long)(row_ok ? i : 0)) * (unsigned long long)T;
    AttnDrop drop;
    drop.init(a, rowbase, inv_keep);

    constexpr int NKV = TK * CPR / 256, NBI = TQ * (TK / 8) / 256;
    uint4 rk[NKV], rv[NKV], rbi[NBI];
    auto load_tile = [&](int j0) {
#pragma unroll
        for (int c = 0; c < NKV; ++c) {
            const int cid = tid + 256 * c, row = cid / CPR, c8 = cid % CPR;
            rk[c] = make_uint4(0, 0, 0, 0); rv[c] = rk[c];
            if (j0 + row < T) {
                rk[c] = *(const uint4*)(K + (long)(j0 + row) * a.ldk + c8 * 8);
                rv[c] = *(const uint4*)(V + (long)(j0 + row) * a.ldk + c8 * 8);
            }
        }
#pragma unroll
        for (int c = 0; c < NBI; ++c) {
            const int cid = tid + 256 * c, row = cid >> 3, c8 = cid & 7;
            rbi[c] = make_uint4(0, 0, 0, 0);
            if (i0 + row < T && j0 + c8 * 8 < T) rbi[c] = *(const uint4*)(Bi + (long)(i0 + row) * T + j0 + c8 * 8);
        }
    };
    load_tile(0);
    if constexpr (POS) {                                          // "row -1" of the slab
        if (tid < PB / 8) *(uint4*)&sB[-PB + tid * 8] = make_uint4(0, 0, 0, 0);
    }
    for (int j0 = 0; j0 < T; j0 += TK) {
#pragma unroll
        for (int c = 0; c < NKV; ++c) {
            const int cid = tid + 256 * c, row = cid / CPR, c8 = cid % CPR;
            *(uint4*)&sK[row * PK + c8 * 8] = rk[c];                         // as saved: the scores are recomputed on the forward's operands
            *(uint4*)&sKt[row * PT + c8 * 8] = recode8<TA, bf16>(rk[c]);     // bf16 copies meet the bf16 gradients (dQ += dS K, dP = dO V^T)
            *(uint4*)&sV[row * PK + c8 * 8] = recode8<TA, bf16>(rv[c]);
        }
#pragma unroll
        for (int c = 0; c < NBI; ++c) {                           // (POS: into columns j0 .. j0+63 of the slab, which keeps every tile's dS)
            const int cid = tid + 256 * c, row = cid >> 3, c8 = cid & 7;
            *(uint4*)&sB[row * PB + (POS ? j0 : 0) + c8 * 8] = rbi[c];
        }
        __syncthreads();
        if (j0 + TK < T) load_tile(j0 + TK);
        f32x16 s[2], dp[2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[f][r] = 0.f; dp[f][r] = 0.f; }
#pragma unroll
        for (int st = 0; st < DH / 16; ++st)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 kf = *(const bf16x8*)&sK[(f * 32 + (lane & 31)) * PK + st * 16 + half * 8];
                const bf16x8 vf = *(const bf16x8*)&sV[(f * 32 + (lane & 31)) * PK + st * 16 + half * 8];
                s[f] = mfma16<TA>(kf, fq[st], s[f]);
                dp[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, fdo[st], dp[f], 0, 0, 0);
            }
        // dScore = scale * p * (keep * dP - D)   (lane = query row, registers = keys); staged to LDS as the d(bias) tile
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int jl = f * 32 + 8 * g + 4 * half;
                uint16_t* bp = &sB[(wave * 32 + (lane & 31)) * PB + (POS ? j0 : 0) + jl];
                const uint2 bu = *(const uint2*)bp;
                const float bv[4] = {H16<TA>::lo(bu.x), H16<TA>::hi(bu.x), H16<TA>::lo(bu.y), H16<TA>::hi(bu.y)};
                float ds[4];
                float kp[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (a.p_drop > 0.f) drop.keep4(rowbase + (unsigned long long)(j0 + jl), kp);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = j0 + jl + e;
                    const float x = (s[f][4 * g + e] + (j == i + 1 ? 0.f : bv[e])) * sc2;
                    const float p = j < T ? exp2_raw(x - lse) : 0.f;
                    const float keep = kp[e];
                    ds[e] = a.scale * p * (keep * dp[f][4 * g + e] - dsum);
                    s[f][4 * g + e] = ds[e];
                }
                uint2 ou;
                ou.x = pack2_bf16(ds[0], ds[1]);
                ou.y = pack2_bf16(ds[2], ds[3]);
                *(uint2*)bp = ou;                              // each lane overwrites exactly the 4 bias values it has just read
            }
        // dQ^T += K^T dS^T
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            bf16x8 pf[2];
            acc_to_operand<bf16>(s[f], pf);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int c = 0; c < DH / 32; ++c) {
                    const bf16x8 kf = tr_frag<PT>(sKt, f * 32 + st * 16, c * 32, lane);
                    dq[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, pf[st], dq[c], 0, 0, 0);
                }
        }
        __syncthreads();
        if constexpr (!POS) {
            // d(bias) tile -> global, 16-byte row pieces (the element (i, i+1) is the padding zero of the shift: its gradient is dropped
            // by the consumer, which never reads it back through the shifted addressing)
#pragma unroll
            for (int c = 0; c < TQ * (TK / 8) / 256; ++c) {
                const int cid = tid + 256 * c, row = cid >> 3, c8 = cid & 7;
                if (i0 + row < T && j0 + c8 * 8 < T) *(uint4*)(dBi + (long)(i0 + row) * T + j0 + c8 * 8) = *(const uint4*)&sB[row * PB + c8 * 8];
            }
            __syncthreads();
        }
    }
    if (row_ok) {
        h16* out = a.dqu + ((long)b * T + i) * a.lddq + h * DH;
#pragma unroll
        for (int c = 0; c < DH / 32; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u;
                u.x = pack2_bf16(dq[c][4 * g + 0], dq[c][4 * g + 1]);
                u.y = pack2_bf16(dq[c][4 * g + 2], dq[c][4 * g + 3]);
                *(uint2*)(out + c * 32 + 8 * g + 4 * half) = u;
            }
    }
    if constexpr (POS) {
        // sB now holds dS[il][j] (bf16) for the tile's 128 rows and all T keys, rows PB apart.  Re-pitched to T (through registers, in
        // place) the slab is the un-shifted gradient laid out flat:  dR[i0 + rl][m]  sits at halfword  F = rl * (T+1) + m + i0 - (T-1)
        // - the reference's pad-and-reshape read backwards - with row -1 (zeros) in front for the entries the previous tile owns.
        const int il = wave * 32 + (lane & 31);
        const int ntile = gridDim.x, q = blockIdx.x;
        const int r1 = i0 + TQ;                                   // first row of the next tile: its "upper" part lives in this tile's last row
        const bool has_extra = r1 < T;
        constexpr int NPART = 256 / DH;
        const int xc = tid % DH, xp = tid / DH;
        float xacc = 0.f;
        // the head's positional projection and the tile's 129 qv rows: one request burst, consumed after the re-pitch
        const h16* Pm = a.pos + h * DH;
        const h16* QV = a.qv + (long)b * T * a.ldq + h * DH;
        constexpr int NQV = ((TQ + 16) * CPR + 255) / 256;
        uint4 rp[CPR], rqv[NQV];
#pragma unroll
        for (int c = 0; c < CPR; ++c) {
            const int row = tid / CPR + c * (256 / CPR);
            rp[c] = row < T ? *(const uint4*)(Pm + (long)row * a.ldp + (tid % CPR) * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < NQV; ++c) {
            const int cid = tid + 256 * c, row = cid / CPR, c8 = cid % CPR;
            rqv[c] = (row <= TQ && i0 + row < T) ? *(const uint4*)(QV + (long)(i0 + row) * a.ldq + c8 * 8) : make_uint4(0, 0, 0, 0);
        }
        {
            const int cpr = T >> 3, total = TQ * cpr;
            uint4 t[16];                                          // (128 rows x T <= 256 columns: at most 16 pieces per thread)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int cid = tid + 256 * u, row = cid / cpr, c8 = cid - row * cpr;
                t[u] = cid < total ? *(const uint4*)&sB[row * PB + c8 * 8] : make_uint4(0, 0, 0, 0);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int cid = tid + 256 * u, row = cid / cpr, c8 = cid - row * cpr;
                if (cid < total) *(uint4*)&sB[row * T + c8 * 8] = t[u];
            }
        }
        // ---- dqv^T[c][r] = sum_m P^T[c][m] dR[r][m]: a lane's 8 consecutive m are 8 consecutive halfwords of the flat slab at an
        //      arbitrary 2-byte phase - five aligned dwords and a funnel shift
        f32x16 dqv[DH / 32];
#pragma unroll
        for (int c = 0; c < DH / 32; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqv[c][r] = 0.f;
        uint16_t* sP = sK;                                        // [128 positions][PT] bf16
        static_assert(2 * TK * PK + TK * PT >= 128 * PT, "position stage must fit the K / V buffers");
        const uint32_t* sBw = (const uint32_t*)sB;
        const int fa0 = il * T + i - (T - 1);                     // + m
        const uint32_t sh = (uint32_t)(fa0 & 1) * 16u;            // (m advances in steps of 8: the phase is the lane's for the whole pass)
#pragma unroll
        for (int stage = 0; stage < 2; ++stage) {
            if (stage * 128 < T) {
#pragma unroll
                for (int c = 0; c < CPR / 2; ++c) {
                    const int row = tid / CPR + c * (256 / CPR);
                    *(uint4*)&sP[row * PT + (tid % CPR) * 8] = recode8<TA, bf16>(rp[stage * (CPR / 2) + c]);
                }
                __syncthreads();
                for (int ks = 0; ks < 8; ++ks) {
                    const int m0 = stage * 128 + ks * 16 + half * 8;
                    if (stage * 128 + ks * 16 >= T) break;
                    const int dw = (fa0 + m0) >> 1;
                    uint32_t dd[5];
#pragma unroll
                    for (int x = 0; x < 5; ++x) dd[x] = sBw[dw + x];
                    union { uint32_t u[4]; bf16x8 b; } fr;
#pragma unroll
                    for (int x = 0; x < 4; ++x) fr.u[x] = m0 < T ? __builtin_amdgcn_alignbit(dd[x + 1], dd[x], sh) : 0u;   // (T % 8 == 0: all in or all out)
#pragma unroll
                    for (int c = 0; c < DH / 32; ++c) {
                        const bf16x8 pf = tr_frag<PT>(sP, ks * 16, c * 32, lane);
                        dqv[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, fr.b, dqv[c], 0, 0, 0);
                    }
                }
                if (has_extra) {                                  // the next tile's first row: plain dot products
                    for (int ml = xp; ml < 128; ml += NPART) {
                        const int m = stage * 128 + ml;
                        if (m <= T - 2 - r1) xacc += bf16_bits_to_f32(sB[(TQ - 1) * T + m + r1 + 1]) * bf16_bits_to_f32(sP[ml * PT + xc]);
                    }
                }
                __syncthreads();
            }
        }
        if (row_ok) {
            h16* out = a.dqv + ((long)b * T + i) * a.lddqv + h * DH;
#pragma unroll
            for (int c = 0; c < DH / 32; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 u;
                    u.x = pack2_bf16(dqv[c][4 * g + 0], dqv[c][4 * g + 1]);
                    u.y = pack2_bf16(dqv[c][4 * g + 2], dqv[c][4 * g + 3]);
                    *(uint2*)(out + c * 32 + 8 * g + 4 * half) = u;
                }
        }
        if (q >= 1 && il == 0) {                                  // this tile's share of its first row, unrounded
            float* fx = a.dqv_fix + ((long)(bh * ntile + q) * 2 + 0) * DH;
#pragma unroll
            for (int c = 0; c < DH / 32; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4*)(fx + c * 32 + 8 * g + 4 * half) = make_float4(dqv[c][4 * g], dqv[c][4 * g + 1], dqv[c][4 * g + 2], dqv[c][4 * g + 3]);
        }
        float* red = (float*)sK;
        red[tid] = xacc;
        __syncthreads();
        if (has_extra && tid < DH) {
            float t = 0.f;
#pragma unroll
            for (int pp = 0; pp < NPART; ++pp) t += red[pp * DH + tid];
            a.dqv_fix[((long)(bh * ntile + q + 1) * 2 + 1) * DH + tid] = t;
        }
        __syncthreads();
        // ---- dpos^T[c][m] = sum_rl qv^T[c][i0 + rl] dR[i0 + rl][m], rl = 0 .. 128: lane = position, 8 consecutive rows are T+1 halfwords apart
        uint16_t* sQ = sK;                                        // [144][PT] bf16: 129 qv rows, zeros behind them
        static_assert(2 * TK * PK + TK * PT >= (TQ + 16) * PT, "qv rows must fit the K / V buffers");
#pragma unroll
        for (int c = 0; c < NQV; ++c) {
            const int cid = tid + 256 * c, row = cid / CPR, c8 = cid % CPR;
            if (cid < (TQ + 16) * CPR) {
                uint4 w = rqv[c];
                if (a.vb && row <= TQ && i0 + row < T) w = addb8<TA>(w, a.vb + h * DH + c8 * 8);
                *(uint4*)&sQ[row * PT + c8 * 8] = recode8<TA, bf16>(w);
            }
        }
        __syncthreads();
        h16* DP = a.dpos_part + ((long)(b * ntile + q) * T) * ((long)a.H * DH) + h * DH;
        const int nks = (min(TQ, T - i0) + 15) >> 4;
        for (int mblk = wave; mblk * 32 < T; mblk += 4) {
            const int m = mblk * 32 + (lane & 31);
            const int fb0 = m + i0 - (T - 1);
            f32x16 acc[DH / 32];
#pragma unroll
            for (int c = 0; c < DH / 32; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#ifdef ATTN_EXP_NO_B
            if (a.T > 0) return;
#endif
            for (int ks = 0; ks < nks; ++ks) {
                const int base = fb0 + (ks * 16 + half * 8) * (T + 1);
                union { uint32_t u[4]; bf16x8 b; } fr;
#pragma unroll
                for (int e = 0; e < 8; e += 2)
                    fr.u[e >> 1] = (uint32_t)sB[base + e * (T + 1)] | ((uint32_t)sB[base + (e + 1) * (T + 1)] << 16);
#pragma unroll
                for (int c = 0; c < DH / 32; ++c) {
                    const bf16x8 qf = tr_frag<PT>(sQ, ks * 16, c * 32, lane);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, fr.b, acc[c], 0, 0, 0);
                }
            }
            if (has_extra) {                                      // row i0 + 128: only its "upper" part (m <= T-2-r1) lives in this slab
                const int f = fb0 + TQ * (T + 1);
                const uint32_t ld = (uint32_t)sB[min(f, TQ * T - 1)];
                union { uint32_t u[4]; bf16x8 b; } fr;
                fr.u[0] = (half == 0 && m <= T - 2 - r1) ? ld : 0u;
                fr.u[1] = 0u; fr.u[2] = 0u; fr.u[3] = 0u;
#pragma unroll
                for (int c = 0; c < DH / 32; ++c) {
                    const bf16x8 qf = tr_frag<PT>(sQ, TQ, c * 32, lane);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, fr.b, acc[c], 0, 0, 0);
                }
            }
            if (m < T) {
#pragma unroll
                for (int c = 0; c < DH / 32; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 u;
                        u.x = pack2_bf16(acc[c][4 * g + 0], acc[c][4 * g + 1]);
                        u.y = pack2_bf16(acc[c][4 * g + 2], acc[c][4 * g + 3]);
                        *(uint2*)(DP + (long)m * ((long)a.H * DH) + c * 32 + 8 * g + 4 * half) = u;
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------- backward, part 2: dK and dV, per key tile
// One workgroup per (batch, head, 128 keys), one wave per 32 keys; loops over 64-query tiles.  Lane = key row.
template <int DH, typename TA>
__global__ __launch_bounds__(256) void relpos_attn_bwd_kv_kernel(AttnArgs a) {
    constexpr int TKB = 128, TQ = 64;
    constexpr int PK = DH + 8, PT = DH + 32, PB = TKB + 8;
    constexpr int CPR = DH / 8;
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * TQ * PK + 2 * TQ * PT + TQ * PB];
    __shared__ float sStat[2 * TQ];
    uint16_t* sQ = smem;                 // [query][c]  b128 reads (S^T)
    uint16_t* sDO = sQ + TQ * PK;        // [query][c]  b128 reads (dP^T)
    uint16_t* sQt = sDO + TQ * PK;       // transpose-read copies (dK += dS^T Q, dV += P^T dO)
    uint16_t* sDOt = sQt + TQ * PT;
    uint16_t* sB = sDOt + TQ * PT;       // bias tile [query][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H, T = a.T;
    const int j0 = blockIdx.x * TKB;
    const int j = j0 + wave * 32 + (lane & 31);                // this lane's key row
    const bool key_ok = j < T;
    const h16* Q = a.qu + (long)b * T * a.ldq + h * DH;
    const h16* DO = a.dctx + (long)b * T * a.lddc + h * DH;
    const h16* Bi = a.bias + (long)bh * T * T;
    bf16x8 fk[DH / 16], fv[DH / 16];
    {
        const h16* kp = a.k + ((long)b * T + (key_ok ? j : 0)) * a.ldk + h * DH + half * 8;
        const h16* vp = a.v + ((long)b * T + (key_ok ? j : 0)) * a.ldk + h * DH + half * 8;
#pragma unroll
        for (int s = 0; s < DH / 16; ++s) {
            fk[s] = __builtin_bit_cast(bf16x8, key_ok ? *(const uint4*)(kp + s * 16) : make_uint4(0, 0, 0, 0));
            fv[s] = __builtin_bit_cast(bf16x8, recode8<TA, bf16>(key_ok ? *(const uint4*)(vp + s * 16) : make_uint4(0, 0, 0, 0)));   // meets dO (bf16)
        }
    }
    f32x16 dk[DH / 32], dv[DH / 32];
#pragma unroll
    for (int c = 0; c < DH / 32; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[c][r] = 0.f; dv[c][r] = 0.f; }
    const float sc2 = a.scale * LOG2E;
    const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    AttnDrop drop;
    drop.init(a, (unsigned long long)bh * T * (unsigned long long)T, inv_keep);

    constexpr int NQD = TQ * CPR / 256, NBI = TQ * (TKB / 8) / 256;
    uint4 rq[NQD], rd[NQD], rbi[NBI];
    float rstat = 0.f;
    auto load_tile = [&](int i0) {
#pragma unroll
        for (int c = 0; c < NQD; ++c) {
            const int cid = tid + 256 * c, row = cid / CPR, c8 = cid % CPR;
            rq[c] = make_uint4(0, 0, 0, 0); rd[c] = rq[c];
            if (i0 + row < T) {
                rq[c] = *(const uint4*)(Q + (long)(i0 + row) * a.ldq + c8 * 8);
                rd[c] = *(const uint4*)(DO + (long)(i0 + row) * a.lddc + c8 * 8);
            }
        }
#pragma unroll
        for (int c = 0; c < NBI; ++c) {
            const int cid = tid + 256 * c, row = cid >> 4, c8 = cid & 15;
            rbi[c] = make_uint4(0, 0, 0, 0);
            if (i0 + row < T && j0 + c8 * 8 < T) rbi[c] = *(const uint4*)(Bi + (long)(i0 + row) * T + j0 + c8 * 8);
        }
        rstat = 0.f;
        if (tid < 2 * TQ) {
            const int il = tid & (TQ - 1);
            if (i0 + il < T) rstat = tid < TQ ? a.lse[(long)bh * T + i0 + il] : a.dsum[(long)bh * T + i0 + il];
        }
    };
    load_tile(0);
    for (int i0 = 0; i0 < T; i0 += TQ) {
#pragma unroll
        for (int c = 0; c < NQD; ++c) {
            const int cid = tid + 256 * c, row = cid / CPR, c8 = cid % CPR;
            uint4 qw = rq[c];
            if (a.ub && i0 + row < T) qw = addb8<TA>(qw, a.ub + h * DH + c8 * 8);   // (q + u formed here when the caller passed plain q)
            *(uint4*)&sQ[row * PK + c8 * 8] = qw;                            // as saved (score recomputation)
            *(uint4*)&sQt[row * PT + c8 * 8] = recode8<TA, bf16>(qw);        // bf16 copy for dK += dS^T Q
            *(uint4*)&sDO[row * PK + c8 * 8] = rd[c];
            *(uint4*)&sDOt[row * PT + c8 * 8] = rd[c];
        }
#pragma unroll
        for (int c = 0; c < NBI; ++c) {
            const int cid = tid + 256 * c, row = cid >> 4, c8 = cid & 15;
            *(uint4*)&sB[row * PB + c8 * 8] = rbi[c];
        }
        if (tid < 2 * TQ) sStat[tid] = rstat;
        __syncthreads();
        if (i0 + TQ < T) load_tile(i0 + TQ);
        // S = Q K^T and dP = dO V^T with lane = key, registers = queries
        f32x16 s[2], dp[2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[f][r] = 0.f; dp[f][r] = 0.f; }
#pragma unroll
        for (int st = 0; st < DH / 16; ++st)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 qf = *(const bf16x8*)&sQ[(f * 32 + (lane & 31)) * PK + st * 16 + half * 8];
                const bf16x8 df = *(const bf16x8*)&sDO[(f * 32 + (lane & 31)) * PK + st * 16 + half * 8];
                s[f] = mfma16<TA>(qf, fk[st], s[f]);
                dp[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, fv[st], dp[f], 0, 0, 0);
            }
        // P (dropped, for dV) -> dp registers are reused for it after dS has been formed in s
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = f * 32 + 8 * (r >> 2) + 4 * half + (r & 3);     // query row inside the tile
                const int i = i0 + il;
                const float bv = H16<TA>::lo(sB[il * PB + wave * 32 + (lane & 31)]);
                const float x = (s[f][r] + (j == i + 1 ? 0.f : bv)) * sc2;
                const float p = (i < T && key_ok) ? exp2_raw(x - sStat[il]) : 0.f;
                float keep = 1.0f;
                if (a.p_drop > 0.f)
                    keep = drop.keep1(((unsigned long long)bh * T + (unsigned long long)(i < T ? i : 0)) * (unsigned long long)T + (unsigned long long)(key_ok ? j : 0));
                s[f][r] = a.scale * p * (keep * dp[f][r] - sStat[TQ + il]);       // dScore^T
                dp[f][r] = p * keep;                                               // dropped probability
            }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            bf16x8 sf[2], pf[2];
            acc_to_operand<bf16>(s[f], sf);
            acc_to_operand<bf16>(dp[f], pf);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int c = 0; c < DH / 32; ++c) {
                    const bf16x8 qf = tr_frag<PT>(sQt, f * 32 + st * 16, c * 32, lane);
                    const bf16x8 df = tr_frag<PT>(sDOt, f * 32 + st * 16, c * 32, lane);
                    dk[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, sf[st], dk[c], 0, 0, 0);
                    dv[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, pf[st], dv[c], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    if (key_ok) {
        h16* ok_ = a.dk + ((long)b * T + j) * a.lddk + h * DH;
        h16* ov_ = a.dv + ((long)b * T + j) * a.lddk + h * DH;
#pragma unroll
        for (int c = 0; c < DH / 32; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u, w;
                u.x = pack2_bf16(dk[c][4 * g + 0], dk[c][4 * g + 1]); u.y = pack2_bf16(dk[c][4 * g + 2], dk[c][4 * g + 3]);
                w.x = pack2_bf16(dv[c][4 * g + 0], dv[c][4 * g + 1]); w.y = pack2_bf16(dv[c][4 * g + 2], dv[c][4 * g + 3]);
                *(uint2*)(ok_ + c * 32 + 8 * g + 4 * half) = u;
                *(uint2*)(ov_ + c * 32 + 8 * g + 4 * half) = w;
            }
    }
    if (a.dqv_fix && blockIdx.x >= 1 && tid < DH) {               // the dQ kernel's boundary row (see there): both halves, rounded once
        const float* fx = a.dqv_fix + ((long)(bh * gridDim.x + blockIdx.x) * 2) * DH;
        a.dqv[((long)b * T + j0) * a.lddqv + h * DH + tid] = (h16)(pack2_bf16(fx[tid] + fx[DH + tid], 0.f) & 0xffffu);
    }
    if (a.dq_sum) {
        __syncthreads();                                          // (the boundary row above is one of the addends)
        for (int cid = tid; cid < TKB * CPR; cid += 256) {
            const int row = cid / CPR, c8 = cid % CPR;
            if (j0 + row < T) {
                const long r = (long)b * T + j0 + row;
                f8 x = unpack8<bf16>(*(const uint4*)(a.dqu + r * a.lddq + h * DH + c8 * 8));
                const f8 y = unpack8<bf16>(*(const uint4*)(a.dqv + r * a.lddqv + h * DH + c8 * 8));
#pragma unroll
                for (int e = 0; e < 8; ++e) x.v[e] += y.v[e];
                *(uint4*)(a.dq_sum + r * a.lddq_sum + h * DH + c8 * 8) = pack8<bf16>(x);
            }
        }
    }
}

// C ABI ----------------------------------------------------------------------------------------------------------------------
static int attn_check(int B, int H, int T, int dh, long ldq, long ldk, const char* name) {
    if (!(B > 0 && H > 0 && T > 0 && (dh == 32 || dh == 64 || dh == 128) && T % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0)) {
        sarssl_set_error("%s: unsupported shape (B=%d H=%d T=%d dh=%d): needs dh in {32,64,128}, T %% 8 == 0, 16-byte aligned rows",
                         name, B, H, T, dh);
        return -1;
    }
    return 0;
}

// 1 when sarssl_relpos_attn_{fwd,bwd} support the shape (the caller otherwise uses the unfused GEMM + softmax path).
extern "C" int sarssl_relpos_attn_supported(int T, int dh) { return (T > 0 && T % 8 == 0 && (dh == 32 || dh == 64 || dh == 128)) ? 1 : 0; }

// qu, k, v: bf16 [B*T][ld] with head h at column h*dh; bias: bf16 (B,H,T,T) shifted positional scores (unscaled);
// ctx: bf16 [B*T][ldc]; ctx32 (optional, needed for backward): f32 [B*T][H*dh]; lse: f32 (B,H,T) (log2 domain, saved for backward).
extern "C" int sarssl_relpos_attn_fwd(const void* qu, long ldq, const void* k, const void* v, long ldk, const void* bias, void* ctx,
                                      long ldc, float* ctx32, float* lse, int B, int H, int T, int dh, float scale, float p_drop,
                                      unsigned long long seed, int dtype, void* stream) {
    if (attn_check(B, H, T, dh, ldq, ldk, "sarssl_relpos_attn_fwd")) return -1;
    SARSSL_REQUIRE(ldc % 4 == 0 && lse != nullptr && (dtype == SARSSL_BF16 || dtype == SARSSL_F16), "sarssl_relpos_attn_fwd");
    AttnArgs a = {};
    a.qu = (const h16*)qu; a.ldq = ldq; a.k = (const h16*)k; a.v = (const h16*)v; a.ldk = ldk; a.bias = (const h16*)bias;
    a.ctx = (h16*)ctx; a.ldc = ldc; a.ctx32 = ctx32; a.lse = lse; a.B = B; a.H = H; a.T = T; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.salt = sarssl_dropout_salt();
    dim3 grid((T + 127) / 128, B * H);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SARSSL_F16) {
        if (dh == 128) relpos_attn_fwd_kernel<128, f16><<<grid, 256, 0, st>>>(a);
        else if (dh == 64) relpos_attn_fwd_kernel<64, f16><<<grid, 256, 0, st>>>(a);
        else relpos_attn_fwd_kernel<32, f16><<<grid, 256, 0, st>>>(a);
    } else {
        if (dh == 128) relpos_attn_fwd_kernel<128, bf16><<<grid, 256, 0, st>>>(a);
        else if (dh == 64) relpos_attn_fwd_kernel<64, bf16><<<grid, 256, 0, st>>>(a);
        else relpos_attn_fwd_kernel<32, bf16><<<grid, 256, 0, st>>>(a);
    }
    SARSSL_CHECK_LAUNCH("relpos_attn_fwd_kernel");
    return 0;
}

// The same forward with the shifted positional score formed inside the kernel (attention.py:87-89 + 105-113 fused): qv = q + v_bias
// [B*T][ldq], pos = positional projection [T][ldp] (head h at column h*dh).  bias_out (optional, (B,H,T,T)) receives the shifted score
// for backward kernels that read it.  T <= 256, T % 8 == 0 (sarssl_relpos_attn_pos_supported).
static int sarssl_attn_fwd_waves() {          // SARSSL_ATTN_FWD_WAVES=4: the four-wave forward for d_head 64 as well (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("SARSSL_ATTN_FWD_WAVES"); v = (e && atoi(e) == 4) ? 4 : 8; }
    return v;
}
extern "C" int sarssl_relpos_attn_pos_supported(int T, int dh) { return (T > 0 && T <= 256 && T % 8 == 0 && (dh == 32 || dh == 64 || dh == 128)) ? 1 : 0; }
// u_bias / v_bias (both or neither; f32 [H*dh]): qu and qv are then the SAME plain query projection q and the kernels form q + u / q + v
// while loading (the values sarssl_bias2 would have stored).
static int attn_fwd_pos_impl(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldk, const void* pos,
                             long ldp, void* bias_out, void* ctx, void* ctx_lo, long ldc, float* ctx32, float* lse, int B, int H, int T, int dh,
                             float scale, float p_drop, unsigned long long seed, const float* u_bias, const float* v_bias,
                             int dtype, void* stream) {
    SARSSL_REQUIRE((u_bias == nullptr) == (v_bias == nullptr) && (!u_bias || qu == qv), "sarssl_relpos_attn_fwd_pos(u_bias / v_bias)");
    if (attn_check(B, H, T, dh, ldq, ldk, "sarssl_relpos_attn_fwd_pos")) return -1;
    SARSSL_REQUIRE(sarssl_relpos_attn_pos_supported(T, dh) && ldp % 8 == 0 && qv && pos, "sarssl_relpos_attn_fwd_pos(T <= 256)");
    SARSSL_REQUIRE(ldc % 4 == 0 && lse != nullptr && (dtype == SARSSL_BF16 || dtype == SARSSL_F16), "sarssl_relpos_attn_fwd_pos");
    AttnArgs a = {};
    a.qu = (const h16*)qu; a.qv = (const h16*)qv; a.ldq = ldq; a.k = (const h16*)k; a.v = (const h16*)v; a.ldk = ldk;
    a.pos = (const h16*)pos; a.ldp = ldp; a.bias_out = (h16*)bias_out; a.ub = u_bias; a.vb = v_bias; a.ctx_lo = (h16*)ctx_lo;
    a.ctx = (h16*)ctx; a.ldc = ldc; a.ctx32 = ctx32; a.lse = lse; a.B = B; a.H = H; a.T = T; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.salt = sarssl_dropout_salt();
    dim3 grid((T + 127) / 128, B * H);
    hipStream_t st = (hipStream_t)stream;
    if (dh == 64 && T > 128 && sarssl_attn_fwd_waves() == 8) {        // all 256 query rows of a (batch, head) in one eight-wave workgroup
        dim3 grid8(1, B * H);
        if (dtype == SARSSL_F16) relpos_attn_fwd_kernel<64, f16, true, 8><<<grid8, 512, 0, st>>>(a);
        else relpos_attn_fwd_kernel<64, bf16, true, 8><<<grid8, 512, 0, st>>>(a);
        SARSSL_CHECK_LAUNCH("relpos_attn_fwd_kernel<pos, 8 waves>");
        return 0;
    }
    if (dtype == SARSSL_F16) {
        if (dh == 128) relpos_attn_fwd_kernel<128, f16, true><<<grid, 256, 0, st>>>(a);
        else if (dh == 64) relpos_attn_fwd_kernel<64, f16, true><<<grid, 256, 0, st>>>(a);
        else relpos_attn_fwd_kernel<32, f16, true><<<grid, 256, 0, st>>>(a);
    } else {
        if (dh == 128) relpos_attn_fwd_kernel<128, bf16, true><<<grid, 256, 0, st>>>(a);
        else if (dh == 64) relpos_attn_fwd_kernel<64, bf16, true><<<grid, 256, 0, st>>>(a);
        else relpos_attn_fwd_kernel<32, bf16, true><<<grid, 256, 0, st>>>(a);
    }
    SARSSL_CHECK_LAUNCH("relpos_attn_fwd_kernel<pos>");
    return 0;
}

extern "C" int sarssl_relpos_attn_fwd_pos(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldk, const void* pos,
                                          long ldp, void* bias_out, void* ctx, long ldc, float* ctx32, float* lse, int B, int H, int T, int dh,
                                          float scale, float p_drop, unsigned long long seed, const float* u_bias, const float* v_bias,
                                          int dtype, void* stream) {
    return attn_fwd_pos_impl(qu, qv, ldq, k, v, ldk, pos, ldp, bias_out, ctx, nullptr, ldc, ctx32, lse, B, H, T, dh, scale, p_drop, seed, u_bias, v_bias,
                             dtype, stream);
}
// ... with the context also written as an fp16 PAIR (hybrid mode): ctx_lo [B*T][ldc] = fp16(c - fp16(c)) of the unrounded f32 context c
extern "C" int sarssl_relpos_attn_fwd_pos_pair(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldk, const void* pos,
                                               long ldp, void* bias_out, void* ctx, void* ctx_lo, long ldc, float* ctx32, float* lse, int B, int H,
                                               int T, int dh, float scale, float p_drop, unsigned long long seed, const float* u_bias,
                                               const float* v_bias, void* stream) {
    SARSSL_REQUIRE(ctx_lo != nullptr, "sarssl_relpos_attn_fwd_pos_pair");
    return attn_fwd_pos_impl(qu, qv, ldq, k, v, ldk, pos, ldp, bias_out, ctx, ctx_lo, ldc, ctx32, lse, B, H, T, dh, scale, p_drop, seed, u_bias, v_bias,
                             SARSSL_F16, stream);
}

// The same for T > 256 (T % 8 == 0; the slab covers one 256-key block at a time): qu / qv must be the biased projections (no u_bias /
// v_bias); bias_out (may be null) receives the (B,H,T,T) shifted score block by block - what sarssl_relpos_attn_bwd reads.
extern "C" int sarssl_relpos_attn_pos_long_supported(int T, int dh) { return (T > 256 && T <= 4096 && T % 8 == 0 && (dh == 64 || dh == 128)) ? 1 : 0; }
extern "C" int sarssl_relpos_attn_fwd_pos_long(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldk, const void* pos,
                                               long ldp, void* bias_out, void* ctx, long ldc, float* ctx32, float* lse, int B, int H, int T, int dh,
                                               float scale, float p_drop, unsigned long long seed, int dtype, void* stream) {
    if (attn_check(B, H, T, dh, ldq, ldk, "sarssl_relpos_attn_fwd_pos_long")) return -1;
    SARSSL_REQUIRE(sarssl_relpos_attn_pos_long_supported(T, dh) && ldp % 8 == 0 && qv && pos, "sarssl_relpos_attn_fwd_pos_long(T > 256)");
    SARSSL_REQUIRE(ldc % 4 == 0 && lse != nullptr && (dtype == SARSSL_BF16 || dtype == SARSSL_F16), "sarssl_relpos_attn_fwd_pos_long");
    AttnArgs a = {};
    a.qu = (const h16*)qu; a.qv = (const h16*)qv; a.ldq = ldq; a.k = (const h16*)k; a.v = (const h16*)v; a.ldk = ldk;
    a.pos = (const h16*)pos; a.ldp = ldp; a.bias_out = (h16*)bias_out; a.ub = nullptr; a.vb = nullptr;
    a.ctx = (h16*)ctx; a.ldc = ldc; a.ctx32 = ctx32; a.lse = lse; a.B = B; a.H = H; a.T = T; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.salt = sarssl_dropout_salt();
    dim3 grid((T + 127) / 128, B * H);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SARSSL_F16) {
        if (dh == 128) relpos_attn_fwd_kernel<128, f16, true, 4, true><<<grid, 256, 0, st>>>(a);
        else relpos_attn_fwd_kernel<64, f16, true, 4, true><<<grid, 256, 0, st>>>(a);
    } else {
        if (dh == 128) relpos_attn_fwd_kernel<128, bf16, true, 4, true><<<grid, 256, 0, st>>>(a);
        else relpos_attn_fwd_kernel<64, bf16, true, 4, true><<<grid, 256, 0, st>>>(a);
    }
    SARSSL_CHECK_LAUNCH("relpos_attn_fwd_kernel<pos, long>");
    return 0;
}

// Backward of sarssl_relpos_attn_fwd.  dsum: f32 workspace (B,H,T).  Outputs: dqu [B*T][lddq], dk / dv [B*T][lddk] (head h at
// column h*dh), dbias bf16 (B,H,T,T) = gradient of the shifted positional score (the element (i, i+1) carries no meaning).
// Backward with the positional-score gradients formed in the dQ kernel (no d(bias) tensor, no un-shift pass, no batched products
// after it): qv / pos as in sarssl_relpos_attn_fwd_pos, bias = the (B,H,T,T) shifted score that forward wrote.  Outputs besides dqu /
// dk / dv: dqv [B*T][lddqv] (gradient of q + v_bias through the positional score), dpos_part bf16 (B, ntile, T, H*dh) with
// ntile = ceil(T / 128): partial gradients of the positional projection, to be summed over the first two axes.  dqv_fix: f32
// workspace (B*H, ntile, 2, dh).  T <= 256 (sarssl_relpos_attn_pos_supported).
extern "C" int sarssl_relpos_attn_bwd_pos(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldk, const void* pos,
                                          long ldp, const void* bias, const float* ctx32, const float* lse, const void* dctx, long lddc,
                                          void* dqu, long lddq, void* dqv, long lddqv, void* dk, void* dv, long lddk, void* dpos_part,
                                          float* dqv_fix, float* dsum, int B, int H, int T, int dh, float scale, float p_drop,
                                          unsigned long long seed, const float* u_bias, const float* v_bias, void* dq_sum, long lddq_sum,
                                          int dtype, void* stream) {
    if (attn_check(B, H, T, dh, ldq, ldk, "sarssl_relpos_attn_bwd_pos")) return -1;
    SARSSL_REQUIRE((u_bias == nullptr) == (v_bias == nullptr) && (!u_bias || qu == qv), "sarssl_relpos_attn_bwd_pos(u_bias / v_bias)");
    SARSSL_REQUIRE(sarssl_relpos_attn_pos_supported(T, dh) && ldp % 8 == 0 && qv && pos && bias && dqv && dpos_part && dqv_fix, "sarssl_relpos_attn_bwd_pos(T <= 256)");
    SARSSL_REQUIRE(dtype == SARSSL_BF16 || dtype == SARSSL_MIX16, "sarssl_relpos_attn_bwd_pos(dtype: bf16, or MIX16 = fp16 forward tensors + bf16 gradients)");
    SARSSL_REQUIRE(lddc % 8 == 0 && lddq % 4 == 0 && lddqv % 4 == 0 && lddk % 4 == 0 && dsum != nullptr && lse != nullptr && ctx32 != nullptr, "sarssl_relpos_attn_bwd_pos");
    AttnArgs a = {};
    a.qu = (const h16*)qu; a.qv = (const h16*)qv; a.ldq = ldq; a.k = (const h16*)k; a.v = (const h16*)v; a.ldk = ldk; a.bias = (const h16*)bias;
    a.pos = (const h16*)pos; a.ldp = ldp; a.ub = u_bias; a.vb = v_bias;
    a.ctx32 = (float*)ctx32; a.lse = (float*)lse; a.dctx = (const h16*)dctx; a.lddc = lddc;
    a.dqu = (h16*)dqu; a.lddq = lddq; a.dk = (h16*)dk; a.dv = (h16*)dv; a.lddk = lddk; a.dsum = dsum;
    a.dqv = (h16*)dqv; a.lddqv = lddqv; a.dpos_part = (h16*)dpos_part; a.dqv_fix = dqv_fix;
    SARSSL_REQUIRE(!dq_sum || (lddq % 8 == 0 && lddqv % 8 == 0 && lddq_sum % 8 == 0 && dq_sum != dqu && dq_sum != dqv), "sarssl_relpos_attn_bwd_pos(dq_sum)");
    a.dq_sum = (h16*)dq_sum; a.lddq_sum = lddq_sum;
    a.B = B; a.H = H; a.T = T; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.salt = sarssl_dropout_salt();
    hipStream_t st = (hipStream_t)stream;
    dim3 gq((T + 127) / 128, B * H), gk((T + 127) / 128, B * H);
#define ATTN_BWD(DHv, TAv) do { relpos_attn_bwd_q_kernel<DHv, TAv, true><<<gq, 256, 0, st>>>(a); relpos_attn_bwd_kv_kernel<DHv, TAv><<<gk, 256, 0, st>>>(a); } while (0)
    if (dtype == SARSSL_MIX16) {
        if (dh == 128) ATTN_BWD(128, f16); else if (dh == 64) ATTN_BWD(64, f16); else ATTN_BWD(32, f16);
    } else {
        if (dh == 128) ATTN_BWD(128, bf16); else if (dh == 64) ATTN_BWD(64, bf16); else ATTN_BWD(32, bf16);
    }
#undef ATTN_BWD
    SARSSL_CHECK_LAUNCH("relpos_attn_bwd_kernel<pos>");
    return 0;
}

extern "C" int sarssl_relpos_attn_bwd(const void* qu, long ldq, const void* k, const void* v, long ldk, const void* bias,
                                      const float* ctx32, const float* lse, const void* dctx, long lddc, void* dqu, long lddq,
                                      void* dk, void* dv, long lddk, void* dbias, float* dsum, int B, int H, int T, int dh,
                                      float scale, float p_drop, unsigned long long seed, int dtype, void* stream) {
    if (attn_check(B, H, T, dh, ldq, ldk, "sarssl_relpos_attn_bwd")) return -1;
    SARSSL_REQUIRE(dtype == SARSSL_BF16 || dtype == SARSSL_MIX16, "sarssl_relpos_attn_bwd(dtype: bf16, or MIX16 = fp16 forward tensors + bf16 gradients)");
    SARSSL_REQUIRE(lddc % 8 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && dsum != nullptr && lse != nullptr && ctx32 != nullptr, "sarssl_relpos_attn_bwd");
    AttnArgs a = {};
    a.qu = (const h16*)qu; a.ldq = ldq; a.k = (const h16*)k; a.v = (const h16*)v; a.ldk = ldk; a.bias = (const h16*)bias;
    a.ctx32 = (float*)ctx32; a.lse = (float*)lse; a.dctx = (const h16*)dctx; a.lddc = lddc;
    a.dqu = (h16*)dqu; a.lddq = lddq; a.dk = (h16*)dk; a.dv = (h16*)dv; a.lddk = lddk; a.dbias = (h16*)dbias; a.dsum = dsum;
    a.B = B; a.H = H; a.T = T; a.scale = scale; a.p_drop = p_drop; a.seed = seed; a.salt = sarssl_dropout_salt();
    hipStream_t st = (hipStream_t)stream;
    dim3 gq((T + 127) / 128, B * H), gk((T + 127) / 128, B * H);
#define ATTN_BWD(DHv, TAv) do { relpos_attn_bwd_q_kernel<DHv, TAv><<<gq, 256, 0, st>>>(a); relpos_attn_bwd_kv_kernel<DHv, TAv><<<gk, 256, 0, st>>>(a); } while (0)
    if (dtype == SARSSL_MIX16) {
        if (dh == 128) ATTN_BWD(128, f16); else if (dh == 64) ATTN_BWD(64, f16); else ATTN_BWD(32, f16);
    } else {
        if (dh == 128) ATTN_BWD(128, bf16); else if (dh == 64) ATTN_BWD(64, bf16); else ATTN_BWD(32, bf16);
    }
#undef ATTN_BWD
    SARSSL_CHECK_LAUNCH("relpos_attn_bwd_kernel");
    return 0;
}
