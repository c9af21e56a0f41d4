// 3x3 / stride 1 / pad 1, 64 -> 64 channel convolution of the MC-Conformer CNN stem
// (code/model.py:54-59) as implicit GEMM on bf16 MFMA, channels-last activations (B, F, T, 64).
//
//   conv3x3_fwd   : out[p][co] = sum_{tap,ci} W[tap][co][ci] * z[p + tap][ci],  z = prologue(in)
//                   prologue = identity, or relu(in*scale[ci] + shift[ci]) (the previous
//                   BatchNorm2d + ReLU applied while the tile is staged - they never cost an HBM
//                   round trip).  The same kernel is the data-gradient with flipped/transposed
//                   weights.
//   conv3x3_wgrad : dW[tap][co][ci] = sum_p dy[p][co] * z[p + tap][ci]   (contraction over
//                   pixels: both MFMA operands come from [pixel][channel] LDS tiles through the
//                   gfx950 transpose read ds_read_b64_tr_b16).
//
// Persistent workgroups (one per CU, 8 waves): the 9x64x64 bf16 weight tensor (72 KiB) stays in
// LDS for the whole launch next to an (8+2)x(64+2)-pixel input tile (82.5 KiB); tile t+1 is
// prefetched into registers while tile t is on the matrix cores.  LDS tiles are [pixel][64 ch]
// with the 16-byte chunk index XOR-swizzled by (pixel>>1)&7 so that the 16-lane groups of
// ds_read_b128 / tr reads touch 16 distinct 16-byte slots.
//
// f32 ("precise") storage: operands are split hi/lo to bf16 at staging time and the host runs
// three accumulating passes (see gemm.hip).
#include "common.h"
#include <stdlib.h>

#define TR 8
#define TCOL 64
#define HR (TR + 2)
#define HC (TCOL + 2)
#define NPIX_H (HR * HC)            // 660 halo pixels
#define NCHUNK_H (NPIX_H * 8)       // 5280 16-byte chunks
#define X_ITERS 11                  // ceil(5280 / 512)
#define W_ELEMS (9 * 64 * 64)
#define X_ELEMS (NPIX_H * 64)
#define Y_ELEMS (TR * TCOL * 64)

struct ConvArgs {
    const void* in; const void* w; void* out;
    float* acc_ws; int acc_in, acc_out;
    const float* scale; const float* shift; int prologue;
    int nb, F, T;
    int part_in, part_w;
    double* stats;      // optional f64[128]: per-channel sum / sum of squares of the (bf16-rounded) output, for the next BatchNorm
    // data-gradient mode of the ping-pong kernel: the output is dz = dL/d relu(bn(y)) of the PREVIOUS layer; with bn_y / bn_aff set,
    // stats instead receives that BatchNorm's backward sums [sum g | sum g*xhat], g = dz * relu'(bn(y)) (the cl_bn_bwd_reduce pass)
    const void* bn_y;   // (B,F,T,64) pre-BN activations of the layer whose input gradient this launch produces
    const float* bn_aff;    // [4][64]: scale, shift, mean, rstd
    int prio;           // != 0: raise the wave's issue priority for its MFMA phase (s_setprio), see sarssl_mfma_prio()
    // first-layer INPUT mode of the ping-pong kernel (C1IN): `in` is the stem's 4-channel input a0 (B,F,T,4) and the convolution runs
    // on relu(bn1(W1 a0)) formed while staging (c1_w = W1 f32[64][4], scale / shift = bn1's affine) - the 64-channel output of the
    // first 1x1 layer is never stored or read (2 x 537 MB per encoder and pass at B = 64)
    const float* c1_w;
    // first-layer REDUCTION mode of the ping-pong kernel (C1RED): the launch is the data gradient of the first 3x3 convolution and its
    // result dz1 = dL/d relu(bn1(W1 a0)) is only needed for the first layer's parameter gradients, so it is never stored: the epilogue
    // masks it with relu'(bn1(W1 a0)) (one MFMA per accumulator tile recomputes the pre-activation from c1_a0 / c1_w / scale / shift)
    // and contracts it over the pixels against [a0 | 1] on the matrix cores: c1_red (f64[644], sarssl_stem_c1_bwd's layout) receives
    // G[co][c] = sum_p g a0[c] and s1[co] = sum_p g
    const void* c1_a0; double* c1_red;
    // optional clock probe (sarssl_conv_clock_probe): thread 0 of workgroup 0 stores {s_memtime, s_memrealtime} at kernel entry and
    // exit - shader-clock ticks over constant-rate ticks = the effective shader clock this launch ran at (bench.py reports it next to
    // the launch durations: the chip clocks down under matrix load, NOTES.md 6)
    unsigned long long* clk;
#ifdef CONV_STAMPS
    unsigned long long* stamps;
#endif
};

// C1IN staging: 8 channels of relu(scale * (W1 a) + shift) for one pixel from its packed 4-channel input (lo = channels 0,1; hi = 2,3).
// wp[c][h] = scale-folded weights of the channel pair h for input channel c, shp[h] = shifts: 16 packed FMAs + 8 max + 4 converts per
// chunk - the same VALU work as unpack + affine + ReLU of a stored chunk.
struct C1Const { sarssl_f32x2 wp[4][4], shp[4]; };
__device__ __forceinline__ void c1_setup(C1Const& k, const float* W1, const float* scale, const float* shift, int c0) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int ch = c0 + 2 * h;
        k.shp[h] = sarssl_f32x2{shift[ch], shift[ch + 1]};
#pragma unroll
        for (int c = 0; c < 4; ++c) k.wp[c][h] = sarssl_f32x2{scale[ch] * W1[ch * 4 + c], scale[ch + 1] * W1[(ch + 1) * 4 + c]};
    }
}
template <typename TI = bf16, typename TO = TI>      // TI: encoding of the 4-channel input, TO: of the operand written to LDS
__device__ __forceinline__ uint4 c1_chunk(uint32_t lo, uint32_t hi, bool valid, const C1Const& k) {
    if (!valid) return make_uint4(0, 0, 0, 0);
    const float a0 = H16<TI>::lo(lo), a1 = H16<TI>::hi(lo);
    const float a2 = H16<TI>::lo(hi), a3 = H16<TI>::hi(hi);
    uint32_t o[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const sarssl_f32x2 y = k.wp[0][h] * a0 + k.wp[1][h] * a1 + k.wp[2][h] * a2 + k.wp[3][h] * a3 + k.shp[h];
        o[h] = H16<TO>::pack(fmaxf(y.x, 0.f), fmaxf(y.y, 0.f));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ int swz(int p, int chunk) { return (p * 8 + (chunk ^ ((p >> 1) & 7))) * 8; }
// input-tile variant: XOR term from the pixel's COLUMN in the halo tile only (same conflict-free ds_read_b128 pattern within a row,
// and fragment addresses become lane constant + affine offset)
__device__ __forceinline__ int swzx(int p, int col, int chunk) { return (p * 8 + (chunk ^ ((col >> 1) & 7))) * 8; }

// raw chunk in registers: 8 channels of one pixel
template <typename T> struct Chunk;
template <> struct Chunk<bf16> { uint4 u; };
template <> struct Chunk<f16> { uint4 u; };
template <> struct Chunk<float> { f8 v; };

template <typename T>
__device__ __forceinline__ Chunk<T> load_chunk(const T* p, bool valid) {
    Chunk<T> c;
    if constexpr (sizeof(T) == 2) c.u = valid ? *(const uint4*)p : make_uint4(0, 0, 0, 0);
    else {
        if (valid) c.v = ld8(p);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) c.v.v[e] = 0.f;
        }
    }
    return c;
}

// Prefetch load of the 8 channels at c8 of pixel (b, f, t) with the coordinates CLAMPED into the image: always a valid address, so
// the load is unconditional.  A load under a runtime condition is a serialised load - hipcc branches around it and waits for it before
// issuing the next one: cycle stamps (tools/conv_stamps.py) showed 4.3 k cycles per tile for "issuing" 11 predicated loads, i.e. 11
// back-to-back memory round trips.  Out-of-image chunks are zeroed when the tile is written to LDS (xform_chunk's `valid`).
template <typename T>
__device__ __forceinline__ Chunk<T> load_chunk_clamped(const T* __restrict__ base, int b, int f, int t, int F, int Tn, int c8) {
    f = min(max(f, 0), F - 1);
    t = min(max(t, 0), Tn - 1);
    return load_chunk<T>(base + (((long)b * F + f) * Tn + t) * 64 + c8, true);
}

// prologue + conversion to the 8 MFMA-operand elements written to LDS (16-bit storage T -> TO; f32 storage: split bf16 parts)
template <typename T, typename TO = T>
__device__ __forceinline__ uint4 xform_chunk(const Chunk<T>& c, bool valid, int prologue, const float* sc, const float* sh, int part) {
    if (!valid) return make_uint4(0, 0, 0, 0);
    if constexpr (sizeof(T) == 2) {
        if (!prologue) return recode8<T, TO>(c.u);
        f8 v = unpack8<T>(c.u);
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] = fmaxf(fmaf(v.v[e], sc[e], sh[e]), 0.f);
        return pack8<TO>(v);
    } else {
        f8 v = c.v;
        if (prologue) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] = fmaxf(fmaf(v.v[e], sc[e], sh[e]), 0.f);
        }
        return pack8_part(v, part);
    }
}

// halo-tile geometry helper: chunk q -> (pixel p, image coords, validity, global offset)
struct TileCoord { int b, f0, t0; };
__device__ __forceinline__ TileCoord tile_coord(int tile, int tiles_f, int tiles_t) {
    TileCoord c;
    c.t0 = (tile % tiles_t) * TCOL; tile /= tiles_t;
    c.f0 = (tile % tiles_f) * TR;
    c.b = tile / tiles_f;
    return c;
}

// XCD-aware persistent tile order: workgroup b runs on XCD b % 8 (observed dispatch rule; used for speed only).  In every
// round of gridDim.x tiles each XCD gets a CONTIGUOUS run of tiles (neighbouring tiles of one image), so the halo rows /
// columns shared by adjacent tiles are served by that XCD's L2 instead of being fetched from HBM twice.
__device__ __forceinline__ int xcd_tile(int it, int bid, int grid) {
    if ((grid & 7) == 0) return it * grid + (bid & 7) * (grid >> 3) + (bid >> 3);
    return it * grid + bid;
}

template <typename T, typename TW>
__global__ __launch_bounds__(512) void conv3x3_fwd_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) uint16_t sW[W_ELEMS];
    __shared__ __attribute__((aligned(16))) uint16_t sX[X_ELEMS];
    __shared__ float sStats[8][128];               // per-wave partial statistics, folded in wave order at the end (deterministic)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int F = a.F, Tn = a.T;
    const int tiles_f = (F + TR - 1) / TR, tiles_t = (Tn + TCOL - 1) / TCOL;
    const int ntiles = a.nb * tiles_f * tiles_t;
    const T* in = (const T*)a.in;
    for (int q = tid; q < 8 * 128; q += 512) (&sStats[0][0])[q] = 0.f;
    const int cch = tid & 7;                        // this thread's 8-channel chunk (fixed: 512 % 8 == 0)

    // weights -> LDS once ([tap][co][ci], ci contiguous)
    {
        const TW* w = (const TW*)a.w;
        for (int q = tid; q < 9 * 64 * 8; q += 512) {
            const int p = q >> 3, c = q & 7;
            uint4 u;
            if constexpr (sizeof(TW) == 2) u = *(const uint4*)(w + (long)p * 64 + c * 8);
            else { f8 v = ld8(w + (long)p * 64 + c * 8); u = pack8_part(v, a.part_w); }
            *(uint4*)&sW[swz(p, c)] = u;
        }
    }
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = a.prologue ? a.scale[cch * 8 + e] : 1.f; sh[e] = a.prologue ? a.shift[cch * 8 + e] : 0.f; }

    // Halo tile staging.  Thread = (pixel column pc = tid >> 3, 8-channel chunk cch): halo row i of columns 0..63 for i = 0..9, plus
    // one of the 160 chunks of halo columns 64 / 65 for tid < 160 - no per-chunk division, one pointer bump per row.
    // lane-constant element offsets of the MFMA fragments: W rows (tap*64 + i*32 + lane&31) keep the (row>>1)&7 swizzle, which only
    // depends on the lane; the input tile uses the column swizzle swzx so that its XOR term only depends on (lane, kw)
    int laneW[4], laneX[3][4];
    {
        const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            laneW[kc] = swz(l31, kc * 2 + hi);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) laneX[kw][kc] = swzx(wave * HC + l31 + kw, l31 + kw, kc * 2 + hi);
        }
    }
    Chunk<T> regs[X_ITERS];
    const int pc = tid >> 3;
    auto issue_loads = [&](int tile) {
        const TileCoord tc = tile_coord(tile, tiles_f, tiles_t);
        const int t = tc.t0 - 1 + pc;
#pragma unroll
        for (int i = 0; i < HR; ++i) regs[i] = load_chunk_clamped<T>(in, tc.b, tc.f0 - 1 + i, t, F, Tn, cch * 8);
        {
            const int hr = pc >> 1, te = tc.t0 + TCOL - 1 + (pc & 1);       // (threads >= 160: an unused, harmless extra chunk)
            regs[HR] = load_chunk_clamped<T>(in, tc.b, tc.f0 - 1 + hr, te, F, Tn, cch * 8);
        }
    };
    auto write_tile = [&](int tile) {
        const TileCoord tc = tile_coord(tile, tiles_f, tiles_t);
        const int t = tc.t0 - 1 + pc;
        const bool tv = t >= 0 && t < Tn;
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const int f = tc.f0 - 1 + i;
            *(uint4*)&sX[swzx(i * HC + pc, pc, cch)] = xform_chunk<T>(regs[i], tv && f >= 0 && f < F, a.prologue, sc, sh, a.part_in);
        }
        if (tid < 160) {
            const int hr = pc >> 1, f = tc.f0 - 1 + hr, te = tc.t0 + TCOL - 1 + (pc & 1);
            *(uint4*)&sX[swzx(hr * HC + TCOL + (pc & 1), TCOL + (pc & 1), cch)] = xform_chunk<T>(regs[HR], f >= 0 && f < F && te < Tn, a.prologue, sc, sh, a.part_in);
        }
    };

    const int nrounds = (ntiles + gridDim.x - 1) / gridDim.x;
    if (xcd_tile(0, blockIdx.x, gridDim.x) < ntiles) issue_loads(xcd_tile(0, blockIdx.x, gridDim.x));
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): see the note after the MFMA loop (keeps the loop-carried wait state clean)
    for (int it = 0; it < nrounds; ++it) {
        const int tile = xcd_tile(it, blockIdx.x, gridDim.x);
        if (tile >= ntiles) break;                 // (whole workgroup takes the same branch)
        write_tile(tile);
        __syncthreads();
        const int next = (it + 1 < nrounds) ? xcd_tile(it + 1, blockIdx.x, gridDim.x) : ntiles;
        if (next < ntiles) issue_loads(next);

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        {
            // 36 k-steps (9 taps x 4 chunks of 16 channels); fragments of step s+1 are fetched before the MFMAs of step s.
            // LDS addresses are tile-invariant: recomputing them (a few VALU ops hidden under the MFMAs) instead of letting
            // LICM keep ~40 of them live across the tile loop keeps the kernel well under 256 VGPRs.
            // Fragment addresses = lane constant (laneX[kw][kc], laneW[kc]; computed once per kernel) + compile-time offset of the
            // (tap, row half): no per-step address arithmetic next to the MFMAs (it cost ~6 VALU ops per MFMA).
            bf16x8 wf[2][2], xf[2][2];
            auto fetch = [&](int s, int buf) {
                const int tap = s >> 2, kc = s & 3;
                const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[buf][i] = *(const bf16x8*)(sW + laneW[kc] + (tap * 64 + i * 32) * 64);
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[buf][j] = *(const bf16x8*)(sX + laneX[kw][kc] + (kh * HC + j * 32) * 64);
            };
            fetch(0, 0);
#pragma unroll
            for (int s = 0; s < 36; ++s) {
                const int cur = s & 1;
                if (s + 1 < 36) fetch(s + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);             // keep the prefetch ABOVE this step's MFMAs (hipcc otherwise sinks
                                                               // the ds_reads next to their use and waits lgkmcnt(0) every 2 MFMAs)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][i], xf[cur][j], acc[i][j], 0, 0, 0);
            }
        }

        // gfx9 counts loads AND stores in vmcnt and they may complete out of order with respect to each other: once this tile's
        // output stores are in flight, the first use of the prefetched registers (top of the next iteration) would have to wait
        // vmcnt(0), i.e. for the stores' HBM round trip.  The prefetch was issued a whole MFMA phase ago - retire it here, before
        // any store is issued, so the stores drain behind the next tile's staging and matrix work instead.
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0), expcnt / lgkmcnt untouched
        const TileCoord tc = tile_coord(tile, tiles_f, tiles_t);
        const int f = tc.f0 + wave;
        if constexpr (sizeof(T) == 2) {
            // Output path (bf16).  The accumulators (lane = pixel, 4 consecutive channels per register group) are transposed
            // through the wave's own 8 KiB slice of the now idle input-tile LDS so that 8 consecutive lanes hold one pixel's
            // 128-byte line: 8 fully coalesced 16-byte stores per lane (1 KiB contiguous per instruction) instead of 16
            // scattered 8-byte stores that each touch 32 cache lines (store-issue bound: ~40 % of the kernel).
            __syncthreads();                                   // every wave is done reading the input tile
            uint16_t* stg = sX + wave * (64 * 64);             // [64 px][64 co], 16-byte chunk index XOR (px & 7)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int px = j * 32 + (lane & 31);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = i * 32 + 8 * g + 4 * (lane >> 5);
                        uint2 w2;
                        w2.x = pack2_bf16(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1]);
                        w2.y = pack2_bf16(acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
                        *(uint2*)&stg[px * 64 + (((co >> 3) ^ (px & 7)) << 3) + (co & 7)] = w2;
                    }
                }
            // the staging slice is private to this wave: its own LDS writes only have to land (no workgroup barrier)
            __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            float ssum[8], ssq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
            if (f < F) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int px = (lane >> 3) + 8 * k;
                    const int t = tc.t0 + px;
                    const uint4 o = *(const uint4*)&stg[px * 64 + (((lane & 7) ^ (px & 7)) << 3)];
                    if (t < Tn) {
                        *(uint4*)((uint16_t*)a.out + (((long)tc.b * F + f) * Tn + t) * 64 + (lane & 7) * 8) = o;
                        if (a.stats) {          // BatchNorm statistics of exactly what was stored (8 channels of this lane's chunk)
                            const uint32_t w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float lo = bf16_bits_to_f32(w[q] & 0xffffu), hi = __uint_as_float(w[q] & 0xffff0000u);
                                ssum[2 * q] += lo; ssq[2 * q] += lo * lo; ssum[2 * q + 1] += hi; ssq[2 * q + 1] += hi * hi;
                            }
                        }
                    }
                }
            }
            if (a.stats) {
                // lanes that share a channel chunk are 8 apart: butterfly over lane bits 3..5, then one LDS atomic per channel
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ssum[e] += __shfl_xor(ssum[e], 8, 64); ssum[e] += __shfl_xor(ssum[e], 16, 64); ssum[e] += __shfl_xor(ssum[e], 32, 64);
                    ssq[e] += __shfl_xor(ssq[e], 8, 64); ssq[e] += __shfl_xor(ssq[e], 16, 64); ssq[e] += __shfl_xor(ssq[e], 32, 64);
                }
                if (lane < 8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sStats[wave][lane * 8 + e] += ssum[e]; sStats[wave][64 + lane * 8 + e] += ssq[e]; }   // this wave's own slot: no race
                }
            }
        } else if (f < F) {
            // f32 storage (precise mode): lane = pixel, 4 consecutive co per register group; split-pass accumulation workspace
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t = tc.t0 + j * 32 + (lane & 31);
                if (t < Tn) {
                    const long pix = (((long)tc.b * F + f) * Tn + t) * 64;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int co = i * 32 + 8 * g + 4 * (lane >> 5);
                            float4 v = make_float4(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
                            if (a.acc_in) {
                                const float4 w4 = *(const float4*)(a.acc_ws + pix + co);
                                v.x += w4.x; v.y += w4.y; v.z += w4.z; v.w += w4.w;
                            }
                            if (a.acc_out) *(float4*)(a.acc_ws + pix + co) = v;
                            else st4((T*)a.out + pix + co, v);
                        }
                }
            }
        }
        __syncthreads();
    }
    if (a.stats && tid < 128) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += sStats[w][tid];
        atomicAdd(&a.stats[tid], (double)s);
    }
}


// ------------------------------------------------------------------------------------ forward / dgrad, "ping-pong" variant (bf16)
// Same arithmetic as conv3x3_fwd_kernel, different schedule.  Per-wave cycle stamps showed that with one 8-wave workgroup per CU the
// matrix pipe is ~95 % busy during the MFMA phase but that phase is only ~half of a tile: staging, the two workgroup barriers and
// the output epilogue run with the pipe idle, because all eight waves are always in the same phase.  Here the workgroup is split in
// two independent halves of four waves (waves w and w+4 share a SIMD, so each half has one wave per SIMD).  Each half owns an
// (8+2)x(32+2)-pixel input buffer (2 x 42.5 KiB next to the 72 KiB of weights = 157 KiB) and walks its own sequence of 8x32-pixel
// tiles; the halves only synchronise among their own four waves (an LDS arrival counter - s_barrier would couple all eight).
// The SIMD arbiter favours the older wave, so the halves drift out of phase by themselves: while one is on the matrix cores the
// other stages its next tile / drains its outputs.
#ifdef CONV_STAMPS
// per-phase cycle stamps of workgroup 0 (tools/conv_stamps.py; compiled into a separate probe library only)
// (the buffer pointer travels as a kernel argument: fetching it from a __device__ global would put a load + vmcnt(0) into every stamp
//  and serialise exactly the prefetch loads / output stores the stamps are meant to time)
static unsigned long long* g_conv_stamps_host = nullptr;
extern "C" int sarssl_conv_stamp_buffer(void* p) { g_conv_stamps_host = (unsigned long long*)p; return 0; }
#define STAMP(k) do { if (blockIdx.x == 0 && it < 8 && lane == 0 && a.stamps) a.stamps[(wave * 8 + it) * 12 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
#define PTC 32
#define PHC (PTC + 2)
#define PX_ELEMS (HR * PHC * 64)                   // 21760 bf16 per half

__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* s, int pix_base, int ch_base, int lane) {
    // rows = pixels (contraction), cols = channels: lane gets channel ch_base + (lane&31), 8 consecutive pixels
    // starting at pix_base + (lane>>5)*8.  Two transpose reads of 4 pixels each.
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const int col = ch_base + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
    const int chunk = col >> 3, within = col & 7;
    bf16x8 out;
    s16x4 q[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = pix_base + (lane >> 5) * 8 + h * 4 + ((lane & 15) >> 2);
        const uint16_t* addr = s + swz(p, chunk) + within;
        q[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
    }
    union { s16x4 v[2]; bf16x8 b; } u;
    u.v[0] = q[0]; u.v[1] = q[1];
    out = u.b;
    return out;
}

__device__ __forceinline__ void half_barrier(unsigned* cnt, unsigned& epoch, int lane) {
    __builtin_amdgcn_s_waitcnt(0xC07F);            // my LDS traffic has landed (lgkmcnt(0))
    if (lane == 0) atomicAdd(cnt, 1u);
    epoch += 4;
    // poll through an LDS-address-space pointer: a generic (flat) load would make every poll wait for the outstanding global
    // loads / stores as well (flat operations count in vmcnt)
    volatile __attribute__((address_space(3))) unsigned* c3 = (volatile __attribute__((address_space(3))) unsigned*)cnt;
    while (*c3 < epoch) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// TM: encoding of the input / weights / LDS operands / output (bf16, or fp16 for the forward launches of the fp16-forward mode);
// TY: encoding of the tensors SAVED BY THE FORWARD PASS that the gradient epilogues read (bn_y, c1_a0)
template <bool BNRED, bool C1IN = false, bool C1RED = false, typename TM = bf16, typename TY = bf16>
__global__ __launch_bounds__(512) void conv3x3_fwd_pp_kernel(ConvArgs a) {
    typedef TM T;
    static_assert(!C1RED || __is_same(TM, bf16), "gradient launches contract in bf16");
    __shared__ __attribute__((aligned(16))) uint16_t sW[W_ELEMS];
    __shared__ __attribute__((aligned(16))) uint16_t sXh[2][PX_ELEMS];
    __shared__ float sAff[BNRED ? 256 : 1];
    // C1RED: MFMA "A" fragments of the scale-folded first-layer weights, [co][16 k]: k 0..3 = bf16 high parts of scale*W1[co][c], 4 = of
    // shift, 8..12 = the low parts (the input fragment repeats [a0 | 1] in both k halves: one MFMA gives the f32-accurate pre-activation)
    __shared__ __attribute__((aligned(16))) uint16_t sC1[C1RED ? 64 * 16 : 8];
    __shared__ unsigned sSync[2];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform ids as SCALARS (hipcc cannot prove tid >> 6 uniform): the tile coordinates (two integer divisions), row bases and
    // validity tests derived from them then live on the scalar unit instead of costing ~250 vector instructions per tile
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, hw = wave & 3, htid = tid & 255;
    const int F = a.F, Tn = a.T;
    const int tiles_f = (F + TR - 1) / TR, tiles_t = (Tn + PTC - 1) / PTC;
    const int ntiles = a.nb * tiles_f * tiles_t;
    const int npairs = (ntiles + 1) >> 1;
    const T* in = (const T*)a.in;
    uint16_t* sX = sXh[half];
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[0] = __builtin_amdgcn_s_memtime(); a.clk[1] = __builtin_amdgcn_s_memrealtime(); }
    if (tid < 2) sSync[tid] = 0u;
    if (BNRED && tid < 256) sAff[tid] = a.bn_aff[tid];
    if (C1RED) {
        for (int q = tid; q < 64 * 16; q += 512) {
            const int co = q >> 4, k = q & 15, c = k & 7;
            const float v = c < 4 ? a.scale[co] * a.c1_w[co * 4 + c] : (c == 4 ? a.shift[co] : 0.f);
            const uint32_t hi = f32_to_bf16_bits(v);
            sC1[q] = (uint16_t)(k < 8 ? hi : f32_to_bf16_bits(v - bf16_bits_to_f32(hi)));
        }
    }
    const int cch = tid & 7;
    {
        const T* w = (const T*)a.w;
        for (int q = tid; q < 9 * 64 * 8; q += 512) {
            const int p = q >> 3, c = q & 7;
            *(uint4*)&sW[swz(p, c)] = *(const uint4*)(w + (long)p * 64 + c * 8);
        }
    }
    __syncthreads();                                // weights, counters (the only workgroup-wide barrier before the end)
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = a.prologue ? a.scale[cch * 8 + e] : 1.f; sh[e] = a.prologue ? a.shift[cch * 8 + e] : 0.f;
    }
    C1Const kc1;
    if (C1IN) c1_setup(kc1, a.c1_w, a.scale, a.shift, cch * 8);

    // lane-constant fragment addresses; this wave computes rows 2*hw and 2*hw+1 (32 pixels x 64 channels each)
    int laneW[4], laneX[3][4];
    {
        const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            laneW[kc] = swz(l31, kc * 2 + hi);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) laneX[kw][kc] = swzx(2 * hw * PHC + l31 + kw, l31 + kw, kc * 2 + hi);
        }
    }
    auto tile_of = [&](int it) { const int pr = xcd_tile(it, blockIdx.x, gridDim.x); return pr < npairs ? pr * 2 + half : ntiles; };
    auto coord = [&](int tile) { TileCoord c; c.t0 = (tile % tiles_t) * PTC; tile /= tiles_t; c.f0 = (tile % tiles_f) * TR; c.b = tile / tiles_f; return c; };

    // staging: thread of the half = (pixel column pc = htid >> 3 of 32, chunk): halo rows 0..9 + one chunk of halo columns 32 / 33
    Chunk<T> regs[X_ITERS];
    const int pc = htid >> 3;
    // addresses: row bases are scalar (tile coordinates are wave-uniform), each thread adds ONE byte offset (its clamped frame and chunk)
    auto issue_loads = [&](const TileCoord tc) {
        const int tcl = min(max(tc.t0 - 1 + pc, 0), Tn - 1);
        const unsigned voff = C1IN ? (unsigned)tcl * 8u : (unsigned)(tcl * 64 + cch * 8) * 2u;
        // one 64-bit base per tile (the image), 32-bit row offsets inside it (an image is < 4 GB): round-3 stamps showed this phase at
        // 2.8 k cycles per half tile for 12 loads - ~14 scalar instructions of 64-bit multiply / add per row sat in front of every load
        constexpr unsigned PXB = C1IN ? 8u : 128u;                               // bytes per pixel
        const char* pimg = (const char*)in + (long)tc.b * F * (long)Tn * PXB;
        const unsigned rowbytes = (unsigned)Tn * PXB;
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const int f = min(max(tc.f0 - 1 + i, 0), F - 1);                      // (clamped: unconditional loads, see load_chunk_clamped)
            const char* prow = pimg + (unsigned)f * rowbytes;
            if (C1IN) { const uint2 q = *(const uint2*)(prow + voff); regs[i].u.x = q.x; regs[i].u.y = q.y; }
            else regs[i].u = *(const uint4*)(prow + voff);
        }
        {
            const int hr = pc >> 1, te = tc.t0 + PTC - 1 + (pc & 1);        // (threads >= 160: an unused, harmless extra chunk)
            if (C1IN) {
                const int f = min(max(tc.f0 - 1 + hr, 0), F - 1), t = min(max(te, 0), Tn - 1);
                const uint2 q = *(const uint2*)(in + (((long)tc.b * F + f) * Tn + t) * 4);
                regs[HR].u.x = q.x; regs[HR].u.y = q.y;
            } else regs[HR] = load_chunk_clamped<T>(in, tc.b, tc.f0 - 1 + hr, te, F, Tn, cch * 8);
        }
    };
    auto write_tile = [&](const TileCoord tc) {
        const int t = tc.t0 - 1 + pc;
        const bool tv = t >= 0 && t < Tn;
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const int f = tc.f0 - 1 + i;
            const bool ok = tv && f >= 0 && f < F;
            *(uint4*)&sX[swzx(i * PHC + pc, pc, cch)] = C1IN ? c1_chunk<T>(regs[i].u.x, regs[i].u.y, ok, kc1)
                                                             : xform_chunk<T>(regs[i], ok, a.prologue, sc, sh, 0);
        }
        if (htid < 160) {
            const int hr = pc >> 1, f = tc.f0 - 1 + hr, te = tc.t0 + PTC - 1 + (pc & 1);
            const bool ok = f >= 0 && f < F && te < Tn;
            *(uint4*)&sX[swzx(hr * PHC + PTC + (pc & 1), PTC + (pc & 1), cch)] = C1IN ? c1_chunk<T>(regs[HR].u.x, regs[HR].u.y, ok, kc1)
                                                                                       : xform_chunk<T>(regs[HR], ok, a.prologue, sc, sh, 0);
        }
    };

    unsigned epoch = 0;
    unsigned* cnt = &sSync[half];
    float ssum[8], ssq[8];                         // this thread's 8 output channels ((lane & 7) * 8 + e), summed over all its tiles
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    float bthr[8];                                 // BNRED: relu'(y*sc + sh) as a threshold test on y (see cl_bn_bwd_reduce), lane-constant
    unsigned bsgn = 0u;
    if (BNRED) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = (lane & 7) * 8 + e;
            const float sc_ = sAff[c], sh_ = sAff[64 + c];
            float thr = sc_ != 0.f ? -sh_ / sc_ : (sh_ > 0.f ? -INFINITY : INFINITY);
            if (sc_ < 0.f) { thr = -thr; bsgn |= 1u << e; }
            bthr[e] = thr;
        }
    }
    float gacc[2][4];                              // C1RED: lanes < 32: G[c = r][co = nb*32 + lane]; lanes >= 32: r = 0: s1[co = nb*32 + lane - 32]
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) gacc[nb][r] = 0.f;
    const int nrounds = (npairs + gridDim.x - 1) / gridDim.x;
    int tile = tile_of(0);
    TileCoord tc = coord(tile < ntiles ? tile : 0);
    if (tile < ntiles) issue_loads(tc);
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0) (see conv3x3_fwd_kernel)
    for (int it = 0; it < nrounds; ++it) {
        if (tile >= ntiles) break;                 // (all four waves of the half take the same branch)
        STAMP(0);
        write_tile(tc);
        STAMP(1);
        half_barrier(cnt, epoch, lane);
        STAMP(2);
        const int next = (it + 1 < nrounds) ? tile_of(it + 1) : ntiles;
        const TileCoord tcn = coord(next < ntiles ? next : 0);
        if (next < ntiles) issue_loads(tcn);
        STAMP(3);

        f32x16 acc[2][2];                          // [co half][row]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        {
            bf16x8 wf[2][2], xf[2][2];
            auto fetch = [&](int s, int buf) {
                const int tap = s >> 2, kc = s & 3;
                const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[buf][i] = *(const bf16x8*)(sW + laneW[kc] + (tap * 64 + i * 32) * 64);
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[buf][j] = *(const bf16x8*)(sX + laneX[kw][kc] + ((kh + j) * PHC) * 64);
            };
            // the wave that is on the matrix cores gets issue priority over the other half's wave on the same SIMD (which is staging
            // its next tile / draining its outputs on the VALU): its MFMAs and fragment reads are never queued behind that work
            if (a.prio) __builtin_amdgcn_s_setprio(3);
            fetch(0, 0);
#pragma unroll
            for (int s = 0; s < 36; ++s) {
                const int cur = s & 1;
                if (s + 1 < 36) fetch(s + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma16<TM>(wf[cur][i], xf[cur][j], acc[i][j]);
            }
            if (a.prio) __builtin_amdgcn_s_setprio(0);
        }
        STAMP(4);
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // retire the prefetch before any output store is issued
        STAMP(5);
        // BatchNorm-backward mode: fetch the matching pre-BN activations now - their round trip runs under the half barrier and the
        // accumulator hand-over below - and retire them BEFORE the first output store is issued (loads and stores share vmcnt and
        // complete out of order on gfx9).  Unconditional loads from clamped addresses (out-of-image chunks are skipped in the drain).
        uint4 yv[BNRED ? 8 : 1];
        if (BNRED) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int px = (lane >> 3) + 8 * k;
                const int f = min(tc.f0 + 2 * hw + (px >> 5), F - 1), t = min(tc.t0 + (px & 31), Tn - 1);
                yv[k] = *(const uint4*)((const uint16_t*)a.bn_y + (((long)tc.b * F + f) * Tn + t) * 64 + (lane & 7) * 8);
            }
        }
        uint2 av[C1RED ? 2 : 1];                               // C1RED: the 4 input channels of this lane's pixel (lane & 31) in both rows
        bool pv[C1RED ? 2 : 1];
        if (C1RED) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int f = tc.f0 + 2 * hw + j, t = tc.t0 + (lane & 31);
                pv[j] = f < F && t < Tn;
                av[j] = *(const uint2*)((const uint16_t*)a.c1_a0 + (((long)tc.b * F + min(f, F - 1)) * Tn + min(t, Tn - 1)) * 4);
            }
        }
        half_barrier(cnt, epoch, lane);                        // the half is done reading its input tile
        STAMP(6);
        uint16_t* stg = sX + hw * (64 * 64);                   // this wave's [64 px][64 co] slice (px = row * 32 + column)
        if (C1RED) {
            // (1) mask: pre-activation tile of the first layer in the accumulators' own layout (co x pixel), one MFMA per tile
            __builtin_amdgcn_s_waitcnt(0x0F70);                // av landed (the next tile's prefetch too: it had the whole MFMA loop)
            if constexpr (!__is_same(TY, bf16)) {              // saved fp16 input -> the bf16 operand of the two first-layer contractions below
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    av[j].x = pack2_bf16(H16<TY>::lo(av[j].x), H16<TY>::hi(av[j].x));
                    av[j].y = pack2_bf16(H16<TY>::lo(av[j].y), H16<TY>::hi(av[j].y));
                }
            }
            uint16_t* a0t = sX + 4 * (64 * 64) + hw * 256;     // [4 c][64 px] of this wave, in the part of the input tile no slice uses
            {
                const int j = lane >> 5;                       // lanes < 32 file row 0, lanes >= 32 row 1 (both hold both rows' pixels)
                const uint2 q = j ? av[1] : av[0];
                uint16_t* d = a0t + j * 32 + (lane & 31);
                d[0] = (uint16_t)(q.x & 0xffffu); d[64] = (uint16_t)(q.x >> 16); d[128] = (uint16_t)(q.y & 0xffffu); d[192] = (uint16_t)(q.y >> 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 wfr = *(const bf16x8*)&sC1[(i * 32 + (lane & 31)) * 16 + (lane >> 5) * 8];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    union { uint4 u; bf16x8 b; } bx;
                    bx.u = make_uint4(av[j].x, av[j].y, 0x00003f80u, 0u);          // [a0_0..a0_3, 1, 0, 0, 0]
                    f32x16 y;
#pragma unroll
                    for (int r = 0; r < 16; ++r) y[r] = 0.f;
                    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr, bx.b, y, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = (y[r] > 0.f && pv[j]) ? acc[i][j][r] : 0.f;
                }
            }
            // (2) masked gradient tile, bf16, [64 px][64 co] in the transpose-read layout (swz)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        uint2 w2;
                        w2.x = pack2_bf16(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1]);
                        w2.y = pack2_bf16(acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
                        *(uint2*)(stg + swz(j * 32 + (lane & 31), i * 4 + g) + 4 * (lane >> 5)) = w2;
                    }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            // (3) G[c'][co] += sum_px [a0 | 1][px][c'] * g[px][co]: 4 k-steps of 16 pixels x 2 channel halves
            f32x16 d2[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) d2[nb][r] = 0.f;
            const int m = lane & 31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                union { uint4 u; bf16x8 b; } ax;
                ax.u = *(const uint4*)(a0t + min(m, 3) * 64 + ks * 16 + (lane >> 5) * 8);
                if (m == 4) ax.u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
                else if (m > 4) ax.u = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    d2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax.b, tr_frag(stg, ks * 16, nb * 32, lane), d2[nb], 0, 0, 0);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) gacc[nb][r] += d2[nb][r];
        } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // chunk (i*4+g) ^ (px & 7) only depends on the lane (px & 7 == lane & 7 for both rows): one address per (i, g)
                uint16_t* q = stg + (lane & 31) * 64 + ((((i * 4 + g) ^ (lane & 7)) << 3) | (4 * (lane >> 5)));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint2 w2;
                    w2.x = H16<TM>::pack(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1]);
                    w2.y = H16<TM>::pack(acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
                    *(uint2*)(q + j * 32 * 64) = w2;
                }
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (BNRED) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        STAMP(7);
        // Interior tiles (every tile of a 256-multiple image): branch-free drain - all eight LDS reads in flight at once, one scalar base
        // per wave, lane-constant byte offset, the row / column-block offsets immediate or scalar.  Round-3 stamps: the per-chunk validity
        // branches (s_and_saveexec + branch around every store) made this phase a chain of eight LDS round trips, 2.4-4.5 k cycles per tile.
        const bool interior = !BNRED && (tc.f0 + TR <= F) && (tc.t0 + PTC <= Tn);
        if (interior) {
            uint4 o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = *(const uint4*)&stg[((lane >> 3) + 8 * k) * 64 + (((lane & 7) ^ (lane >> 3)) << 3)];
            char* obase = (char*)a.out + (((long)tc.b * F + tc.f0 + 2 * hw) * Tn + tc.t0) * 128;
            const unsigned loff = (unsigned)((lane >> 3) * 128 + (lane & 7) * 16), rowb = (unsigned)Tn * 128u;
#pragma unroll
            for (int k = 0; k < 8; ++k) *(uint4*)(obase + loff + (k & 3) * (8 * 128) + (k >> 2) * rowb) = o[k];
            if (a.stats) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t w[4] = {o[k].x, o[k].y, o[k].z, o[k].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = H16<TM>::lo(w[q]), hi = H16<TM>::hi(w[q]);
                        ssum[2 * q] += lo; ssq[2 * q] += lo * lo; ssum[2 * q + 1] += hi; ssq[2 * q + 1] += hi * hi;
                    }
                }
            }
        } else
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int px = (lane >> 3) + 8 * k;
            const int f = tc.f0 + 2 * hw + (px >> 5), t = tc.t0 + (px & 31);
            // (px & 7) == (lane >> 3) for every k: lane-constant chunk, k only moves the row offset
            const uint4 o = *(const uint4*)&stg[px * 64 + (((lane & 7) ^ (lane >> 3)) << 3)];
            if (f < F && t < Tn) {
                *(uint4*)((uint16_t*)a.out + (((long)tc.b * F + f) * Tn + t) * 64 + (lane & 7) * 8) = o;
                if (BNRED) {            // sums of g and g*y; turned into rstd*(sum g*y - mean*sum g) per tile below
                    const uint32_t w[4] = {o.x, o.y, o.z, o.w};
                    const uint32_t yw[4] = {yv[k].x, yv[k].y, yv[k].z, yv[k].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float d0 = bf16_bits_to_f32(w[q] & 0xffffu), d1 = __uint_as_float(w[q] & 0xffff0000u);
                        const float y0 = H16<TY>::lo(yw[q]), y1 = H16<TY>::hi(yw[q]);
                        const float t0_ = __uint_as_float(__float_as_uint(y0) ^ (((bsgn >> (2 * q)) & 1u) << 31));
                        const float t1_ = __uint_as_float(__float_as_uint(y1) ^ (((bsgn >> (2 * q + 1)) & 1u) << 31));
                        const float g0 = t0_ > bthr[2 * q] ? d0 : 0.f, g1 = t1_ > bthr[2 * q + 1] ? d1 : 0.f;
                        ssum[2 * q] += g0; ssq[2 * q] = fmaf(g0, y0, ssq[2 * q]);
                        ssum[2 * q + 1] += g1; ssq[2 * q + 1] = fmaf(g1, y1, ssq[2 * q + 1]);
                    }
                } else if (a.stats) {
                    const uint32_t w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = H16<TM>::lo(w[q]), hi = H16<TM>::hi(w[q]);
                        ssum[2 * q] += lo; ssq[2 * q] += lo * lo; ssum[2 * q + 1] += hi; ssq[2 * q + 1] += hi * hi;
                    }
                }
            }
        }
        }
        STAMP(8);
        STAMP(9);
        half_barrier(cnt, epoch, lane);                        // slices drained before the next tile overwrites the buffer
        STAMP(10);
        tile = next; tc = tcn;
    }
    // per-channel sums: the thread's 8 channels were accumulated over ALL its tiles in registers (folding them per tile - 48 lane
    // exchanges + 16 LDS atomics - showed as ~2.5 k of a half-tile's ~22 k cycles in the stamps); one fold per launch.
    // The fold is ORDERED (round 3): every wave parks its 128 partial sums in a slot of its own (the weight table is free once all
    // eight waves are past their last tile) and thread c adds the eight slots in wave order - the round-2 f32 LDS atomics summed them
    // in arrival order, which flipped bf16 roundings downstream from run to run.  What leaves the workgroup is one f64 atomic per
    // channel of f32-valued terms: exact (order-free) as long as the terms' exponents span < 2^20.
    __syncthreads();                               // every wave is done with sW
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[2] = __builtin_amdgcn_s_memtime(); a.clk[3] = __builtin_amdgcn_s_memrealtime(); }
    float* part = (float*)sW;                      // [8 waves][128] statistics, then [8][320] first-layer sums (C1RED)
    if (BNRED || a.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ssum[e] += __shfl_xor(ssum[e], 8, 64); ssum[e] += __shfl_xor(ssum[e], 16, 64); ssum[e] += __shfl_xor(ssum[e], 32, 64);
            ssq[e] += __shfl_xor(ssq[e], 8, 64); ssq[e] += __shfl_xor(ssq[e], 16, 64); ssq[e] += __shfl_xor(ssq[e], 32, 64);
        }
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s2 = ssq[e];
                if (BNRED) s2 = sAff[192 + lane * 8 + e] * (s2 - sAff[128 + lane * 8 + e] * ssum[e]);     // rstd * (sum g*y - mean * sum g)
                part[wave * 128 + lane * 8 + e] = ssum[e]; part[wave * 128 + 64 + lane * 8 + e] = s2;
            }
        }
        __syncthreads();
        if (a.stats && tid < 128) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += part[w * 128 + tid];
            atomicAdd(&a.stats[tid], (double)s);
        }
    }
    if (C1RED) {                                   // fold the 8 waves through LDS in wave order, then f64 atomics
        float* fold = part + 1024;                 // [8][320]: [0,256) G[co][c], [256,320) s1[co]
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int co = nb * 32 + (lane & 31);
            if (lane < 32) {
#pragma unroll
                for (int r = 0; r < 4; ++r) fold[wave * 320 + co * 4 + r] = gacc[nb][r];
            } else fold[wave * 320 + 256 + co] = gacc[nb][0];
        }
        __syncthreads();
        for (int q = tid; q < 320; q += 512) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += fold[w * 320 + q];
            atomicAdd(&a.c1_red[q < 256 ? q : 512 + (q - 256)], (double)s);
        }
    }
}

// Role-split variant of conv3x3_fwd_pp_kernel (12 waves, three per SIMD, <= 168 registers): waves 0-7 are the two tile groups of the
// ping-pong kernel WITHOUT their staging phases - wait for the tile, MFMA phase, accumulators -> LDS, drain / statistics / epilogues,
// release the buffer; waves 8-11 only stage: for each group in turn wait until its buffer is released, write the prefetched tile
// (BatchNorm + ReLU or the first-layer prologue), publish it, request the group's next tile.  Staging (2.6 k cycles), its barrier and
// the prefetch issue (2.1 k) leave the MFMA waves' 16.7 k-cycle iteration.  LDS arrival counters: sReady[g] (4 staging waves per tile),
// sFree[g] (4 MFMA waves per tile), sSync[g] (the group's own barrier between its MFMA phase and the accumulator hand-over).
template <bool BNRED, bool C1IN = false, bool C1RED = false, typename TM = bf16, typename TY = bf16>
__global__ __launch_bounds__(768) void conv3x3_fwd_ws_kernel(ConvArgs a) {
    typedef TM T;
    static_assert(!C1RED || __is_same(TM, bf16), "gradient launches contract in bf16");
    __shared__ __attribute__((aligned(16))) uint16_t sW[W_ELEMS];
    __shared__ __attribute__((aligned(16))) uint16_t sXh[2][PX_ELEMS];
    __shared__ float sAff[BNRED ? 256 : 1];
    // C1RED: MFMA "A" fragments of the scale-folded first-layer weights, [co][16 k]: k 0..3 = bf16 high parts of scale*W1[co][c], 4 = of
    // shift, 8..12 = the low parts (the input fragment repeats [a0 | 1] in both k halves: one MFMA gives the f32-accurate pre-activation)
    __shared__ __attribute__((aligned(16))) uint16_t sC1[C1RED ? 64 * 16 : 8];
    __shared__ unsigned sSync[2], sReady[2], sFree[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave_all >= 8;
    // wave-uniform ids as SCALARS (hipcc cannot prove tid >> 6 uniform): the tile coordinates (two integer divisions), row bases and
    // validity tests derived from them then live on the scalar unit instead of costing ~250 vector instructions per tile
    const int wave = wave_all & 7;                 // MFMA waves: group = wave >> 2; staging waves: wave_all - 8
    const int half = wave >> 2, hw = wave & 3, htid = tid & 255;      // (staging waves: htid = their thread id, 0..255)
    const int F = a.F, Tn = a.T;
    const int tiles_f = (F + TR - 1) / TR, tiles_t = (Tn + PTC - 1) / PTC;
    const int ntiles = a.nb * tiles_f * tiles_t;
    const int npairs = (ntiles + 1) >> 1;
    const T* in = (const T*)a.in;
    uint16_t* sX = sXh[half];                      // MFMA waves: their group's tile buffer
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[0] = __builtin_amdgcn_s_memtime(); a.clk[1] = __builtin_amdgcn_s_memrealtime(); }
    if (tid < 2) { sSync[tid] = 0u; sReady[tid] = 0u; sFree[tid] = 0u; }
    if (BNRED && tid < 256) sAff[tid] = a.bn_aff[tid];
    if (C1RED) {
        for (int q = tid; q < 64 * 16; q += 768) {
            const int co = q >> 4, k = q & 15, c = k & 7;
            const float v = c < 4 ? a.scale[co] * a.c1_w[co * 4 + c] : (c == 4 ? a.shift[co] : 0.f);
            const uint32_t hi = f32_to_bf16_bits(v);
            sC1[q] = (uint16_t)(k < 8 ? hi : f32_to_bf16_bits(v - bf16_bits_to_f32(hi)));
        }
    }
    const int cch = tid & 7;
    {
        const T* w = (const T*)a.w;
        for (int q = tid; q < 9 * 64 * 8; q += 768) {
            const int p = q >> 3, c = q & 7;
            *(uint4*)&sW[swz(p, c)] = *(const uint4*)(w + (long)p * 64 + c * 8);
        }
    }
    __syncthreads();                                // weights, counters (the only workgroup-wide barrier before the end)
    // ------------------------------------------------------------------------------------------------ staging waves
    if (producer) {
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = a.prologue ? a.scale[cch * 8 + e] : 1.f; sh[e] = a.prologue ? a.shift[cch * 8 + e] : 0.f;
        }
        C1Const kc1;
        if (C1IN) c1_setup(kc1, a.c1_w, a.scale, a.shift, cch * 8);

        // staging: thread of the half = (pixel column pc = htid >> 3 of 32, chunk): halo rows 0..9 + one chunk of halo columns 32 / 33
        Chunk<T> regsg[2][X_ITERS];                    // one prefetched tile per group
        const int pc = htid >> 3;
        // addresses: row bases are scalar (tile coordinates are wave-uniform), each thread adds ONE byte offset (its clamped frame and chunk)
        auto issue_loads = [&](Chunk<T> (&regs)[X_ITERS], const TileCoord tc) __attribute__((always_inline)) {
            const int tcl = min(max(tc.t0 - 1 + pc, 0), Tn - 1);
            const unsigned voff = C1IN ? (unsigned)tcl * 8u : (unsigned)(tcl * 64 + cch * 8) * 2u;
            // one 64-bit base per tile (the image), 32-bit row offsets inside it (an image is < 4 GB): round-3 stamps showed this phase at
            // 2.8 k cycles per half tile for 12 loads - ~14 scalar instructions of 64-bit multiply / add per row sat in front of every load
            constexpr unsigned PXB = C1IN ? 8u : 128u;                               // bytes per pixel
            const char* pimg = (const char*)in + (long)tc.b * F * (long)Tn * PXB;
            const unsigned rowbytes = (unsigned)Tn * PXB;
#pragma unroll
            for (int i = 0; i < HR; ++i) {
                const int f = min(max(tc.f0 - 1 + i, 0), F - 1);                      // (clamped: unconditional loads, see load_chunk_clamped)
                const char* prow = pimg + (unsigned)f * rowbytes;
                if (C1IN) { const uint2 q = *(const uint2*)(prow + voff); regs[i].u = make_uint4(q.x, q.y, 0u, 0u); }
                else regs[i].u = *(const uint4*)(prow + voff);
            }
            {
                const int hr = pc >> 1, te = tc.t0 + PTC - 1 + (pc & 1);        // (threads >= 160: an unused, harmless extra chunk)
                if (C1IN) {
                    const int f = min(max(tc.f0 - 1 + hr, 0), F - 1), t = min(max(te, 0), Tn - 1);
                    const uint2 q = *(const uint2*)(in + (((long)tc.b * F + f) * Tn + t) * 4);
                    regs[HR].u = make_uint4(q.x, q.y, 0u, 0u);
                } else regs[HR] = load_chunk_clamped<T>(in, tc.b, tc.f0 - 1 + hr, te, F, Tn, cch * 8);
            }
        };
        auto write_tile = [&](const Chunk<T> (&regs)[X_ITERS], const TileCoord tc, uint16_t* __restrict__ sX) __attribute__((always_inline)) {
            const int t = tc.t0 - 1 + pc;
            const bool tv = t >= 0 && t < Tn;
#pragma unroll
            for (int i = 0; i < HR; ++i) {
                const int f = tc.f0 - 1 + i;
                const bool ok = tv && f >= 0 && f < F;
                *(uint4*)&sX[swzx(i * PHC + pc, pc, cch)] = C1IN ? c1_chunk<T>(regs[i].u.x, regs[i].u.y, ok, kc1)
                                                                 : xform_chunk<T>(regs[i], ok, a.prologue, sc, sh, 0);
            }
            if (htid < 160) {
                const int hr = pc >> 1, f = tc.f0 - 1 + hr, te = tc.t0 + PTC - 1 + (pc & 1);
                const bool ok = f >= 0 && f < F && te < Tn;
                *(uint4*)&sX[swzx(hr * PHC + PTC + (pc & 1), PTC + (pc & 1), cch)] = C1IN ? c1_chunk<T>(regs[HR].u.x, regs[HR].u.y, ok, kc1)
                                                                                           : xform_chunk<T>(regs[HR], ok, a.prologue, sc, sh, 0);
            }
        };


        auto coord = [&](int tile) __attribute__((always_inline)) { TileCoord c; c.t0 = (tile % tiles_t) * PTC; tile /= tiles_t; c.f0 = (tile % tiles_f) * TR; c.b = tile / tiles_f; return c; };
        auto tile_g = [&](int it, int g) __attribute__((always_inline)) { const int pr = xcd_tile(it, blockIdx.x, gridDim.x); return pr < npairs ? pr * 2 + g : ntiles; };
        volatile __attribute__((address_space(3))) unsigned* free3 = (volatile __attribute__((address_space(3))) unsigned*)sFree;
        const int nrounds = (npairs + gridDim.x - 1) / gridDim.x;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int t0_ = tile_g(0, g);
            if (t0_ < ntiles) issue_loads(regsg[g], coord(t0_));
        }
        for (int it = 0; it < nrounds; ++it) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int tile = tile_g(it, g);
                if (tile >= ntiles) continue;
                while (free3[g] < (unsigned)it * 4u) __builtin_amdgcn_s_sleep(1);      // the group has drained its previous tile
                asm volatile("" ::: "memory");
                write_tile(regsg[g], coord(tile), sXh[g]);
                __builtin_amdgcn_s_waitcnt(0xC07F);                                     // my LDS writes have landed (lgkmcnt(0))
                if (lane == 0) atomicAdd(&sReady[g], 1u);
                const int next = (it + 1 < nrounds) ? tile_g(it + 1, g) : ntiles;
                if (next < ntiles) issue_loads(regsg[g], coord(next));
            }
        }
        // the MFMA waves' final folds run behind workgroup barriers: take part in the same sequence
        __syncthreads();
        if (BNRED || a.stats) __syncthreads();
        if (C1RED) __syncthreads();
        return;
    }
    // ------------------------------------------------------------------------------------------------ MFMA waves
    // lane-constant fragment addresses; this wave computes rows 2*hw and 2*hw+1 (32 pixels x 64 channels each)
    int laneW[4], laneX[3][4];
    {
        const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            laneW[kc] = swz(l31, kc * 2 + hi);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) laneX[kw][kc] = swzx(2 * hw * PHC + l31 + kw, l31 + kw, kc * 2 + hi);
        }
    }
    auto tile_of = [&](int it) { const int pr = xcd_tile(it, blockIdx.x, gridDim.x); return pr < npairs ? pr * 2 + half : ntiles; };
    auto coord = [&](int tile) { TileCoord c; c.t0 = (tile % tiles_t) * PTC; tile /= tiles_t; c.f0 = (tile % tiles_f) * TR; c.b = tile / tiles_f; return c; };

    unsigned epoch = 0;
    unsigned* cnt = &sSync[half];
    float ssum[8], ssq[8];                         // this thread's 8 output channels ((lane & 7) * 8 + e), summed over all its tiles
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    float bthr[8];                                 // BNRED: relu'(y*sc + sh) as a threshold test on y (see cl_bn_bwd_reduce), lane-constant
    unsigned bsgn = 0u;
    if (BNRED) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = (lane & 7) * 8 + e;
            const float sc_ = sAff[c], sh_ = sAff[64 + c];
            float thr = sc_ != 0.f ? -sh_ / sc_ : (sh_ > 0.f ? -INFINITY : INFINITY);
            if (sc_ < 0.f) { thr = -thr; bsgn |= 1u << e; }
            bthr[e] = thr;
        }
    }
    float gacc[2][4];                              // C1RED: lanes < 32: G[c = r][co = nb*32 + lane]; lanes >= 32: r = 0: s1[co = nb*32 + lane - 32]
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) gacc[nb][r] = 0.f;
    const int nrounds = (npairs + gridDim.x - 1) / gridDim.x;
    volatile __attribute__((address_space(3))) unsigned* ready3 = (volatile __attribute__((address_space(3))) unsigned*)&sReady[half];
    for (int it = 0; it < nrounds; ++it) {
        const int tile = tile_of(it);
        if (tile >= ntiles) break;                 // (all four waves of the group take the same branch)
        const TileCoord tc = coord(tile);
        while (*ready3 < (unsigned)(it + 1) * 4u) __builtin_amdgcn_s_sleep(1);      // the staging waves have published this tile
        asm volatile("" ::: "memory");

        f32x16 acc[2][2];                          // [co half][row]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        {
            bf16x8 wf[2][2], xf[2][2];
            auto fetch = [&](int s, int buf) {
                const int tap = s >> 2, kc = s & 3;
                const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[buf][i] = *(const bf16x8*)(sW + laneW[kc] + (tap * 64 + i * 32) * 64);
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[buf][j] = *(const bf16x8*)(sX + laneX[kw][kc] + ((kh + j) * PHC) * 64);
            };
            // the wave that is on the matrix cores gets issue priority over the other half's wave on the same SIMD (which is staging
            // its next tile / draining its outputs on the VALU): its MFMAs and fragment reads are never queued behind that work
            if (a.prio) __builtin_amdgcn_s_setprio(3);
            fetch(0, 0);
#pragma unroll
            for (int s = 0; s < 36; ++s) {
                const int cur = s & 1;
                if (s + 1 < 36) fetch(s + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma16<TM>(wf[cur][i], xf[cur][j], acc[i][j]);
            }
            if (a.prio) __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // retire the prefetch before any output store is issued
        // BatchNorm-backward mode: fetch the matching pre-BN activations now - their round trip runs under the half barrier and the
        // accumulator hand-over below - and retire them BEFORE the first output store is issued (loads and stores share vmcnt and
        // complete out of order on gfx9).  Unconditional loads from clamped addresses (out-of-image chunks are skipped in the drain).
        uint4 yv[BNRED ? 8 : 1];
        if (BNRED) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int px = (lane >> 3) + 8 * k;
                const int f = min(tc.f0 + 2 * hw + (px >> 5), F - 1), t = min(tc.t0 + (px & 31), Tn - 1);
                yv[k] = *(const uint4*)((const uint16_t*)a.bn_y + (((long)tc.b * F + f) * Tn + t) * 64 + (lane & 7) * 8);
            }
        }
        uint2 av[C1RED ? 2 : 1];                               // C1RED: the 4 input channels of this lane's pixel (lane & 31) in both rows
        bool pv[C1RED ? 2 : 1];
        if (C1RED) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int f = tc.f0 + 2 * hw + j, t = tc.t0 + (lane & 31);
                pv[j] = f < F && t < Tn;
                av[j] = *(const uint2*)((const uint16_t*)a.c1_a0 + (((long)tc.b * F + min(f, F - 1)) * Tn + min(t, Tn - 1)) * 4);
            }
        }
        half_barrier(cnt, epoch, lane);                        // the half is done reading its input tile
        uint16_t* stg = sX + hw * (64 * 64);                   // this wave's [64 px][64 co] slice (px = row * 32 + column)
        if (C1RED) {
            // (1) mask: pre-activation tile of the first layer in the accumulators' own layout (co x pixel), one MFMA per tile
            __builtin_amdgcn_s_waitcnt(0x0F70);                // av landed (the next tile's prefetch too: it had the whole MFMA loop)
            if constexpr (!__is_same(TY, bf16)) {              // saved fp16 input -> the bf16 operand of the two first-layer contractions below
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    av[j].x = pack2_bf16(H16<TY>::lo(av[j].x), H16<TY>::hi(av[j].x));
                    av[j].y = pack2_bf16(H16<TY>::lo(av[j].y), H16<TY>::hi(av[j].y));
                }
            }
            uint16_t* a0t = sX + 4 * (64 * 64) + hw * 256;     // [4 c][64 px] of this wave, in the part of the input tile no slice uses
            {
                const int j = lane >> 5;                       // lanes < 32 file row 0, lanes >= 32 row 1 (both hold both rows' pixels)
                const uint2 q = j ? av[1] : av[0];
                uint16_t* d = a0t + j * 32 + (lane & 31);
                d[0] = (uint16_t)(q.x & 0xffffu); d[64] = (uint16_t)(q.x >> 16); d[128] = (uint16_t)(q.y & 0xffffu); d[192] = (uint16_t)(q.y >> 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 wfr = *(const bf16x8*)&sC1[(i * 32 + (lane & 31)) * 16 + (lane >> 5) * 8];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    union { uint4 u; bf16x8 b; } bx;
                    bx.u = make_uint4(av[j].x, av[j].y, 0x00003f80u, 0u);          // [a0_0..a0_3, 1, 0, 0, 0]
                    f32x16 y;
#pragma unroll
                    for (int r = 0; r < 16; ++r) y[r] = 0.f;
                    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr, bx.b, y, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = (y[r] > 0.f && pv[j]) ? acc[i][j][r] : 0.f;
                }
            }
            // (2) masked gradient tile, bf16, [64 px][64 co] in the transpose-read layout (swz)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        uint2 w2;
                        w2.x = pack2_bf16(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1]);
                        w2.y = pack2_bf16(acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
                        *(uint2*)(stg + swz(j * 32 + (lane & 31), i * 4 + g) + 4 * (lane >> 5)) = w2;
                    }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            // (3) G[c'][co] += sum_px [a0 | 1][px][c'] * g[px][co]: 4 k-steps of 16 pixels x 2 channel halves
            f32x16 d2[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) d2[nb][r] = 0.f;
            const int m = lane & 31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                union { uint4 u; bf16x8 b; } ax;
                ax.u = *(const uint4*)(a0t + min(m, 3) * 64 + ks * 16 + (lane >> 5) * 8);
                if (m == 4) ax.u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
                else if (m > 4) ax.u = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    d2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax.b, tr_frag(stg, ks * 16, nb * 32, lane), d2[nb], 0, 0, 0);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) gacc[nb][r] += d2[nb][r];
        } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // chunk (i*4+g) ^ (px & 7) only depends on the lane (px & 7 == lane & 7 for both rows): one address per (i, g)
                uint16_t* q = stg + (lane & 31) * 64 + ((((i * 4 + g) ^ (lane & 7)) << 3) | (4 * (lane >> 5)));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint2 w2;
                    w2.x = H16<TM>::pack(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1]);
                    w2.y = H16<TM>::pack(acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
                    *(uint2*)(q + j * 32 * 64) = w2;
                }
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (BNRED) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        // Interior tiles (every tile of a 256-multiple image): branch-free drain - all eight LDS reads in flight at once, one scalar base
        // per wave, lane-constant byte offset, the row / column-block offsets immediate or scalar.  Round-3 stamps: the per-chunk validity
        // branches (s_and_saveexec + branch around every store) made this phase a chain of eight LDS round trips, 2.4-4.5 k cycles per tile.
        const bool interior = !BNRED && (tc.f0 + TR <= F) && (tc.t0 + PTC <= Tn);
        if (interior) {
            uint4 o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = *(const uint4*)&stg[((lane >> 3) + 8 * k) * 64 + (((lane & 7) ^ (lane >> 3)) << 3)];
            char* obase = (char*)a.out + (((long)tc.b * F + tc.f0 + 2 * hw) * Tn + tc.t0) * 128;
            const unsigned loff = (unsigned)((lane >> 3) * 128 + (lane & 7) * 16), rowb = (unsigned)Tn * 128u;
#pragma unroll
            for (int k = 0; k < 8; ++k) *(uint4*)(obase + loff + (k & 3) * (8 * 128) + (k >> 2) * rowb) = o[k];
            if (a.stats) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t w[4] = {o[k].x, o[k].y, o[k].z, o[k].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = H16<TM>::lo(w[q]), hi = H16<TM>::hi(w[q]);
                        ssum[2 * q] += lo; ssq[2 * q] += lo * lo; ssum[2 * q + 1] += hi; ssq[2 * q + 1] += hi * hi;
                    }
                }
            }
        } else
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int px = (lane >> 3) + 8 * k;
            const int f = tc.f0 + 2 * hw + (px >> 5), t = tc.t0 + (px & 31);
            // (px & 7) == (lane >> 3) for every k: lane-constant chunk, k only moves the row offset
            const uint4 o = *(const uint4*)&stg[px * 64 + (((lane & 7) ^ (lane >> 3)) << 3)];
            if (f < F && t < Tn) {
                *(uint4*)((uint16_t*)a.out + (((long)tc.b * F + f) * Tn + t) * 64 + (lane & 7) * 8) = o;
                if (BNRED) {            // sums of g and g*y; turned into rstd*(sum g*y - mean*sum g) per tile below
                    const uint32_t w[4] = {o.x, o.y, o.z, o.w};
                    const uint32_t yw[4] = {yv[k].x, yv[k].y, yv[k].z, yv[k].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float d0 = bf16_bits_to_f32(w[q] & 0xffffu), d1 = __uint_as_float(w[q] & 0xffff0000u);
                        const float y0 = H16<TY>::lo(yw[q]), y1 = H16<TY>::hi(yw[q]);
                        const float t0_ = __uint_as_float(__float_as_uint(y0) ^ (((bsgn >> (2 * q)) & 1u) << 31));
                        const float t1_ = __uint_as_float(__float_as_uint(y1) ^ (((bsgn >> (2 * q + 1)) & 1u) << 31));
                        const float g0 = t0_ > bthr[2 * q] ? d0 : 0.f, g1 = t1_ > bthr[2 * q + 1] ? d1 : 0.f;
                        ssum[2 * q] += g0; ssq[2 * q] = fmaf(g0, y0, ssq[2 * q]);
                        ssum[2 * q + 1] += g1; ssq[2 * q + 1] = fmaf(g1, y1, ssq[2 * q + 1]);
                    }
                } else if (a.stats) {
                    const uint32_t w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = H16<TM>::lo(w[q]), hi = H16<TM>::hi(w[q]);
                        ssum[2 * q] += lo; ssq[2 * q] += lo * lo; ssum[2 * q + 1] += hi; ssq[2 * q + 1] += hi * hi;
                    }
                }
            }
        }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // my reads of the slice have landed (lgkmcnt(0)) ...
        if (lane == 0) atomicAdd(&sFree[half], 1u);            // ... the staging waves may overwrite the buffer once all four say so
    }
    // per-channel sums: the thread's 8 channels were accumulated over ALL its tiles in registers (folding them per tile - 48 lane
    // exchanges + 16 LDS atomics - showed as ~2.5 k of a half-tile's ~22 k cycles in the stamps); one fold per launch.
    // The fold is ORDERED (round 3): every wave parks its 128 partial sums in a slot of its own (the weight table is free once all
    // eight waves are past their last tile) and thread c adds the eight slots in wave order - the round-2 f32 LDS atomics summed them
    // in arrival order, which flipped bf16 roundings downstream from run to run.  What leaves the workgroup is one f64 atomic per
    // channel of f32-valued terms: exact (order-free) as long as the terms' exponents span < 2^20.
    __syncthreads();                               // every wave is done with sW
    if (a.clk && blockIdx.x == 0 && tid == 0) { a.clk[2] = __builtin_amdgcn_s_memtime(); a.clk[3] = __builtin_amdgcn_s_memrealtime(); }
    float* part = (float*)sW;                      // [8 waves][128] statistics, then [8][320] first-layer sums (C1RED)
    if (BNRED || a.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ssum[e] += __shfl_xor(ssum[e], 8, 64); ssum[e] += __shfl_xor(ssum[e], 16, 64); ssum[e] += __shfl_xor(ssum[e], 32, 64);
            ssq[e] += __shfl_xor(ssq[e], 8, 64); ssq[e] += __shfl_xor(ssq[e], 16, 64); ssq[e] += __shfl_xor(ssq[e], 32, 64);
        }
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s2 = ssq[e];
                if (BNRED) s2 = sAff[192 + lane * 8 + e] * (s2 - sAff[128 + lane * 8 + e] * ssum[e]);     // rstd * (sum g*y - mean * sum g)
                part[wave * 128 + lane * 8 + e] = ssum[e]; part[wave * 128 + 64 + lane * 8 + e] = s2;
            }
        }
        __syncthreads();
        if (a.stats && tid < 128) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += part[w * 128 + tid];
            atomicAdd(&a.stats[tid], (double)s);
        }
    }
    if (C1RED) {                                   // fold the 8 waves through LDS in wave order, then f64 atomics
        float* fold = part + 1024;                 // [8][320]: [0,256) G[co][c], [256,320) s1[co]
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int co = nb * 32 + (lane & 31);
            if (lane < 32) {
#pragma unroll
                for (int r = 0; r < 4; ++r) fold[wave * 320 + co * 4 + r] = gacc[nb][r];
            } else fold[wave * 320 + 256 + co] = gacc[nb][0];
        }
        __syncthreads();
        for (int q = tid; q < 320; q += 512) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += fold[w * 320 + q];
            atomicAdd(&a.c1_red[q < 256 ? q : 512 + (q - 256)], (double)s);
        }
    }
}

// ------------------------------------------------------------------------------------ wgrad
struct WgradArgs {
    const void* dy;       // (B,F,T,64) gradient w.r.t. the conv output
    const void* zin;      // (B,F,T,64) conv input before prologue
    const float* scale; const float* shift; int prologue;
    float* partial;       // [gridDim.x * 2][9][64][64] f32
    int nb, F, T;
    int part_dy, part_z;
    const float* c1_w;    // first-layer input mode (see ConvArgs::c1_w): zin = a0 (B,F,T,4), the operand relu(bn1(W1 a0)) is formed while staging
};

// Weight-gradient tiles use a column-parity swizzle: 16-byte chunk index XOR 4*((column >> 1) & 1).  A 32-lane transpose read
// touches 4 consecutive pixels x 4 consecutive chunks; pixels x and x+2 share a bank half, and this flips which four chunks of it
// they use - conflict-free for every tap (the forward kernel's (pixel>>1)&7 swizzle is 2-way conflicted for these reads), and the
// XOR term only depends on the column within a 16-pixel block, so fragment addresses are lane constant + affine offset.
__device__ __forceinline__ int swzc(int p, int col, int chunk) { return (p * 8 + (chunk ^ (((col >> 1) & 1) << 2))) * 8; }
__device__ __forceinline__ bf16x8 tr_pair(const uint16_t* p0, const uint16_t* p1) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    union { s16x4 v[2]; bf16x8 b; } u;
    u.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    u.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    return u.b;
}

// 8 waves = (co half) x (ci half) x (tap group: taps 0-4 | taps 5-8).  Every wave walks ALL pixels of the tile, so it only
// needs 5 (4) accumulator fragments = 80 VGPRs; the registers that frees hold the NEXT tile (z halo + dy, 19 x 16 B per
// thread) which is fetched from HBM while the current tile is on the matrix cores.
template <typename T>
__global__ __launch_bounds__(512) void conv3x3_wgrad_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) uint16_t sY[Y_ELEMS];     // dy tile  [8*64 px][64 co]
    __shared__ __attribute__((aligned(16))) uint16_t sX[X_ELEMS];     // z halo tile [660 px][64 ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = (wave >> 1) & 1, wt = wave >> 2;     // co half, ci half, tap group
    const int tap0 = wt ? 5 : 0, ntap = wt ? 4 : 5;
    const int F = a.F, Tn = a.T;
    const int tiles_f = (F + TR - 1) / TR, tiles_t = (Tn + TCOL - 1) / TCOL;
    const int ntiles = a.nb * tiles_f * tiles_t;
    const T* zin = (const T*)a.zin;
    const T* dy = (const T*)a.dy;
    const int cch = tid & 7;

    f32x16 acc[5];
#pragma unroll
    for (int t9 = 0; t9 < 5; ++t9)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t9][r] = 0.f;

    // lane-constant parts of the transpose-read addresses (element offsets into sY / sX) for fragment halves h = 0, 1:
    // A (dy): channels wi*32.., pixels +loff(h);  B (z) for this wave's tap tt: channels wj*32.., pixel (kh, kw + loff(h))
    int baseA[2], baseB[5][2];
    {
        const int chA = wi * 32 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4, chB = wj * 32 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int loff = (lane >> 5) * 8 + h * 4 + ((lane & 15) >> 2);
            baseA[h] = swzc(loff, loff, chA >> 3) + (chA & 7);
#pragma unroll
            for (int tt = 0; tt < 5; ++tt) {
                const int tap = tap0 + (tt < ntap ? tt : 0);
                const int kh = tap / 3, kw = tap - kh * 3;
                baseB[tt][h] = swzc(kh * HC + kw + loff, kw + loff, chB >> 3) + (chB & 7);
            }
        }
    }

    // staging as in the forward kernel: thread = (pixel column pc, 8-channel chunk), halo rows 0..9 + one chunk of columns 64/65
    Chunk<T> rz[X_ITERS], ry[8];
    const int pc = tid >> 3;
    auto issue_loads = [&](int tile) {
        const TileCoord tc = tile_coord(tile, tiles_f, tiles_t);
        const int t = tc.t0 - 1 + pc;
#pragma unroll
        for (int i = 0; i < HR; ++i) rz[i] = load_chunk_clamped<T>(zin, tc.b, tc.f0 - 1 + i, t, F, Tn, cch * 8);
        {
            const int hr = pc >> 1, te = tc.t0 + TCOL - 1 + (pc & 1);       // (threads >= 160: an unused, harmless extra chunk)
            rz[HR] = load_chunk_clamped<T>(zin, tc.b, tc.f0 - 1 + hr, te, F, Tn, cch * 8);
        }
        const int ty = tc.t0 + pc;
#pragma unroll
        for (int i = 0; i < 8; ++i) ry[i] = load_chunk_clamped<T>(dy, tc.b, tc.f0 + i, ty, F, Tn, cch * 8);
    };
    auto write_tile = [&](int tile) {
        const TileCoord tc = tile_coord(tile, tiles_f, tiles_t);
        float sc[8], sh[8];                                    // live only while staging
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = a.prologue ? a.scale[cch * 8 + e] : 1.f; sh[e] = a.prologue ? a.shift[cch * 8 + e] : 0.f; }
        const int t = tc.t0 - 1 + pc;
        const bool tv = t >= 0 && t < Tn;
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const int f = tc.f0 - 1 + i;
            *(uint4*)&sX[swzc(i * HC + pc, pc, cch)] = xform_chunk<T>(rz[i], tv && f >= 0 && f < F, a.prologue, sc, sh, a.part_z);
        }
        if (tid < 160) {
            const int hr = pc >> 1, f = tc.f0 - 1 + hr, te = tc.t0 + TCOL - 1 + (pc & 1);
            *(uint4*)&sX[swzc(hr * HC + TCOL + (pc & 1), TCOL + (pc & 1), cch)] = xform_chunk<T>(rz[HR], f >= 0 && f < F && te < Tn, a.prologue, sc, sh, a.part_z);
        }
        const int ty = tc.t0 + pc;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *(uint4*)&sY[swzc(i * 64 + pc, pc, cch)] = xform_chunk<T>(ry[i], tc.f0 + i < F && ty < Tn, 0, sc, sh, a.part_dy);
    };

    const int nrounds = (ntiles + gridDim.x - 1) / gridDim.x;
    if (xcd_tile(0, blockIdx.x, gridDim.x) < ntiles) issue_loads(xcd_tile(0, blockIdx.x, gridDim.x));
    for (int it = 0; it < nrounds; ++it) {
        const int tile = xcd_tile(it, blockIdx.x, gridDim.x);
        if (tile >= ntiles) break;
        write_tile(tile);
        __syncthreads();
        const int next = (it + 1 < nrounds) ? xcd_tile(it + 1, blockIdx.x, gridDim.x) : ntiles;
        if (next < ntiles) issue_loads(next);
        // Transpose-read addresses are lane constant + (row, column block) offset (see swzc): one VGPR add per fragment half and
        // row, the column block goes into the instruction's immediate offset.
#pragma unroll 1
        for (int r = 0; r < TR; ++r) {
            const uint16_t* ya[2] = {sY + baseA[0] + r * (64 * 64), sY + baseA[1] + r * (64 * 64)};
            const uint16_t* xb[5][2];
#pragma unroll
            for (int tt = 0; tt < 5; ++tt) { xb[tt][0] = sX + baseB[tt][0] + r * (HC * 64); xb[tt][1] = sX + baseB[tt][1] + r * (HC * 64); }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int co_ = cb * 16 * 64;                  // 16 pixels x 64 channels further on
                const bf16x8 fa = tr_pair(ya[0] + co_, ya[1] + co_);
                bf16x8 fb[2];
                fb[0] = tr_pair(xb[0][0] + co_, xb[0][1] + co_);
#pragma unroll
                for (int tt = 0; tt < 5; ++tt) {
                    if (tt + 1 < 5 && tt + 1 < ntap) fb[(tt + 1) & 1] = tr_pair(xb[tt + 1][0] + co_, xb[tt + 1][1] + co_);
                    __builtin_amdgcn_sched_barrier(0);         // pin the next tap's transpose reads above this MFMA
                    if (tt < ntap) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[tt & 1], acc[tt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // D[co][ci]: lane: ci = lane&31, co = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* P = a.partial + (long)blockIdx.x * W_ELEMS;
#pragma unroll
    for (int tt = 0; tt < 5; ++tt) {
        if (tt < ntap) {
            const int tap = tap0 + tt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ci = wj * 32 + (lane & 31);
                P[(tap * 64 + co) * 64 + ci] = acc[tt][r];
            }
        }
    }
}

// ---- weight gradient, double-buffered variant (bf16).  In conv3x3_wgrad_kernel all eight waves stage, barrier, run their MFMAs,
// barrier: the matrix pipe idles for the whole staging phase (one tile = ~25 k cycles of which 2 x 5.1 k are MFMA, mfma_busy 0.38 in
// the round-2 counters) and the 150 KB of one 8 x 64-pixel tile pair leave no room for a second buffer.  Here the tile is 8 x 32 pixels
// (z halo 43.5 KB + dy 32.8 KB), LDS holds TWO of them, the next tile is written into the other buffer right behind this tile's
// MFMAs and there is ONE barrier per tile; the waves of a SIMD drift against each other inside a tile, so one wave's staging (VALU,
// LDS writes) runs under the other's MFMAs.  Same wave roles / accumulators / partial-sum output as conv3x3_wgrad_kernel.
#define WTC 32
#define WHC (WTC + 2)
#define WX_ELEMS (HR * WHC * 64)
#define WY_ELEMS (TR * WTC * 64)
template <bool C1IN = false, typename TA = bf16>      // TA: encoding of the saved forward operand (zin / a0); dy and the contraction are bf16
__global__ __launch_bounds__(512) void conv3x3_wgrad_db_kernel(WgradArgs a) {
    typedef bf16 T;
    __shared__ __attribute__((aligned(16))) uint16_t sYb[2][WY_ELEMS];   // dy tiles  [8*32 px][64 co]
    __shared__ __attribute__((aligned(16))) uint16_t sXb[2][WX_ELEMS];   // z halo tiles [340 px][64 ci]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = (wave >> 1) & 1, wt = wave >> 2;     // co half, ci half, tap group
    const int tap0 = wt ? 5 : 0, ntap = wt ? 4 : 5;
    const int F = a.F, Tn = a.T;
    const int tiles_f = (F + TR - 1) / TR, tiles_t = (Tn + WTC - 1) / WTC;
    const int ntiles = a.nb * tiles_f * tiles_t;
    const TA* zin = (const TA*)a.zin;
    const T* dy = (const T*)a.dy;
    const int cch = tid & 7;

    f32x16 acc[5];
#pragma unroll
    for (int t9 = 0; t9 < 5; ++t9)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t9][r] = 0.f;

    int baseA[2], baseB[5][2];
    {
        const int chA = wi * 32 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4, chB = wj * 32 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int loff = (lane >> 5) * 8 + h * 4 + ((lane & 15) >> 2);
            baseA[h] = swzc(loff, loff, chA >> 3) + (chA & 7);
#pragma unroll
            for (int tt = 0; tt < 5; ++tt) {
                const int tap = tap0 + (tt < ntap ? tt : 0);
                const int kh = tap / 3, kw = tap - kh * 3;
                baseB[tt][h] = swzc(kh * WHC + kw + loff, kw + loff, chB >> 3) + (chB & 7);
            }
        }
    }
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = a.prologue ? a.scale[cch * 8 + e] : 1.f; sh[e] = a.prologue ? a.shift[cch * 8 + e] : 0.f; }
    C1Const kc1;
    if (C1IN) c1_setup(kc1, a.c1_w, a.scale, a.shift, cch * 8);
    auto load_z = [&](int b, int f, int t) {                    // clamped, unconditional (load_chunk_clamped); C1IN: the pixel's 4 input channels
        Chunk<TA> c;
        if (C1IN) {
            f = min(max(f, 0), F - 1); t = min(max(t, 0), Tn - 1);
            const uint2 q = *(const uint2*)(zin + (((long)b * F + f) * Tn + t) * 4);
            c.u.x = q.x; c.u.y = q.y;
        } else c = load_chunk_clamped<TA>(zin, b, f, t, F, Tn, cch * 8);
        return c;
    };
    auto xform_z = [&](const Chunk<TA>& c, bool ok) { return C1IN ? c1_chunk<TA, bf16>(c.u.x, c.u.y, ok, kc1) : xform_chunk<TA, bf16>(c, ok, a.prologue, sc, sh, 0); };

    // staging: thread = (row parity pr, pixel column pcol of 32, 8-channel chunk): halo rows pr, pr+2, .. pr+8 and dy rows pr, pr+2, ..
    // pr+6 of its column; threads < 160 also one chunk of halo columns 32 / 33
    // (round 3: ~1100 vector instructions per wave and tile, a third of them addresses - 64-bit products per load, the LDS swizzle per
    //  store.  The row parity is wave-uniform, so row bases live on the scalar unit: one 64-bit image base per tile + 32-bit row offsets,
    //  one vector byte offset per thread for all rows of a tensor, one lane-constant LDS base per tensor with the row step as an
    //  immediate: 1426 -> 1019 vector issue slots per tile.  The launch time did not move (379 us alone, +0.2 % on the step): like the
    //  operand-read and look-ahead experiments in tools/conv_ng3/, it says this kernel is bound by none of them.)
    Chunk<TA> rz[6]; Chunk<T> ry[4];
    const int pcol = (tid >> 3) & 31;
    const int pr = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int lbX = swzc(pr * WHC + pcol, pcol, cch), lbY = swzc(pr * WTC + pcol, pcol, cch);     // LDS element offsets of row pr; row pr + 2k: + k * 2 * W?C * 64
    auto coord = [&](int tile) { TileCoord c; c.t0 = (tile % tiles_t) * WTC; tile /= tiles_t; c.f0 = (tile % tiles_f) * TR; c.b = tile / tiles_f; return c; };
    auto issue_loads = [&](const TileCoord tc) {
        constexpr unsigned PXB = C1IN ? 8u : 128u;                               // bytes per pixel of zin
        const char* zimg = (const char*)zin + (long)tc.b * F * (long)Tn * PXB;   // (an image is < 4 GB: 32-bit offsets inside it)
        const char* yimg = (const char*)dy + (long)tc.b * F * (long)Tn * 128;
        const unsigned zrow = (unsigned)Tn * PXB, yrow = (unsigned)Tn * 128u;
        const int tz = min(max(tc.t0 - 1 + pcol, 0), Tn - 1), ty = min(tc.t0 + pcol, Tn - 1);      // clamped: unconditional loads
        const unsigned vz = C1IN ? (unsigned)tz * 8u : (unsigned)(tz * 64 + cch * 8) * 2u, vy = (unsigned)(ty * 64 + cch * 8) * 2u;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int f = min(max(tc.f0 - 1 + pr + 2 * k, 0), F - 1);
            const char* prow = zimg + (unsigned)f * zrow;
            if (C1IN) { const uint2 q = *(const uint2*)(prow + vz); rz[k].u.x = q.x; rz[k].u.y = q.y; }
            else rz[k].u = *(const uint4*)(prow + vz);
        }
        {
            const int q = tid >> 3, hr = q >> 1, te = tc.t0 + WTC - 1 + (q & 1);        // (threads >= 160: an unused, harmless extra chunk)
            rz[5] = load_z(tc.b, tc.f0 - 1 + hr, te);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int f = min(tc.f0 + pr + 2 * k, F - 1);
            ry[k].u = *(const uint4*)(yimg + (unsigned)f * yrow + vy);
        }
    };
    // the tile is written in three pieces (halo rows 0-2 of this thread | halo rows 3-4 + edge columns | dy rows) so that the pieces can be
    // placed between the row iterations of the PREVIOUS tile's MFMA loop
    auto write_piece = [&](const int piece, const TileCoord tc, uint16_t* __restrict__ sX, uint16_t* __restrict__ sY) {
        if (piece < 2) {
            const int t = tc.t0 - 1 + pcol;
            const bool tv = t >= 0 && t < Tn;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                if ((piece == 0) != (k < 3)) continue;
                const int f = tc.f0 - 1 + pr + 2 * k;
                *(uint4*)&sX[lbX + k * (2 * WHC * 64)] = xform_z(rz[k], tv && f >= 0 && f < F);
            }
            if (piece == 1 && tid < 160) {
                const int q = tid >> 3, hr = q >> 1, f = tc.f0 - 1 + hr, te = tc.t0 + WTC - 1 + (q & 1);
                *(uint4*)&sX[swzc(hr * WHC + WTC + (q & 1), WTC + (q & 1), cch)] = xform_z(rz[5], f >= 0 && f < F && te < Tn);
            }
        } else {
            const int ty = tc.t0 + pcol;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *(uint4*)&sY[lbY + k * (2 * WTC * 64)] = xform_chunk<T>(ry[k], tc.f0 + pr + 2 * k < F && ty < Tn, 0, sc, sh, 0);
            }
        }
    };
    auto write_tile = [&](const TileCoord tc, uint16_t* __restrict__ sX, uint16_t* __restrict__ sY) {
        write_piece(0, tc, sX, sY); write_piece(1, tc, sX, sY); write_piece(2, tc, sX, sY);
    };

    const int nrounds = (ntiles + gridDim.x - 1) / gridDim.x;
    int tile = xcd_tile(0, blockIdx.x, gridDim.x);
    TileCoord tc = coord(tile < ntiles ? tile : 0);
    if (tile < ntiles) {
        issue_loads(tc);
        write_tile(tc, sXb[0], sYb[0]);
    }
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < nrounds; ++it) {
        if (tile >= ntiles) break;
        const int next = (it + 1 < nrounds) ? xcd_tile(it + 1, blockIdx.x, gridDim.x) : ntiles;
        const TileCoord tcn = coord(next < ntiles ? next : 0);
        if (next < ntiles) issue_loads(tcn);
        const uint16_t* sX = sXb[cur];
        const uint16_t* sY = sYb[cur];
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const uint16_t* ya[2] = {sY + baseA[0] + r * (WTC * 64), sY + baseA[1] + r * (WTC * 64)};
            const uint16_t* xb[5][2];
#pragma unroll
            for (int tt = 0; tt < 5; ++tt) { xb[tt][0] = sX + baseB[tt][0] + r * (WHC * 64); xb[tt][1] = sX + baseB[tt][1] + r * (WHC * 64); }
#pragma unroll
            for (int cb = 0; cb < WTC / 16; ++cb) {
                const int co_ = cb * 16 * 64;                  // 16 pixels x 64 channels further on
                const bf16x8 fa = tr_pair(ya[0] + co_, ya[1] + co_);
                bf16x8 fb[2];
                fb[0] = tr_pair(xb[0][0] + co_, xb[0][1] + co_);
#pragma unroll
                for (int tt = 0; tt < 5; ++tt) {
                    if (tt + 1 < 5 && tt + 1 < ntap) fb[(tt + 1) & 1] = tr_pair(xb[tt + 1][0] + co_, xb[tt + 1][1] + co_);
                    __builtin_amdgcn_sched_barrier(0);         // pin the next tap's transpose reads above this MFMA
                    if (tt < ntap) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[tt & 1], acc[tt], 0, 0, 0);
                }
            }
            // next tile -> the other buffer, in pieces behind the later rows (its loads were issued before this tile's first MFMA and
            // have had rows 0-4 to land): this wave's VALU / LDS-write work runs under the MFMAs it has just queued and the other
            // wave's on the same SIMD
            if (next < ntiles && r >= 5) write_piece(r - 5, tcn, sXb[cur ^ 1], sYb[cur ^ 1]);
        }
        __syncthreads();
        cur ^= 1; tile = next; tc = tcn;
    }
    float* P = a.partial + (long)blockIdx.x * W_ELEMS;
#pragma unroll
    for (int tt = 0; tt < 5; ++tt) {
        if (tt < ntap) {
            const int tap = tap0 + tt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ci = wj * 32 + (lane & 31);
                P[(tap * 64 + co) * 64 + ci] = acc[tt][r];
            }
        }
    }
}

// Role-split variant of conv3x3_wgrad_db_kernel: 16 waves, waves 0-7 only run the MFMA loop (same decomposition: co half x ci half x tap
// group), waves 8-15 only stage (global loads one tile ahead in registers, BatchNorm + ReLU / first-layer prologue, LDS writes).  Ablation
// builds of the lockstep kernel ran 254 us without its staging and 268 us without its MFMAs, 378 us with both in the same waves.
template <bool C1IN = false, typename TA = bf16>
__global__ __launch_bounds__(1024) void conv3x3_wgrad_ws_kernel(WgradArgs a) {
    typedef bf16 T;
    __shared__ __attribute__((aligned(16))) uint16_t sYb[2][WY_ELEMS];   // dy tiles  [8*32 px][64 co]
    __shared__ __attribute__((aligned(16))) uint16_t sXb[2][WX_ELEMS];   // z halo tiles [340 px][64 ci]
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave_all >= 8;
    const int tid = threadIdx.x & 511;                                  // thread id inside the role (8 waves each)
    const int wave = wave_all & 7;
    const int wi = wave & 1, wj = (wave >> 1) & 1, wt = wave >> 2;     // consumers: co half, ci half, tap group
    const int tap0 = wt ? 5 : 0, ntap = wt ? 4 : 5;
    const int F = a.F, Tn = a.T;
    const int tiles_f = (F + TR - 1) / TR, tiles_t = (Tn + WTC - 1) / WTC;
    const int ntiles = a.nb * tiles_f * tiles_t;
    const TA* zin = (const TA*)a.zin;
    const T* dy = (const T*)a.dy;
    const int cch = tid & 7;

    const int nrounds = (ntiles + gridDim.x - 1) / gridDim.x;
    int tile = xcd_tile(0, blockIdx.x, gridDim.x);
    if (producer) {
        // ---- staging waves: tile `it + 1` is written into the free buffer while the MFMA waves are on tile `it`; the loads of tile
        // `it + 2` are issued right behind the writes and have the rest of the iteration (the wait at the barrier) to land
        float sc[8], sh[8];
    #pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = a.prologue ? a.scale[cch * 8 + e] : 1.f; sh[e] = a.prologue ? a.shift[cch * 8 + e] : 0.f; }
        C1Const kc1;
        if (C1IN) c1_setup(kc1, a.c1_w, a.scale, a.shift, cch * 8);
        auto load_z = [&](int b, int f, int t) __attribute__((always_inline)) {                    // clamped, unconditional (load_chunk_clamped); C1IN: the pixel's 4 input channels
            Chunk<TA> c;
            if (C1IN) {
                f = min(max(f, 0), F - 1); t = min(max(t, 0), Tn - 1);
                const uint2 q = *(const uint2*)(zin + (((long)b * F + f) * Tn + t) * 4);
                c.u = make_uint4(q.x, q.y, 0u, 0u);      // (whole object: a partially written chunk carried over the loop edge stays on the stack)
            } else c = load_chunk_clamped<TA>(zin, b, f, t, F, Tn, cch * 8);
            return c;
        };
        auto xform_z = [&](const Chunk<TA>& c, bool ok) __attribute__((always_inline)) { return C1IN ? c1_chunk<TA, bf16>(c.u.x, c.u.y, ok, kc1) : xform_chunk<TA, bf16>(c, ok, a.prologue, sc, sh, 0); };

        // staging: thread = (row parity pr, pixel column pcol of 32, 8-channel chunk): halo rows pr, pr+2, .. pr+8 and dy rows pr, pr+2, ..
        // pr+6 of its column; threads < 160 also one chunk of halo columns 32 / 33
        // (round 3: ~1100 vector instructions per wave and tile, a third of them addresses - 64-bit products per load, the LDS swizzle per
        //  store.  The row parity is wave-uniform, so row bases live on the scalar unit: one 64-bit image base per tile + 32-bit row offsets,
        //  one vector byte offset per thread for all rows of a tensor, one lane-constant LDS base per tensor with the row step as an
        //  immediate: 1426 -> 1019 vector issue slots per tile.  The launch time did not move (379 us alone, +0.2 % on the step): like the
        //  operand-read and look-ahead experiments in tools/conv_ng3/, it says this kernel is bound by none of them.)
        Chunk<TA> rz[6]; Chunk<T> ry[4];
        const int pcol = (tid >> 3) & 31;
        const int pr = __builtin_amdgcn_readfirstlane(tid >> 8);
        const int lbX = swzc(pr * WHC + pcol, pcol, cch), lbY = swzc(pr * WTC + pcol, pcol, cch);     // LDS element offsets of row pr; row pr + 2k: + k * 2 * W?C * 64
        auto coord = [&](int tile) __attribute__((always_inline)) { TileCoord c; c.t0 = (tile % tiles_t) * WTC; tile /= tiles_t; c.f0 = (tile % tiles_f) * TR; c.b = tile / tiles_f; return c; };
        auto issue_loads = [&](const TileCoord tc) __attribute__((always_inline)) {
            constexpr unsigned PXB = C1IN ? 8u : 128u;                               // bytes per pixel of zin
            const char* zimg = (const char*)zin + (long)tc.b * F * (long)Tn * PXB;   // (an image is < 4 GB: 32-bit offsets inside it)
            const char* yimg = (const char*)dy + (long)tc.b * F * (long)Tn * 128;
            const unsigned zrow = (unsigned)Tn * PXB, yrow = (unsigned)Tn * 128u;
            const int tz = min(max(tc.t0 - 1 + pcol, 0), Tn - 1), ty = min(tc.t0 + pcol, Tn - 1);      // clamped: unconditional loads
            const unsigned vz = C1IN ? (unsigned)tz * 8u : (unsigned)(tz * 64 + cch * 8) * 2u, vy = (unsigned)(ty * 64 + cch * 8) * 2u;
    #pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int f = min(max(tc.f0 - 1 + pr + 2 * k, 0), F - 1);
                const char* prow = zimg + (unsigned)f * zrow;
                if (C1IN) { const uint2 q = *(const uint2*)(prow + vz); rz[k].u = make_uint4(q.x, q.y, 0u, 0u); }
                else rz[k].u = *(const uint4*)(prow + vz);
            }
            {
                const int q = tid >> 3, hr = q >> 1, te = tc.t0 + WTC - 1 + (q & 1);        // (threads >= 160: an unused, harmless extra chunk)
                rz[5] = load_z(tc.b, tc.f0 - 1 + hr, te);
            }
    #pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int f = min(tc.f0 + pr + 2 * k, F - 1);
                ry[k].u = *(const uint4*)(yimg + (unsigned)f * yrow + vy);
            }
        };
        // the tile is written in three pieces (halo rows 0-2 of this thread | halo rows 3-4 + edge columns | dy rows) so that the pieces can be
        // placed between the row iterations of the PREVIOUS tile's MFMA loop
        auto write_piece = [&](const int piece, const TileCoord tc, uint16_t* __restrict__ sX, uint16_t* __restrict__ sY) __attribute__((always_inline)) {
            if (piece < 2) {
                const int t = tc.t0 - 1 + pcol;
                const bool tv = t >= 0 && t < Tn;
    #pragma unroll
                for (int k = 0; k < 5; ++k) {
                    if ((piece == 0) != (k < 3)) continue;
                    const int f = tc.f0 - 1 + pr + 2 * k;
                    *(uint4*)&sX[lbX + k * (2 * WHC * 64)] = xform_z(rz[k], tv && f >= 0 && f < F);
                    }
                if (piece == 1 && tid < 160) {
                    const int q = tid >> 3, hr = q >> 1, f = tc.f0 - 1 + hr, te = tc.t0 + WTC - 1 + (q & 1);
                    *(uint4*)&sX[swzc(hr * WHC + WTC + (q & 1), WTC + (q & 1), cch)] = xform_z(rz[5], f >= 0 && f < F && te < Tn);
                }
            } else {
                const int ty = tc.t0 + pcol;
    #pragma unroll
                for (int k = 0; k < 4; ++k) {
                    *(uint4*)&sY[lbY + k * (2 * WTC * 64)] = xform_chunk<T>(ry[k], tc.f0 + pr + 2 * k < F && ty < Tn, 0, sc, sh, 0);
                }
            }
        };
        auto write_tile = [&](const TileCoord tc, uint16_t* __restrict__ sX, uint16_t* __restrict__ sY) __attribute__((always_inline)) {
            write_piece(0, tc, sX, sY); write_piece(1, tc, sX, sY); write_piece(2, tc, sX, sY);
        };

        TileCoord tc = coord(tile < ntiles ? tile : 0);
        if (tile < ntiles) { issue_loads(tc); write_tile(tc, sXb[0], sYb[0]); }
        int next = (1 < nrounds) ? xcd_tile(1, blockIdx.x, gridDim.x) : ntiles;
        TileCoord tcn = coord(next < ntiles ? next : 0);
        if (next < ntiles) issue_loads(tcn);
        __syncthreads();
        int cur = 0;
        for (int it = 0; it < nrounds; ++it) {
            if (tile >= ntiles) break;
            if (next < ntiles) write_tile(tcn, sXb[cur ^ 1], sYb[cur ^ 1]);
            const int nn = (it + 2 < nrounds) ? xcd_tile(it + 2, blockIdx.x, gridDim.x) : ntiles;
            const TileCoord tcnn = coord(nn < ntiles ? nn : 0);
            if (nn < ntiles) issue_loads(tcnn);
            __syncthreads();
            cur ^= 1; tile = next; next = nn; tcn = tcnn;
        }
        return;
    }
    // ---- MFMA waves
    f32x16 acc[5];
#pragma unroll
    for (int t9 = 0; t9 < 5; ++t9)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t9][r] = 0.f;

    int baseA[2], baseB[5][2];
    {
        const int chA = wi * 32 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4, chB = wj * 32 + 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int loff = (lane >> 5) * 8 + h * 4 + ((lane & 15) >> 2);
            baseA[h] = swzc(loff, loff, chA >> 3) + (chA & 7);
#pragma unroll
            for (int tt = 0; tt < 5; ++tt) {
                const int tap = tap0 + (tt < ntap ? tt : 0);
                const int kh = tap / 3, kw = tap - kh * 3;
                baseB[tt][h] = swzc(kh * WHC + kw + loff, kw + loff, chB >> 3) + (chB & 7);
            }
        }
    }
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < nrounds; ++it) {
        if (tile >= ntiles) break;
        const int next = (it + 1 < nrounds) ? xcd_tile(it + 1, blockIdx.x, gridDim.x) : ntiles;
        const uint16_t* sX = sXb[cur];
        const uint16_t* sY = sYb[cur];
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            const uint16_t* ya[2] = {sY + baseA[0] + r * (WTC * 64), sY + baseA[1] + r * (WTC * 64)};
            const uint16_t* xb[5][2];
#pragma unroll
            for (int tt = 0; tt < 5; ++tt) { xb[tt][0] = sX + baseB[tt][0] + r * (WHC * 64); xb[tt][1] = sX + baseB[tt][1] + r * (WHC * 64); }
#pragma unroll
            for (int cb = 0; cb < WTC / 16; ++cb) {
                const int co_ = cb * 16 * 64;                  // 16 pixels x 64 channels further on
                const bf16x8 fa = tr_pair(ya[0] + co_, ya[1] + co_);
                bf16x8 fb[2];
                fb[0] = tr_pair(xb[0][0] + co_, xb[0][1] + co_);
#pragma unroll
                for (int tt = 0; tt < 5; ++tt) {
                    if (tt + 1 < 5 && tt + 1 < ntap) fb[(tt + 1) & 1] = tr_pair(xb[tt + 1][0] + co_, xb[tt + 1][1] + co_);
                    __builtin_amdgcn_sched_barrier(0);         // pin the next tap's transpose reads above this MFMA
                    if (tt < ntap) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[tt & 1], acc[tt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        cur ^= 1; tile = next;
    }
    float* P = a.partial + (long)blockIdx.x * W_ELEMS;
#pragma unroll
    for (int tt = 0; tt < 5; ++tt) {
        if (tt < ntap) {
            const int tap = tap0 + tt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ci = wj * 32 + (lane & 31);
                P[(tap * 64 + co) * 64 + ci] = acc[tt][r];
            }
        }
    }
}

// dW[e] (+)= sum_p partial[p][e]; workgroup = 64 elements x 4 part-slots, 4-way unrolled loads
// grad_oihw != null: the sum is ADDED to the parameter-gradient buffer in nn.Conv2d's own (co, ci, kh, kw) layout instead.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ dW,
                                                           int accumulate, float* __restrict__ grad_oihw = nullptr) {
    __shared__ float sred[4][64];
    const int col = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = slot;
    for (; p + 12 < nparts; p += 16) {
        s0 += partial[(long)p * W_ELEMS + e]; s1 += partial[(long)(p + 4) * W_ELEMS + e];
        s2 += partial[(long)(p + 8) * W_ELEMS + e]; s3 += partial[(long)(p + 12) * W_ELEMS + e];
    }
    for (; p < nparts; p += 4) s0 += partial[(long)p * W_ELEMS + e];
    sred[slot][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slot == 0) {
        const float t = (sred[0][col] + sred[1][col]) + (sred[2][col] + sred[3][col]);
        if (grad_oihw) { const int tap = e >> 12, coci = e & 4095; grad_oihw[coci * 9 + tap] += t; }
        else dW[e] = accumulate ? dW[e] + t : t;
    }
}

// Workgroups of the persistent convolution launches.  kind 0 = forward launches, 1 = data / weight gradients.  One per CU, or
// SARSSL_CONV_CUS[_FWD | _BWD] of them: a convolution workgroup takes a CU's whole LDS and nearly all of its registers, so while a
// launch covers every CU the other encoder's stream stands still; a launch that leaves an eighth of the CUs free lets that stream's
// short latency-bound kernels run next to it (measured on the step, see NOTES.md 4.7).  The grid is then trimmed so that the last
// round of tiles is as full as the others (a multiple of 8 keeps the XCD-aware tile order).
// Per-context override of the workgroup count of the gradient launches (0 = the default rule below): the engine uses it to give the
// stem backward that runs LAST - alone on the chip, the other encoder's stream has drained - all CUs (sarssl_ctx_set_conv_cus).
static int conv_cus_override() { const sarssl_ctx* c = sarssl_current(); return c ? c->conv_cus_bwd : 0; }     // (sarssl_ctx_set_conv_cus)
static int conv_cus(int kind) {
    static const int lim[2] = {
        []() { const char* e = getenv("SARSSL_CONV_CUS_FWD"); if (!e) e = getenv("SARSSL_CONV_CUS"); return e ? atoi(e) : 0; }(),
        []() { const char* e = getenv("SARSSL_CONV_CUS_BWD"); if (!e) e = getenv("SARSSL_CONV_CUS"); return e ? atoi(e) : 0; }()};
    const int ncu = sarssl_cu_count();
    const int ovr = conv_cus_override();
    if (kind == 1 && ovr > 0) return ovr < ncu ? ovr : ncu;
    if (lim[kind] > 0) return lim[kind] < ncu ? lim[kind] : ncu;
    // default: forward launches on every CU, gradient launches on 7/8 of them (same-box A/B at B = 64, three rounds: 5 510 - 5 750
    // segments/s with 256 of 256, 5 736 - 5 750 with 224 - the step gains ~2 % although each gradient launch alone is ~12 % slower)
    return kind == 1 && ncu >= 64 ? (ncu * 7 / 8) & ~7 : ncu;
}
// Clock probe buffer of the current context (device memory, 5 slots x 4 u64: forward with BN prologue | data gradient | data gradient +
// BN sums | forward from the 4-channel input | data gradient consumed in its epilogue); null = off.  Set by bench.py around its
// event-timed launches (sarssl_ctx_set_clock_probe).
static unsigned long long* conv_clk() { const sarssl_ctx* c = sarssl_current(); return c ? c->conv_clk : nullptr; }  // (sarssl_ctx_set_clock_probe)
extern "C" long sarssl_wall_clock_khz() {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess) return 0;
    return (long)khz;
}
// forward / data-gradient kernel: the ping-pong kernel, or (SARSSL_CONV_WS=1) its role-split variant
// Which launches take it (SARSSL_CONV_WS, bit mask: 1 plain forward / data gradient, 2 forward from the 4-channel input, 4 data gradient
// consumed in its epilogue, 8 data gradient + BatchNorm sums; default 4).  In-step launch times, ping-pong -> role-split, same box:
// BatchNorm-prologue forward 0.298 -> 0.314 ms and 4-channel-input forward 0.351 -> 0.350 (one 4-wave staging team serves both groups:
// with a prologue to compute it becomes the bottleneck), data gradient + BatchNorm sums 0.361 -> 0.56 (80 B of scratch at the 168-register
// cap), data gradient consumed in its epilogue (identity prologue, no output drain) 0.345 -> **0.314**: step 10.72 -> 10.68 ms.
static bool conv_ws(int variant_bit) {
    static const int mask = []() { const char* e = getenv("SARSSL_CONV_WS"); return e ? atoi(e) : 4; }();
    return (mask & variant_bit) != 0;
}
static int conv_persistent_grid(int nunits, int kind) {
    const int cus = conv_cus(kind);
    if (nunits <= cus) return nunits;
    const int rounds = (nunits + cus - 1) / cus;
    int g = (nunits + rounds - 1) / rounds;
    g = (g + 7) & ~7;
    return g < cus ? g : cus;
}
static int wgrad_db_grid(int nb, int F, int T) {
    const int ntiles = nb * ((F + TR - 1) / TR) * ((T + WTC - 1) / WTC);
    return conv_persistent_grid(ntiles, 1);
}
static int conv_grid(int nb, int F, int T) {
    int ntiles = nb * ((F + TR - 1) / TR) * ((T + TCOL - 1) / TCOL);
    const int ncu = sarssl_cu_count();              // persistent: one workgroup per CU
    return ntiles < ncu ? ntiles : ncu;
}

// in/out: (B,F,T,64) channels-last, dtype 0 f32 / 1 bf16 (same for both).  w: [9][64][64] ([tap][co][ci]) of
// dtype w_dtype.  scale/shift: f32[64] prologue affine (+ReLU) or null for identity.
// precise (f32 only): 3-pass split with ws = f32 (B,F,T,64).
static int conv3x3_launch(const void* in, const void* w, void* out, int dtype, int w_dtype, int nb, int F, int T,
                          const float* scale, const float* shift, int precise, float* ws, double* stats, const void* bn_y,
                          const float* bn_aff, void* stream, int y_dtype = SARSSL_BF16);

extern "C" int sarssl_conv3x3_fwd(const void* in, const void* w, void* out, int dtype, int w_dtype, int nb, int F, int T,
                                  const float* scale, const float* shift, int precise, float* ws, double* stats, void* stream) {
    return conv3x3_launch(in, w, out, dtype, w_dtype, nb, F, T, scale, shift, precise, ws, stats, nullptr, nullptr, stream);
}

// Data gradient of a 3x3 convolution (w = flipped / transposed taps, no prologue) that also returns, in red = f64[128], the
// backward sums [sum g | sum g*xhat] of the BatchNorm + ReLU in front of that convolution (g = dz * relu'(bn(y)), y = its pre-BN
// activations, aff = [scale | shift | mean | rstd], 4 x 64 f32).  bf16 (ping-pong kernel).
// y_dtype: encoding of the saved activations y (SARSSL_BF16, or SARSSL_F16 in the fp16-forward mode); dy / w / dz are bf16.
extern "C" int sarssl_conv3x3_dgrad_bnred(const void* dy, const void* w, void* dz, int nb, int F, int T, const void* y,
                                          const float* aff, double* red, int y_dtype, void* stream) {
    SARSSL_REQUIRE(y != nullptr && aff != nullptr && red != nullptr && (y_dtype == SARSSL_BF16 || y_dtype == SARSSL_F16), "sarssl_conv3x3_dgrad_bnred");
    return conv3x3_launch(dy, w, dz, SARSSL_BF16, SARSSL_BF16, nb, F, T, nullptr, nullptr, 0, nullptr, red, y, aff, stream, y_dtype);
}

// 3x3 convolution of relu(bn1(W1 a0)) straight from the stem's 4-channel input a0 (B,F,T,4) bf16: W1 f32[64][4], scale / shift = bn1's
// affine; out (B,F,T,64) bf16 and, optionally, stats = [sum | sum of squares] of the stored output.  bf16 (ping-pong kernel).
// dtype: encoding of a0, w and out (SARSSL_BF16 or SARSSL_F16).
extern "C" int sarssl_conv3x3_fwd_c1(const void* a0, const float* W1, const float* scale, const float* shift, const void* w, void* out,
                                     int nb, int F, int T, double* stats, int dtype, void* stream) {
    SARSSL_REQUIRE(nb > 0 && F > 0 && T > 0 && a0 && W1 && scale && shift && (dtype == SARSSL_BF16 || dtype == SARSSL_F16), "sarssl_conv3x3_fwd_c1");
    if (stats && SARSSL_ZERO(stats, 128 * sizeof(double), (hipStream_t)stream) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    ConvArgs a = {};
#ifdef CONV_STAMPS
    a.stamps = g_conv_stamps_host;
#endif
    a.prio = (sarssl_mfma_prio() == 1 || sarssl_mfma_prio() == 2);
    a.clk = conv_clk() ? conv_clk() + 4 * 3 : nullptr;
    a.stats = stats;
    a.in = a0; a.w = w; a.out = out; a.scale = scale; a.shift = shift; a.prologue = 1; a.c1_w = W1;
    a.nb = nb; a.F = F; a.T = T;
    const int npairs = (nb * ((F + TR - 1) / TR) * ((T + PTC - 1) / PTC) + 1) / 2;
    if (dtype == SARSSL_F16) {
        if (conv_ws(2)) conv3x3_fwd_ws_kernel<false, true, false, f16><<<conv_persistent_grid(npairs, 0), 768, 0, (hipStream_t)stream>>>(a);
        else conv3x3_fwd_pp_kernel<false, true, false, f16><<<conv_persistent_grid(npairs, 0), 512, 0, (hipStream_t)stream>>>(a);
    } else if (conv_ws(2)) conv3x3_fwd_ws_kernel<false, true><<<conv_persistent_grid(npairs, 0), 768, 0, (hipStream_t)stream>>>(a);
    else conv3x3_fwd_pp_kernel<false, true><<<conv_persistent_grid(npairs, 0), 512, 0, (hipStream_t)stream>>>(a);
    SARSSL_CHECK_LAUNCH("conv3x3_fwd_pp_kernel<c1in>");
    return 0;
}

// Data gradient of the first 3x3 convolution (w = its flipped / transposed taps, dy = the gradient w.r.t. its output) whose result is
// consumed in the epilogue instead of being stored: red (f64[644], the layout of sarssl_stem_c1_bwd, zeroed here) receives
// G[co][c] = sum_p g[p][co] a0[p][c] at [co*4 + c] and s1[co] = sum_p g[p][co] at [512 + co], g = dz1 * relu'(scale * (W1 a0) + shift);
// the remaining entries follow from the input's moments (sarssl_stem_c1_bwd_finalize_mom).  bf16 (ping-pong kernel).
// a0_dtype: encoding of the saved input a0 (SARSSL_BF16 or SARSSL_F16); dy / w are bf16.
extern "C" int sarssl_conv3x3_dgrad_c1red(const void* dy, const void* w, const void* a0, const float* W1, const float* scale,
                                          const float* shift, int nb, int F, int T, double* red, int a0_dtype, void* stream) {
    SARSSL_REQUIRE(nb > 0 && F > 0 && T > 0 && dy && w && a0 && W1 && scale && shift && red && (a0_dtype == SARSSL_BF16 || a0_dtype == SARSSL_F16),
                   "sarssl_conv3x3_dgrad_c1red");
    if (SARSSL_ZERO(red, 644 * sizeof(double), (hipStream_t)stream) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    ConvArgs a = {};
#ifdef CONV_STAMPS
    a.stamps = g_conv_stamps_host;
#endif
    a.prio = (sarssl_mfma_prio() == 1 || sarssl_mfma_prio() == 2);
    a.clk = conv_clk() ? conv_clk() + 4 * 4 : nullptr;
    a.in = dy; a.w = w; a.out = nullptr; a.scale = scale; a.shift = shift; a.prologue = 0; a.c1_w = W1; a.c1_a0 = a0; a.c1_red = red;
    a.nb = nb; a.F = F; a.T = T;
    const int npairs = (nb * ((F + TR - 1) / TR) * ((T + PTC - 1) / PTC) + 1) / 2;
    if (a0_dtype == SARSSL_F16) {
        if (conv_ws(4)) conv3x3_fwd_ws_kernel<false, false, true, bf16, f16><<<conv_persistent_grid(npairs, 1), 768, 0, (hipStream_t)stream>>>(a);
        else conv3x3_fwd_pp_kernel<false, false, true, bf16, f16><<<conv_persistent_grid(npairs, 1), 512, 0, (hipStream_t)stream>>>(a);
    } else if (conv_ws(4)) conv3x3_fwd_ws_kernel<false, false, true><<<conv_persistent_grid(npairs, 1), 768, 0, (hipStream_t)stream>>>(a);
    else conv3x3_fwd_pp_kernel<false, false, true><<<conv_persistent_grid(npairs, 1), 512, 0, (hipStream_t)stream>>>(a);
    SARSSL_CHECK_LAUNCH("conv3x3_fwd_pp_kernel<c1red>");
    return 0;
}

static int conv3x3_launch(const void* in, const void* w, void* out, int dtype, int w_dtype, int nb, int F, int T,
                          const float* scale, const float* shift, int precise, float* ws, double* stats, const void* bn_y,
                          const float* bn_aff, void* stream, int y_dtype) {
    SARSSL_REQUIRE(nb > 0 && F > 0 && T > 0, "sarssl_conv3x3_fwd");
    SARSSL_REQUIRE(stats == nullptr || dtype == SARSSL_BF16 || dtype == SARSSL_F16, "sarssl_conv3x3_fwd(fused statistics: 16-bit storage only)");
    if (stats && SARSSL_ZERO(stats, 128 * sizeof(double), (hipStream_t)stream) != hipSuccess) { sarssl_set_error("memset"); return -2; }
    ConvArgs a = {};
#ifdef CONV_STAMPS
    a.stamps = g_conv_stamps_host;
#endif
    a.prio = (sarssl_mfma_prio() == 1 || sarssl_mfma_prio() == 2);
    a.clk = conv_clk() ? conv_clk() + 4 * (bn_y ? 2 : (scale != nullptr ? 0 : 1)) : nullptr;
    a.bn_y = bn_y; a.bn_aff = bn_aff;
    a.stats = stats;
    a.in = in; a.w = w; a.out = out; a.acc_ws = nullptr; a.acc_in = 0; a.acc_out = 0;
    a.scale = scale; a.shift = shift; a.prologue = (scale != nullptr);
    a.nb = nb; a.F = F; a.T = T; a.part_in = 0; a.part_w = 0;
    hipStream_t st = (hipStream_t)stream;
    const int grid = conv_grid(nb, F, T);
    if (dtype == SARSSL_BF16 && w_dtype == SARSSL_BF16) {
        const int npairs = (nb * ((F + TR - 1) / TR) * ((T + PTC - 1) / PTC) + 1) / 2;
        const int g = conv_persistent_grid(npairs, scale != nullptr ? 0 : 1);      // (no prologue = a data-gradient launch)
        if (bn_y && y_dtype == SARSSL_F16) {
            if (conv_ws(8)) conv3x3_fwd_ws_kernel<true, false, false, bf16, f16><<<g, 768, 0, st>>>(a);
            else conv3x3_fwd_pp_kernel<true, false, false, bf16, f16><<<g, 512, 0, st>>>(a);
        } else if (bn_y && conv_ws(8)) conv3x3_fwd_ws_kernel<true><<<g, 768, 0, st>>>(a);
        else if (!bn_y && conv_ws(1)) conv3x3_fwd_ws_kernel<false><<<g, 768, 0, st>>>(a);
        else if (bn_y) conv3x3_fwd_pp_kernel<true><<<g, 512, 0, st>>>(a);
        else conv3x3_fwd_pp_kernel<false><<<g, 512, 0, st>>>(a);
    } else if (dtype == SARSSL_F16 && w_dtype == SARSSL_F16) {              // forward launches of the fp16-forward mode
        SARSSL_REQUIRE(bn_y == nullptr, "sarssl_conv3x3_fwd(fp16: forward launches only)");
        const int npairs = (nb * ((F + TR - 1) / TR) * ((T + PTC - 1) / PTC) + 1) / 2;
        const int g = conv_persistent_grid(npairs, scale != nullptr ? 0 : 1);
        if (conv_ws(1)) conv3x3_fwd_ws_kernel<false, false, false, f16><<<g, 768, 0, st>>>(a);
        else conv3x3_fwd_pp_kernel<false, false, false, f16><<<g, 512, 0, st>>>(a);
    } else if (dtype == SARSSL_F32 && w_dtype == SARSSL_F32) {
        if (!precise) conv3x3_fwd_kernel<float, float><<<grid, 512, 0, st>>>(a);
        else {
            SARSSL_REQUIRE(ws != nullptr, "sarssl_conv3x3_fwd(precise needs workspace)");
            a.acc_ws = ws;
            ConvArgs p = a;
            p.part_in = 0; p.part_w = 1; p.acc_in = 0; p.acc_out = 1; conv3x3_fwd_kernel<float, float><<<grid, 512, 0, st>>>(p);
            p.part_in = 1; p.part_w = 0; p.acc_in = 1; p.acc_out = 1; conv3x3_fwd_kernel<float, float><<<grid, 512, 0, st>>>(p);
            p.part_in = 0; p.part_w = 0; p.acc_in = 1; p.acc_out = 0; conv3x3_fwd_kernel<float, float><<<grid, 512, 0, st>>>(p);
        }
    } else { sarssl_set_error("sarssl_conv3x3_fwd: unsupported dtypes (%d,%d)", dtype, w_dtype); return -1; }
    SARSSL_CHECK_LAUNCH("conv3x3_fwd_kernel");
    return 0;
}

extern "C" long sarssl_conv3x3_wgrad_workspace_bytes(int nb, int F, int T) {
    const int g1 = conv_grid(nb, F, T), g2 = wgrad_db_grid(nb, F, T);          // (either weight-gradient kernel may run)
    return (long)(g1 > g2 ? g1 : g2) * W_ELEMS * sizeof(float);
}

// dW: f32 [9][64][64] ([tap][co][ci]).  partial: workspace of sarssl_conv3x3_wgrad_workspace_bytes.
// Re-laid-out taps of a (64, 64, 3, 3) f32 convolution weight in ONE launch (a permute + flip + 2 casts = 5 torch launches per
// convolution otherwise, redone every step because the weights move): fwd [9][co][ci] and dgr [9][ci][co] with flipped taps
// (= W.flip(2,3).permute(2,3,1,0)), as f32 (dtype 0) or bf16 (dtype 1).
template <typename T, typename TD>
__global__ void conv_taps_kernel(const float* __restrict__ W, T* __restrict__ fwd, TD* __restrict__ dgr) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 9 * 4096) return;
    const int tap = e >> 12, a = (e >> 6) & 63, b = e & 63;
    st_f(fwd + e, W[(a * 64 + b) * 9 + tap]);                    // fwd[tap][co = a][ci = b]
    st_f(dgr + e, W[(b * 64 + a) * 9 + (8 - tap)]);              // dgr[tap][ci = a][co = b] = W[co][ci][2 - kh][2 - kw]
}
extern "C" int sarssl_conv_taps(const float* W, void* fwd, void* dgr, int dtype, void* stream) {
    if (dtype == SARSSL_BF16) conv_taps_kernel<bf16, bf16><<<144, 256, 0, (hipStream_t)stream>>>(W, (bf16*)fwd, (bf16*)dgr);
    else if (dtype == SARSSL_MIX16) conv_taps_kernel<f16, bf16><<<144, 256, 0, (hipStream_t)stream>>>(W, (f16*)fwd, (bf16*)dgr);     // fp16 forward taps, bf16 gradient taps
    else if (dtype == SARSSL_F32) conv_taps_kernel<float, float><<<144, 256, 0, (hipStream_t)stream>>>(W, (float*)fwd, (float*)dgr);
    else { sarssl_set_error("sarssl_conv_taps: unsupported dtype %d", dtype); return -1; }
    SARSSL_CHECK_LAUNCH("conv_taps_kernel");
    return 0;
}
// (d, 4, F, 1) f32 frame-patch convolution weight -> [d][f * 4 + c] (the GEMM operand over the (B, T, F, 4) activations)
template <typename T>
__global__ void patch_w_kernel(const float* __restrict__ W, T* __restrict__ out, int d, int F) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)d * F * 4) return;
    const int c = e & 3; const long of = e >> 2; const int f = of % F; const long o = of / F;
    st_f(out + e, W[(o * 4 + c) * F + f]);
}
extern "C" int sarssl_patch_w(const float* W, void* out, int d, int F, int dtype, void* stream) {
    const long n = (long)d * F * 4;
    if (dtype == SARSSL_BF16) patch_w_kernel<bf16><<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, (bf16*)out, d, F);
    else if (dtype == SARSSL_F32) patch_w_kernel<float><<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, (float*)out, d, F);
    else if (dtype == SARSSL_F16) patch_w_kernel<f16><<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, (f16*)out, d, F);
    else { sarssl_set_error("sarssl_patch_w: unsupported dtype %d", dtype); return -1; }
    SARSSL_CHECK_LAUNCH("patch_w_kernel");
    return 0;
}
// grad (d, 4, F, 1) f32 += sum_s g[s][d][f * 4 + c]  (the frame-patch weight gradient back in nn.Conv2d layout; g = nslice split-K
// partial products, so the fold of the partials and the re-layout are one pass and nothing has to be zeroed)
__global__ void patch_wgrad_accum_kernel(const float* __restrict__ g, int nslice, float* __restrict__ grad, int d, int F) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)d * F * 4;
    if (e >= n) return;
    float s = 0.f;
    for (int k = 0; k < nslice; ++k) s += g[k * n + e];
    const int c = e & 3; const long of = e >> 2; const int f = of % F; const long o = of / F;
    grad[(o * 4 + c) * F + f] += s;
}
extern "C" int sarssl_patch_wgrad_accum(const float* g, int nslice, float* grad, int d, int F, void* stream) {
    SARSSL_REQUIRE(nslice > 0, "sarssl_patch_wgrad_accum");
    const long n = (long)d * F * 4;
    patch_wgrad_accum_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(g, nslice, grad, d, F);
    SARSSL_CHECK_LAUNCH("patch_wgrad_accum_kernel");
    return 0;
}

// bf16 weight gradient: the role-split kernel (16 waves: 8 stage, 8 run the MFMA loop), or with SARSSL_WGRAD_WS=0 (A/B) the lockstep
// double-buffered one.  Same box, three interleaved rounds, both weight-gradient launches of a stem: 10.87 -> 10.72 ms per step; alone
// 361 -> 311 us (its MFMA loop alone: 254 us).
template <bool C1IN>
static void wgrad_bf16_launch(const WgradArgs& a, int g2, hipStream_t st, bool z_f16) {
    static const bool ws = []() { const char* e = getenv("SARSSL_WGRAD_WS"); return !(e && atoi(e) == 0); }();
    if (z_f16) {
        if (ws) conv3x3_wgrad_ws_kernel<C1IN, f16><<<g2, 1024, 0, st>>>(a);
        else conv3x3_wgrad_db_kernel<C1IN, f16><<<g2, 512, 0, st>>>(a);
    } else if (ws) conv3x3_wgrad_ws_kernel<C1IN><<<g2, 1024, 0, st>>>(a);
    else conv3x3_wgrad_db_kernel<C1IN><<<g2, 512, 0, st>>>(a);
}

extern "C" int sarssl_conv3x3_wgrad(const void* dy, const void* zin, int dtype, int nb, int F, int T,
                                    const float* scale, const float* shift, float* dW, float* partial, int precise,
                                    void* stream) {
    SARSSL_REQUIRE(nb > 0 && F > 0 && T > 0, "sarssl_conv3x3_wgrad");
    WgradArgs a = {};
    a.dy = dy; a.zin = zin; a.scale = scale; a.shift = shift; a.prologue = (scale != nullptr);
    a.partial = partial; a.nb = nb; a.F = F; a.T = T; a.part_dy = 0; a.part_z = 0;
    hipStream_t st = (hipStream_t)stream;
    const int grid = conv_grid(nb, F, T);
    const int rblocks = W_ELEMS / 64;
    if (dtype == SARSSL_BF16 || dtype == SARSSL_MIX16) {        // MIX16: dy bf16, zin fp16 (saved by the fp16 forward pass)
        const int g2 = wgrad_db_grid(nb, F, T);
        wgrad_bf16_launch<false>(a, g2, st, dtype == SARSSL_MIX16);
        wgrad_reduce_kernel<<<rblocks, 256, 0, st>>>(partial, g2, dW, 0);
    } else if (dtype == SARSSL_F32) {
        const int npass = precise ? 3 : 1;
        for (int pass = 0; pass < npass; ++pass) {
            WgradArgs p = a;
            if (precise) { p.part_dy = (pass == 1); p.part_z = (pass == 0); }     // hi*lo, lo*hi, hi*hi
            conv3x3_wgrad_kernel<float><<<grid, 512, 0, st>>>(p);
            wgrad_reduce_kernel<<<rblocks, 256, 0, st>>>(partial, grid, dW, pass > 0);
        }
    } else { sarssl_set_error("sarssl_conv3x3_wgrad: unsupported dtype %d", dtype); return -1; }
    SARSSL_CHECK_LAUNCH("conv3x3_wgrad_kernel");
    return 0;
}

// bf16 weight gradient ADDED straight into the f32 (64, 64, 3, 3) parameter-gradient buffer (nn.Conv2d layout): no [9][64][64]
// intermediate, no permute-add pass.
// z_dtype: encoding of zin (SARSSL_BF16 or SARSSL_F16); dy is bf16.
extern "C" int sarssl_conv3x3_wgrad_acc(const void* dy, const void* zin, int nb, int F, int T, const float* scale, const float* shift,
                                        float* grad_oihw, float* partial, int z_dtype, void* stream) {
    SARSSL_REQUIRE(nb > 0 && F > 0 && T > 0 && grad_oihw && partial && (z_dtype == SARSSL_BF16 || z_dtype == SARSSL_F16), "sarssl_conv3x3_wgrad_acc");
    WgradArgs a = {};
    a.dy = dy; a.zin = zin; a.scale = scale; a.shift = shift; a.prologue = (scale != nullptr);
    a.partial = partial; a.nb = nb; a.F = F; a.T = T; a.part_dy = 0; a.part_z = 0;
    hipStream_t st = (hipStream_t)stream;
    const int g2 = wgrad_db_grid(nb, F, T);
    wgrad_bf16_launch<false>(a, g2, st, z_dtype == SARSSL_F16);
    wgrad_reduce_kernel<<<W_ELEMS / 64, 256, 0, st>>>(partial, g2, nullptr, 0, grad_oihw);
    SARSSL_CHECK_LAUNCH("conv3x3_wgrad_kernel(acc)");
    return 0;
}
// The same with the input operand relu(bn1(W1 a0)) formed from the stem's 4-channel input a0 (B,F,T,4) bf16 while staging (W1 f32[64][4],
// scale / shift = bn1's affine): the first layer's 64-channel output is not read.  Double-buffered bf16 kernel.
extern "C" int sarssl_conv3x3_wgrad_c1_acc(const void* dy, const void* a0, const float* W1, int nb, int F, int T, const float* scale,
                                           const float* shift, float* grad_oihw, float* partial, int a0_dtype, void* stream) {
    SARSSL_REQUIRE(nb > 0 && F > 0 && T > 0 && grad_oihw && partial && W1 && scale && shift && (a0_dtype == SARSSL_BF16 || a0_dtype == SARSSL_F16),
                   "sarssl_conv3x3_wgrad_c1_acc");
    WgradArgs a = {};
    a.dy = dy; a.zin = a0; a.scale = scale; a.shift = shift; a.prologue = 1; a.c1_w = W1;
    a.partial = partial; a.nb = nb; a.F = F; a.T = T;
    hipStream_t st = (hipStream_t)stream;
    const int g2 = wgrad_db_grid(nb, F, T);
    wgrad_bf16_launch<true>(a, g2, st, a0_dtype == SARSSL_F16);
    wgrad_reduce_kernel<<<W_ELEMS / 64, 256, 0, st>>>(partial, g2, nullptr, 0, grad_oihw);
    SARSSL_CHECK_LAUNCH("conv3x3_wgrad_db_kernel<c1in>");
    return 0;
}
