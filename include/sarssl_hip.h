/* C ABI of libsarssl_hip.so - the MI355X (gfx950) kernels behind SAR-SSL's pretraining hot path.
 *
 * The reference (Audio-WestlakeU/SAR-SSL) is pure Python on PyTorch ATen ops: it has no FFI of its own, so each entry
 * point below names the reference code whose device work it replaces (paths relative to the reference repository).
 * Conventions: plain pointers and sizes only (no torch types); every pointer is DEVICE memory owned by the caller,
 * including workspaces; kernels are enqueued on `stream` (a hipStream_t passed as void*) and never synchronise;
 * return 0 on success, negative on error with a message in sarssl_last_error().
 * dtype: 0 = f32, 1 = bf16, 2 = int16, 3 = fp16, 4 = "mixed 16": a backward kernel whose GRADIENT tensors (in / out) are bf16 while the
 * tensors SAVED BY THE FORWARD PASS it reads are fp16 - the fp16-forward / bf16-backward numeric mode (abi version 2: fp16 has 3 more
 * mantissa bits than bf16 at the same MFMA rate, which is what puts the forward pass inside the reference's 1e-3 per-bin tolerance;
 * gradients keep bf16's exponent range, so no loss scaling - the reference's own AMP is fp16 with a GradScaler, code/learner.py:46-50).
 * 5 = "mixed f32" (hybrid mode): bf16 gradient tensors next to an f32 tensor saved by the forward pass.
 * "precise" != 0 (f32 storage only) runs every MFMA contraction as three split-bf16 passes (hi*hi + hi*lo + lo*hi).
 */
#ifndef SARSSL_HIP_H
#define SARSSL_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* sarssl_last_error(void);
int sarssl_abi_version(void);

/* ---- contexts ("no global mutable state except an opaque sarssl_ctx* created per device").  A context owns everything a
 *      caller can configure - the workgroup count of the 3x3 gradient launches, the clock-probe buffer, the attached step state (dropout
 *      salt of a captured step), the range of the host-zeroed accumulator arena.  Kernels are launched under the context that is
 *      CURRENT ON THE CALLING THREAD (sarssl_make_current; none current = library defaults), so two contexts on one device - two models,
 *      a training thread and a validation thread - never see each other's settings.  The library keeps no other mutable state (the
 *      error string of sarssl_last_error is per thread). */
typedef struct sarssl_ctx sarssl_ctx;
sarssl_ctx* sarssl_create(int device);              /* NULL on error */
int sarssl_destroy(sarssl_ctx* ctx);
int sarssl_make_current(sarssl_ctx* ctx);           /* ctx or NULL becomes the calling thread's current context */
int sarssl_ctx_device(const sarssl_ctx* ctx);
int sarssl_ctx_set_conv_cus(sarssl_ctx* ctx, int ncus);       /* 3x3 data / weight gradient launches: workgroups; 0 = 7/8 of the CUs */
/* fp16 range guard: kernels that encode externally scaled data as fp16 (sarssl_mask_inputs, sarssl_stft_frontend_pairs_masked with
 * dtype fp16: the spectrum divided by mean|X_0| + eps, code/learner.py:539-542) set a device word of the context when a value does not fit
 * (|v| > 65 504).  The next sarssl_masked_mse_* launch under the context then reports the loss as NaN and clears the word - an overflowed
 * forward does not reach the loss by itself (BatchNorm makes NaN of inf, ReLU makes 0 of NaN) - and the guarded Adam launches skip the
 * step.  sarssl_ctx_fp16_overflow reads (and optionally clears) the word from the host; synchronises `stream`. */
int sarssl_ctx_fp16_overflow(sarssl_ctx* ctx, int clear, void* stream);
int sarssl_ctx_get_conv_cus(const sarssl_ctx* ctx);
int sarssl_ctx_set_clock_probe(sarssl_ctx* ctx, void* buf);   /* see the measurement aid below; NULL = off */
int sarssl_ctx_attach_step_state(sarssl_ctx* ctx, void* state);   /* device SarsslStepState* or NULL: dropout launches add its salt */
int sarssl_ctx_zero_arena(sarssl_ctx* ctx, const void* base, long bytes);   /* accumulators inside [base, base+bytes) are already zero */
int sarssl_device_info(int device, char* name_out, int name_len, int* cu_count, long* lds_bytes);

/* ---- front-end: code/common/utils_module.py:49-72 (STFT.forward), code/learner.py:525-553 (data_preprocess),
 *      code/common/utils_module.py:128-134 (AddChToBatch 'M').  sig: (B, nsample, nch) f32|int16.
 *      U: workspace (B, nch, 257, nt, 2) f32; magsum: workspace f64[B]; out: (B*(nch-1), 2, 256, nt, 2) f32. */
int sarssl_stft_frontend(const void* sig, int sig_dtype, int nb, long nsample, int nch, int win_len, int hop, int nfft,
                         int nt, float eps, float* U, double* magsum, float* out, void* stream);
/*      pair_mode 0 = AddChToBatch 'M' (as above), 1 = 'MM' (code/common/utils_module.py:136-143): all nch(nch-1)/2 pairs,
 *      out: (B*npair, 2, 256, nt, 2). */
int sarssl_stft_frontend_pairs(const void* sig, int sig_dtype, int nb, long nsample, int nch, int win_len, int hop, int nfft,
                               int nt, float eps, int pair_mode, float* U, double* magsum, float* out, void* stream);
/* the same followed by sarssl_mask_inputs(mode 0) in ONE pass over the spectrum (code/learner.py:533-551 + code/model.py:541, :563):
 * mp (B*npair, nt) u8 frame mask, mch (B*npair) i32 masked channel; spec / spat (B*npair, 256, nt, 4) of `dtype` */
int sarssl_stft_frontend_pairs_masked(const void* sig, int sig_dtype, int nb, long nsample, int nch, int win_len, int hop, int nfft,
                                      int nt, float eps, int pair_mode, float* U, double* magsum, float* out, const unsigned char* mp,
                                      const int* mch, void* spec, void* spat, int dtype, void* stream);
/*      out: complex64 (B, 257, nt, nch) interleaved, the STFT.forward return value. */
int sarssl_stft_raw(const void* sig, int sig_dtype, int nb, long nsample, int nch, int win_len, int hop, int nfft, int nt,
                    float* U, double* magsum, float* out, void* stream);
/* ---- on-disk segments: `{idx}.wav` PCM-16 files read per item by soundfile.read in FixMicSigDataset.__getitem__
 *      (code/dataset.py:147-151).  Host code only (no GPU work): fills out[n][nsample][nch] int16 (e.g. a pinned batch
 *      buffer) with samples [offset, offset+nsample) of each file using nthreads positional readers. */
int sarssl_wav_probe(const char* path, int* nch, int* fs, long* nsample);
int sarssl_wav_read_batch(const char* const* paths, int n, long nsample, int nch, int fs, long offset, short* out, int nthreads);
/* ---- host-side mask RNG: PatchMask.forward / gen_mask_idx (code/common/utils_module.py:255-273, 305-308) with Python's own
 *      MT19937 state (random.getstate()[1]: 624 words + position) - bit-identical index / channel stream, no interpreter loop.
 *      idx: int64[nbatch][nmasked], ch: int64[nbatch]. */
int sarssl_mask_sample(unsigned int* mt, int* pos, int nbatch, int npatch, int nmasked, int nmic, long* idx, long* ch);
/* ---- inverse STFT: code/common/utils_module.py:74-113 (ISTFT.forward = torch.istft, rectangular window, center = inv).
 *      spec: complex64 (B, 257, nt, nch) interleaved; sig: (B, nsample, nch) f32, nsample = (nt+1)*hop (center 0) or
 *      (nt-1)*hop (center 1); frames_ws: sarssl_istft_workspace_bytes(nb, nch, nt) bytes. */
long sarssl_istft_workspace_bytes(int nb, int nch, int nt);
int sarssl_istft(const float* spec, int nb, int nch, int nt, int win_len, int hop, int nfft, int center, float* frames_ws,
                 float* sig, void* stream);

/* ---- generic batched MFMA GEMM with fused epilogue: nn.Linear / Conv1d(k=1) / patch conv / attention bmm
 *      (code/common/conformer/modules.py:35-48, feed_forward.py:47-54, attention.py:82-103, convolution.py:138,143,
 *      code/model.py:63, 296-301).  See csrc/gemm.hip for the layout flags.  split_k > 0: C(f32) += alpha*A*B.
 *      c_row_shift != 0 (M == N == ldc): row m of every batch matrix is stored m + 1 - N elements further, i.e. the product is
 *      written directly in the layout of RelativeMultiHeadAttention._relative_shift (attention.py:105-113). */
int sarssl_gemm(const void* A, const void* B, void* C, int dtA, int dtB, int dtC, int a_kc, int b_kc, int M, int N, int K,
                long lda, long ldb, long ldc, int nbatch, int batch_inner, long sA0, long sA1, long sB0, long sB1, long sC0,
                long sC1, float alpha, float out_scale, const float* bias, int act, const void* resid, long ldr, long sR0,
                long sR1, float res_scale, void* preact, const void* aux, int aux_act, int aux_dtype, float p_drop,
                unsigned long long seed, int precise, float* ws, int split_k, int c_row_shift, void* stream);
/*      operand dtypes: (bf16,bf16) and (fp16,fp16) contract on the matrix cores as stored, (f32,f32) as split-bf16 parts; a bf16 / fp16
 *      pair (gradient x saved activation) contracts in bf16, the fp16 operand re-encoded while staged.  aux_dtype: dtype of `aux`
 *      (= dtC, or fp16 next to a bf16 C). */

/* ---- downstream heads (round 6): SARSSL.forward's downstream branch (code/model.py:667-719: mean over the frames, then
 *      nn.Sequential(LayerNorm, Linear) or (LayerNorm, Linear, ReLU, Linear)) and SARSSL_MultiCH.head_mch (code/model.py:793-821).  f32 tensors
 *      of a few hundred rows at most; LayerNorm = sarssl_layernorm_fwd / _bwd.  sarssl_mean_rows: x (B, Tn, d) of `dtype` -> out f32 (B, d);
 *      _bwd: dx (B, Tn, d) of `dtype` (f32 | bf16) = dy / Tn.  sarssl_small_linear_fwd: y [M][N] = act(x [M][K] W[N][K]^T + bias), act 0 | 1 (relu),
 *      any N >= 1.  _bwd: dx (may be NULL) = dz W, dW += dz^T x, db (may be NULL) += column sums of dz, dz = dy (act 0) or dy * [y > 0] (act 1,
 *      written to dz_ws [M][N]). */
int sarssl_mean_rows(const void* x, int B, int Tn, int d, float* out, int dtype, void* stream);
int sarssl_mean_rows_bwd(const float* dy, int B, int Tn, int d, void* dx, int dtype, void* stream);
int sarssl_small_linear_fwd(const float* x, const float* W, const float* bias, int M, int N, int K, int act, float* y, void* stream);
int sarssl_small_linear_bwd(const float* dy, const float* y, const float* x, const float* W, int M, int N, int K, int act, float* dz_ws, float* dx,
                            float* dW, float* db, void* stream);

/* ---- "hybrid" numeric mode (round 6): fp16 CNN stem + f32 residual stream in the Conformer blocks / decoder - the mode that meets the
 *      1e-3 per-bin tolerance against the reference's f32 path (code/learner.py:100-103 runs the model in f32 by default) at 16-bit
 *      matrix-core speed.  An f32 tensor enters a product as an fp16 PAIR hi = fp16(x), lo = fp16(x - hi), a weight likewise.
 *      sarssl_gemm_split: nn.Linear forward (code/common/conformer/modules.py:35-48 and every use of it: feed_forward.py:47-54,
 *      attention.py:82-85, :113, convolution.py:138, :143, code/model.py:63, 296-301): C = epilogue(A B^T [+ A_lo B^T] [+ A B_lo^T]), f32
 *      accumulation across the segments, epilogue of sarssl_gemm (bias, act 1 relu | 2 swish, saved pre-activation, dropout, scaled
 *      residual).  A, A_lo [M][K] fp16 (row stride lda), B, B_lo [N][K] fp16 (row stride ldb); A_lo needs B_lo; C / resid / preact of
 *      dtC (fp16 | f32). */
int sarssl_gemm_split(const void* A, const void* A_lo, const void* B, const void* B_lo, void* C, int dtC, int M, int N, int K, long lda,
                      long ldb, long ldc, float out_scale, const float* bias, int act, const void* resid, long ldr, float res_scale,
                      void* preact, float p_drop, unsigned long long seed, void* stream);
/*      LayerNorm on f32 rows with the result written as an fp16 pair (and, y32 != NULL, in f32): the operand of the Linear layer behind it.
 *      y_lo (z_lo below) may be NULL: the hi half only. */
int sarssl_layernorm_fwd_pair(const float* x, long ldx, long M, int d, const float* gamma, const float* beta, float eps, void* y_hi,
                              void* y_lo, long ldy, float* y32, long ldy32, float* mean, float* rstd, void* stream);
/*      y = LN_a(x) in f32 (a Conformer block's closing LayerNorm, code/common/Conformer.py:88-90), z = LN_b(y) as a pair (the next
 *      block's feed_forward.py:48) in one launch. */
int sarssl_layernorm_fwd2_pair(const float* x, long ldx, long M, int d, const float* gamma_a, const float* beta_a, float eps_a, float* y,
                               long ldy, float* mean_a, float* rstd_a, const float* gamma_b, const float* beta_b, float eps_b, void* z_hi,
                               void* z_lo, long ldz, float* mean_b, float* rstd_b, void* stream);
/*      LayerNorm backward on the f32 stream: x / resid / dx f32, dy (the branch gradient) bf16 | f32 (dy_dtype); dx2 (optional, [M][d]
 *      bf16) = dx * dropmask(p_drop, seed) * gscale, the matrix-core operand of the next module of the backward chain.  partial /
 *      dgamma / dbeta as sarssl_layernorm_bwd. */
int sarssl_layernorm_bwd_stream(const void* dy, int dy_dtype, long lddy, const float* x, long ldx, long M, int d, const float* gamma,
                                const float* mean, const float* rstd, const float* resid, long ldr, float* dx, long lddx, float* dgamma,
                                float* dbeta, float* partial, void* dx2, float p_drop, unsigned long long seed, float gscale, void* stream);
/*      fused feed-forward module on the f32 stream (d = 256; csrc/ffn2h.hip): FeedForwardModule.forward (feed_forward.py:47-54) under
 *      the half-step residual (Conformer.py:60-67) with its LayerNorm in the prologue - y = x + out_scale * drop2((W2h + W2l) drop1(
 *      swish((W1h + W1l) LN(x) + b1)) + b2), LN(x) held on the CU as an fp16 pair; written for the backward pass: ln_hi [M][d] fp16,
 *      ln_mean / ln_rstd [M], preact / hidden [M][4d] fp16.  w1h / w1l / w2h / w2l: sarssl_ffn_pack of the weights' hi / lo shadows. */
int sarssl_ffn2h_supported(long M, int d);
int sarssl_ffn2h_fwd(const float* x, long ldx, const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_hi, float* ln_mean,
                     float* ln_rstd, const void* w1h, const void* w1l, const void* w2h, const void* w2l, const float* b1, const float* b2,
                     void* preact, void* hidden, float* y, long ldy, long M, int d, float p1, unsigned long long s1, float p2,
                     unsigned long long s2, float out_scale, int act_pair, void* stream);
/*      act_pair != 0: LN(x) enters the first product as a pair (hi hi + lo hi + hi lo), else as its hi half (hi hi + hi lo) - its output is
 *      an fp16 tensor either way, whose own rounding is of the size of what the lo half adds. */
/*      the stem's 64 -> 4 convolution with its f32 result as a pair (y4_hi = what sarssl_stem_c4_fwd stores, y4_lo the remainder;
 *      stats8 = BatchNorm(4) sums of the pair's value, may be NULL), and BatchNorm(4) affine + ReLU on the pair -> pair: the f32 operand
 *      of the frame-patch product (code/model.py:60-63). */
int sarssl_stem_c4_fwd_pair(const void* y3, const float* W4, const float* scale, const float* shift, int nb, int F, int Tn, void* y4_hi,
                            void* y4_lo, double* stats8, int dtype, void* stream);
int sarssl_cl_affine_act_pair(const void* x_hi, const void* x_lo, long n, int C, const float* scale, const float* shift, int act, void* z_hi,
                              void* z_lo, void* stream);
/*      the guarded Adam launches with a third shadow: pl16 = fp16(p - fp16(p)), the weights' lo halves, rewritten in the same pass */
int sarssl_adam_step_guard_lo(float* p, const float* g, float* m, float* v, void* p16, void* ph16, void* pl16, long n, float gscale, float lr,
                              float beta1, float beta2, float eps, int step, const float* guard, int* nskipped, void* stream);
int sarssl_adam_step_dev_guard_lo(float* p, float* g, float* m, float* v, void* p16, void* ph16, void* pl16, long n, float gscale, void* state,
                                  float eps, int zero_grad, const float* guard, void* stream);
/*      src (f32 | fp16 | bf16; n % 4 == 0) -> hi = fp16(src) (may be NULL), lo = fp16(src - hi): the weights' lo shadow, stem outputs. */
int sarssl_split_pair(const void* src, int src_dtype, long n, void* hi, void* lo, void* stream);

/* several split-K products reduced in one launch: C_q[m][n] += sum_s ws_q[s][m][n].  Pair with sarssl_gemm(split_k > 0, C = NULL,
 * bf16 operands), which then only writes the partials (split count = ceil(K / (ceil(ceil(K / split_k) / 64) * 64))). */
int sarssl_splitk_reduce_multi(const float* const* ws, const int* nsplit, const int* M, const int* N, float* const* C, const long* ldc,
                               int n_prob, void* stream);
/* up to 12 independent weight-gradient products dY_q^T X_q (the nn.Linear weight gradients autograd computes one by one for the
 * layers of a Conformer block, code/common/conformer/*.py) as ONE launch: A_q = dY [K_q][M_q], B_q = X [K_q][N_q], bf16, row strides
 * lda / ldb; f32 partial sums of split_out[q] K-slices to ws[q] (size it for split_k[q] slices), folded by
 * sarssl_splitk_reduce_multi.  csum_ws (may be null, entries may be null): csum_ws[q] = f32 [split_k[q]][M_q] receives the per-slice
 * column sums of dY_q - the layer's bias gradient, folded by the same sarssl_splitk_reduce_multi launch (M = 1, N = M_q).
 * Returns 1 without launching when a shape is ragged (M, N % 128 or a K slice % 64). */
int sarssl_gemm_group_tn(const void* const* A, const void* const* B, float* const* ws, const int* M, const int* N, const int* K,
                         const long* lda, const long* ldb, const int* split_k, int* split_out, float* const* csum_ws, int n_prob,
                         int dtB, void* stream);       /* dtB: dtype of every X_q (bf16 | fp16); dY_q is bf16 */

/* ---- fused feed-forward module (round 5): FeedForwardModule.forward, code/common/conformer/feed_forward.py:47-54, under the half-step
 *      residual of ConformerBlock (code/common/Conformer.py:60-67) - Linear(d, 4d) + Swish + Dropout + Linear(4d, d) + Dropout as ONE
 *      launch per 64-row tile (csrc/ffn2.hip): the [64, 4d] hidden tile never travels through HBM as an operand.  Replaces the pair of
 *      sarssl_gemm launches (bias + Swish + dropout + saved pre-activation | bias + dropout + scaled residual); dropout masks are the same
 *      function of (seed, row * N + column).  M % 64 == 0, d in {256, 512} (sarssl_ffn2_supported), else use sarssl_gemm.
 *      Weights are read from PACKS in MFMA fragment order (sarssl_ffn_pack: block (n / 32, k / 16) = 64 lanes x 8 elements, lane l =
 *      row 32 nb + (l & 31), k = 16 ks + 8 (l >> 5) ..): src(n, k) = src[n * rs + k * cs], so rs / cs select a matrix or its transpose;
 *      up to 64 matrices per launch; 16-bit elements.
 *      forward:  y = resid + out_scale * drop(p2, s2)(W2 drop(p1, s1)(swish(W1 ln + b1)) + b2); preact, hidden: [M][4d] saved for backward.
 *                w1p = pack(W1 [4d x d]), w2p = pack(W2 [d x 4d]); dtype SARSSL_F16 | SARSSL_BF16 (every 16-bit tensor).
 *      backward: dh = (dz2 W2) * dropmask(p1, s1) * swish'(preact) [M][4d] (operand of both weight-gradient products), dln = dh W1 [M][d];
 *                w2tp = pack(W2^T [4d x d]), w1tp = pack(W1^T [d x 4d]); dtype SARSSL_BF16 | SARSSL_MIX16 (bf16 gradients, fp16 preact). */
int sarssl_ffn2_supported(long M, int d);
int sarssl_ffn_pack(const void* const* src, void* const* dst, const int* N, const int* K, const long* rs, const long* cs, int n_mat,
                    void* stream);
/*      LayerNorm in the same launch (feed_forward.py:48 and its backward; arithmetic of sarssl_layernorm_fwd / _bwd):
 *      forward,  x_ln != NULL: ln = LayerNorm(x_ln; gamma, beta, eps) is formed in the launch's prologue (bit-identical to
 *                sarssl_layernorm_fwd), written to ln_out [M][d] with ln_mean / ln_rstd [M]; the `ln` argument is ignored.
 *      backward, x_ln != NULL (the module's saved input, dtype of preact): the epilogue runs sarssl_layernorm_bwd(_drop) on the second
 *                product - `dln` receives dx = LayerNorm'(dh W1) + resid, dx2 (optional, [M][d]) = dx * dropmask(p2, s2) * gscale2,
 *                ln_partial [M / 64][2][d] the per-tile dgamma | dbeta sums (fold: sarssl_ln_param_reduce_multi, nparts = M / 64). */
int sarssl_ffn2_fwd(const void* ln, long ldln, const void* w1p, const void* w2p, const float* b1, const float* b2, void* preact,
                    void* hidden, void* y, long ldy, const void* resid, long ldr, long M, int d, float p1, unsigned long long s1,
                    float p2, unsigned long long s2, float out_scale, const void* x_ln, long ldx, const float* ln_gamma,
                    const float* ln_beta, float ln_eps, void* ln_out, float* ln_mean, float* ln_rstd, int dtype, void* stream);
int sarssl_ffn2_bwd(const void* dz2, long lddz, const void* w2tp, const void* w1tp, const void* preact, void* dh, void* dln, long lddln,
                    long M, int d, float p1, unsigned long long s1, const void* x_ln, long ldx, const float* ln_gamma,
                    const float* ln_mean, const float* ln_rstd, const void* resid, long ldr, void* dx2, float p2, unsigned long long s2,
                    float gscale2, float* ln_partial, int dtype, void* stream);

/* ---- row-tile-resident Linear layers of the d = 256 Conformer blocks (round 5, csrc/lin256.hip): nn.Linear / Conv1d(k = 1) forward and
 *      data gradient (code/common/conformer/attention.py:82-85, :113 q/k/v and output projections; convolution.py:138, :143 pointwise
 *      convolutions) as ONE launch per layer with the 64 x K input tile resident in LDS and the weights read from fragment-order packs
 *      (sarssl_ffn_pack) - optionally together with the LayerNorm in front of the layer (attention.py:146, convolution.py:137; forward:
 *      prologue, bit-identical to sarssl_layernorm_fwd; backward: sarssl_layernorm_bwd(_drop) in the epilogue).  Replaces sarssl_gemm (+ the
 *      LayerNorm launches) for these shapes: M % 64 == 0, N and K multiples of 256, K <= 768, K == 256 or N == 256.
 *      fwd: y = resid + out_scale * drop(p, seed)(a W^T + bias), wp = pack(W [N x K]); dtype SARSSL_F16 | SARSSL_BF16.
 *      bwd: dx = dy Wt^T, wtp = pack(W^T [N x K]) with N = the layer's input width; dtype SARSSL_BF16 | SARSSL_MIX16 (fp16 x_ln). */
int sarssl_lin256_supported(long M, int N, int K);
int sarssl_lin256_fwd(const void* a, long lda, const void* wp, const float* bias, void* y, long ldy, const void* resid, long ldr, long M,
                      int N, int K, float p, unsigned long long seed, float out_scale, const void* x_ln, long ldx, const float* ln_gamma,
                      const float* ln_beta, float ln_eps, void* ln_out, float* ln_mean, float* ln_rstd, int dtype, void* stream);
int sarssl_lin256_bwd(const void* dy, long lddy, const void* wtp, void* dx, long lddx, long M, int N, int K, const void* x_ln, long ldx,
                      const float* ln_gamma, const float* ln_mean, const float* ln_rstd, const void* resid, long ldr, void* dx2, float p2,
                      unsigned long long s2, float gscale2, float* ln_partial, int dtype, void* stream);

/* ---- OCP fp8 (e4m3fn) GEMM path (BASELINE.json config 5; no reference counterpart - the reference is fp32 / fp16-AMP,
 *      code/learner.py:46-50): per-tensor scales chosen on the device, block-scaled MFMA with unit block scales, same fused epilogue as
 *      sarssl_gemm.  sarssl_fp8_quantize: x [rows][cols] (f32 | bf16, row stride ld) -> q fp8 [rows][cols] or (transpose) [cols][rows],
 *      inv_scale = amax / 448 (device float), amax_ws = one device word.  sarssl_gemm_fp8: C = epilogue(alpha * sa * sb * A8 B8^T). */
int sarssl_fp8_quantize(const void* x, int dtype, long rows, long cols, long ld, void* q, long ldq, float* amax_ws, float* inv_scale,
                        int transpose, void* stream);
int sarssl_gemm_fp8(const void* A8, const void* B8, const float* sa, const float* sb, void* C, int dtC, int M, int N, int K, long lda,
                    long ldb, long ldc, float alpha, float out_scale, const float* bias, int act, const void* resid, long ldr,
                    float res_scale, void* preact, const void* aux, int aux_act, float p_drop, unsigned long long seed, void* stream);

/* ---- fused relative-position attention, bf16 (RelativeMultiHeadAttention.forward, attention.py:87-101, and its backward):
 *      softmax(((q+u) k^T + bias) * scale) -> dropout -> @ v per (batch, head) without materialising scores / probabilities.
 *      qu, k, v: bf16 [B*T][ld], head h at column h*dh; bias: bf16 (B,H,T,T) = shifted positional score (a sarssl_gemm with
 *      c_row_shift), element (i, i+1) ignored; ctx: bf16 [B*T][ldc]; ctx32: f32 [B*T][H*dh] unrounded copy for backward; lse: f32 (B,H,T).  sarssl_relpos_attn_supported: shapes the
 *      fused kernels take (T % 8 == 0, dh in {32, 64, 128}); otherwise use the GEMM + sarssl_softmax_relshift_fwd path.
 *      Backward: dsum = f32 (B,H,T) workspace; dbias bf16 (B,H,T,T) in the shifted layout (feed sarssl_relshift_bwd). */
int sarssl_relpos_attn_supported(int T, int dh);
int sarssl_relpos_attn_fwd(const void* qu, long ldq, const void* k, const void* v, long ldk, const void* bias, void* ctx, long ldc,
                           float* ctx32, float* lse, int B, int H, int T, int dh, float scale, float p_drop, unsigned long long seed,
                           int dtype, void* stream);      /* dtype of qu / k / v / bias / ctx: bf16 | fp16 */
/* The same forward with the shifted positional score formed INSIDE the kernel (attention.py:87-89 and the pad-and-reshape shift of
 * :105-113 fused into the score): qv = q + v_bias [B*T][ldq], pos = positional projection [T][ldp], head h at column h*dh.  bias_out
 * (optional, (B,H,T,T)): the shifted score as the kernel used it, for sarssl_relpos_attn_bwd.  sarssl_relpos_attn_pos_supported:
 * T <= 256 (the score tile of 128 query rows lives in LDS), T % 8 == 0. */
int sarssl_relpos_attn_pos_supported(int T, int dh);
int sarssl_relpos_attn_fwd_pos(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldk, const void* pos, long ldp,
                               void* bias_out, void* ctx, long ldc, float* ctx32, float* lse, int B, int H, int T, int dh, float scale,
                               float p_drop, unsigned long long seed, const float* u_bias, const float* v_bias, int dtype, void* stream);
/*      hybrid mode: the context also leaves as an fp16 pair - ctx_lo [B*T][ldc] = fp16(c - fp16(c)) of the unrounded f32 context c, the
 *      lo operand of the output projection (attention.py:101) - instead of a sarssl_split_pair pass over ctx32.  fp16 tensors. */
int sarssl_relpos_attn_fwd_pos_pair(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldk, const void* pos, long ldp,
                                    void* bias_out, void* ctx, void* ctx_lo, long ldc, float* ctx32, float* lse, int B, int H, int T, int dh,
                                    float scale, float p_drop, unsigned long long seed, const float* u_bias, const float* v_bias, void* stream);
/*      round 6: the same for T > 256 (T % 8 == 0, d_head 64 | 128; BASELINE config 5: T = 624) - the slab of the shifted score covers one
 *      256-key block at a time, the position tiles that can reach it stream from memory.  qu / qv: the biased projections (q + u_bias,
 *      q + v_bias as sarssl_bias2 stores them); bias_out (may be NULL) = the (B,H,T,T) shifted score sarssl_relpos_attn_bwd reads.
 *      Replaces the positional-score sarssl_gemm(c_row_shift) launch and the attention kernel's read of its result. */
int sarssl_relpos_attn_pos_long_supported(int T, int dh);
int sarssl_relpos_attn_fwd_pos_long(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldk, const void* pos, long ldp,
                                    void* bias_out, void* ctx, long ldc, float* ctx32, float* lse, int B, int H, int T, int dh, float scale,
                                    float p_drop, unsigned long long seed, int dtype, void* stream);
/* u_bias / v_bias (both or neither, f32 [H*dh]; attention.py:54-55): qu == qv == the plain query projection q and the kernels form
 * q + u / q + v while loading their rows (the values sarssl_bias2 would have stored); same for sarssl_relpos_attn_bwd_pos */
/* Backward with the positional-score gradients formed in the dQ kernel - no d(bias) tensor, no sarssl_relshift_bwd pass, no batched
 * products after it (attention.py:87-89, 105-113 backward).  bias: the (B,H,T,T) shifted score sarssl_relpos_attn_fwd_pos wrote.
 * dqv [B*T][lddqv]: gradient of q + v_bias through the positional score; dpos_part bf16 (B, ntile, T, H*dh), ntile = ceil(T/128):
 * partial gradients of the positional projection (sum over the first two axes); dqv_fix: f32 workspace (B*H, ntile, 2, dh). */
int sarssl_relpos_attn_bwd_pos(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldk, const void* pos, long ldp,
                               const void* bias, const float* ctx32, const float* lse, const void* dctx, long lddc, void* dqu, long lddq,
                               void* dqv, long lddqv, void* dk, void* dv, long lddk, void* dpos_part, float* dqv_fix, float* dsum,
                               int B, int H, int T, int dh, float scale, float p_drop, unsigned long long seed, const float* u_bias,
                               const float* v_bias, void* dq_sum, long lddq_sum, int dtype, void* stream);
/* dq_sum (optional, [B*T][lddq_sum], not aliasing dqu / dqv): dqu + dqv - the gradient of the query projection - written by the dK / dV
 * kernel (what sarssl_axpby2d would make of the two tensors in a launch of its own) */
int sarssl_relpos_attn_bwd(const void* qu, long ldq, const void* k, const void* v, long ldk, const void* bias, const float* ctx32,
                           const float* lse, const void* dctx, long lddc, void* dqu, long lddq, void* dk, void* dv,
                           long lddk, void* dbias, float* dsum, int B, int H, int T, int dh, float scale, float p_drop,
                           unsigned long long seed, int dtype, void* stream);      /* bf16 | mixed 16 (qu / k / v / bias fp16) */

/* ---- CNN stem, channels-last (B,F,T,C): code/model.py:50-64 (patch_embed), masking code/model.py:533-564 */
int sarssl_mask_inputs(const float* x, const unsigned char* mp, const int* mch, int nb, int F, int Tn, int mode, void* spec,
                       void* spat, int dtype, void* stream);
int sarssl_stem_c1_fwd(const void* a0, const float* W1, long npix, void* y1, double* stats, int dtype, void* stream);
int sarssl_stem_c1_wgrad(const void* dy1, const void* a0, long npix, double* dW1d, int dtype, void* stream);
/*      same, from (dz1, y1): the first BatchNorm's backward normalisation is applied in registers (dy1 is never stored) */
/*      the whole backward of the first stem layer (BatchNorm sums, dgamma/dbeta, dW1) in one pass over (dz1, y1, a0): red = f64[644]
 *      workspace, dW1 (64x4) / dgamma / dbeta (64) are f32 buffers accumulated into */
int sarssl_stem_c1_bwd(const void* dz1, const void* y1, const void* a0, long npix, const float* aff, int use_stats, double* red,
                       float* dW1, float* dgamma, float* dbeta, int dtype, void* stream);
int sarssl_stem_c1_wgrad_bn(const void* dz1, const void* y1, const void* a0, long npix, const float* aff, const double* bnred,
                            int use_stats, double* dW1d, int dtype, void* stream);
int sarssl_conv3x3_fwd(const void* in, const void* w, void* out, int dtype, int w_dtype, int nb, int F, int T,
                       const float* scale, const float* shift, int precise, float* ws, double* stats, void* stream);
/*      data gradient (w = flipped/transposed taps) that also accumulates the BatchNorm-backward sums of the layer in front:
 *      red = f64[128] = [sum g | sum g*xhat], g = dz * relu'(bn(y)); aff = [scale|shift|mean|rstd] (4 x 64 f32); bf16 only.
 */
int sarssl_conv3x3_dgrad_bnred(const void* dy, const void* w, void* dz, int nb, int F, int T, const void* y, const float* aff,
                               double* red, int y_dtype, void* stream);     /* y_dtype: bf16 | fp16 (dy, w, dz: bf16) */
/*      measurement aid (no reference counterpart): buf = device memory, 5 slots x 4 u64; thread 0 of workgroup 0 of every bf16 3x3
 *      forward / data-gradient launch stores {s_memtime, s_memrealtime} at kernel entry and exit into the slot of its variant (0 forward
 *      with BN prologue, 1 data gradient, 2 data gradient + BN sums, 3 forward from the 4-channel input, 4 data gradient consumed in its
 *      epilogue): effective shader clock = d(memtime) / d(memrealtime) * sarssl_wall_clock_khz().  Set per context: sarssl_ctx_set_clock_probe. */
/*      scheduling aid (no reference counterpart): sarssl_ctx_set_conv_cus - workgroup count of the 3x3 gradient launches (data and weight
 *      gradients) issued under the context; 0 = the default rule (7/8 of the CUs, leaving room for the other encoder's stream). */
long sarssl_wall_clock_khz();
/*      sarssl_stamp: a one-thread launch that stores the constant-rate device clock (sarssl_wall_clock_khz ticks per ms) to *dst -
 *      capturable into the step graph: timeline markers of an unprofiled replay (tools/step_stamps.py) */
int sarssl_stamp(unsigned long long* dst, void* stream);
long sarssl_conv3x3_wgrad_workspace_bytes(int nb, int F, int T);
int sarssl_conv3x3_wgrad(const void* dy, const void* zin, int dtype, int nb, int F, int T, const float* scale,
                         const float* shift, float* dW, float* partial, int precise, void* stream);
/*      bf16: the weight gradient ADDED straight into the f32 (64,64,3,3) parameter-gradient buffer (nn.Conv2d layout) */
int sarssl_conv3x3_wgrad_acc(const void* dy, const void* zin, int nb, int F, int T, const float* scale, const float* shift,
                             float* grad_oihw, float* partial, int z_dtype, void* stream);      /* z_dtype: bf16 | fp16 (dy: bf16) */
/*      re-laid-out operands of the stem convolutions, one launch each (they follow the weights every step): conv_taps: (64,64,3,3)
 *      f32 -> fwd [9][co][ci] and dgr [9][ci][co] with flipped taps, f32 | bf16 | mixed 16 (fwd fp16, dgr bf16); patch_w: the frame-patch conv weight (d,4,F,1)
 *      (code/model.py:63) -> GEMM operand [d][f*4+c]; patch_wgrad_accum: grad (d,4,F,1) += sum over the nslice
 *      split-K partial products g [nslice][d][f*4+c]. */
int sarssl_conv_taps(const float* W, void* fwd, void* dgr, int dtype, void* stream);
int sarssl_patch_w(const float* W, void* out, int d, int F, int dtype, void* stream);
int sarssl_patch_wgrad_accum(const float* g, int nslice, float* grad, int d, int F, void* stream);
/* The first stem layer without its 64-channel output (reference: code/model.py:50-64, patch_embed[0:3] = Conv2d(4,64,1) + BatchNorm2d
 * + ReLU): stem_c1_stats = BatchNorm sums [sum | sum sq] f64[128] of y1 = W1 a0 from the moments of a0 (mom14: f64[16] scratch);
 * conv3x3_fwd_c1 / conv3x3_wgrad_c1_acc = the following 3x3 convolution (patch_embed[3]) and its weight gradient with the operand
 * relu(bn1(W1 a0)) formed from a0 (B,F,T,4) bf16 while staging;
 * stem_c1_bwd_a0 = sarssl_stem_c1_bwd with y1 recomputed (npix % 64 == 0, bf16). */
int sarssl_stem_c1_stats(const void* a0, long npix, const float* W1, double* mom14, double* sums128, int dtype, void* stream);
/* sarssl_stem_c1_stats followed by sarssl_bn_finalize (C = 64, N = npix) in two launches instead of three: aff (4 x 64) = scale | shift |
 * mean | rstd of BatchNorm(1), running statistics and the batch counter updated (pass null for eval-style use) */
int sarssl_stem_c1_stats_affine(const void* a0, long npix, const float* W1, double* mom14, const float* gamma, const float* beta,
                                float eps, float momentum, float* running_mean, float* running_var, long* nbt, float* aff,
                                int dtype, void* stream);
int sarssl_conv3x3_fwd_c1(const void* a0, const float* W1, const float* scale, const float* shift, const void* w, void* out,
                          int nb, int F, int T, double* stats, int dtype, void* stream);      /* dtype of a0 / w / out: bf16 | fp16 */
int sarssl_conv3x3_wgrad_c1_acc(const void* dy, const void* a0, const float* W1, int nb, int F, int T, const float* scale,
                                const float* shift, float* grad_oihw, float* partial, int a0_dtype, void* stream);
int sarssl_stem_c1_bwd_a0(const void* dz1, const void* a0, const float* W1, long npix, const float* aff, int use_stats,
                          double* red, float* dW1, float* dgamma, float* dbeta, int dtype, void* stream);     /* bf16 | mixed 16 (a0 fp16) */
/* conv3x3_dgrad_c1red: data gradient of patch_embed[3] consumed in its epilogue (masked with relu'(bn1(W1 a0)) and contracted over the
 * pixels against [a0 | 1] on the matrix cores): red f64[644] gets G at [co*4+c] and s1 at [512+co], nothing is stored;
 * stem_c1_bwd_finalize_mom completes dW1 / dgamma / dbeta from (red, the input's moments mom14 of sarssl_stem_c1_stats). */
int sarssl_conv3x3_dgrad_c1red(const void* dy, const void* w, const void* a0, const float* W1, const float* scale, const float* shift,
                               int nb, int F, int T, double* red, int a0_dtype, void* stream);
int sarssl_stem_c1_bwd_finalize_mom(const double* red, const double* mom14, const float* W1, long npix, const float* aff, int use_stats,
                                    float* dW1, float* dgamma, float* dbeta, void* stream);
int sarssl_stem_c4_fwd(const void* y3, const float* W4, const float* scale, const float* shift, int nb, int F, int Tn, void* y4,
                       int dtype, void* stream);
/* sarssl_stem_c4_fwd that also returns the BatchNorm sums of its stored output: stats8 f64[8] = [sum (4) | sum of squares (4)] */
int sarssl_stem_c4_fwd_stats(const void* y3, const float* W4, const float* scale, const float* shift, int nb, int F, int Tn, void* y4,
                             double* stats8, int dtype, void* stream);
int sarssl_stem_c4_bwd(const void* y3, const void* dy4, const float* W4, const float* scale, const float* shift,
                       const float* mean, const float* rstd, int nb, int F, int Tn, void* g3, double* red, int dtype,
                       void* stream);
/*      two-phase form (1.7 GB instead of 2.7 GB of traffic per encoder): phase 1 = sums only (red as above), phase 2 writes the
 *      BatchNorm(3)-input gradient dy3 directly (the 64->4 contraction is recomputed); use_stats = 0: eval-mode BatchNorm. */
int sarssl_stem_c4_bwd_sums(const void* y3, const void* dy4, const float* W4, const float* scale, const float* shift,
                            const float* mean, const float* rstd, int nb, int F, int Tn, double* red, int dtype, void* stream);
int sarssl_stem_c4_bwd_apply(const void* y3, const void* dy4, const float* W4, const float* scale, const float* shift,
                             const float* mean, const float* rstd, int nb, int F, int Tn, const double* red, int use_stats,
                             void* dy3, int dtype, void* stream);
/* the same pass; workgroup 0 also adds the parameter gradients from the finished sums: gW4 (4,64) += red[0:256], dbeta += red[256:320],
 * dgamma += red[320:384] */
int sarssl_stem_c4_bwd_apply_pg(const void* y3, const void* dy4, const float* W4, const float* scale, const float* shift,
                                const float* mean, const float* rstd, int nb, int F, int Tn, const double* red, int use_stats, void* dy3,
                                float* gW4, float* dgamma, float* dbeta, int dtype, void* stream);

/* ---- BatchNorm{1,2}d on channels-last [N][C] tensors (training statistics, running-stat update, backward):
 *      nn.BatchNorm2d in code/model.py:52-61, nn.BatchNorm1d in code/common/conformer/convolution.py:142 */
int sarssl_cl_stats(const void* x, long N, int C, double* sums, int dtype, void* stream);
int sarssl_bn_finalize(const double* sums, long N, int C, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, long* nbt, float* scale, float* shift, float* mean,
                       float* rstd, void* stream);
int sarssl_bn_eval_affine(int C, const float* gamma, const float* beta, float eps, const float* running_mean,
                          const float* running_var, float* scale, float* shift, float* mean, float* rstd, void* stream);
int sarssl_cl_affine_act(const void* x, long N, int C, const float* scale, const float* shift, int act, void* z, int dtype,
                         void* stream);
/* sarssl_bn_finalize + sarssl_cl_affine_act in one launch (training-mode BatchNorm + activation of the convolution module,
 * conformer/convolution.py:141-142): every thread forms its channels' affine from the sums, workgroup 0 writes the rows and moves the
 * running statistics; bit-identical to the two launches.  C % 8 == 0, C >= 64. */
int sarssl_cl_bn_train_act(const void* x, long N, int C, const double* sums, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, long* nbt, float* scale, float* shift, float* mean,
                           float* rstd, int act, void* z, int dtype, void* stream);
int sarssl_cl_bn_bwd_reduce(const void* dz, const void* y, long N, int C, const float* scale, const float* shift,
                            const float* mean, const float* rstd, int act, double* red, int dtype, void* stream);
int sarssl_cl_bn_bwd_apply(const void* dz, const void* y, long N, int C, const float* scale, const float* shift,
                           const float* mean, const float* rstd, int act, int g_is_masked, int use_stats, const double* red,
                           void* dy, int dtype, void* stream);
/* the same pass; workgroup 0 also adds the BatchNorm parameter gradients dbeta += red[0:C], dgamma += red[C:2C] */
int sarssl_cl_bn_bwd_apply_pg(const void* dz, const void* y, long N, int C, const float* scale, const float* shift, const float* mean,
                              const float* rstd, int act, int g_is_masked, int use_stats, const double* red, void* dy, float* dgamma,
                              float* dbeta, int dtype, void* stream);

/* ---- depthwise-conv part of the Conformer convolution module as LDS tiles (convolution.py:139-143 and its backward): GLU fused
 *      into the tile load, BatchNorm1d batch sums in the epilogue; the data gradient fused with the GLU backward; the weight gradient
 *      recomputes the GLU output.  h: [nb*Tn][2d] pointwise-conv output, c / dc: [nb*Tn][d], w / dw: f32 [d][31], sums: f64[2d]. */
int sarssl_dwglu_fwd(const void* h, const float* w, int nb, int Tn, int d, int ksize, void* c, double* sums, int dtype, void* stream);
int sarssl_dwglu_bwd(const void* dc, const void* h, const float* w, int nb, int Tn, int d, int ksize, void* dh, int dtype, void* stream);
long sarssl_dwglu_wgrad_workspace_bytes(int nb, int Tn, int d);
int sarssl_dwglu_wgrad(const void* dc, const void* h, int nb, int Tn, int d, int ksize, float* dw, float* partial, int dtype,
                       void* stream);

/* ---- Conformer row / elementwise kernels: LayerNorm (feed_forward.py:48, attention.py:139, convolution.py:137,
 *      Conformer.py:87), GLU (activation.py:31-42), depthwise conv k=31 (convolution.py:140), relative-shift softmax
 *      (attention.py:87-113), u/v bias add (attention.py:87-88) */
int sarssl_layernorm_fwd(const void* x, long ldx, long M, int d, const float* gamma, const float* beta, float eps, void* y,
                         long ldy, float* mean, float* rstd, int dtype, void* stream);
/* a Conformer block's closing LayerNorm followed by the first LayerNorm of the next block's feed-forward module (Conformer.py:88-90,
 * feed_forward.py:48) in one launch: y = LN_a(x), z = LN_b(y as stored); bit-identical to two sarssl_layernorm_fwd launches */
int sarssl_layernorm_fwd2(const void* x, long ldx, long M, int d, const float* gamma_a, const float* beta_a, float eps_a, void* y, long ldy,
                          float* mean_a, float* rstd_a, const float* gamma_b, const float* beta_b, float eps_b, void* z, long ldz,
                          float* mean_b, float* rstd_b, int dtype, void* stream);
long sarssl_layernorm_bwd_workspace_bytes(long M, int d);
int sarssl_layernorm_bwd(const void* dy, long lddy, const void* x, long ldx, long M, int d, const float* gamma,
                         const float* mean, const float* rstd, const void* resid, long ldr, void* dx, long lddx,
                         float* dgamma, float* dbeta, float* partial, int dtype, void* stream);
/*      + dx2 [M][d] = dx * dropout_mask(seed, m*d + c) * gscale, i.e. the sarssl_act_bwd(act = 0) pass the next module of the backward
 *      chain (Dropout backward of feed_forward.py:53 / attention.py:151 / convolution.py:145 + the half-step factor) starts with */
int sarssl_layernorm_bwd_drop(const void* dy, long lddy, const void* x, long ldx, long M, int d, const float* gamma,
                              const float* mean, const float* rstd, const void* resid, long ldr, void* dx, long lddx,
                              float* dgamma, float* dbeta, float* partial, void* dx2, float p_drop, unsigned long long seed,
                              float gscale, int dtype, void* stream);
/*      dgamma == NULL with partial != NULL: the [nparts][2 d] partial sums only (nparts = sarssl_layernorm_bwd_nparts(M)); the folds
 *      of up to 8 such launches (the LayerNorms of one Conformer block) then run as one launch: dgamma_q / dbeta_q += ... */
int sarssl_layernorm_bwd_nparts(long M);
int sarssl_ln_param_reduce_multi(const float* const* partial, const int* nparts, const int* d, float* const* dgamma,
                                 float* const* dbeta, int n_prob, void* stream);
int sarssl_glu_fwd(const void* h, long M, int d, void* g, int dtype, void* stream);
int sarssl_glu_bwd(const void* dg, const void* h, long M, int d, void* dh, int dtype, void* stream);
int sarssl_dwconv_fwd(const void* x, const float* w, int nb, int Tn, int d, int ksize, int flip, void* y, int dtype,
                      void* stream);
long sarssl_dwconv_wgrad_workspace_bytes(int nb, int Tn, int d);
int sarssl_dwconv_wgrad(const void* dy, const void* x, int nb, int Tn, int d, int ksize, float* dw, float* partial, int dtype,
                        void* stream);
int sarssl_softmax_relshift_fwd(const float* content, const float* pos, long nmat, int Tn, float scale, void* p, void* pd,
                                float p_drop, unsigned long long seed, int dtype, void* stream);
int sarssl_softmax_bwd(const float* dpd, const void* p, long nmat, int Tn, float scale, float p_drop, unsigned long long seed,
                       void* dscore, int dtype, void* stream);
int sarssl_relshift_bwd(const void* dscore, long nmat, int Tn, void* dpos, int dtype, void* stream);
int sarssl_bias2(const void* q, long ldq, long M, int d, const float* u, const float* v, void* qu, void* qv, int dtype,
                 void* stream);
int sarssl_axpby(const void* x, const void* y, float a, float b, long n, void* out, int dtype, void* stream);
int sarssl_axpby2d(const void* x, long ldx, const void* y, long ldy, float a, float b, long M, int N, void* out, long ldo,
                   int dtype, void* stream);
/* out[n] (x's dtype) = sum_m x[m][n]: few rows, many columns (batch sum of the positional-score gradient, model.py:RelPositionMultiHeadAttention) */
int sarssl_colsum_store(const void* x, long ldx, long M, int N, void* out, int dtype, void* stream);
/* up to 24 independent column sums in one launch (the bias gradients of one backward stage; autograd of nn.Linear / Conv1d biases,
   conformer/modules.py:35-48): parts[q][s][n] = sum over row slice s of xs[q][m][n], s < sarssl_colsum_slices(Ms[q], Ns[q]); every
   element is written (no atomics: run-to-run reproducible), sarssl_splitk_reduce_multi (M = 1, nsplit = slices) adds them to the
   gradient in slice order */
int sarssl_colsum_slices(long M, int N);
int sarssl_colsum_multi_partials(const void* const* xs, const long* ldxs, const long* Ms, const int* Ns, float* const* parts, int n,
                                 int dtype, void* stream);
int sarssl_act_bwd(const void* dz, const void* h, long n, int act, float p_drop, unsigned long long seed, float gscale,
                   void* dh, int dtype, void* stream);
int sarssl_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream);
int sarssl_f64_accum(const double* src, float* dst, int n, float scale, void* stream);
int sarssl_f64_accum2(const double* src, float* dst1, float* dst2, int n, void* stream);     /* dst1 += src[:n], dst2 += src[n:2n] */
/* f64 accumulators (BatchNorm / loss sums) handed to the reduction entry points are zeroed by a memset in front of each launch - unless
 * they lie inside [base, base + bytes), an arena the caller zeroes itself once per forward / backward pass (NULL unregisters). */

/* ---- loss: code/model.py:585-592 (channel select) + 721-747 (gen_loss).  fwd: sums = f64[128] scratch, out = f32[2]
 * (loss, diff); F <= 480. */
int sarssl_masked_mse_fwd(const void* pred, const float* x, const int* idx, const int* mch, int nb, int F, int Tn, int nm,
                          double* sums, float* out, int dtype, void* stream);
/*      the same; the finalize launch also copies (loss, diff) to out_keep (f32[2]) and adds them to acc (f64[2]); either may be null
 *      (the captured step's running sums, learner.py:104-110) */
int sarssl_masked_mse_fwd_acc(const void* pred, const float* x, const int* idx, const int* mch, int nb, int F, int Tn, int nm,
                              double* sums, float* out, float* out_keep, double* acc, int dtype, void* stream);
/* loss and, in the same pass, its gradient w.r.t. pred for an incoming gradient of 1 (= sarssl_masked_mse_bwd with gscale 1):
 * dpred (nb, Tn, F*4) of the gradient dtype; dtype: bf16 | f32 | mixed 16 (pred fp16, dpred bf16) */
int sarssl_masked_mse_fwd_bwd(const void* pred, const float* x, const int* idx, const int* mch, int nb, int F, int Tn, int nm,
                              double* sums, float* out, float* out_keep, double* acc, void* dpred, int dtype, void* stream);
int sarssl_masked_mse_bwd(const void* pred, const float* x, const unsigned char* mp, const int* mch, int nb, int F, int Tn,
                          int nm, float gscale, const float* gscale_dev, void* dpred, int dtype, void* stream);
/* ---- the decoder on the masked frames only (round 5).  gen_loss (code/model.py:721-747) reads the prediction at the masked frames of
 *      the masked channel; EmbedDecoder ['', 'fc'] (code/model.py:296-301, :321-334) is Linear - ReLU - Linear on every frame separately.
 *      Rows of unmasked frames therefore get a zero gradient and add nothing to a parameter gradient: the training step runs the decoder
 *      on the nb * nm gathered rows (exact; half the decoder's forward, data-gradient and weight-gradient work) and forms the full
 *      prediction only on request (vis).  idx [nb][nm] int32 ASCENDING per item; rows are 16-bit or f32 (dtype), d % 8 == 0.
 *      sarssl_gather_rows: dst [nb * nm][d] <- src rows (b, idx[b][j]); sarssl_scatter_rows: dst [nb * Tn][d] (zeroed first) <- src rows.
 *      sarssl_masked_mse_compact: loss / diff (and dpred_c for an incoming gradient of 1 when dpred_c != NULL) on pred_c [nb][nm][F * 4];
 *      sarssl_masked_mse_bwd_compact: the stand-alone gradient, scaled by gscale * (*gscale_dev). */
int sarssl_gather_rows(const void* src, long ld_src, const int* idx, int nb, int Tn, int nm, int d, void* dst, int dtype, void* stream);
int sarssl_scatter_rows(const void* src, const int* idx, int nb, int Tn, int nm, int d, void* dst, long ld_dst, int dtype, void* stream);
/*      hybrid mode: scatter of an f32 gradient [nb * nm][d] that also writes the bf16 operand of the next module of the backward chain with
 *      its dropout backward applied - dst16 = bf16(bf16(v) * gscale * keep(seed, index in the FULL tensor)), zeros elsewhere (both
 *      destinations [nb * Tn][d] contiguous, zeroed first); replaces sarssl_scatter_rows + sarssl_cast + sarssl_act_bwd over the full tensor */
int sarssl_scatter_rows_drop16(const float* src, const int* idx, int nb, int Tn, int nm, int d, float* dst32, void* dst16, float p_drop,
                               unsigned long long seed, float gscale, void* stream);
int sarssl_masked_mse_compact(const void* pred_c, const float* x, const int* idx, const int* mch, int nb, int F, int Tn, int nm, double* sums,
                              float* out, float* out_keep, double* acc, void* dpred_c, int dtype, void* stream);
int sarssl_masked_mse_bwd_compact(const void* pred_c, const float* x, const int* idx, const int* mch, int nb, int F, int Tn, int nm,
                                  float gscale, const float* gscale_dev, void* dpred_c, int dtype, void* stream);

/* ---- optimiser: torch.optim.Adam(betas=(0.9,0.999), weight_decay=0) at code/learner.py:83, over one flat buffer */
/*      p16 / ph16 (either may be null): bf16 / fp16 shadow copies of the updated parameters, the GEMM / convolution operands */
int sarssl_adam_step(float* p, const float* g, float* m, float* v, void* p16, void* ph16, long n, float gscale, float lr, float beta1,
                     float beta2, float eps, int step, void* stream);
/*      the same behind a device-side guard: `guard` -> the step's loss (f32 on the device).  Not finite -> the update is skipped as a
 *      whole (parameters, moments, shadow copies untouched; *nskipped += 1 when given): torch.cuda.amp.GradScaler.step of the reference's
 *      fp16 autocast path (code/learner.py:105-108: a step whose forward overflowed is not an optimizer step) without a host sync. */
int sarssl_adam_step_guard(float* p, const float* g, float* m, float* v, void* p16, void* ph16, long n, float gscale, float lr, float beta1,
                           float beta2, float eps, int step, const float* guard, int* nskipped, void* stream);

/* ---- device-resident step state: what varies from step to step when the whole step (code/learner.py:93-115: forward, backward,
 *      optimizer.step(), optimizer.zero_grad()) is replayed from a hipGraph with frozen launch arguments.  `state` is
 *      sarssl_step_state_bytes() of device memory: dropout salt (added to every launch's seed), Adam step count, lr, betas and the
 *      bias-correction factors.  sarssl_step_tick is the first node of a step (next salt, step += 1); sarssl_step_state_reset is the
 *      "optimizer re-created at the start of every epoch" of code/learner.py:83; attach(state) makes the launch wrappers hand the
 *      salt pointer to the dropout-drawing kernels (attach(NULL) detaches - keep it attached only around graph capture). */
long sarssl_step_state_bytes(void);
int sarssl_step_state_init(void* state, unsigned long long salt, float lr, float beta1, float beta2, void* stream);
int sarssl_step_state_reset(void* state, float lr, float beta1, float beta2, void* stream);
int sarssl_step_tick(void* state, void* stream);
int sarssl_adam_step_dev(float* p, float* g, float* m, float* v, void* p16, void* ph16, long n, float gscale, const void* state,
                         float eps, int zero_grad, void* stream);
/*      guarded form (see sarssl_adam_step_guard): a skipped step still clears the gradient buffer (zero_grad), takes the state's step
 *      count back by one and counts itself in the state (sarssl_step_state_skipped) */
int sarssl_adam_step_dev_guard(float* p, float* g, float* m, float* v, void* p16, void* ph16, long n, float gscale, void* state, float eps,
                               int zero_grad, const float* guard, void* stream);
int sarssl_step_state_skipped(const void* state, void* stream);      /* synchronises `stream`; -> steps skipped since init, < 0 on error */

/* ---- collectives: the gradient exchange of data-parallel training (replaces torch.nn.DataParallel's per-step replicate / gather /
 *      reduce, code/learner.py:25-31, :102).  One communicator per (process, device); the 128-byte id is generated by rank 0 and moved
 *      to the other ranks by the caller (any side channel: a file, a socket, torch.distributed's store).  RCCL is resolved with dlopen
 *      at first use - the library has no link-time dependency on it.  sarssl_allreduce_bucket: in-place f32 sum of one contiguous
 *      bucket of the flat gradient buffer over all ranks, enqueued on `stream` (asynchronous, capturable into a hipGraph); the
 *      1/world scaling is the `gscale` argument of sarssl_adam_step*. */
int sarssl_comm_available(void);                /* 1 when librccl.so.1 could be resolved in this process */
int sarssl_comm_rccl_version(void);             /* RCCL's version code, or -1 */
int sarssl_comm_unique_id(void* id128);
void* sarssl_comm_create(int nranks, int rank, const void* id128);     /* on the current HIP device; collective; NULL on failure */
int sarssl_comm_destroy(void* comm);
int sarssl_comm_size(void* comm);
int sarssl_allreduce_bucket(void* comm, float* bucket, long count, void* stream);

#ifdef __cplusplus
}
#endif
#endif
