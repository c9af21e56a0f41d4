// Error plumbing + version/probe entry points of the C-ABI library.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void sarssl_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* sarssl_last_error() { return g_err; }
extern "C" int sarssl_abi_version() { return 2; }      // 2: fp16 / mixed-16 dtypes, dtype arguments of the 16-bit-only entry points

// Device probe: returns 0 and fills name/arch info when a gfx950 device is usable.
extern "C" int sarssl_device_info(int device, char* name_out, int name_len, int* cu_count, long* lds_bytes) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) { sarssl_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return -2; }
    snprintf(name_out, name_len, "%s|%s", p.name, p.gcnArchName);
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (long)p.sharedMemPerBlock;
    return 0;
}

// Compute-unit count of the current device (cached per device): persistent kernels launch one workgroup per CU.
int sarssl_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// ---- device-resident step state ----------------------------------------------------------------------------------------------------
// A training step captured into a hipGraph replays with frozen kernel arguments, so everything that changes from step to step lives
// in device memory: the dropout salt (added to each launch's static seed) and the Adam step count / bias corrections.
// (thread_local: the pointer is attached by the thread that captures a step and must only reach the launches THAT thread issues - a loader
//  or validation thread launching kernels at the same time keeps seeing "no salt")
const unsigned long long* sarssl_dropout_salt() { sarssl_ctx* c = sarssl_current(); return c ? c->salt : nullptr; }
// state: device pointer to a SarsslStepState (or null to detach).  While attached to a context, every launch issued under that context
// that draws dropout masks reads the salt through this pointer - attach only around graph capture: the pointer is baked into the
// captured launches.  (The context is current per THREAD: a loader or validation thread launching kernels at the same time under
// another context - or none - keeps seeing "no salt".)
extern "C" int sarssl_ctx_attach_step_state(sarssl_ctx* ctx, void* state) {
    SARSSL_REQUIRE(ctx != nullptr, "sarssl_ctx_attach_step_state");
    ctx->salt = state ? &((const SarsslStepState*)state)->salt : nullptr;
    return 0;
}
extern "C" long sarssl_step_state_bytes() { return (long)sizeof(SarsslStepState); }

__global__ void step_state_init_kernel(SarsslStepState* s, unsigned long long salt, float lr, float beta1, float beta2) {
    s->salt = salt; s->step = 0; s->lr = lr; s->beta1 = beta1; s->beta2 = beta2; s->step_size = 0.f; s->inv_bc2_sqrt = 1.f; s->nskipped = 0;
}
// (re)start: step count 0 (a fresh torch.optim.Adam, code/learner.py:83), learning rate, betas; salt_seed != 0 also reseeds the salt
__global__ void step_state_reset_kernel(SarsslStepState* s, float lr, float beta1, float beta2) {
    s->step = 0; s->lr = lr; s->beta1 = beta1; s->beta2 = beta2;
}
extern "C" int sarssl_step_state_init(void* state, unsigned long long salt, float lr, float beta1, float beta2, void* stream) {
    step_state_init_kernel<<<1, 1, 0, (hipStream_t)stream>>>((SarsslStepState*)state, salt, lr, beta1, beta2);
    SARSSL_CHECK_LAUNCH("step_state_init_kernel");
    return 0;
}
extern "C" int sarssl_step_state_reset(void* state, float lr, float beta1, float beta2, void* stream) {
    step_state_reset_kernel<<<1, 1, 0, (hipStream_t)stream>>>((SarsslStepState*)state, lr, beta1, beta2);
    SARSSL_CHECK_LAUNCH("step_state_reset_kernel");
    return 0;
}
// once per step, first node of the graph: next salt (SplitMix64 increment), next Adam step and its bias corrections - computed in
// double exactly like the host does for sarssl_adam_step, so both paths produce the same f32 factors
__global__ void step_tick_kernel(SarsslStepState* s) {
    s->salt += 0x9E3779B97F4A7C15ull;
    const int t = s->step + 1;
    s->step = t;
    const double bc1 = 1.0 - pow((double)s->beta1, (double)t), bc2 = 1.0 - pow((double)s->beta2, (double)t);
    s->step_size = (float)((double)s->lr / bc1);
    s->inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
}
// Measurement aid: one 64-bit store of the constant-rate device clock (s_memrealtime: sarssl_wall_clock_khz ticks per ms) - a marker that
// can be captured into the step graph between two launches of a stream, so that the timeline of an UNPROFILED replay can be read back
// (tools/step_stamps.py: rocprofv3 delays the second hardware queue's packets, its traces under-state the two-stream overlap).
__global__ void stamp_kernel(unsigned long long* dst) { *dst = __builtin_amdgcn_s_memrealtime(); }
extern "C" int sarssl_stamp(unsigned long long* dst, void* stream) {
    stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>(dst);
    SARSSL_CHECK_LAUNCH("stamp_kernel");
    return 0;
}
extern "C" int sarssl_step_state_skipped(const void* state, void* stream) {
    SarsslStepState h;
    if (hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return h.nskipped;
}
extern "C" int sarssl_step_tick(void* state, void* stream) {
    step_tick_kernel<<<1, 1, 0, (hipStream_t)stream>>>((SarsslStepState*)state);
    SARSSL_CHECK_LAUNCH("step_tick_kernel");
    return 0;
}

// ---- pre-zeroed arena ----------------------------------------------------------------------------------------------------------------
// The f64 accumulators of the reductions (BatchNorm sums, backward sums, loss sums) are zeroed by a hipMemsetAsync in front of every
// launch: ~26 memset nodes per training step.  The host side can instead hand out slices of ONE arena it zeroes once per forward /
// backward pass (hip.py: sums_zeroed); a pointer inside the registered range is taken as already zero and its memset is skipped.
// One range per context (the host keeps one arena per GPU).  base == null clears it.
extern "C" int sarssl_ctx_zero_arena(sarssl_ctx* ctx, const void* base, long bytes) {
    SARSSL_REQUIRE(ctx != nullptr, "sarssl_ctx_zero_arena");
    ctx->zero_lo = (const char*)base; ctx->zero_hi = base ? (const char*)base + bytes : nullptr;
    return 0;
}
bool sarssl_prezeroed(const void* p) {
    const sarssl_ctx* c = sarssl_current();
    return c && c->zero_lo && (const char*)p >= c->zero_lo && (const char*)p < c->zero_hi;
}

// ---- contexts ------------------------------------------------------------------------------------------------------------------------
// "no global mutable state except an opaque sarssl_ctx* created per device".  A context owns the caller-configurable
// state (gradient-convolution workgroup count, clock-probe buffer, attached step state, zeroed-arena range); kernels are launched
// under the context that is current on the calling thread, so two contexts on one device - two models, a training loop and a
// validation thread - do not see each other's settings.  No context current = library defaults.
static thread_local sarssl_ctx* t_ctx = nullptr;
sarssl_ctx* sarssl_current() { return t_ctx; }
extern "C" sarssl_ctx* sarssl_create(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { sarssl_set_error("sarssl_create: no device %d", device); return nullptr; }
    sarssl_ctx* c = (sarssl_ctx*)calloc(1, sizeof(sarssl_ctx));
    if (c) {
        c->device = device;
        // fp16-overflow word (see common.h): 4 bytes of device memory per context, allocated on the context's device
        int cur = 0;
        if (hipGetDevice(&cur) == hipSuccess && (cur == device || hipSetDevice(device) == hipSuccess)) {
            if (hipMalloc((void**)&c->ovf_flag, sizeof(int)) != hipSuccess || hipMemset(c->ovf_flag, 0, sizeof(int)) != hipSuccess) c->ovf_flag = nullptr;
            if (cur != device) (void)hipSetDevice(cur);
        }
    }
    return c;
}
// Requires that NO other thread has this context current (the pointer it holds would dangle): destroy after the threads that used it
// have made another context (or null) current.  The library keeps no list of threads.
extern "C" int sarssl_destroy(sarssl_ctx* ctx) {
    if (ctx && t_ctx == ctx) t_ctx = nullptr;
    if (ctx && ctx->ovf_flag) (void)hipFree(ctx->ovf_flag);
    free(ctx);
    return 0;
}
int* sarssl_overflow_flag() { sarssl_ctx* c = sarssl_current(); return c ? c->ovf_flag : nullptr; }
// 1 when a kernel issued under ctx has met a value outside fp16's range since the flag was last cleared (by a loss launch or by this
// call); synchronises `stream`.  -1 on error.
extern "C" int sarssl_ctx_fp16_overflow(sarssl_ctx* ctx, int clear, void* stream) {
    if (!ctx || !ctx->ovf_flag) return 0;
    int h = 0;
    if (hipMemcpyAsync(&h, ctx->ovf_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (clear && hipMemsetAsync(ctx->ovf_flag, 0, sizeof(int), (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return h != 0;
}
// ctx (or null) becomes the calling thread's current context
extern "C" int sarssl_make_current(sarssl_ctx* ctx) { t_ctx = ctx; return 0; }
extern "C" int sarssl_ctx_device(const sarssl_ctx* ctx) { return ctx ? ctx->device : -1; }
// workgroup count of the 3x3 gradient launches (data and weight gradients) issued under ctx; 0 = the default rule (7/8 of the CUs)
extern "C" int sarssl_ctx_set_conv_cus(sarssl_ctx* ctx, int ncus) {
    SARSSL_REQUIRE(ctx != nullptr, "sarssl_ctx_set_conv_cus");
    ctx->conv_cus_bwd = ncus > 0 ? ncus : 0;
    return 0;
}
extern "C" int sarssl_ctx_get_conv_cus(const sarssl_ctx* ctx) { return ctx ? ctx->conv_cus_bwd : 0; }
// clock-probe buffer (device memory, 5 slots x 4 u64) of the 3x3 forward / data-gradient launches issued under ctx; null = off
extern "C" int sarssl_ctx_set_clock_probe(sarssl_ctx* ctx, void* buf) {
    SARSSL_REQUIRE(ctx != nullptr, "sarssl_ctx_set_clock_probe");
    ctx->conv_clk = (unsigned long long*)buf;
    return 0;
}

// waves raise their issue priority (s_setprio) during MFMA phases: 2 = the ping-pong convolution only (measured in round 2: -2.4 ... -3.5 %
// on those launches alone; no effect on the GEMM kernels)
int sarssl_mfma_prio() { return 2; }
