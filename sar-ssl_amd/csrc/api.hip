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
static thread_local const unsigned long long* g_salt = nullptr;
const unsigned long long* sarssl_dropout_salt() { return g_salt; }
// state: device pointer to a SarsslStepState (or null to detach).  While attached, every launch that draws dropout masks reads the
// salt through this pointer - attach only around graph capture: the pointer is baked into the captured launches.
extern "C" int sarssl_step_state_attach(void* state) {
    g_salt = state ? &((const SarsslStepState*)state)->salt : nullptr;
    return 0;
}
extern "C" long sarssl_step_state_bytes() { return (long)sizeof(SarsslStepState); }

__global__ void step_state_init_kernel(SarsslStepState* s, unsigned long long salt, float lr, float beta1, float beta2) {
    s->salt = salt; s->step = 0; s->lr = lr; s->beta1 = beta1; s->beta2 = beta2; s->step_size = 0.f; s->inv_bc2_sqrt = 1.f;
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
extern "C" int sarssl_step_tick(void* state, void* stream) {
    step_tick_kernel<<<1, 1, 0, (hipStream_t)stream>>>((SarsslStepState*)state);
    SARSSL_CHECK_LAUNCH("step_tick_kernel");
    return 0;
}

// ---- pre-zeroed arena ----------------------------------------------------------------------------------------------------------------
// The f64 accumulators of the reductions (BatchNorm sums, backward sums, loss sums) are zeroed by a hipMemsetAsync in front of every
// launch: ~26 memset nodes per training step.  The host side can instead hand out slices of ONE arena it zeroes once per forward /
// backward pass (hip.py: sums_zeroed); a pointer inside the registered range is taken as already zero and its memset is skipped.
// One range per device (the host keeps one arena per GPU; with a single process-global range the second device's arena used to evict the
// first one's, whose slices then silently got all their memsets back).  base == null clears the table.
#define ZERO_ARENA_MAX 16
static struct { const char* lo; const char* hi; } g_zero[ZERO_ARENA_MAX];
static int g_nzero = 0;
extern "C" int sarssl_zero_arena(const void* base, long bytes) {
    if (!base) { g_nzero = 0; return 0; }
    for (int i = 0; i < g_nzero; ++i)
        if (g_zero[i].lo == (const char*)base) { g_zero[i].hi = (const char*)base + bytes; return 0; }
    SARSSL_REQUIRE(g_nzero < ZERO_ARENA_MAX, "sarssl_zero_arena(table full)");
    g_zero[g_nzero].lo = (const char*)base; g_zero[g_nzero].hi = (const char*)base + bytes; ++g_nzero;
    return 0;
}
bool sarssl_prezeroed(const void* p) {
    for (int i = 0; i < g_nzero; ++i)
        if ((const char*)p >= g_zero[i].lo && (const char*)p < g_zero[i].hi) return true;
    return false;
}

// waves raise their issue priority (s_setprio) during MFMA phases: 2 = the ping-pong convolution only (measured in round 2: -2.4 ... -3.5 %
// on those launches alone; no effect on the GEMM kernels)
int sarssl_mfma_prio() { return 2; }
