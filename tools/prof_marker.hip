// Profiling aid (NOT part of the product library): an empty kernel whose grid size encodes a tag, launched between groups of
// kernels so that rocprofv3's per-dispatch CSV rows can be attributed to the group they belong to (tools/prof_counters.py).
#include <hip/hip_runtime.h>
__global__ void prof_marker_kernel() {}
extern "C" void prof_marker(int id, void* stream) { prof_marker_kernel<<<id, 64, 0, (hipStream_t)stream>>>(); }
