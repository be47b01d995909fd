// Probe for tools/stress_streams.py: the simplest kernel with long-lived per-lane state.  Every thread adds n/threads values of x in a fixed
// order into TWO accumulators (one plain, one behind a data-dependent select) and stores them; nothing is shared between lanes, no LDS,
// no atomics.  Next to another stream it must return bit for bit what it returns alone.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(256) void k_longsum(const float* __restrict__ x, float* __restrict__ out, int64_t n, int rounds) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
    float a = 0.0f, b = 0.0f;
    for (int r = 0; r < rounds; ++r)
        for (int64_t i = tid * 4; i + 3 < n; i += nt * 4) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            a += (v.x + v.y) + (v.z + v.w);
            b += (v.x > 0.0f) ? v.y : v.z;
        }
    out[2 * tid] = a;
    out[2 * tid + 1] = b;
}

extern "C" int probe_longsum(const float* x, float* out, int64_t n, int blocks, int rounds, void* stream) {
    hipLaunchKernelGGL(k_longsum, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, n, rounds);
    return (int)hipGetLastError();
}
