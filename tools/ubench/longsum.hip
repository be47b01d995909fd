// Probe for tools/stress_streams.py: the simplest kernel with long-lived per-lane state.  Every thread adds n/threads values of x in a fixed
// order into THREE accumulators (one plain, one behind a data-dependent select, one inside a divergent branch) and stores them; nothing is shared between lanes, no LDS,
// no atomics.  Next to another stream it must return bit for bit what it returns alone.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(256) void k_longsum(const float* __restrict__ x, float* __restrict__ out, int64_t n, int rounds) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
    float a = 0.0f, b = 0.0f, c = 0.0f;
    for (int r = 0; r < rounds; ++r)
        for (int64_t i = tid * 4; i + 3 < n; i += nt * 4) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            a += (v.x + v.y) + (v.z + v.w);
            b += (v.x > 0.0f) ? v.y : v.z;
            // a REAL divergent region (a dependent load inside keeps the compiler from turning it into a select): executed under a
            // per-lane EXEC mask, the form that went wrong in k_mulq_bwd
            if (v.y > 0.25f) {
                const float w = x[(i + 4 * (int64_t)(v.z > 0.0f ? 3 : 5)) % n];
                c += w * v.w;
            }
        }
    out[3 * tid] = a;
    out[3 * tid + 1] = b;
    out[3 * tid + 2] = c;
}

extern "C" int probe_longsum(const float* x, float* out, int64_t n, int blocks, int rounds, void* stream) {
    hipLaunchKernelGGL(k_longsum, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, n, rounds);
    return (int)hipGetLastError();
}
