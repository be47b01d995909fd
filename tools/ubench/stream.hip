// Micro-benchmark (GPU box): what do the streaming shapes of the cfg-2 step cost with different store policies / work splits?
//   copy of R rows x 4000 fp32 (ld 4000), one workgroup per row (16 B per lane per access) vs grid-stride with more loads in flight;
//   plain vs sc1 (write-through) vs nt stores; back-to-back launches timed with HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int POL> __device__ __forceinline__ void st16(float* p, f32x4 v) {
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
    else if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
// loads stay plain C++ (an asm load's destination registers are not protected until the data lands: the compiler may reuse them)
template <int POL> __device__ __forceinline__ f32x4 ld16(const float* p) {
    if (POL == 0) return *reinterpret_cast<const f32x4*>(p);
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
}
// one workgroup per row, 4 float4 per thread, all loads first
template <int SP, int LP> __global__ __launch_bounds__(256) void k_row(const float* __restrict__ x, float* __restrict__ y, int M, int ld) {
    const float* xr = x + (size_t)blockIdx.x * ld;
    float* yr = y + (size_t)blockIdx.x * ld;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { int m = 4 * threadIdx.x + 1024 * i; if (m < M) v[i] = ld16<LP>(xr + m); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { int m = 4 * threadIdx.x + 1024 * i; if (m < M) { v[i] = v[i] * 1.0001f; st16<SP>(yr + m, v[i]); } }
}
// R consecutive rows per workgroup
template <int SP, int LP, int R> __global__ __launch_bounds__(256) void k_rows(const float* __restrict__ x, float* __restrict__ y, int M, int ld) {
    for (int r = 0; r < R; ++r) {
        const size_t row = (size_t)blockIdx.x * R + r;
        const float* xr = x + row * ld;
        float* yr = y + row * ld;
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { int m = 4 * threadIdx.x + 1024 * i; if (m < M) v[i] = ld16<LP>(xr + m); }
    #pragma unroll
        for (int i = 0; i < 4; ++i) { int m = 4 * threadIdx.x + 1024 * i; if (m < M) { v[i] = v[i] * 1.0001f; st16<SP>(yr + m, v[i]); } }
    }
}
// read-only reduction (no big output)
__global__ __launch_bounds__(256) void k_read(const float* __restrict__ x, float* __restrict__ y, int M, int ld) {
    const float* xr = x + (size_t)blockIdx.x * ld;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { int m = 4 * threadIdx.x + 1024 * i; if (m < M) v[i] = ld16<0>(xr + m); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { int m = 4 * threadIdx.x + 1024 * i; if (m < M) s += v[i][0] + v[i][1] + v[i][2] + v[i][3]; }
    if (s == 123.456f) y[blockIdx.x] = s;
}
// write-only
template <int SP> __global__ __launch_bounds__(256) void k_write(float* __restrict__ y, int M, int ld) {
    float* yr = y + (size_t)blockIdx.x * ld;
    f32x4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) { int m = 4 * threadIdx.x + 1024 * i; if (m < M) st16<SP>(yr + m, v); }
}
__global__ void k_empty() {}

template <typename F> float timeit(F f, int n = 50) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / n;
}

int main() {
    const int M = 3999, ld = 4000;
    for (int rows : {1024, 4096}) {
        const size_t n = (size_t)rows * ld;
        // ring of buffers > 256 MiB so that nothing is served from the Infinity Cache
        const int NB = (int)((600ull << 20) / (n * 4)) + 2;
        std::vector<float*> xs(NB), ys(NB);
        for (int i = 0; i < NB; ++i) { CK(hipMalloc(&xs[i], n * 4)); CK(hipMalloc(&ys[i], n * 4)); CK(hipMemset(xs[i], 0, n * 4)); CK(hipMemset(ys[i], 0, n * 4)); }
        int it = 0;
        auto nxt = [&]() { it = (it + 1) % NB; return it; };
        const double mb = n * 4 / 1e6;
        printf("rows %d: %.1f MB per tensor, ring of %d buffer pairs\n", rows, mb, NB);
        float t;
        t = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); });
        printf("  empty kernel back-to-back              %6.2f us\n", t);
#define RUN(name, mbs, ...) t = timeit([&] { int i = nxt(); (void)i; __VA_ARGS__; }); printf("  %-38s %6.2f us  %7.1f GB/s\n", name, t, (mbs) / t * 1e3 / 1e3 * 1e3 / 1e3);
        RUN("copy row/WG plain", 2 * mb, hipLaunchKernelGGL((k_row<0, 0>), dim3(rows), dim3(256), 0, 0, xs[i], ys[i], M, ld));
        RUN("copy row/WG sc1 store", 2 * mb, hipLaunchKernelGGL((k_row<1, 0>), dim3(rows), dim3(256), 0, 0, xs[i], ys[i], M, ld));
        RUN("copy row/WG nt store", 2 * mb, hipLaunchKernelGGL((k_row<2, 0>), dim3(rows), dim3(256), 0, 0, xs[i], ys[i], M, ld));
        RUN("copy row/WG sc0sc1 store", 2 * mb, hipLaunchKernelGGL((k_row<3, 0>), dim3(rows), dim3(256), 0, 0, xs[i], ys[i], M, ld));
        RUN("copy row/WG nt load + nt store", 2 * mb, hipLaunchKernelGGL((k_row<2, 1>), dim3(rows), dim3(256), 0, 0, xs[i], ys[i], M, ld));
        RUN("copy row/WG nt load + plain store", 2 * mb, hipLaunchKernelGGL((k_row<0, 1>), dim3(rows), dim3(256), 0, 0, xs[i], ys[i], M, ld));
        RUN("copy 4 rows/WG plain", 2 * mb, hipLaunchKernelGGL((k_rows<0, 0, 4>), dim3(rows / 4), dim3(256), 0, 0, xs[i], ys[i], M, ld));
        RUN("copy 4 rows/WG nt store", 2 * mb, hipLaunchKernelGGL((k_rows<2, 0, 4>), dim3(rows / 4), dim3(256), 0, 0, xs[i], ys[i], M, ld));
        RUN("copy 2 rows/WG plain", 2 * mb, hipLaunchKernelGGL((k_rows<0, 0, 2>), dim3(rows / 2), dim3(256), 0, 0, xs[i], ys[i], M, ld));
        RUN("read only row/WG", mb, hipLaunchKernelGGL(k_read, dim3(rows), dim3(256), 0, 0, xs[i], ys[i], M, ld));
        RUN("write only row/WG plain", mb, hipLaunchKernelGGL((k_write<0>), dim3(rows), dim3(256), 0, 0, ys[i], M, ld));
        RUN("write only row/WG sc1", mb, hipLaunchKernelGGL((k_write<1>), dim3(rows), dim3(256), 0, 0, ys[i], M, ld));
        RUN("write only row/WG nt", mb, hipLaunchKernelGGL((k_write<2>), dim3(rows), dim3(256), 0, 0, ys[i], M, ld));
        // producer -> consumer pairs: does the consumer pay for the producer's dirty lines?
        RUN("write(plain) then read pair", 2 * mb, { hipLaunchKernelGGL((k_write<0>), dim3(rows), dim3(256), 0, 0, ys[i], M, ld); hipLaunchKernelGGL(k_read, dim3(rows), dim3(256), 0, 0, ys[i], xs[i], M, ld); });
        RUN("write(nt) then read pair", 2 * mb, { hipLaunchKernelGGL((k_write<2>), dim3(rows), dim3(256), 0, 0, ys[i], M, ld); hipLaunchKernelGGL(k_read, dim3(rows), dim3(256), 0, 0, ys[i], xs[i], M, ld); });
        RUN("write(sc1) then read pair", 2 * mb, { hipLaunchKernelGGL((k_write<1>), dim3(rows), dim3(256), 0, 0, ys[i], M, ld); hipLaunchKernelGGL(k_read, dim3(rows), dim3(256), 0, 0, ys[i], xs[i], M, ld); });
        for (int i = 0; i < NB; ++i) { CK(hipFree(xs[i])); CK(hipFree(ys[i])); }
    }
    return 0;
}
