// Issue rate of v_fmac_f32 against v_fmac_f32_dpp row_newbcast (one source broadcast from a lane of each 16-lane row), and of an LDS
// ds_read_b128 in which every lane of a 16-lane row reads its own 16 bytes against one in which a quad of lanes reads the same 16 bytes
// (the two ways an LSTM step can hand h_{t-1} to 512 gate rows).  One workgroup of NW waves per CU, cycles by s_memtime.
//   ./dpp_fma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k_rate(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float hs[1024];
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) hs[e] = 0.001f * e;
    __syncthreads();
    float acc[8], w[8], h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = 0.f; w[i] = 1.f + i + threadIdx.x; h[i] = 0.5f * i + threadIdx.x; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 2 || MODE == 3) {
            // MODE 2: quad split (lane reads the 16 bytes of its k-quarter: 4 distinct addresses per wave); MODE 3: row split (16 distinct)
            const int off = MODE == 2 ? (threadIdx.x & 3) * 36 : (threadIdx.x & 15) * 4;
#pragma unroll
            for (int i = 0; i < 8; i += 4) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 v;
                const unsigned addr = (unsigned)(size_t)(hs) + 4u * (off + 64 * (i / 4) + (it & 1) * 256);
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
                h[i] = v.x; h[i + 1] = v.y; h[i + 2] = v.z; h[i + 3] = v.w;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (MODE == 1 || MODE == 3) {
                    switch (r) {
#define C(n) case n: asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(h[i]), "v"(w[i])); break;
                        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
                    }
                } else {
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(h[i]), "v"(w[i]));
                }
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
    CHECK(hipMalloc(&cyc, 256 * sizeof(unsigned long long)));
    const int iters = 2000;
    const char* names[4] = {"v_fmac_f32", "v_fmac_f32_dpp row_newbcast", "quad-split b128 x2 + 128 fmac", "row-split b128 x2 + 128 fmac_dpp"};
    for (int nw : {4, 8, 16}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(64 * nw), 0, 0, out, cyc, iters); break;
                    case 1: hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(64 * nw), 0, 0, out, cyc, iters); break;
                    case 2: hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(64 * nw), 0, 0, out, cyc, iters); break;
                    default: hipLaunchKernelGGL(k_rate<3>, dim3(256), dim3(64 * nw), 0, 0, out, cyc, iters); break;
                }
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::vector<unsigned long long> h(256);
            CHECK(hipMemcpy(h.data(), cyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double m = 0;
            for (auto v : h) m += (double)v;
            m /= 256;
            // 128 FMAs per wave and iteration; nw / 4 waves per SIMD
            printf("%2d waves/CU  %-34s %8.1f ticks / iteration  = %5.2f ticks, %5.2f ns per FMA and SIMD (kernel %.1f us: %.2f ticks / ns)\n", nw,
                   names[mode], m / iters, m / iters / 128.0 / (nw / 4.0), ms * 1e6 / iters / 128.0 / (nw / 4.0), ms * 1e3, m / (ms * 1e6));
        }
    }
    return 0;
}
