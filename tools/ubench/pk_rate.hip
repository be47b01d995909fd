// Issue rate of the fp32 vector forms a quantizer loop is made of, wave64 on gfx950: scalar v_fma_f32 / v_mul_f32 / v_add_f32 against
// the packed v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 in their DEFAULT operand selection (op_sel:[0,0] op_sel_hi:[1,1]: low with low,
// high with high -- the only form the library admits, tests/test_host.py scans for any other), and two mixes (a packed fma beside a
// v_rndne / v_cvt pair, the shape of a quantizer's inner loop).  1 / 2 / 4 / 8 waves per SIMD, eight independent chains per wave, one
// or two workgroups per CU; each case alone AND beside a bf16-MFMA loop resident on the same CUs from another stream (the companion the
// op_sel hazard of profiles/r03_pk_opsel_repro.txt needs; the packed results are also checked, bit for bit, against the scalar forms).
//   ./pk_rate            -> one line per (form, waves per SIMD, alone | beside MFMA): cycles per wave-instruction and ELEMENTS per cycle and SIMD
// "settle packed fp32 with a committed microbenchmark".
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { S_FMA, S_MUL, S_ADD, P_FMA, P_MUL, P_ADD, MIX_S, MIX_P, NMODE };
static const char* NAMES[NMODE] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
                                   "2x(v_fma + v_rndne + v_mul) scalar", "v_pk_fma + 2 v_rndne + v_pk_mul"};
static const int ELEMS[NMODE] = {1, 1, 1, 2, 2, 2, 2, 2};       // elements one "group" of the unrolled body retires per lane
static const int INSTS[NMODE] = {1, 1, 1, 1, 1, 1, 6, 4};       // vector instructions of one group

// one group on chain i.  Values stay bounded: x <- x * w + h with |w| < 1.
template <int MODE>
__device__ __forceinline__ void group(f32x2& a, const f32x2 w, const f32x2 h) {
    if constexpr (MODE == S_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a.x) : "v"(w.x), "v"(h.x));
    if constexpr (MODE == S_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a.x) : "v"(w.x));
    if constexpr (MODE == S_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a.x) : "v"(h.x));
    if constexpr (MODE == P_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(w), "v"(h));
    if constexpr (MODE == P_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(w));
    if constexpr (MODE == P_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(h));
    if constexpr (MODE == MIX_S)
        asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %4, %5\n\tv_rndne_f32 %0, %0\n\tv_rndne_f32 %1, %1\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %4"
                     : "+v"(a.x), "+v"(a.y) : "v"(w.x), "v"(h.x), "v"(w.y), "v"(h.y));
    if constexpr (MODE == MIX_P) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(w), "v"(h));
        asm volatile("v_rndne_f32 %0, %0\n\tv_rndne_f32 %1, %1" : "+v"(a.x), "+v"(a.y));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(w));
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_rate(float* out, unsigned long long* cyc, int iters) {
    f32x2 a[8], w[8], h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = f32x2{0.25f + 0.01f * i, -0.5f + 0.02f * i};
        w[i] = f32x2{0.75f - 0.03f * i, -0.625f + 0.01f * i};
        h[i] = f32x2{0.125f * (i + 1), -0.0625f * (i + 1)};
    }
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) group<MODE>(a[i], w[i], h[i]);
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {          // every wave: its loop's start / end on the 100 MHz clock and its shader-clock ticks
        unsigned long long* c = cyc + 3 * ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        c[0] = r0; c[1] = r1; c[2] = t1 - t0;
    }
}

// the packed forms against the scalar ones on the same operands, bit for bit (every lane its own operands)
__global__ __launch_bounds__(256) void k_check(unsigned* bad, int iters) {
    unsigned hsh = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, nbad = 0;
    f32x2 acc = {0.25f, -0.5f};
    for (int i = 0; i < iters; ++i) {
        hsh = hsh * 1664525u + 1013904223u;
        const f32x2 t = {__uint_as_float(0x3f800000u | (hsh >> 9)) - 1.5f, __uint_as_float(0x3f800000u | ((hsh * 2246822519u) >> 9)) - 1.5f};
        const f32x2 c = {0.125f, -0.375f};
        f32x2 pf, pm, pa; float e0, e1;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pf) : "v"(acc), "v"(t), "v"(c));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pm) : "v"(acc), "v"(t));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pa) : "v"(acc), "v"(t));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(acc.x), "v"(t.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(acc.y), "v"(t.y), "v"(c.y));
        nbad += (__float_as_uint(pf.x) != __float_as_uint(e0)) + (__float_as_uint(pf.y) != __float_as_uint(e1));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(acc.x), "v"(t.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(acc.y), "v"(t.y));
        nbad += (__float_as_uint(pm.x) != __float_as_uint(e0)) + (__float_as_uint(pm.y) != __float_as_uint(e1));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(acc.x), "v"(t.x));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(acc.y), "v"(t.y));
        nbad += (__float_as_uint(pa.x) != __float_as_uint(e0)) + (__float_as_uint(pa.y) != __float_as_uint(e1));
        acc.x = e0 * 0.5f + 0.25f; acc.y = e1 * 0.5f - 0.125f;
    }
    if (nbad) atomicAdd(bad, nbad);
}

// companion: bf16 MFMAs from registers, one wave per SIMD (100 KB of dynamic LDS keep it to one workgroup per CU), a FIXED trip count
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    union { unsigned u[4]; bf16x8 v; } a, b;
    for (int k = 0; k < 4; ++k) { a.u[k] = 0x3c003c00u + threadIdx.x; b.u[k] = 0x3b803b80u + k; }
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

typedef void (*rate_t)(float*, unsigned long long*, int);
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    float *out, *out2; unsigned long long* cyc; unsigned* bad;
    CK(hipMalloc(&out, (size_t)2 * ncu * 1024 * 4)); CK(hipMalloc(&out2, (size_t)64 * ncu * 256 * 4));
    CK(hipMalloc(&cyc, (size_t)2 * ncu * 16 * 3 * 8)); CK(hipMalloc(&bad, 4));
    CK(hipFuncSetAttribute((const void*)k_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));   // one companion workgroup per CU at a time
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const rate_t kern[NMODE] = {k_rate<S_FMA>, k_rate<S_MUL>, k_rate<S_ADD>, k_rate<P_FMA>, k_rate<P_MUL>, k_rate<P_ADD>, k_rate<MIX_S>, k_rate<MIX_P>};
    printf("# %s, %d CUs, property clock %d MHz, %d x 64 groups per wave, eight independent chains per wave.\n"
           "# span us = mean over workgroups of (last wave's loop end - first wave's loop start) on s_memrealtime (100 MHz): the time the SIMDs\n"
           "# needed for ALL the workgroup's waves, without the launch delay a companion adds to the event time; first us = the first wave's own\n"
           "# loop time (the arbiter favours the oldest wave: not a throughput); GHz = s_memtime ticks / ns of that wave; cyc/inst = span x GHz /\n"
           "# (instructions per wave x waves per SIMD); el/cyc, el/ns = elements per cycle / ns and SIMD.\n", prop.gcnArchName, ncu, prop.clockRate / 1000, iters);
    printf("%-38s %3s %6s | %9s %9s %9s %5s | %8s %7s %7s\n", "form", "w/S", "beside", "event us", "span us", "first us", "GHz", "cyc/inst", "el/cyc", "el/ns");
    std::vector<unsigned long long> hc((size_t)2 * ncu * 16 * 3);
    for (int beside = 0; beside < 2; ++beside)
        for (int mode = 0; mode < NMODE; ++mode)
            for (int wps : {1, 2, 4, 8}) {
                // wps waves per SIMD: one workgroup of 256 * wps threads per CU, or two of 1024 for 8 (64 VGPRs each: the kernel needs ~50)
                const int threads = wps == 8 ? 1024 : 256 * wps, grid = wps == 8 ? 2 * ncu : ncu;
                float ms = 0.f;
                for (int rep = 0; rep < 2; ++rep) {
                    // companion: 16 workgroups of 4 waves per CU in turn (its grid outlasts the case; its workgroups come and go, so the
                    // dispatcher interleaves the two kernels' workgroups on every CU)
                    if (beside) hipLaunchKernelGGL(k_mfma, dim3(64 * ncu), dim3(256), 100 * 1024, s2, out2, iters * 4);
                    CK(hipEventRecord(e0, s1));
                    hipLaunchKernelGGL(kern[mode], dim3(grid), dim3(threads), 0, s1, out, cyc, iters);
                    CK(hipEventRecord(e1, s1));
                    CK(hipEventSynchronize(e1));
                    const bool still = beside && hipStreamQuery(s2) == hipErrorNotReady;
                    CK(hipDeviceSynchronize());
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (beside && !still && rep == 1) printf("# (companion finished first in the next line)\n");
                }
                const int nw = threads / 64;
                CK(hipMemcpy(hc.data(), cyc, (size_t)grid * nw * 3 * 8, hipMemcpyDeviceToHost));
                double span = 0, first = 0, ticks = 0;
                for (int g = 0; g < grid; ++g) {
                    unsigned long long lo = ~0ull, hi = 0;
                    for (int w = 0; w < nw; ++w) { const unsigned long long* c = &hc[3 * ((size_t)g * nw + w)]; lo = c[0] < lo ? c[0] : lo; hi = c[1] > hi ? c[1] : hi; }
                    span += (hi - lo) * 10.0; first += (hc[3 * (size_t)g * nw + 1] - hc[3 * (size_t)g * nw]) * 10.0; ticks += hc[3 * (size_t)g * nw + 2];
                }
                span /= grid; first /= grid; ticks /= grid;                                     // ns, ns, shader ticks
                const double ghz = ticks / first;
                // the two workgroups of the 8-wave case share a CU: their spans overlap, each SIMD runs 8 waves over (about) one span
                const double insts = (double)iters * 64 * INSTS[mode] * wps, el = (double)iters * 64 * ELEMS[mode] * 64 * wps;
                printf("%-38s %3d %6s | %9.1f %9.1f %9.1f %5.2f | %8.2f %7.2f %7.2f\n", NAMES[mode], wps, beside ? "mfma" : "alone", ms * 1e3, span * 1e-3,
                       first * 1e-3, ghz, span * ghz / insts, el / (span * ghz), el / span);
            }
    for (int beside = 0; beside < 2; ++beside) {
        CK(hipMemset(bad, 0, 4));
        if (beside) hipLaunchKernelGGL(k_mfma, dim3(64 * ncu), dim3(256), 100 * 1024, s2, out2, iters * 4);
        hipLaunchKernelGGL(k_check, dim3(4 * ncu), dim3(256), 0, s1, bad, 20000);
        CK(hipDeviceSynchronize());
        unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("packed (default op_sel) vs scalar, %s: %u mismatching halves of %.3g\n", beside ? "beside the MFMA loop" : "alone", hb, 6.0 * 4 * ncu * 256 * 20000);
    }
    return 0;
}
