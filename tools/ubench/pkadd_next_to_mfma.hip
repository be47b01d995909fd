// Reproducer for docs/history/DESIGN_rounds_1-5.md 9 (round 3): on gfx950 a packed fp32 add whose SECOND source has its halves swapped,
//     v_pk_add_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0]          (D.lo = A.lo + B.hi, D.hi = A.hi + B.lo)
// returns a wrong sum in lanes 48-63 while a kernel on ANOTHER stream is resident on the same CUs; alone it is exact.
//   hipcc --offload-arch=gfx950 -O3 -o pkadd_next_to_mfma pkadd_next_to_mfma.hip -ldl
//   ./pkadd_next_to_mfma [rounds] [pkadd|mfma|tgemm] [path of libfqss_hip.so for tgemm]
// Victim: every lane runs a chain of ONE packed instruction per step (inline asm: exactly the form named) and checks both halves, bit
// for bit, against scalar v_add_f32 / v_mul_f32 / v_fma_f32 of the same operands; no LDS, no atomics until the end.  Aggressors:
// "pkadd" = nothing but default-form v_pk_add_f32 chains; "mfma" = bf16 MFMA + LDS (ds_read_b64_tr_b16) loop, 64 KB of LDS;
// "tgemm" = the library's teacher GEMM (where the effect was first seen).  A mismatch is also classified: does the wrong half equal what
// the DEFAULT operand selection (A.lo + B.lo / A.hi + B.hi) would have produced?
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define PK2(name, mods) asm volatile(name " %0, %1, %2" mods : "=v"(p) : "v"(acc), "v"(t))
#define PK3(name, mods) asm volatile(name " %0, %1, %2, %3" mods : "=v"(p) : "v"(acc), "v"(t), "v"(c))
#define SC2(name, d, a, b) asm volatile(name " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define SC3(name, d, a, b, cc) asm volatile(name " %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(cc))
// OP 0 add | 1 mul | 2 fma;  SL / SH: bit i set = the LOW / HIGH result takes the HIGH half of source i (default form: SL 0, SH 3)
template <int OP, int SL, int SH>
__global__ __launch_bounds__(256) void k_victim(unsigned* bad, int iters) {
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, nbad = 0, ndef = 0, first = 0xffffffffu;
    f32x2 acc = {0.25f, -0.5f};
    const f32x2 c = {0.125f, -0.375f};
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        f32x2 t = {__uint_as_float(0x3f800000u | (h >> 9)) - 1.5f, __uint_as_float(0x3f800000u | ((h * 2246822519u) >> 9)) - 1.5f};
        f32x2 p;
        if (OP == 0 && SL == 0 && SH == 3) PK2("v_pk_add_f32", "");
        if (OP == 0 && SL == 2 && SH == 1) PK2("v_pk_add_f32", " op_sel:[0,1] op_sel_hi:[1,0]");
        if (OP == 0 && SL == 1 && SH == 2) PK2("v_pk_add_f32", " op_sel:[1,0] op_sel_hi:[0,1]");
        if (OP == 0 && SL == 2 && SH == 3) PK2("v_pk_add_f32", " op_sel:[0,1] op_sel_hi:[1,1]");
        if (OP == 0 && SL == 0 && SH == 1) PK2("v_pk_add_f32", " op_sel_hi:[1,0]");
        if (OP == 1 && SL == 2 && SH == 1) PK2("v_pk_mul_f32", " op_sel:[0,1] op_sel_hi:[1,0]");
        if (OP == 2 && SL == 2 && SH == 1) PK3("v_pk_fma_f32", " op_sel:[0,1,0] op_sel_hi:[1,0,1]");
        const float a0 = (SL & 1) ? acc.y : acc.x, b0 = (SL & 2) ? t.y : t.x, a1 = (SH & 1) ? acc.y : acc.x, b1 = (SH & 2) ? t.y : t.x;
        float e0, e1, d0, d1;      // expected halves; d = what the DEFAULT selection would give
        if (OP == 0) { SC2("v_add_f32", e0, a0, b0); SC2("v_add_f32", e1, a1, b1); SC2("v_add_f32", d0, acc.x, t.x); SC2("v_add_f32", d1, acc.y, t.y); }
        if (OP == 1) { SC2("v_mul_f32", e0, a0, b0); SC2("v_mul_f32", e1, a1, b1); SC2("v_mul_f32", d0, acc.x, t.x); SC2("v_mul_f32", d1, acc.y, t.y); }
        if (OP == 2) { SC3("v_fma_f32", e0, a0, b0, c.x); SC3("v_fma_f32", e1, a1, b1, c.y); SC3("v_fma_f32", d0, acc.x, t.x, c.x); SC3("v_fma_f32", d1, acc.y, t.y, c.y); }
        const bool w0 = __float_as_uint(p.x) != __float_as_uint(e0), w1 = __float_as_uint(p.y) != __float_as_uint(e1);
        nbad += w0 + w1;
        if ((w0 || w1) && first == 0xffffffffu) first = i;
        ndef += (w0 && __float_as_uint(p.x) == __float_as_uint(d0)) + (w1 && __float_as_uint(p.y) == __float_as_uint(d1));
        acc.x = (h & 64u) ? e0 * 0.5f : acc.x * 0.5f + 0.25f;     // bounded chain; the select mimics the v_cndmask of the kernel it was found in
        acc.y = e1 * 0.5f - 0.125f;
    }
    if (nbad) { atomicAdd(&bad[0], nbad); atomicAdd(&bad[1 + (threadIdx.x & 63) / 16], nbad); atomicAdd(&bad[5], ndef); atomicAdd(&bad[6], 1u);
                atomicMin(&bad[7], first); atomicMax(&bad[8], first); atomicAdd(&bad[9 + (first * 8) / iters], 1u); }
}

__global__ __launch_bounds__(256) void k_pkadd_hog(float* out, int iters) {       // default-form packed adds, nothing else
    f32x2 a = {1.0f + threadIdx.x, 2.0f}, b = {0.5f, 0.25f}, c = {3.0f, 1.0f}, d = {0.f, blockIdx.x * 1.0f};
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(c) : "v"(a));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d) : "v"(b));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b) : "v"(d));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a.x + a.y + c.x + c.y + d.x + d.y + b.x + b.y;
}

__global__ __launch_bounds__(256, 2) void k_mfma_hog(float* out, int iters) {     // bf16 MFMA + transposed LDS reads, 64 KB dynamic LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* sm = reinterpret_cast<uint4*>(smem);
    for (int i = threadIdx.x; i < 2048; i += 256) sm[i] = make_uint4(0x3f803f80u + i, 0x3f803f80u, 0x3c003c00u, 0x3f803f80u);
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
        union { uint4 u; bf16x8 v; s16x4 h[2]; } a, b;
        a.u = sm[(threadIdx.x + 7 * i) & 2047];
        b.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(smem + ((threadIdx.x * 8 + 64 * i) & 32760)));
        b.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(smem + ((threadIdx.x * 8 + 64 * i + 512) & 32760)));
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// nothing but bf16 MFMAs, registers named explicitly: accumulators v[2:65], A / B fragments at v[AB .. AB+15] ("mfma_lo": AB = 68, "mfma_hi": 172 --
// where the teacher GEMM's register allocation puts them)
#define MFMA4(a0, a1, b0, b1) "v_mfma_f32_32x32x16_bf16 v[50:65], v[" a0 "], v[" b0 "], v[50:65]\n v_mfma_f32_32x32x16_bf16 v[34:49], v[" a0 "], v[" b1 "], v[34:49]\n" \
                              "v_mfma_f32_32x32x16_bf16 v[18:33], v[" a1 "], v[" b0 "], v[18:33]\n v_mfma_f32_32x32x16_bf16 v[2:17], v[" a1 "], v[" b1 "], v[2:17]\n"
#define CLOB_ACC "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33", \
                 "v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65"
template <bool HI>
__global__ __launch_bounds__(256, 2) void k_mfma_asm(float* out, int iters) {
    for (int i = 0; i < iters; ++i) {
        if (HI) asm volatile(MFMA4("172:175", "176:179", "180:183", "184:187") MFMA4("188:191", "192:195", "196:199", "200:203") ::: CLOB_ACC,
                             "v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187",
                             "v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203");
        else asm volatile(MFMA4("68:71", "72:75", "76:79", "80:83") MFMA4("84:87", "88:91", "92:95", "96:99") ::: CLOB_ACC,
                          "v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83",
                          "v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99");
    }
    if (iters < 0) out[0] = 1.f;
}

// the skeleton of the teacher GEMM (its FQSS_TDIAG=30 build still triggers the effect): 128 x 128 x 32 tiles, 4 waves as 2 x 2, three bf16 planes
// per operand in 60 KB of static LDS, per k-tile 2 x (12 fragment reads + 24 MFMAs) between workgroup barriers, FOUR k-tiles per workgroup
__global__ __launch_bounds__(256, 2) void k_gemm_skel(float* out, int ktiles) {
    __shared__ __attribute__((aligned(16))) unsigned short As[3][128][40], Bs[3][32][160];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 3 * 128 * 40; i += 256) (&As[0][0][0])[i] = 0x3c00 + (i & 255);
    for (int i = tid; i < 3 * 32 * 160; i += 256) (&Bs[0][0][0])[i] = 0x3b80 + (i & 127);
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a >> 1][a & 1][r] = 0.f;
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[3][2], bfr[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[p][i] = *reinterpret_cast<const bf16x8*>(&As[p][wm * 64 + i * 32 + lr][ks * 16 + 8 * lh]);
                    union { bf16x8 v; s16x4 h[2]; } u;
                    u.h[0] = *reinterpret_cast<const s16x4*>(&Bs[p][ks * 16 + 8 * lh][wn * 64 + i * 32 + (lr & 28)]);
                    u.h[1] = *reinterpret_cast<const s16x4*>(&Bs[p][ks * 16 + 8 * lh + 4][wn * 64 + i * 32 + (lr & 28)]);
                    bfr[p][i] = u.v;
                }
            constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int sp = 0; sp < 6; ++sp)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a >> 1][a & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[IA[sp]][a >> 1], bfr[IB[sp]][a & 1], acc[a >> 1][a & 1], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < ktiles) { As[kt % 3][tid >> 1][(tid & 1) * 16] = (unsigned short)kt; __syncthreads(); }
    }
    if (acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3] == 12345.678f) out[0] = 1.0f;
}

typedef int (*tgemm_t)(const uint16_t*, const float*, int, int, int, int, int64_t, int, const double*, const float*, const float*, float, const float*,
                       const float*, int, const float*, double*, int, float*, const float*, int64_t, float*, const float*, int64_t, void*);
typedef void (*victim_t)(unsigned*, int);
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 10;
    const char* bg = argc > 2 ? argv[2] : "pkadd";
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    unsigned* bad; CK(hipMalloc(&bad, 20 * 4)); float* out; CK(hipMalloc(&out, (size_t)200000 * 256 * 4));
    CK(hipFuncSetAttribute((const void*)k_mfma_hog, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    tgemm_t tgemm = nullptr; uint16_t* planes = nullptr; float *x = nullptr, *c1 = nullptr, *bias = nullptr, *slope = nullptr; double* st = nullptr;
    const int B = 8, Ci = 128, Co = 512, M = 3999, ld = 4000;
    if (!strcmp(bg, "tgemm")) {
        void* lib = dlopen(argc > 3 ? argv[3] : "fqss_amd/csrc/libfqss_hip.so", RTLD_NOW);
        if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
        tgemm = (tgemm_t)dlsym(lib, "fqss_tgemm");
        CK(hipMalloc(&planes, 3 * Co * Ci * 2)); CK(hipMemset(planes, 0x3c, 3 * Co * Ci * 2));      // bf16 0x3c3c: a small finite value
        CK(hipMalloc(&x, (size_t)B * Ci * ld * 4)); CK(hipMemset(x, 0, (size_t)B * Ci * ld * 4));
        CK(hipMalloc(&c1, (size_t)B * Co * ld * 4)); CK(hipMalloc(&bias, Co * 4)); CK(hipMemset(bias, 0, Co * 4));
        CK(hipMalloc(&slope, 4)); CK(hipMemset(slope, 0, 4)); CK(hipMalloc(&st, B * 32 * 16 * 8)); CK(hipMemset(st, 0, B * 32 * 16 * 8));
    }
    if (!strcmp(bg, "rate")) {      // what bf16-MFMA rate do these loops sustain alone?  (docs/history/DESIGN_rounds_1-5.md 4: the practical matrix-pipe ceiling of this part)
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int which = 0; which < 3; ++which) {
            float ms = 0.f; double flop = 0.0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, s1));
                if (which == 0) { hipLaunchKernelGGL(k_mfma_hog, dim3(4096), dim3(256), 64 * 1024, s1, out, 3000); flop = 4096.0 * 4 * 3000 * 4 * 32768; }
                if (which == 1) { hipLaunchKernelGGL(k_gemm_skel, dim3(2016), dim3(256), 0, s1, out, 64); flop = 2016.0 * 4 * 64 * 48 * 32768; }
                if (which == 2) { hipLaunchKernelGGL(k_mfma_asm<false>, dim3(2048), dim3(256), 0, s1, out, 3000); flop = 2048.0 * 4 * 3000 * 8 * 32768; }
                CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("%s: %.3f ms, %.0f TFLOP/s bf16\n", which == 0 ? "k_mfma_hog (LDS-fed, 4 accumulators)" : which == 1 ? "k_gemm_skel (GEMM skeleton, 24 MFMA per 12 fragment reads)" : "k_mfma_asm (registers only)", ms, flop / ms * 1e-9);
        }
        return 0;
    }
    const victim_t victims[7] = {k_victim<0, 0, 3>, k_victim<0, 2, 1>, k_victim<0, 1, 2>, k_victim<0, 2, 3>, k_victim<0, 0, 1>, k_victim<1, 2, 1>, k_victim<2, 2, 1>};
    const char* names[7] = {"v_pk_add_f32 (default)", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]",
                            "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1]", "v_pk_add_f32 op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]",
                            "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]"};
    for (int v = 0; v < 7; ++v) {
        unsigned tot[2][20] = {}; unsigned fmin = 0xffffffffu, fmax = 0;
        for (int phase = 0; phase < 2; ++phase)              // phase 0: the victim alone; phase 1: next to the aggressor
            for (int r = 0; r < rounds; ++r) {
                CK(hipMemsetAsync(bad, 0, 20 * 4, s1)); CK(hipMemsetAsync(bad + 7, 0xff, 4, s1)); CK(hipStreamSynchronize(s1));
                if (phase == 1 && !strcmp(bg, "pkadd")) hipLaunchKernelGGL(k_pkadd_hog, dim3(4096), dim3(256), 0, s2, out, 20000);
                if (phase == 1 && !strcmp(bg, "mfma_lo")) hipLaunchKernelGGL(k_mfma_asm<false>, dim3(2048), dim3(256), 0, s2, out, 3000);
                if (phase == 1 && !strcmp(bg, "mfma_hi")) hipLaunchKernelGGL(k_mfma_asm<true>, dim3(2048), dim3(256), 0, s2, out, 3000);
                if (phase == 1 && !strcmp(bg, "skel")) for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(k_gemm_skel, dim3(2016), dim3(256), 0, s2, out, 4);
                if (phase == 1 && !strcmp(bg, "skel_long")) hipLaunchKernelGGL(k_gemm_skel, dim3(2016), dim3(256), 0, s2, out, 24);
                if (phase == 1 && !strcmp(bg, "mfma_short")) hipLaunchKernelGGL(k_mfma_hog, dim3(200000), dim3(256), 64 * 1024, s2, out, 8);
                if (phase == 1 && !strcmp(bg, "mfma")) hipLaunchKernelGGL(k_mfma_hog, dim3(4096), dim3(256), 64 * 1024, s2, out, 3000);
                if (phase == 1 && tgemm) for (int k = 0; k < 6; ++k)
                    if (tgemm(planes, x, B, Ci, Co, M, ld, 0, nullptr, nullptr, nullptr, 1e-8f, nullptr, bias, 2, slope, st, Co, c1, nullptr, ld, nullptr, nullptr, 0, s2)) return 1;
                hipLaunchKernelGGL(victims[v], dim3(1024), dim3(256), 0, s1, bad, 4000);
                CK(hipDeviceSynchronize());
                unsigned hb[20]; CK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
                for (int i = 0; i < 20; ++i) tot[phase][i] += hb[i];
                if (hb[0]) { fmin = hb[7] < fmin ? hb[7] : fmin; fmax = hb[8] > fmax ? hb[8] : fmax; }
            }
        printf("%-46s alone: %u | next to %s: %u mismatching halves in %u lanes-launches (lanes 0-15: %u, 16-31: %u, 32-47: %u, 48-63: %u); "
               "%u of them equal the DEFAULT-selection result\n", names[v], tot[0][0], bg, tot[1][0], tot[1][6], tot[1][1], tot[1][2], tot[1][3], tot[1][4], tot[1][5]);
        if (tot[1][0]) printf("      first bad step of a lane: min %u max %u of 4000; by eighth of the run: %u %u %u %u %u %u %u %u\n", fmin, fmax, tot[1][9], tot[1][10], tot[1][11],
                              tot[1][12], tot[1][13], tot[1][14], tot[1][15], tot[1][16]);
    }
    return 0;
}
