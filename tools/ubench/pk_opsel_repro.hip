// gfx950 (MI355X, ROCm 7.2): a packed fp32 VALU op whose LOW result takes the HIGH half of its second source (op_sel[1] = 1), e.g.
//     v_pk_add_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0]          (D.lo = A.lo + B.hi, D.hi = A.hi + B.lo)
// returns a wrong value in lanes 48-63 -- a few times per million wave-instructions -- while ANOTHER stream runs a bf16-MFMA GEMM loop on
// the same CUs.  Alone it is exact; the default operand selection is exact next to the same aggressor.  Self-contained:
//   hipcc --offload-arch=gfx950 -O3 -o pk_opsel_repro pk_opsel_repro.hip && ./pk_opsel_repro          (exit code 1 = reproduced)
// Victim: one packed add per step per lane (inline asm), checked bit for bit against two scalar v_add_f32 of the same operands; no LDS,
// no atomics before the end.  Aggressor: the skeleton of a 128 x 128 x 32-tiled bf16 GEMM (LDS fragment reads + 48 MFMAs per k-tile
// between workgroup barriers, 4 k-tiles per workgroup); it touches no memory the victim uses.  hipcc's SLP vectorizer emits these forms
// for code like `acc0 += t1; acc1 += t0;` -- build with -fno-slp-vectorize (see fqss_amd/csrc/Makefile, docs/history/DESIGN_rounds_1-5.md 9).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <bool SWAP>   // false: default halves (control); true: op_sel:[0,1] op_sel_hi:[1,0]
__global__ __launch_bounds__(256) void k_victim(unsigned* bad, int iters) {
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, nbad = 0;
    f32x2 acc = {0.25f, -0.5f};
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        f32x2 t = {__uint_as_float(0x3f800000u | (h >> 9)) - 1.5f, __uint_as_float(0x3f800000u | ((h * 2246822519u) >> 9)) - 1.5f}, p;
        if (SWAP) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(acc), "v"(t));
        else asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(acc), "v"(t));
        const float b0 = SWAP ? t.y : t.x, b1 = SWAP ? t.x : t.y;
        float e0, e1;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(acc.x), "v"(b0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(acc.y), "v"(b1));
        nbad += (__float_as_uint(p.x) != __float_as_uint(e0)) + (__float_as_uint(p.y) != __float_as_uint(e1));
        acc.x = e0 * 0.5f;
        acc.y = e1 * 0.5f - 0.125f;
    }
    if (nbad) { atomicAdd(&bad[0], nbad); atomicAdd(&bad[1 + (threadIdx.x & 63) / 16], nbad); }
}

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

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    unsigned* bad; CK(hipMalloc(&bad, 8 * 4)); float* out; CK(hipMalloc(&out, 4));
    unsigned tot[2][2][5] = {};      // [form][alone | next to the GEMM][total, four quarters of the wave]
    for (int form = 0; form < 2; ++form)
        for (int beside = 0; beside < 2; ++beside)
            for (int r = 0; r < rounds; ++r) {
                CK(hipMemsetAsync(bad, 0, 8 * 4, s1)); CK(hipStreamSynchronize(s1));
                if (beside) hipLaunchKernelGGL(k_gemm_skel, dim3(2016), dim3(256), 0, s2, out, 24);
                if (form) hipLaunchKernelGGL(k_victim<true>, dim3(1024), dim3(256), 0, s1, bad, 4000);
                else hipLaunchKernelGGL(k_victim<false>, dim3(1024), dim3(256), 0, s1, bad, 4000);
                CK(hipDeviceSynchronize());
                unsigned hb[5]; CK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
                for (int i = 0; i < 5; ++i) tot[form][beside][i] += hb[i];
            }
    const char* names[2] = {"v_pk_add_f32 (default halves)", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]"};
    for (int form = 0; form < 2; ++form)
        printf("%-44s wrong halves alone: %u | next to the bf16-MFMA GEMM: %u (lanes 0-15: %u, 16-31: %u, 32-47: %u, 48-63: %u) of %.1e\n", names[form],
               tot[form][0][0], tot[form][1][0], tot[form][1][1], tot[form][1][2], tot[form][1][3], tot[form][1][4], 2.0 * rounds * 1024 * 256 * 4000);
    return tot[1][1][0] ? 1 : 0;
}
