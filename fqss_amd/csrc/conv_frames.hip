// conv_frames.hip -- frame gather / overlap-add kernels of the general (strided, dilated, zero-padded) convolutions of the
// HTDemucs layers: Conv1d k8 s4 p2 and the dilated k3 convs of DConv (hdemucsq.py:72-162, demucsq.py:110-182),
// Conv2d (8,1)/(4,1) along frequency and the 3x3 `rewrite` convs of the decoder (hdemucsq.py:261-347), and their transposed forms.
//
// A convolution is   frames = gather(x)  ->  pointwise GEMM (the existing fqss_pwconv_* / q-GEMM kernels)  and a transposed
// convolution is   pointwise GEMM -> overlap-add(frames); each of the two data movements is the other's adjoint, so the four
// passes (conv fwd / bwd-data, convT fwd / bwd-data) need exactly these two kernels.  Both are pure HBM streams:
//   * k_frames_gather : one thread per FRAME element, consecutive lanes walk the innermost (time) axis of the output, so the
//     store is fully coalesced and the loads are unit-stride (stride s_w for a strided 1-D conv; the k/s-fold re-read of a
//     tap row comes from L2);
//   * k_frames_ola    : one thread per SIGNAL element gathers its <= ceil(kh/sh)*ceil(kw/sw) contributing frame elements in a
//     fixed order (tap row ascending, tap column ascending): the overlap-add is deterministic, no atomics.
#define FQSS_USES_GRAD_ADD   // the fp32 gradient atomics of this file go through grad_add (fqss_dev.h: FQSS_DETERMINISTIC=1)
#include <cstdlib>
#include "fqss_dev.h"

namespace fqss {

struct FrameGeom {
    int64_t B, C, H, W;          // the signal [B][C][H][W]
    int64_t sb, sc, sh_;         // its strides in elements (unit stride along W)
    int kh, kw, st_h, st_w, ph, pw, dh, dw;
    int64_t Ho, Wo, ld;          // frames [B][C*kh*kw][Ho*Wo], row stride ld
};

// grid: x = chunks of the frame row (m = ho*Wo + wo), y = frame rows (b, c, tap); everything but (ho, wo) is uniform per workgroup
// q = n / d, r = n % d for 0 <= n < 2^24 (host check) by a float reciprocal and one correction step -- the integer division by a
// run-time divisor is ~25 instructions per element in kernels that move one element per thread
__device__ __forceinline__ void div_small(int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; r -= d; }
}

__global__ __launch_bounds__(256) void k_frames_gather(const float* __restrict__ x, float* __restrict__ f, const FrameGeom g) {
    const int M = (int)(g.Ho * g.Wo), Wo = (int)g.Wo;
    const float inv_wo = 1.0f / (float)Wo;
    const int64_t rows = g.B * g.C * g.kh * g.kw;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        const int j = (int)(r % g.kw);
        const int64_t r1 = r / g.kw;
        const int ti = (int)(r1 % g.kh);
        const int64_t bc = r1 / g.kh, c = bc % g.C, b = bc / g.C;
        const float* xp = x + b * g.sb + c * g.sc;
        float* fp = f + r * g.ld;
        const int h0 = ti * g.dh - g.ph, w0 = j * g.dw - g.pw;
        for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
            int ho, wo;
            div_small(m, Wo, inv_wo, ho, wo);
            const int h = ho * g.st_h + h0, w = wo * g.st_w + w0;
            float v = 0.0f;
            if (h >= 0 && h < (int)g.H && w >= 0 && w < (int)g.W) v = xp[(int64_t)h * g.sh_ + w];
            fp[m] = v;
        }
    }
}

// ... four consecutive frame positions per thread and ONE 16-B store (frame rows are 16-B aligned, ld % 4 == 0: the positions past M
// land in the row's own padding, as zeros): one division per group while it stays inside a frame-grid row, and with unit column
// stride one (unaligned) 16-B load where the whole group lies inside the signal.  Round 6: the one-element form above moved 2.5 TB/s.
struct __attribute__((packed, aligned(4))) G4U { float x, y, z, w; };
__global__ __launch_bounds__(256) void k_frames_gather4(const float* __restrict__ x, float* __restrict__ f, const FrameGeom g) {
    const int M = (int)(g.Ho * g.Wo), Wo = (int)g.Wo, H = (int)g.H, W = (int)g.W;
    const float inv_wo = 1.0f / (float)Wo;
    const int64_t rows = g.B * g.C * g.kh * g.kw;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        const int j = (int)(r % g.kw);
        const int64_t r1 = r / g.kw;
        const int ti = (int)(r1 % g.kh);
        const int64_t bc = r1 / g.kh, c = bc % g.C, b = bc / g.C;
        const float* xp = x + b * g.sb + c * g.sc;
        float* fp = f + r * g.ld;
        const int h0 = ti * g.dh - g.ph, w0 = j * g.dw - g.pw;
        for (int m0 = (blockIdx.x * 256 + threadIdx.x) * 4; m0 < M; m0 += gridDim.x * 1024) {
            int ho, wo;
            div_small(m0, Wo, inv_wo, ho, wo);
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (wo + 3 < Wo && m0 + 3 < M) {          // the group inside one row of the frame grid
                const int h = ho * g.st_h + h0;
                if (h >= 0 && h < H) {
                    const float* xr = xp + (int64_t)h * g.sh_;
                    const int w = wo * g.st_w + w0;
                    if (g.st_w == 1 && w >= 0 && w + 3 < W) {
                        const G4U t = *reinterpret_cast<const G4U*>(xr + w);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int we = w + e * g.st_w;
                            if (we >= 0 && we < W) v[e] = xr[we];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (m0 + e < M) {
                        int he, we;
                        div_small(m0 + e, Wo, inv_wo, he, we);
                        const int h = he * g.st_h + h0, w = we * g.st_w + w0;
                        if (h >= 0 && h < H && w >= 0 && w < W) v[e] = xp[(int64_t)h * g.sh_ + w];
                    }
                }
            }
            *reinterpret_cast<float4*>(fp + m0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// y[b][c][h][w] = bias[c] + sum over taps (ti, j) with (h + ph - ti*dh) = ho*st_h, (w + pw - j*dw) = wo*st_w in range
// grid: x = chunks of the plane (h*W + w), y = planes (b, c)
__global__ __launch_bounds__(256) void k_frames_ola(const float* __restrict__ f, const float* __restrict__ bias, float* __restrict__ y,
                                                     const FrameGeom g) {
    const int HW = (int)(g.H * g.W), W = (int)g.W, Ho = (int)g.Ho, Wo = (int)g.Wo;
    const float inv_w = 1.0f / (float)W;
    // tap rows: with dh = 1 and a power-of-two row stride only ti = (h + ph) mod st_h, + st_h, ... reach row h (k8 s4: two of eight),
    // in the same ascending order -- no division per element and tap row
    const bool fast_h = g.dh == 1 && (g.st_h & (g.st_h - 1)) == 0;
    const int sh_shift = __builtin_ctz((unsigned)g.st_h), ti_step = fast_h ? g.st_h : 1;
    const int64_t planes = g.B * g.C;
    for (int64_t bc = blockIdx.y; bc < planes; bc += gridDim.y) {
        const int64_t c = bc % g.C, b = bc / g.C;
        const float bv = bias != nullptr ? bias[c] : 0.0f;
        const float* fb = f + bc * g.kh * g.kw * g.ld;
        float* yp = y + b * g.sb + c * g.sc;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            int h, w;
            div_small(i, W, inv_w, h, w);
            float acc = 0.0f;
            for (int ti = fast_h ? ((h + g.ph) & (g.st_h - 1)) : 0; ti < g.kh; ti += ti_step) {
                const int hn = h + g.ph - ti * g.dh;
                if (hn < 0) continue;
                const int ho = fast_h ? (hn >> sh_shift) : hn / g.st_h;
                if (ho * g.st_h != hn || ho >= Ho) continue;
                for (int j = 0; j < g.kw; ++j) {
                    const int wn = w + g.pw - j * g.dw;
                    if (wn < 0) continue;
                    const int wo = wn / g.st_w;
                    if (wo * g.st_w != wn || wo >= Wo) continue;
                    acc += fb[(int64_t)(ti * g.kw + j) * g.ld + ho * Wo + wo];
                }
            }
            if (bias != nullptr) acc += bv;
            yp[(int64_t)h * g.sh_ + w] = acc;
        }
    }
}

// k_frames_ola for kernels that are ONE COLUMN wide with unit column stride (the frequency-branch transposed convolutions of HTDemucs,
// (8, 1) / (4, 1): kw = 1, st_w = 1, pw = 0): four consecutive columns per thread -- the tap rows that reach row h are the same for all
// four, each contributes one (unaligned) 16-B load, the result is one 16-B store.  Same terms in the same order as k_frames_ola.
__global__ __launch_bounds__(256) void k_frames_ola_col4(const float* __restrict__ f, const float* __restrict__ bias, float* __restrict__ y,
                                                          const FrameGeom g) {
    const int HW = (int)(g.H * g.W), W = (int)g.W, Ho = (int)g.Ho, Wo = (int)g.Wo;
    const float inv_w = 1.0f / (float)W;
    const bool fast_h = g.dh == 1 && (g.st_h & (g.st_h - 1)) == 0;
    const int sh_shift = __builtin_ctz((unsigned)g.st_h), ti_step = fast_h ? g.st_h : 1;
    const int64_t planes = g.B * g.C;
    for (int64_t bc = blockIdx.y; bc < planes; bc += gridDim.y) {
        const int64_t c = bc % g.C, b = bc / g.C;
        const float bv = bias != nullptr ? bias[c] : 0.0f;
        const float* fb = f + bc * g.kh * g.ld;
        float* yp = y + b * g.sb + c * g.sc;
        for (int i0 = (blockIdx.x * 256 + threadIdx.x) * 4; i0 < HW; i0 += gridDim.x * 1024) {
            int h, w;
            div_small(i0, W, inv_w, h, w);
            const int n = min(4, min(W - w, HW - i0));      // columns of this group inside row h (the rest: the next group's)
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int ti = fast_h ? ((h + g.ph) & (g.st_h - 1)) : 0; ti < g.kh; ti += ti_step) {
                const int hn = h + g.ph - ti * g.dh;
                if (hn < 0) continue;
                const int ho = fast_h ? (hn >> sh_shift) : hn / g.st_h;
                if (ho * g.st_h != hn || ho >= Ho) continue;
                const float* fr = fb + (int64_t)ti * g.ld + (int64_t)ho * Wo + w;
                if (n == 4 && w + 3 < Wo) {
                    const G4U t = *reinterpret_cast<const G4U*>(fr);
                    acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < n && w + e < Wo) acc[e] += fr[e];
                }
            }
            float* yr = yp + (int64_t)h * g.sh_ + w;
            if (bias != nullptr) { acc[0] += bv; acc[1] += bv; acc[2] += bv; acc[3] += bv; }
            if (n == 4) {
                G4U o; o.x = acc[0]; o.y = acc[1]; o.z = acc[2]; o.w = acc[3];
                *reinterpret_cast<G4U*>(yr) = o;
            } else {
                for (int e = 0; e < n; ++e) yr[e] = acc[e];
                // the group straddles a row end: its remaining positions belong to row h + 1 -- done one by one, k_frames_ola's way
                for (int e = n; e < 4 && i0 + e < HW; ++e) {
                    int h2, w2;
                    div_small(i0 + e, W, inv_w, h2, w2);
                    float a = 0.0f;
                    for (int ti = fast_h ? ((h2 + g.ph) & (g.st_h - 1)) : 0; ti < g.kh; ti += ti_step) {
                        const int hn = h2 + g.ph - ti * g.dh;
                        if (hn < 0) continue;
                        const int ho = fast_h ? (hn >> sh_shift) : hn / g.st_h;
                        if (ho * g.st_h != hn || ho >= Ho) continue;
                        if (w2 < Wo) a += fb[(int64_t)ti * g.ld + (int64_t)ho * Wo + w2];
                    }
                    if (bias != nullptr) a += bv;
                    yp[(int64_t)h2 * g.sh_ + w2] = a;
                }
            }
        }
    }
}

// The same sums for signals of FEW LONG ROWS (the 1-D transposed convolutions: H = 1, W = 27 k .. 441 k), row by row: same terms in the
// same order (tap row, then tap column ascending) as k_frames_ola, bit for bit.  grid: x = chunks of a signal row (w), y = planes (b, c), z = slices of the rows (h):
// the tap rows that reach row h are found once per row; SW = the column stride when it is 1, 2 or 4 (0: any) so that the
// divisibility test of a column is a mask, and with dw = 1 only the taps j = (w + pw) mod SW, + SW, ... are visited (k8 s4: two of
// eight; the flattened form divides per element and tap: 146 -> 65 us at [4][48][110250]).  On planes of many short rows (the
// spectrogram, 431 columns) the flattened k_frames_ola is the faster one (dense lanes): measured 1.1-1.6x.
template <int SW>
__global__ __launch_bounds__(256) void k_frames_ola_rows(const float* __restrict__ f, const float* __restrict__ bias, float* __restrict__ y,
                                                     const FrameGeom g) {
    const int W = (int)g.W, H = (int)g.H, Ho = (int)g.Ho, Wo = (int)g.Wo;
    const int stw = SW ? SW : g.st_w;
    const int64_t planes = g.B * g.C;
    for (int64_t bc = blockIdx.y; bc < planes; bc += gridDim.y) {
        const int64_t c = bc % g.C, b = bc / g.C;
        const float bv = bias != nullptr ? bias[c] : 0.0f;
        const float* fb = f + bc * g.kh * g.kw * g.ld;
        float* yp = y + b * g.sb + c * g.sc;
        for (int h = blockIdx.z; h < H; h += gridDim.z) {
            float* yr = yp + (int64_t)h * g.sh_;
            for (int w = blockIdx.x * 256 + threadIdx.x; w < W; w += gridDim.x * 256) {
                float acc = 0.0f;
                for (int ti = 0; ti < g.kh; ++ti) {                       // (uniform per row)
                    const int hn = h + g.ph - ti * g.dh;
                    if (hn < 0) continue;
                    const int ho = hn / g.st_h;
                    if (ho * g.st_h != hn || ho >= Ho) continue;
                    const float* fr = fb + (int64_t)ti * g.kw * g.ld + (int64_t)ho * Wo;
                    if (SW && g.dw == 1) {
                        const int wp = w + g.pw;
                        for (int j = wp & (SW - 1); j < g.kw; j += SW) {
                            const int wn = wp - j;
                            if (wn < 0) break;                            // (larger j only make it more negative)
                            const int wo = SW == 1 ? wn : (SW == 2 ? wn >> 1 : wn >> 2);
                            if (wo < Wo) acc += fr[(int64_t)j * g.ld + wo];
                        }
                    } else {
                        for (int j = 0; j < g.kw; ++j) {
                            const int wn = w + g.pw - j * g.dw;
                            if (wn < 0) continue;
                            const int wo = wn / stw;
                            if (wo * stw != wn || wo >= Wo) continue;
                            acc += fr[(int64_t)j * g.ld + wo];
                        }
                    }
                }
                if (bias != nullptr) acc += bv;
                yr[w] = acc;
            }
        }
    }
}

// out[c] += sum_{b, m} g[b][c][m]  (bias gradient of a transposed convolution); one workgroup per (c, b)
__global__ __launch_bounds__(256) void k_chan_sum(const float* __restrict__ g, float* __restrict__ out, int64_t C, int64_t M, int64_t ld) {
    __shared__ float smem[4];
    const int64_t c = blockIdx.x, b = blockIdx.y;
    const float* row = g + (b * C + c) * ld;
    float v[1] = {0.0f};
    for (int64_t m = (int64_t)blockIdx.z * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.z * 256) v[0] += row[m];   // z: chunks of a long row
    block_sum<float, 1>(v, smem);
    if (threadIdx.x == 0) grad_add(out + c, v[0]);
}

// ... rows of 16-B aligned floats: four float4 requests in flight per thread (the one-element form reads 1.5 TB/s; the frame-path
// convolutions' bias gradients run over 0.3 GB tensors)
__global__ __launch_bounds__(256) void k_chan_sum4(const float* __restrict__ g, float* __restrict__ out, int64_t C, int64_t M, int64_t ld) {
    __shared__ float smem[4];
    const int64_t c = blockIdx.x, b = blockIdx.y;
    const float* row = g + (b * C + c) * ld;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int64_t step = (int64_t)gridDim.z * 1024;
    for (int64_t m0 = ((int64_t)blockIdx.z * 256 + threadIdx.x) * 4; m0 < M; m0 += 4 * step) {
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + i * step;
            v[i] = (m < M) ? *reinterpret_cast<const float4*>(row + m) : make_float4(0.f, 0.f, 0.f, 0.f);      // (ld % 4 == 0: inside the row's stride)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + i * step;
            a0 += v[i].x;
            a1 += (m + 1 < M) ? v[i].y : 0.f;
            a2 += (m + 2 < M) ? v[i].z : 0.f;
            a3 += (m + 3 < M) ? v[i].w : 0.f;
        }
    }
    float v1[1] = {(a0 + a1) + (a2 + a3)};
    block_sum<float, 1>(v1, smem);
    if (threadIdx.x == 0) grad_add(out + c, v1[0]);
}

// ... MANY short rows per channel ([B F = 2048][C][431]: the DConv layers on a spectrogram's rows): a workgroup walks a slice of the
// batch entries of its channel and issues ONE add -- a workgroup per (channel, batch entry) put thousands of adds on each of C addresses
// (0.8 ms for [2048][96][431])
__global__ __launch_bounds__(256) void k_chan_sum_b(const float* __restrict__ g, float* __restrict__ out, int64_t B, int64_t C, int64_t M, int64_t ld,
                                                     int vec) {
    __shared__ float smem[4];
    const int64_t c = blockIdx.x;
    float acc = 0.0f;
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        const float* row = g + (b * C + c) * ld;
        if (vec) {
            for (int64_t m = (int64_t)threadIdx.x * 4; m < M; m += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(row + m);
                acc += v.x;
                acc += (m + 1 < M) ? v.y : 0.f;
                acc += (m + 2 < M) ? v.z : 0.f;
                acc += (m + 3 < M) ? v.w : 0.f;
            }
        } else {
            for (int64_t m = threadIdx.x; m < M; m += 256) acc += row[m];
        }
    }
    float v1[1] = {acc};
    block_sum<float, 1>(v1, smem);
    if (threadIdx.x == 0) grad_add(out + c, v1[0]);
}

static int check_geom(const FrameGeom& g) {
    FQSS_REQUIRE(g.B > 0 && g.C > 0 && g.H > 0 && g.W > 0, "empty signal");
    FQSS_REQUIRE(g.kh >= 1 && g.kw >= 1 && g.st_h >= 1 && g.st_w >= 1 && g.dh >= 1 && g.dw >= 1 && g.ph >= 0 && g.pw >= 0, "bad geometry");
    // (The frame grid is the caller's: for a convolution it is the geometry's own, Ho = (H + 2 ph - dh (kh - 1) - 1) / st_h + 1; when the
    // signal is a WINDOW of a transposed convolution's output -- cut at the back, padded at the front -- it is that convolution's input
    // grid.  Both kernels bound every index on both sides: the gather reads zeros outside the signal, the overlap-add drops what lands
    // there, frames outside the grid are neither produced nor read.)
    FQSS_REQUIRE(g.H + 2 * g.ph >= (int64_t)g.dh * (g.kh - 1) + 1 && g.W + 2 * g.pw >= (int64_t)g.dw * (g.kw - 1) + 1, "signal shorter than the kernel");
    FQSS_REQUIRE(g.Ho >= 1 && g.Wo >= 1, "empty frame grid");
    FQSS_REQUIRE(g.Ho <= (g.H + 2 * g.ph - 1) / g.st_h + 1 + g.kh && g.Wo <= (g.W + 2 * g.pw - 1) / g.st_w + 1 + g.kw, "frame grid far past the signal");
    FQSS_REQUIRE(g.ld >= g.Ho * g.Wo && g.sh_ >= g.W && g.sc >= g.sh_ * (g.H - 1) + g.W && g.sb >= g.sc * (g.C - 1) + g.W, "bad strides");
    return FQSS_OK;
}

// x covers a row / plane in chunks of 1024 positions (4 per thread), y walks the rows; ~16 k workgroups at most
static inline dim3 plane_grid(int64_t positions, int64_t rows) {
    int64_t gx = cdiv(positions, 1024);
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    int64_t gy = 16384 / gx;
    if (gy < 1) gy = 1;
    if (gy > rows) gy = rows;
    if (gy > 65535) gy = 65535;
    return dim3((unsigned)gx, (unsigned)gy, 1);
}

// x covers a row of `width` positions in chunks of `chunk`, y walks `rows` (planes / frame rows), z the `depth` rows of a plane:
// ~16 k workgroups at most, at least ~2 k when the problem has them
static inline dim3 row_grid(int64_t width, int64_t chunk, int64_t rows, int64_t depth) {
    int64_t gx = cdiv(width, chunk);
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    int64_t gy = 16384 / gx;
    if (gy < 1) gy = 1;
    if (gy > rows) gy = rows;
    if (gy > 65535) gy = 65535;
    int64_t gz = 16384 / (gx * gy);
    if (gz < 1) gz = 1;
    if (gz > depth) gz = depth;
    if (gz > 65535) gz = 65535;
    return dim3((unsigned)gx, (unsigned)gy, (unsigned)gz);
}

static inline unsigned stream_grid(int64_t n) {
    int64_t b = cdiv(n, 256 * 4);
    if (b < 1) b = 1;
    if (b > 65536) b = 65536;
    return (unsigned)b;
}

}  // namespace fqss

using namespace fqss;

#define FQSS_GEOM_ARGS                                                                                                               \
    int64_t B, int64_t C, int64_t H, int64_t W, int64_t sb, int64_t sc, int64_t sh, int kh, int kw, int st_h, int st_w, int ph, int pw, \
        int dh, int dw, int64_t Ho, int64_t Wo, int64_t ld
#define FQSS_GEOM_INIT \
    FrameGeom g{B, C, H, W, sb, sc, sh, kh, kw, st_h, st_w, ph, pw, dh, dw, Ho, Wo, ld}

extern "C" int fqss_frames_gather(const float* x, float* frames, FQSS_GEOM_ARGS, fqss_stream_t stream) {
    FQSS_REQUIRE(x && frames, "null pointer");
    FQSS_GEOM_INIT;
    if (int rc = check_geom(g)) return rc;
    FQSS_REQUIRE(Ho * Wo < (1ll << 24) && H * W < (1ll << 31), "frame plane too large (positions are split by a float reciprocal: < 2^24)");
    static const bool one = getenv("FQSS_GATHER_V1") != nullptr;       // (A/B knob: the one-element kernel)
    if (!one && ld % 4 == 0 && aligned16(frames))
        hipLaunchKernelGGL(k_frames_gather4, plane_grid(Ho * Wo, B * C * kh * kw), dim3(256), 0, (hipStream_t)stream, x, frames, g);
    else
        hipLaunchKernelGGL(k_frames_gather, plane_grid(Ho * Wo, B * C * kh * kw), dim3(256), 0, (hipStream_t)stream, x, frames, g);
    return launch_status("fqss_frames_gather");
}

extern "C" int fqss_frames_ola(const float* frames, const float* bias, float* y, FQSS_GEOM_ARGS, fqss_stream_t stream) {
    FQSS_REQUIRE(y && frames, "null pointer");
    FQSS_GEOM_INIT;
    if (int rc = check_geom(g)) return rc;
    FQSS_REQUIRE(Ho * Wo < (1ll << 31) && H * W < (1ll << 31), "plane too large for 32-bit position arithmetic");
    FQSS_REQUIRE(H == 1 || H * W < (1ll << 24), "signal plane too large (positions are split by a float reciprocal: < 2^24)");
    if (H == 1 && W >= 4096) {          // few long rows: the row form (no division per element and tap)
        const dim3 grid = row_grid(W, 256, B * C, H);
        if (st_w == 1) hipLaunchKernelGGL(k_frames_ola_rows<1>, grid, dim3(256), 0, (hipStream_t)stream, frames, bias, y, g);
        else if (st_w == 2) hipLaunchKernelGGL(k_frames_ola_rows<2>, grid, dim3(256), 0, (hipStream_t)stream, frames, bias, y, g);
        else if (st_w == 4) hipLaunchKernelGGL(k_frames_ola_rows<4>, grid, dim3(256), 0, (hipStream_t)stream, frames, bias, y, g);
        else hipLaunchKernelGGL(k_frames_ola_rows<0>, grid, dim3(256), 0, (hipStream_t)stream, frames, bias, y, g);
        return launch_status("fqss_frames_ola");
    }
    static const bool one = getenv("FQSS_OLA_V1") != nullptr;       // (A/B knob: the one-element kernel)
    if (!one && kw == 1 && st_w == 1 && pw == 0 && W >= 8)
        hipLaunchKernelGGL(k_frames_ola_col4, plane_grid(H * W, B * C), dim3(256), 0, (hipStream_t)stream, frames, bias, y, g);
    else
        hipLaunchKernelGGL(k_frames_ola, plane_grid(H * W, B * C), dim3(256), 0, (hipStream_t)stream, frames, bias, y, g);
    return launch_status("fqss_frames_ola");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Halo-packed signal of a stride-1 convolution (round 6): xp[b][c] = one plane of `plane` floats = Hp rows of Wp floats, the signal's
// element (h, w) at row h + ph, column w + pw, zeros everywhere else (the halo IS the convolution's zero padding, the tail of the plane
// slack for the last row's taps).  On this image every tap of a stride-1 convolution is a constant shift of the flat plane, so the
// GEMM kernels read their B operand straight from it (k_qgemm<.., IMP>, k_gemm_x3<.., IMP>) and the frame image -- kh kw times the
// signal, written by fqss_frames_gather and read back -- is never made; the pack moves the signal once.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halo_pack(const float* __restrict__ x, float* __restrict__ xp, int64_t planes, int C, int H, int W,
                                                    int64_t sb, int64_t sc, int64_t sh_, int ph, int pw, int Wp, int64_t plane) {
    const int groups = (int)(plane >> 2);
    const float inv_wp = 1.0f / (float)Wp;
    for (int64_t bc = blockIdx.y; bc < planes; bc += gridDim.y) {
        const int64_t b = bc / C, c = bc - b * C;
        const float* xb = x + b * sb + c * sc;
        float* op = xp + bc * plane;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < groups; i += gridDim.x * 256) {
            int hp, wp;
            div_small(4 * i, Wp, inv_wp, hp, wp);       // (Wp % 4 == 0: a group never straddles rows)
            const int h = hp - ph;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (h >= 0 && h < H) {
                const float* xr = xb + (int64_t)h * sh_;
                const int w0 = wp - pw;
                if (w0 >= 0 && w0 + 3 < W) {          // (one possibly unaligned 16-B request)
                    const G4U t = *reinterpret_cast<const G4U*>(xr + w0);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int w = w0 + q;
                        if (w >= 0 && w < W) v[q] = xr[w];
                    }
                }
            }
            *reinterpret_cast<float4*>(op + 4 * (int64_t)i) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

extern "C" int fqss_halo_pack(const float* x, float* xp, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sb, int64_t sc, int64_t sh,
                              int ph, int pw, int64_t Wp, int64_t plane, fqss_stream_t stream) {
    FQSS_REQUIRE(x && xp, "null pointer");
    FQSS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && ph >= 0 && pw >= 0, "bad shape");
    FQSS_REQUIRE(Wp % 4 == 0 && Wp >= W + 2 * pw && plane % 4 == 0 && plane >= (H + 2 * ph) * Wp && plane < (1ll << 24) && aligned16(xp),
                 "packed planes: rows of Wp % 4 == 0 floats, plane % 4 == 0 and < 2^24");
    FQSS_REQUIRE(sh >= W && sc >= sh * (H - 1) + W && (B == 1 || sb >= sc * (C - 1) + W), "bad strides");
    hipLaunchKernelGGL(k_halo_pack, plane_grid(plane, B * C), dim3(256), 0, (hipStream_t)stream, x, xp, B * C, (int)C, (int)H, (int)W, sb, sc, sh, ph, pw,
                       (int)Wp, plane);
    return launch_status("fqss_halo_pack");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Phase-packed signal of a STRIDED convolution along one axis (round 6; kernel k = T s taps, stride s, zero padding p, dilation 1 -- the
// k8 s4 p2 encoder / decoder layers of HTDemucs, hdemucsq.py:72-162, 261-347).  With t - p = s q + r the convolution
//   z[n] = sum_t W[t] x[s n + t - p]   is   sum_r sum_{q'} W[t0(r) + s q'] Y_r[n + q'],   Y_r[m] = x[s (m + q0(r)) + r],
// t0(r) = (r + p) mod s the first tap of phase r and q0(r) = (t0(r) - p - r) / s its offset: a stride-1 convolution with T = k / s taps
// over s C phase channels.  k_phase_pack writes the planes xp[b][c s + r] = Y_r (rows of Wp floats, zeros outside the signal) once --
// the frame image is k / s times that -- and the implicit GEMMs of fqss_conv2_* run on them; k_phase_unpack is the inverse map (every
// signal position lies in exactly one phase plane): the data gradient's way back, and the transposed convolution's way out.
//   axis 0: along H ([B][C][H][W] with (k, 1) kernels: planes of Hy rows x Wp, Hy = Ho + T - 1);  axis 1: along W (H = 1: one row)
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int phase_q0(int r, int s, int p) { return ((r + p) % s - p - r) / s; }      // (exact: the numerator is a multiple of s)

__global__ __launch_bounds__(256) void k_phase_pack(const float* __restrict__ x, float* __restrict__ xp, int64_t planes, int C, int H, int W,
                                                     int64_t sb, int64_t sc, int64_t sh_, int axis, int s, int p, int Wp, int64_t plane) {
    const int groups = (int)(plane >> 2);
    const float inv_wp = 1.0f / (float)Wp;
    for (int64_t pl = blockIdx.y; pl < planes; pl += gridDim.y) {
        const int r = (int)(pl % s);
        const int64_t bc = pl / s, b = bc / C, c = bc - b * C;
        const int q0 = phase_q0(r, s, p);
        const float* xb = x + b * sb + c * sc;
        float* op = xp + pl * plane;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < groups; i += gridDim.x * 256) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (axis == 0) {
                int m, w;
                div_small(4 * i, Wp, inv_wp, m, w);
                const int h = s * (m + q0) + r;
                if (h >= 0 && h < H) {
                    const float* xr = xb + (int64_t)h * sh_ + w;
                    if (w + 3 < W) {
                        const G4U t = *reinterpret_cast<const G4U*>(xr);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (w + e < W) v[e] = xr[e];
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t col = (int64_t)s * (4 * i + e + q0) + r;
                    if (col >= 0 && col < W) v[e] = xb[col];
                }
            }
            *reinterpret_cast<float4*>(op + 4 * (int64_t)i) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// gx[b][c][h][w] (a window starting `off` positions into the strided axis of the full signal) = the phase planes' element of that position
__global__ __launch_bounds__(256) void k_phase_unpack(const float* __restrict__ gy, float* __restrict__ gx, int64_t planes, int C, int H, int W,
                                                       int64_t sb, int64_t sc, int64_t sh_, int axis, int s, int p, int Hy, int Wp, int64_t plane,
                                                       int off, const float* __restrict__ bias) {
    const int HW = H * W;
    const float inv_w = 1.0f / (float)W;
    for (int64_t bc = blockIdx.y; bc < planes; bc += gridDim.y) {
        const int64_t b = bc / C, c = bc - b * C;
        const float* yb = gy + bc * s * plane;
        float* ob = gx + b * sb + c * sc;
        const float bv = bias != nullptr ? bias[c] : 0.0f;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            int h, w;
            div_small(i, W, inv_w, h, w);
            const int pos = (axis == 0 ? h : w) + off;
            const int r = pos % s, m = pos / s - phase_q0(r, s, p);
            float v = 0.0f;
            if (axis == 0) {
                if (m >= 0 && m < Hy) v = yb[(int64_t)r * plane + (int64_t)m * Wp + w];
            } else {
                if (m >= 0 && m < Hy) v = yb[(int64_t)r * plane + m];      // (axis 1: Hy = the phase rows' live length, Lo + T - 1)
            }
            ob[(int64_t)h * sh_ + w] = v + bv;
        }
    }
}

// ... four consecutive columns per thread.  axis 0: the four share their phase plane and row -- one 16-B load (rows of Wp % 4 == 0 floats,
// w % 4 == 0), one (unaligned) 16-B store; axis 1 with s = 4: the four are the four phases of one group -- four loads, one 16-B store.
__global__ __launch_bounds__(256) void k_phase_unpack4(const float* __restrict__ gy, float* __restrict__ gx, int64_t planes, int C, int H, int W,
                                                        int64_t sb, int64_t sc, int64_t sh_, int axis, int s, int p, int Hy, int Wp, int64_t plane,
                                                        int off, const float* __restrict__ bias) {
    const int groups_w = (W + 3) >> 2, total = H * groups_w;
    const float inv_gw = 1.0f / (float)groups_w;
    for (int64_t bc = blockIdx.y; bc < planes; bc += gridDim.y) {
        const int64_t b = bc / C, c = bc - b * C;
        const float* yb = gy + bc * s * plane;
        float* ob = gx + b * sb + c * sc;
        const float bv = bias != nullptr ? bias[c] : 0.0f;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
            int h, gw;
            div_small(i, groups_w, inv_gw, h, gw);
            const int w = 4 * gw;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (axis == 0) {
                const int pos = h + off, r = pos % s, m = pos / s - phase_q0(r, s, p);
                if (m >= 0 && m < Hy) {
                    const float4 t = *reinterpret_cast<const float4*>(yb + (int64_t)r * plane + (int64_t)m * Wp + w);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int pos = w + e + off, r = pos % s, m = pos / s - phase_q0(r, s, p);
                    if (m >= 0 && m < Hy) v[e] = yb[(int64_t)r * plane + m];
                }
            }
            float* o = ob + (int64_t)h * sh_ + w;
            if (w + 3 < W) {
                G4U t; t.x = v[0] + bv; t.y = v[1] + bv; t.z = v[2] + bv; t.w = v[3] + bv;
                *reinterpret_cast<G4U*>(o) = t;
            } else {
                for (int e = 0; e < 4 && w + e < W; ++e) o[e] = v[e] + bv;
            }
        }
    }
}

extern "C" int fqss_phase_pack(const float* x, float* xp, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sb, int64_t sc, int64_t sh, int axis,
                               int s, int p, int64_t Wp, int64_t plane, fqss_stream_t stream) {
    FQSS_REQUIRE(x && xp, "null pointer");
    FQSS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (axis == 0 || (axis == 1 && H == 1)) && s >= 1 && p >= 0, "bad shape");
    FQSS_REQUIRE(Wp % 4 == 0 && plane % 4 == 0 && plane >= Wp && plane < (1ll << 24) && aligned16(xp) && (axis == 1 || Wp >= W), "packed planes");
    FQSS_REQUIRE(sh >= W && sc >= sh * (H - 1) + W && (B == 1 || sb >= sc * (C - 1) + W) && W < (1ll << 30), "bad strides");
    hipLaunchKernelGGL(k_phase_pack, plane_grid(plane, B * C * s), dim3(256), 0, (hipStream_t)stream, x, xp, B * C * s, (int)C, (int)H, (int)W, sb, sc, sh,
                       axis, s, p, (int)Wp, plane);
    return launch_status("fqss_phase_pack");
}

/* the inverse move: gx [B][C][H][W] (strides sb, sc, sh; a window that starts `off` positions into the strided axis) from the phase planes
 * gy [B][C s][plane] (Hy rows of Wp floats); bias [C] (nullable) is added: the tail of a transposed convolution */
extern "C" int fqss_phase_unpack(const float* gy, float* gx, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sb, int64_t sc, int64_t sh, int axis,
                                 int s, int p, int64_t Hy, int64_t Wp, int64_t plane, int off, const float* bias, fqss_stream_t stream) {
    FQSS_REQUIRE(gy && gx, "null pointer");
    FQSS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (axis == 0 || (axis == 1 && H == 1)) && s >= 1 && p >= 0 && off >= 0, "bad shape");
    FQSS_REQUIRE(Wp % 4 == 0 && (axis == 0 ? plane >= Hy * Wp : (Hy <= Wp && plane >= Wp)) && H * W < (1ll << 24), "packed planes");
    FQSS_REQUIRE(sh >= W && sc >= sh * (H - 1) + W && (B == 1 || sb >= sc * (C - 1) + W), "bad strides");
    static const bool one = getenv("FQSS_UNPACK_V1") != nullptr;       // (A/B knob: the one-element kernel)
    if (!one && W >= 8)
        hipLaunchKernelGGL(k_phase_unpack4, plane_grid(H * ((W + 3) / 4) * 4, B * C), dim3(256), 0, (hipStream_t)stream, gy, gx, B * C, (int)C, (int)H, (int)W,
                           sb, sc, sh, axis, s, p, (int)Hy, (int)Wp, plane, off, bias);
    else
        hipLaunchKernelGGL(k_phase_unpack, plane_grid(H * W, B * C), dim3(256), 0, (hipStream_t)stream, gy, gx, B * C, (int)C, (int)H, (int)W, sb, sc, sh, axis,
                           s, p, (int)Hy, (int)Wp, plane, off, bias);
    return launch_status("fqss_phase_unpack");
}

extern "C" int fqss_chan_sum(const float* g, float* out, int64_t B, int64_t C, int64_t M, int64_t ld, fqss_stream_t stream) {
    FQSS_REQUIRE(g && out, "null pointer");
    FQSS_REQUIRE(B > 0 && C > 0 && M > 0 && ld >= M && B <= 65535, "bad shape");
    int64_t gz = cdiv(M, 16384);          // few channels x few samples must still fill 256 CUs
    const int64_t want = cdiv(2048, C * B);
    if (gz > want) gz = want;
    if (gz < 1) gz = 1;
    if (gz > 1024) gz = 1024;
    if (B * C > 4096 && M <= 16384) {       // many short rows: slices of the batch per workgroup, ~2048 workgroups
        int64_t nb = 2048 / C;
        if (nb < 1) nb = 1;
        if (nb > B) nb = B;
        hipLaunchKernelGGL(k_chan_sum_b, dim3((unsigned)C, (unsigned)nb), dim3(256), 0, (hipStream_t)stream, g, out, B, C, M, ld,
                           (ld % 4 == 0 && aligned16(g)) ? 1 : 0);
        return launch_status("fqss_chan_sum");
    }
    if (ld % 4 == 0 && aligned16(g) && M >= 4096)
        hipLaunchKernelGGL(k_chan_sum4, dim3((unsigned)C, (unsigned)B, (unsigned)gz), dim3(256), 0, (hipStream_t)stream, g, out, C, M, ld);
    else
        hipLaunchKernelGGL(k_chan_sum, dim3((unsigned)C, (unsigned)B, (unsigned)gz), dim3(256), 0, (hipStream_t)stream, g, out, C, M, ld);
    return launch_status("fqss_chan_sum");
}
