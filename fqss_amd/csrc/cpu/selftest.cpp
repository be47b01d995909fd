// Driver of the sanitizer build (make asan): every entry point of fqss_cpu.cpp on small RAGGED shapes (row strides > columns, odd sizes,
// empty inputs), buffers sized exactly, a few identities checked.  AddressSanitizer / UBSan do the rest.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fqss.h"
#define CHECK(x) do { if (!(x)) { printf("FAILED: %s (line %d): %s\n", #x, __LINE__, fqss_last_error()); return 1; } } while (0)
static std::vector<float> rnd(size_t n, float s = 1.0f) {
    std::vector<float> v(n);
    for (auto& x : v) x = s * ((float)rand() / RAND_MAX - 0.5f);
    return v;
}
int main() {
    const int B = 2, C = 5, M = 37, ld = 40, Co = 7;
    auto x = rnd((size_t)B * C * ld), g = rnd((size_t)B * C * ld);
    std::vector<float> y((size_t)B * C * ld), gz((size_t)B * C * ld);
    std::vector<uint8_t> idx((size_t)B * C * 48);
    float lo = -0.4f, hi = 0.45f, slope = 0.25f;
    uint32_t obs[2];
    CHECK(fqss_obs_reset(obs, 1, nullptr) == 0);
    CHECK(fqss_actq_fwd(x.data(), y.data(), nullptr, B * C, M, ld, ld, 0, FQSS_ACT_PRELU, &slope, FQSS_Q_OBSERVE, nullptr, nullptr, obs, nullptr) == 0);
    CHECK(fqss_observer_ema(&lo, &hi, obs, 0.9, nullptr) == 0);
    CHECK(fqss_actq_fwd(x.data(), y.data(), idx.data(), B * C, M, ld, ld, 48, FQSS_ACT_RELU, nullptr, FQSS_Q_QUANT, &lo, &hi, nullptr, nullptr) == 0);
    std::vector<float> y2(y.size());
    CHECK(fqss_actq_fwd(y.data(), y2.data(), nullptr, B * C, M, ld, ld, 0, FQSS_ACT_NONE, nullptr, FQSS_Q_QUANT, &lo, &hi, nullptr, nullptr) == 0);
    for (int r = 0; r < B * C; ++r)
        for (int m = 0; m < M; ++m) CHECK(y[r * ld + m] == y2[r * ld + m]);      // the quantizer is idempotent on its own grid
    std::vector<double> gacc(FQSS_GACC_SLOTS * 3, 0.0);
    std::vector<float> gb(C, 0.f);
    float gmin = 0, gmax = 0, gsl = 0;
    CHECK(fqss_actq_bwd(x.data(), g.data(), gz.data(), B * C, M, ld, ld, ld, FQSS_ACT_PRELU, &slope, FQSS_Q_QUANT, &lo, &hi, gacc.data(), gb.data(), C, nullptr) == 0);
    CHECK(fqss_gacc_flush(gacc.data(), &gmin, &gmax, &gsl, nullptr) == 0);
    CHECK(fqss_actq_fwd(nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr) == 0);      // empty input
    CHECK(fqss_minmax(x.data(), B * C, M, ld, obs, nullptr) == 0);
    // weight quantizer, both channel axes
    auto w = rnd((size_t)Co * C * 3);
    std::vector<float> wq(w.size()), gw(w.size()), qlo(C), qhi(C), glo(C), ghi(C);
    std::vector<int8_t> wi(w.size());
    CHECK(fqss_wq_observe(w.data(), Co, C, 3, qlo.data(), qhi.data(), nullptr) == 0);
    CHECK(fqss_wq_fwd(w.data(), wq.data(), wi.data(), Co, C, 3, qlo.data(), qhi.data(), nullptr) == 0);
    CHECK(fqss_wq_bwd(w.data(), wq.data(), gw.data(), glo.data(), ghi.data(), Co, C, 3, qlo.data(), qhi.data(), 0, nullptr) == 0);
    // pointwise conv and its gradients; <gz, W x> == <W^T gz, x>
    auto pw = rnd((size_t)Co * C), bias = rnd(Co), gzc = rnd((size_t)B * Co * ld);
    std::vector<float> z((size_t)B * Co * ld), gx((size_t)B * C * ld), gpw((size_t)Co * C, 0.f);
    CHECK(fqss_pwconv_fwd_x3(x.data(), pw.data(), nullptr, z.data(), B, C, Co, M, ld, ld, nullptr) == 0);
    CHECK(fqss_pwconv_bwd_x(gzc.data(), pw.data(), gx.data(), B, C, Co, M, ld, ld, nullptr) == 0);
    CHECK(fqss_pwconv_bwd_w(gzc.data(), x.data(), gpw.data(), B, C, Co, M, ld, ld, nullptr) == 0);
    double a = 0, b2 = 0;
    for (int b = 0; b < B; ++b) {
        for (int co = 0; co < Co; ++co) for (int m = 0; m < M; ++m) a += (double)gzc[(b * Co + co) * ld + m] * z[(b * Co + co) * ld + m];
        for (int ci = 0; ci < C; ++ci) for (int m = 0; m < M; ++m) b2 += (double)gx[(b * C + ci) * ld + m] * x[(b * C + ci) * ld + m];
    }
    CHECK(fabs(a - b2) <= 1e-4 * (fabs(a) + 1.0));
    CHECK(fqss_pwconv_fwd(x.data(), pw.data(), bias.data(), z.data(), B, C, Co, M, ld, ld, nullptr) == 0);
    // depthwise conv, GroupNorm
    auto dw = rnd((size_t)C * 3), db = rnd(C);
    std::vector<float> gdw((size_t)C * 3, 0.f), mr(2 * B), gg(C, 0.f), gbt(C, 0.f);
    CHECK(fqss_dwconv_fwd(x.data(), dw.data(), db.data(), y.data(), B, C, M, 3, 4, 4, ld, ld, nullptr) == 0);
    CHECK(fqss_dwconv_bwd_x(g.data(), dw.data(), gz.data(), B, C, M, 3, 4, 4, ld, ld, nullptr) == 0);
    CHECK(fqss_dwconv_bwd_w(g.data(), x.data(), gdw.data(), B, C, M, 3, 4, 4, ld, ld, nullptr) == 0);
    auto gam = rnd(C), bet = rnd(C);
    CHECK(fqss_gn_fwd(x.data(), gam.data(), bet.data(), y.data(), mr.data(), B, C, M, ld, ld, 1e-8f, nullptr, nullptr) == 0);
    CHECK(fqss_gn_bwd(g.data(), x.data(), gam.data(), mr.data(), gz.data(), gg.data(), gbt.data(), B, C, M, ld, ld, ld, nullptr, nullptr) == 0);
    // element-wise, masking product
    CHECK(fqss_axpby(x.data(), g.data(), 1.0f, -1.0f, y.data(), B * C, M, ld, ld, ld, nullptr) == 0);
    const int S = 2;
    auto mask = rnd((size_t)B * S * C * ld);
    std::vector<float> mz(mask.size()), gmask(mask.size()), gfeat((size_t)B * C * ld);
    CHECK(fqss_mul_bcast_fwd(mask.data(), x.data(), mz.data(), B, S, C, M, ld, ld, ld, nullptr) == 0);
    CHECK(fqss_mul_bcast_bwd(mz.data(), mask.data(), x.data(), gmask.data(), gfeat.data(), B, S, C, M, ld, ld, ld, ld, ld, nullptr) == 0);
    // splitter, framing conv, overlap-add, weight gradients
    const int K = 16, st = 8;
    const int64_t T = (int64_t)(M - 1) * st + K;
    auto sig = rnd((size_t)B * T, 1.8f);
    std::vector<float> sp((size_t)B * 2 * T), enc((size_t)B * Co * ld), dec((size_t)B * T), gwe((size_t)Co * 2 * K, 0.f), gwd((size_t)Co * K, 0.f);
    CHECK(fqss_obs_reset(obs, 1, nullptr) == 0 && fqss_minmax(sig.data(), B, T, T, obs, nullptr) == 0);
    CHECK(fqss_splitter2(sig.data(), sp.data(), B, T, obs, nullptr) == 0);
    auto ew = rnd((size_t)Co * 2 * K);
    CHECK(fqss_frames_conv_fwd(sp.data(), ew.data(), enc.data(), B, 2, Co, T, K, st, M, ld, nullptr) == 0);
    CHECK(fqss_ola_convtr_fwd(enc.data(), ew.data(), dec.data(), B, Co, M, ld, K, st, T, nullptr) == 0);
    CHECK(fqss_frames_wgrad(enc.data(), sp.data(), gwe.data(), B, Co, 2, M, ld, T, K, st, nullptr) == 0);
    CHECK(fqss_frames_wgrad1(enc.data(), dec.data(), gwd.data(), B, Co, M, ld, T, K, st, nullptr) == 0);
    CHECK(fqss_frames_wgrad1s(enc.data(), sp.data() + T, 2 * T, gwe.data() + K, 2 * K, B, Co, M, ld, T, K, st, nullptr) == 0);
    // loss, clip + Adam
    const int64_t TT = 301;
    auto est = rnd((size_t)B * 2 * TT), fe = rnd((size_t)B * 2 * TT), tg = rnd((size_t)B * 2 * TT);
    std::vector<float> out(4), wv(B), si(B), ge((size_t)B * 2 * TT);
    CHECK(fqss_kd_loss(est.data(), fe.data(), tg.data(), B, TT, 0.1f, nullptr, out.data(), wv.data(), si.data(), ge.data(), nullptr) == 0);
    CHECK(std::isfinite(out[0]) && std::isfinite(ge[5]));
    const int64_t n = 1001;
    auto p = rnd(n), gp = rnd(n);
    std::vector<float> m1(n, 0.f), v1(n, 0.f);
    std::vector<int32_t> t0(n, 0);
    t0[7] = INT32_MAX;
    double ss = 0.0;
    int32_t stp = 0;
    float gn = 0.f;
    CHECK(fqss_sumsq(gp.data(), n, &ss, nullptr) == 0);
    const float p7 = p[7];
    CHECK(fqss_adam_clip(p.data(), gp.data(), m1.data(), v1.data(), n, &ss, 5.0f, 1.0f, 1e-3f, 0.9f, 0.999f, 1e-8f, &stp, t0.data(), &gn, nullptr) == 0);
    CHECK(stp == 1 && p[7] == p7 && fabs(gn - sqrt(ss)) < 1e-3);
    CHECK(fqss_version() == FQSS_VERSION);
    printf("selftest ok\n");
    return 0;
}
