// desc_api.hip -- descriptor-struct forms of the long C entry points (include/fqss.h): validation of the
// descriptors (dtype, rank, contiguity, workspace size) and a call of the flat form.  No kernels here.
#include <cstring>

#include "fqss_dev.h"

using namespace fqss;

namespace {

// a [B][C][M] tensor (or [rows][cols]) whose rows are contiguous and equally spaced: base pointer + row stride
struct Rows {
    void* p;
    int64_t B, C, M, ld;
};

bool rows3(const FqssTensor* t, int dtype, Rows& r, const char*& why) {
    if (t == nullptr || t->data == nullptr) { why = "null tensor"; return false; }
    if (t->dtype != dtype) { why = "wrong dtype"; return false; }
    if (t->ndim < 2 || t->ndim > 3) { why = "tensor must be [B][C][M] or [rows][cols]"; return false; }
    const int n = t->ndim;
    if (t->stride[n - 1] != 1) { why = "innermost stride must be 1"; return false; }
    for (int i = 0; i < n; ++i)
        if (t->shape[i] < 0) { why = "negative extent"; return false; }
    r.p = t->data;
    r.M = t->shape[n - 1];
    r.C = t->shape[n - 2];
    r.B = n == 3 ? t->shape[0] : 1;
    r.ld = t->stride[n - 2];
    if (r.ld < r.M) { why = "row stride smaller than the row"; return false; }
    if (n == 3 && r.B > 1 && t->stride[0] != r.C * r.ld) { why = "batch stride must be C * row stride"; return false; }
    return true;
}

bool same_shape(const Rows& a, const Rows& b) { return a.B == b.B && a.C == b.C && a.M == b.M; }

#define DESC_ROWS(var, tensor, dtype)                                  \
    Rows var{};                                                        \
    {                                                                  \
        const char* why_ = nullptr;                                    \
        if (!rows3(tensor, dtype, var, why_)) {                        \
            ::fqss::set_error("%s: %s: %s", __func__, #tensor, why_); \
            return FQSS_EINVAL;                                        \
        }                                                              \
    }

int64_t ws_bytes_of(const char* op, int64_t B, int64_t C, int64_t M) {
    if (!strcmp(op, "gln_fq_fwd")) return 2 * 64 * B * (int64_t)sizeof(int64_t);
    if (!strcmp(op, "gln_fq_bwd")) return (2 * B * C + 2 * B) * (int64_t)sizeof(double);
    if (!strcmp(op, "pwconv_fq_fwd")) return B * fqss_qpw_stat_slots((int)C, (int)M) * 2 * (int64_t)sizeof(int64_t);
    if (!strcmp(op, "dwconv_fq_fwd")) return B * fqss_dwq_stat_slots((int)C, (int)M) * 2 * (int64_t)sizeof(int64_t);
    if (!strcmp(op, "add_fq_fwd") || !strcmp(op, "add_fq_bwd") || !strcmp(op, "tgemm")) return 0;
    return -1;
}

}  // namespace

extern "C" int64_t fqss_workspace_bytes(const char* op, const int64_t* shape, int ndim) {
    if (op == nullptr || shape == nullptr || ndim != 3 || shape[0] < 0 || shape[1] < 0 || shape[2] < 0 || shape[1] > INT32_MAX ||
        shape[2] > INT32_MAX) {
        set_error("fqss_workspace_bytes: shape must be {B, C, M}");
        return -1;
    }
    const int64_t n = ws_bytes_of(op, shape[0], shape[1], shape[2]);
    if (n < 0) set_error("fqss_workspace_bytes: unknown op '%s'", op);
    return n;
}

extern "C" int fqss_add_fq_fwd(const FqssTensor* a, const FqssQParams* qa, const FqssTensor* b, const FqssQParams* qb, float sb,
                               FqssTensor* y, FqssTensor* y_out, const FqssQParams* q, void* ws, size_t ws_bytes,
                               fqss_stream_t stream) {
    (void)ws; (void)ws_bytes;
    FQSS_REQUIRE(qa && q && qa->qmin && qa->qmax && q->qmin && q->qmax, "null quantizer");
    DESC_ROWS(ra, a, FQSS_DT_U8);
    DESC_ROWS(ry, y, FQSS_DT_U8);
    FQSS_REQUIRE(same_shape(ra, ry), "a / y shapes differ");
    Rows rb{}, ro{};
    const bool b_codes = b != nullptr && b->dtype == FQSS_DT_U8;
    if (b != nullptr) {
        DESC_ROWS(rb_, b, b_codes ? FQSS_DT_U8 : FQSS_DT_F32);
        FQSS_REQUIRE(same_shape(ra, rb_), "a / b shapes differ");
        FQSS_REQUIRE(!b_codes || (qb && qb->qmin && qb->qmax), "coded b needs its quantizer");
        rb = rb_;
    }
    if (y_out != nullptr) {
        DESC_ROWS(ro_, y_out, FQSS_DT_F32);
        FQSS_REQUIRE(same_shape(ra, ro_), "a / y_out shapes differ");
        ro = ro_;
    }
    return fqss_ewq_fwd((const uint8_t*)ra.p, qa->qmin, qa->qmax, b_codes ? (const uint8_t*)rb.p : nullptr, b_codes ? qb->qmin : nullptr,
                        b_codes ? qb->qmax : nullptr, (b != nullptr && !b_codes) ? (const float*)rb.p : nullptr, sb, (uint8_t*)ry.p,
                        (float*)ro.p, ra.B * ra.C, ra.M, ra.ld, rb.ld, rb.ld, ry.ld, ro.ld, q->act, q->slope, q->qmin, q->qmax, stream);
}

extern "C" int fqss_add_fq_bwd(const FqssTensor* a, const FqssQParams* qa, const FqssTensor* b, const FqssQParams* qb, float sb,
                               const FqssTensor* g, FqssTensor* gz, const FqssQParams* q, const FqssProducer* pa, const FqssProducer* pb,
                               void* ws, size_t ws_bytes, fqss_stream_t stream) {
    (void)ws; (void)ws_bytes;
    FQSS_REQUIRE(qa && q && qa->qmin && qa->qmax && q->qmin && q->qmax && q->gacc, "null quantizer / gacc");
    DESC_ROWS(ra, a, FQSS_DT_U8);
    DESC_ROWS(rg, g, FQSS_DT_F32);
    FQSS_REQUIRE(same_shape(ra, rg), "a / g shapes differ");
    Rows rb{}, rz{};
    const bool b_codes = b != nullptr && b->dtype == FQSS_DT_U8;
    if (b != nullptr) {
        DESC_ROWS(rb_, b, b_codes ? FQSS_DT_U8 : FQSS_DT_F32);
        FQSS_REQUIRE(same_shape(ra, rb_), "a / b shapes differ");
        FQSS_REQUIRE(!b_codes || (qb && qb->qmin && qb->qmax), "coded b needs its quantizer");
        rb = rb_;
    }
    if (gz != nullptr) {
        DESC_ROWS(rz_, gz, FQSS_DT_F32);
        FQSS_REQUIRE(same_shape(ra, rz_), "a / gz shapes differ");
        rz = rz_;
    }
    if (pa == nullptr && pb == nullptr) {
        FQSS_REQUIRE(gz != nullptr, "gz missing");
        return fqss_ewq_bwd((const uint8_t*)ra.p, qa->qmin, qa->qmax, b_codes ? (const uint8_t*)rb.p : nullptr, b_codes ? qb->qmin : nullptr,
                            b_codes ? qb->qmax : nullptr, (b != nullptr && !b_codes) ? (const float*)rb.p : nullptr, sb, (const float*)rg.p,
                            (float*)rz.p, ra.B * ra.C, ra.M, ra.ld, rb.ld, rb.ld, rg.ld, rz.ld, q->act, q->slope, q->qmin, q->qmax, q->gacc,
                            stream);
    }
    FQSS_REQUIRE(b == nullptr || b_codes, "fused producers need a coded (or absent) b");
    Rows pz[2] = {{}, {}}, po[2] = {{}, {}};
    const FqssProducer* ps[2] = {pa, pb};
    for (int i = 0; i < 2; ++i) {
        if (ps[i] == nullptr) continue;
        FQSS_REQUIRE(ps[i]->gacc != nullptr, "producer gacc missing");
        DESC_ROWS(z_, ps[i]->z, FQSS_DT_F32);
        DESC_ROWS(o_, ps[i]->out, FQSS_DT_F32);
        FQSS_REQUIRE(same_shape(ra, z_) && same_shape(ra, o_), "producer shapes differ from a");
        pz[i] = z_;
        po[i] = o_;
    }
    FQSS_REQUIRE(gz != nullptr || (pa != nullptr && (b == nullptr || pb != nullptr)), "gz may only be NULL when every operand is fused");
    return fqss_ewq_bwd_p((const uint8_t*)ra.p, qa->qmin, qa->qmax, b_codes ? (const uint8_t*)rb.p : nullptr, b_codes ? qb->qmin : nullptr,
                          b_codes ? qb->qmax : nullptr, sb, (const float*)rg.p, (float*)rz.p, ra.B * ra.C, ra.M, ra.ld, rb.ld, rg.ld, rz.ld,
                          q->act, q->slope, q->qmin, q->qmax, q->gacc, (int)ra.C, (const float*)pz[0].p, pz[0].ld, pa ? pa->act : 0,
                          pa ? pa->slope : nullptr, pa ? pa->gacc : nullptr, pa ? pa->gbias : nullptr, (float*)po[0].p, po[0].ld,
                          (const float*)pz[1].p, pz[1].ld, pb ? pb->act : 0, pb ? pb->slope : nullptr, pb ? pb->gacc : nullptr,
                          pb ? pb->gbias : nullptr, (float*)po[1].p, po[1].ld, stream);
}

extern "C" int fqss_pwconv_fq_fwd(const FqssTensor* x, const FqssQParams* qx, const FqssWCodes* w, const float* bias1, const float* bias2,
                                  FqssTensor* z1, FqssTensor* z2, FqssTensor* y1, FqssTensor* y2, const FqssQParams* q1,
                                  const FqssQParams* q2, void* ws, size_t ws_bytes, fqss_stream_t stream) {
    FQSS_REQUIRE(qx && qx->qmin && qx->qmax && q1 && q1->qmin && q1->qmax, "null quantizer");
    FQSS_REQUIRE(w && w->idx && w->dw && w->rw && w->Co > 0 && w->Ci > 0, "null weight codes");
    DESC_ROWS(rx, x, FQSS_DT_U8);
    DESC_ROWS(rz1, z1, FQSS_DT_F32);
    DESC_ROWS(ry1, y1, FQSS_DT_U8);
    FQSS_REQUIRE(rx.C == w->Ci, "x channels != Ci");
    FQSS_REQUIRE(rz1.B == rx.B && rz1.M == rx.M && same_shape(rz1, ry1), "z1 / y1 shapes");
    const int Co1 = (int)rz1.C;
    int Co2 = 0;
    Rows rz2{}, ry2{};
    if (z2 != nullptr || y2 != nullptr) {
        FQSS_REQUIRE(z2 && y2 && q2 && q2->qmin && q2->qmax, "second layer: z2, y2 and q2 all needed");
        DESC_ROWS(rz2_, z2, FQSS_DT_F32);
        DESC_ROWS(ry2_, y2, FQSS_DT_U8);
        FQSS_REQUIRE(rz2_.B == rx.B && rz2_.M == rx.M && same_shape(rz2_, ry2_), "z2 / y2 shapes");
        rz2 = rz2_;
        ry2 = ry2_;
        Co2 = (int)rz2.C;
    }
    FQSS_REQUIRE(Co1 + Co2 == w->Co, "z1 + z2 channels != Co of the weight codes");
    int64_t* stats = nullptr;
    if (ws != nullptr && Co2 == 0) {
        const int64_t need = ws_bytes_of("pwconv_fq_fwd", rx.B, Co1, rx.M);
        if (need > 0) {
            FQSS_REQUIRE((int64_t)ws_bytes >= need, "workspace smaller than fqss_workspace_bytes(\"pwconv_fq_fwd\")");
            stats = (int64_t*)ws;
        }
    }
    return fqss_qpw_fwdq((const uint8_t*)rx.p, w->idx, w->dw, w->rw, bias1, bias2, qx->qmin, qx->qmax, (float*)rz1.p, (float*)rz2.p, q1->act,
                         q1->slope, q1->qmin, q1->qmax, Co2 ? q2->qmin : nullptr, Co2 ? q2->qmax : nullptr, (uint8_t*)ry1.p, (uint8_t*)ry2.p,
                         (int)rx.B, w->Ci, Co1, Co2, (int)rx.M, rx.ld, rz1.ld, rz2.ld, ry1.ld, ry2.ld, stats, stream);
}

extern "C" int fqss_gln_fq_fwd(const FqssTensor* x, const FqssQParams* qx, const float* gamma, const float* beta, float eps, FqssTensor* y,
                               FqssTensor* y_out, float* mean_rstd, const FqssQParams* q, void* ws, size_t ws_bytes, const int64_t* stats,
                               int nslots, fqss_stream_t stream) {
    FQSS_REQUIRE(qx && qx->qmin && qx->qmax && q && q->qmin && q->qmax && gamma && beta && mean_rstd, "null argument");
    DESC_ROWS(rx, x, FQSS_DT_U8);
    DESC_ROWS(ry, y, FQSS_DT_U8);
    FQSS_REQUIRE(same_shape(rx, ry), "x / y shapes differ");
    Rows ro{};
    if (y_out != nullptr) {
        DESC_ROWS(ro_, y_out, FQSS_DT_F32);
        FQSS_REQUIRE(same_shape(rx, ro_), "x / y_out shapes differ");
        ro = ro_;
    }
    if (stats == nullptr)
        FQSS_REQUIRE(ws != nullptr && (int64_t)ws_bytes >= ws_bytes_of("gln_fq_fwd", rx.B, rx.C, rx.M),
                     "workspace smaller than fqss_workspace_bytes(\"gln_fq_fwd\")");
    return fqss_gnq_fwd((const uint8_t*)rx.p, qx->qmin, qx->qmax, gamma, beta, (uint8_t*)ry.p, (float*)ro.p, mean_rstd, (int)rx.B, (int)rx.C,
                        (int)rx.M, rx.ld, ry.ld, ro.ld, eps, q->qmin, q->qmax, ws, stats, nslots, stream);
}

extern "C" int fqss_tgemm_desc(const FqssTGemmDesc* d, fqss_stream_t stream) {
    FQSS_REQUIRE(d && d->planes && d->planes->data && d->planes->dtype == FQSS_DT_U16 && d->planes->ndim == 3 && d->planes->shape[0] == 3,
                 "planes must be a [3][Co][Ci] u16 tensor");
    const int Co = (int)d->planes->shape[1], Ci = (int)d->planes->shape[2];
    FQSS_REQUIRE(d->planes->stride[2] == 1 && d->planes->stride[1] == Ci && d->planes->stride[0] == (int64_t)Co * Ci, "planes must be dense");
    DESC_ROWS(rx, d->x, FQSS_DT_F32);
    FQSS_REQUIRE(rx.C == Ci, "x channels != Ci");
    DESC_ROWS(c1, d->c1, FQSS_DT_F32);
    FQSS_REQUIRE(c1.B == rx.B && c1.M == rx.M && c1.C == (d->M1 < Co ? d->M1 : Co), "c1 shape");
    Rows c2{}, r1{}, r2{};
    if (d->M1 < Co) {
        DESC_ROWS(c2_, d->c2, FQSS_DT_F32);
        FQSS_REQUIRE(c2_.B == rx.B && c2_.M == rx.M && c2_.C == Co - d->M1, "c2 shape");
        c2 = c2_;
    }
    if (d->r1 != nullptr) {
        DESC_ROWS(r1_, d->r1, FQSS_DT_F32);
        FQSS_REQUIRE(same_shape(r1_, c1) && r1_.ld == c1.ld, "r1 must have the layout of c1");
        r1 = r1_;
    }
    if (d->r2 != nullptr) {
        DESC_ROWS(r2_, d->r2, FQSS_DT_F32);
        FQSS_REQUIRE(same_shape(r2_, c2) && r2_.ld == c2.ld, "r2 must have the layout of c2");
        r2 = r2_;
    }
    return fqss_tgemm((const uint16_t*)d->planes->data, (const float*)rx.p, (int)rx.B, Ci, Co, (int)rx.M, rx.ld, d->pro, d->pro_stats,
                      d->pro_gamma, d->pro_beta, d->pro_eps, d->pro_slope, d->bias, d->act, d->slope, d->stats_out, d->M1, (float*)c1.p,
                      (const float*)r1.p, c1.ld, (float*)c2.p, (const float*)r2.p, c2.ld, stream);
}
