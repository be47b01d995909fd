"""The step's heaviest kernels at their cfg-2 launch shapes (B = 8 x 4 s, M = 3999 frames), shared by
bench.py (live HIP-event timing -> `roofline`) and tools/roofline_probe.py (rocprofv3 / PMC passes).

Every case carries
  launches   launches per QAT step at that shape (24 TCN blocks; profiles/r01_step_steady_state.txt)
  flops      algorithmic flop per launch (2*Co*Ci*B*M for a 1x1 conv)
  rd, wr     algorithmic HBM bytes per launch (bytes = rd + wr): every operand read once + every result written
             once, at the width the kernel's interface moves it (fp32 4 B, 8-bit codes 1 B)
  survey     the same under SURVEY.md 8(d)'s convention 4 B x (in + out elements)
"""
import torch

from . import kernels as K

B, M = 8, 3999
NB, NH = 128, 512          # bottleneck / hidden channels of the TCN blocks
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak (the arithmetic of these GEMMs is fp32-exact)


def _act(C, dev):
    t = K.empty_act((B, C, M), dev)
    t.normal_()
    return t


def _codes(C, dev):
    t = K.empty_codes((B, C, M), dev)
    t.random_(0, 256)
    return t


def build(dev, calib=False):
    """-> list of dicts (kernel, label, bound, launches, flops, bytes, survey, fn)"""
    n = B * M
    lo, hi = torch.tensor([-2.0], device=dev), torch.tensor([2.5], device=dev)
    slope = torch.tensor([0.25], device=dev)
    cases = []

    # ---- fused teacher GEMM (csrc/teacher.hip): T1 conv1+PReLU+stats, T3 GN-prologue res+skip GEMM + residuals
    h, acc, y = _act(NB, dev), _act(NB, dev), _act(NH, dev)
    w1, w3 = K.split3_planes(torch.randn(NH, NB, device=dev) * 0.1), K.split3_planes(torch.randn(2 * NB, NH, device=dev) * 0.05)
    b1, b3 = torch.randn(NH, device=dev), torch.randn(2 * NB, device=dev)
    st = K.tstat_buffer(2, B, dev)
    K.tstats(y, st[0])
    ga, be = torch.ones(NH, device=dev), torch.zeros(NH, device=dev)
    cases.append(dict(kernel="k_tgemm", label="teacher T1: 1x1 conv 128->512 + PReLU + GroupNorm statistics", bound="mfma", launches=24,
                      flops=2.0 * NH * NB * n, rd=4.0 * NB * n, wr=4.0 * NH * n, survey=4.0 * (NB + NH) * n,
                      fn=lambda: K.tgemm(w1, h, b1, act=K.ACT_PRELU, slope=slope, stats_out=st[1])))
    cases.append(dict(kernel="k_tgemm", label="teacher T3: GroupNorm-apply + res|skip 1x1 convs 512->256 + residual adds", bound="mfma", launches=24,
                      flops=2.0 * 2 * NB * NH * n, rd=4.0 * (NH + 2 * NB) * n, wr=4.0 * 2 * NB * n, survey=4.0 * (NH + 2 * NB) * n,
                      fn=lambda: K.tgemm(w3, y, b3, pro=1, pro_stats=st[0], pro_gamma=ga, pro_beta=be, pro_eps=1e-8, M1=NB, r1=h, r2=acc)))

    # ---- student q-GEMMs on 8-bit codes (csrc/qgemm.hip): conv1 of a block (128->512) and the paired res|skip
    #      convs (512 -> 128+128, one GEMM over the concatenated channels)
    xc_b, xc_h = _codes(NB, dev), _codes(NH, dev)
    gz_b, gz_b2, gz_h = _act(NB, dev), _act(NB, dev), _act(NH, dev)
    ones = lambda c: torch.ones(c, 1, 1, device=dev) * 0.2
    wc_up = K.wq_codes(torch.randn(NH, NB, 1, device=dev) * 0.05, -ones(NH), ones(NH))
    wr_, ws_ = (K.wq_codes(torch.randn(NB, NH, 1, device=dev) * 0.05, -ones(NB), ones(NB)) for _ in range(2))
    pc = K.WCodes()
    pc.Co, pc.Ci = 2 * NB, NH
    pc.idx, pc.idxT = torch.cat([wr_.idx, ws_.idx], 0).contiguous(), torch.cat([wr_.idxT, ws_.idxT], 1).contiguous()
    pc.dw, pc.rw = torch.cat([wr_.dw, ws_.dw]), torch.cat([wr_.rw, ws_.rw])
    bu, bd, bd2 = torch.randn(NH, device=dev), torch.randn(NB, device=dev), torch.randn(NB, device=dev)
    gw_up, gw_pair = torch.zeros(NH, NB, device=dev), torch.zeros(2 * NB, NH, device=dev)
    fl = 2.0 * NH * NB * n
    cases.append(dict(kernel="k_qwgrad", label="student wgrad 128->512 (fp32 gz x u8 codes)", bound="hbm", launches=24,
                      flops=fl, rd=(4.0 * NH + NB) * n, wr=4.0 * NH * NB, survey=4.0 * (NH + NB) * n, fn=lambda: K.qpw_bwd_w(gz_h, xc_b, lo, hi, gw_up)))
    cases.append(dict(kernel="k_qwgrad", label="student wgrad res|skip pair 512->128+128 (fp32 gz x u8 codes)", bound="hbm", launches=24,
                      flops=2 * fl, rd=(8.0 * NB + NH) * n, wr=8.0 * NH * NB, survey=4.0 * (NH + 2 * NB) * n,
                      fn=lambda: K.qpw_bwd_w2(gz_b, gz_b2, xc_h, lo, hi, gw_pair)))
    cases.append(dict(kernel="k_qgemm<1>", label="student dgrad of 128->512 (int8 W^T x fp32 gz)", bound="hbm", launches=24,
                      flops=fl, rd=4.0 * NH * n, wr=4.0 * NB * n, survey=4.0 * (NH + NB) * n, fn=lambda: K.qpw_bwd_x(gz_h, wc_up)))
    cases.append(dict(kernel="k_qgemm<1>", label="student dgrad of the res|skip pair (K = 128+128 -> 512)", bound="hbm", launches=24,
                      flops=2 * fl, rd=8.0 * NB * n, wr=4.0 * NH * n, survey=4.0 * (NH + 2 * NB) * n, fn=lambda: K.qpw_bwd_x2(gz_b, gz_b2, pc)))
    cases.append(dict(kernel="k_qgemm<0>", label="student fwd 128->512 (int8 W x u8 codes -> fp32)", bound="hbm", launches=24,
                      flops=fl, rd=1.0 * NB * n, wr=4.0 * NH * n, survey=4.0 * (NH + NB) * n, fn=lambda: K.qpw_fwd(xc_b, wc_up, bu, lo, hi)))
    cases.append(dict(kernel="k_qgemm<0>", label="student fwd res|skip pair 512->128+128 (int8 W x u8 codes -> fp32)", bound="hbm", launches=24,
                      flops=2 * fl, rd=1.0 * NH * n, wr=8.0 * NB * n, survey=4.0 * (NH + 2 * NB) * n,
                      fn=lambda: K.qpw_fwd2(xc_h, pc, bd, bd2, lo, hi, NB)))

    # ---- depthwise layer backward (one launch, gz in LDS) and the activation-quantizer backward
    w_dw, gm = torch.randn(NH, 1, 3, device=dev), torch.ones(NH, device=dev)
    gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
    gbb, gw_dw = torch.zeros(NH, device=dev), torch.zeros(NH, 1, 3, device=dev)
    cases.append(dict(kernel="k_dwq_bwd", label="student depthwise+PReLU+fq backward, C=512 (codes + fp32 g in, fp32 gx out)", bound="hbm", launches=24,
                      flops=0.0, rd=5.0 * NH * n, wr=4.0 * NH * n, survey=12.0 * NH * n,
                      fn=lambda: K.dwq_bwd(xc_h, lo, hi, w_dw, gm, gz_h, 4, 4, 1, slope, lo, hi, gacc, gbb, gw_dw)))
    # ---- gLN backward with the producing conv's epilogue backward fused in (two passes), and the AddQ backward that
    #      also runs the res / skip convs' output-quantizer backward
    z_h, z_b, z_b2 = _act(NH, dev), _act(NB, dev), _act(NB, dev)
    gm_, bt_ = torch.rand(NH, device=dev) + 0.5, torch.randn(NH, device=dev) * 0.1
    _, _, mr = K.gnq_fwd(xc_h, lo, hi, gm_, bt_, 1e-8, lo, hi, False)
    gg, gb2 = torch.zeros(NH, device=dev), torch.zeros(NH, device=dev)
    pgacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
    pgb = torch.zeros(NH, device=dev)
    cases.append(dict(kernel="k_gnq_bwd_rows+coef+apply", label="gLN+fq backward (2 passes) with the producer conv's STE/PReLU/range/bias backward fused, C=512",
                      bound="hbm", launches=24, flops=0.0, rd=14.0 * NH * n, wr=4.0 * NH * n, survey=20.0 * NH * n,
                      fn=lambda: K.gnq_bwd(xc_h, lo, hi, gz_h, gm_, bt_, mr, lo, hi, gacc, gg, gb2, producer=(z_h, 1, slope, pgacc, pgb))))
    cases.append(dict(kernel="k_gnq_bwd_rows+coef+apply", label="gLN+fq backward (2 passes, plain), C=512", bound="hbm", launches=24,
                      flops=0.0, rd=10.0 * NH * n, wr=4.0 * NH * n, survey=16.0 * NH * n,
                      fn=lambda: K.gnq_bwd(xc_h, lo, hi, gz_h, gm_, bt_, mr, lo, hi, gacc, gg, gb2)))
    xc_b2 = _codes(NB, dev)
    pga, pgb_a, pgb_b = torch.zeros_like(pgacc), torch.zeros(NB, device=dev), torch.zeros(NB, device=dev)
    cases.append(dict(kernel="k_ewq_bwd", label="AddQ backward with one operand's conv output-quantizer backward fused, C=128", bound="hbm", launches=48,
                      flops=0.0, rd=10.0 * NB * n, wr=8.0 * NB * n, survey=16.0 * NB * n,
                      fn=lambda: K.ewq_bwd_p(xc_b, lo, hi, xc_b2, lo, hi, 1.0, gz_b, 0, None, lo, hi, gacc, NB, prod_b=(z_b, 0, None, pga, pgb_b))))
    gb_b = torch.zeros(NB, device=dev)
    cases.append(dict(kernel="k_actq_bwd", label="activation fake-quant backward (bottleneck / mask convs), C=128", bound="hbm", launches=4,
                      flops=0.0, rd=8.0 * NB * n, wr=4.0 * NB * n, survey=12.0 * NB * n,
                      fn=lambda: K.actq_bwd(z_b, gz_b, 0, None, 2, lo, hi, gacc, gbias=gb_b, C=NB)))
    for c in cases:
        c["bytes"] = c["rd"] + c["wr"]
        c.setdefault("calib", False)
    if calib:
        # single-pass shapes (one workgroup row/column of tiles: every operand byte is requested exactly once), used by
        # tools/roofline_probe.py to calibrate the FETCH_SIZE byte scale of each kernel's access pattern
        gz64 = K.empty_act((B, 64, M), dev).normal_()
        gw64 = torch.zeros(64, NB, device=dev)
        wq = K.split3_planes(torch.randn(NB, NB, device=dev) * 0.1)
        bq = torch.randn(NB, device=dev)
        extra = [
            dict(kernel="k_qwgrad", label="calibration: wgrad 128->64 (single pass)", rd=(4.0 * 64 + NB) * n, wr=4.0 * 64 * NB,
                 fn=lambda: K.qpw_bwd_w(gz64, xc_b, lo, hi, gw64)),
            dict(kernel="k_tgemm", label="calibration: tgemm 128->128 (single pass)", rd=4.0 * NB * n, wr=4.0 * NB * n,
                 fn=lambda: K.tgemm(wq, h, bq)),
        ]
        for c in extra:
            c.update(bound="hbm", launches=24, flops=0.0, survey=c["rd"] + c["wr"], bytes=c["rd"] + c["wr"], calib=True)
        cases += extra
    return cases


def time_case(case, iters=20, warm=3):
    """average launch duration in ms, HIP events on the stream the kernels are launched on (torch's current stream)"""
    for _ in range(warm):
        case["fn"]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        case["fn"]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def summarize(cases, times_ms):
    """group the per-shape measurements by kernel; -> list sorted by time per step (dominant first)"""
    groups = {}
    for c, ms in zip(cases, times_ms):
        g = groups.setdefault(c["kernel"], dict(kernel=c["kernel"], bound=c["bound"], launches=0, ms_step=0.0, flops=0.0, bytes=0.0,
                                                 survey=0.0, shapes=[]))
        g["launches"] += c["launches"]
        g["ms_step"] += c["launches"] * ms
        g["flops"] += c["launches"] * c["flops"]
        g["bytes"] += c["launches"] * c["bytes"]
        g["survey"] += c["launches"] * c["survey"]
        g["shapes"].append({"label": c["label"], "launches_per_step": c["launches"], "launch_us": round(ms * 1e3, 2)})
    return sorted(groups.values(), key=lambda g: -g["ms_step"])


def roofline_object(g, traffic=None):
    """the `roofline` JSON object of one kernel group: launch-weighted averages over its shapes"""
    n = g["launches"]
    ms = g["ms_step"] / n
    gbps = g["bytes"] / n / (ms * 1e-3) / 1e9
    out = {"kernel": g["kernel"], "bound": g["bound"], "launch_us": round(ms * 1e3, 2), "launches_per_step": n,
           "ms_per_step": round(g["ms_step"], 3), "algorithmic_bytes_per_launch": round(g["bytes"] / n),
           "algorithmic_GBps": round(gbps, 1), "survey_convention_GBps": round(g["survey"] / n / (ms * 1e-3) / 1e9, 1),
           "traffic": traffic, "shapes": g["shapes"]}
    if g["bound"] == "mfma":
        tf = g["flops"] / n / (ms * 1e-3) / 1e12
        out.update(achieved=round(tf, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                   hbm_frac=round(gbps / HBM_PEAK_GBS, 4))
    else:
        out.update(achieved=round(gbps, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbps / HBM_PEAK_GBS, 4))
    return out
