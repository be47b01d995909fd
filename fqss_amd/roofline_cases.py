"""The step's kernels at their cfg-2 launch shapes (B = 8 x 4 s, M = 3999 frames), shared by bench.py (live HIP-event timing ->
`roofline`), tools/roofline_probe.py (rocprofv3 / PMC passes) and tools/kbench.py.

Every case is ONE kernel launch (`group=True` marks the two multi-kernel operations that are timed as a whole and never picked as
`roofline`) and carries
  launches   launches per QAT step at that shape (24 TCN blocks; profiles/r02_step_eager_steady_state.txt)
  flops      algorithmic flop per launch (2*Co*Ci*B*M for a 1x1 conv)
  rd, wr     ALGORITHMIC HBM bytes per launch: every operand read once + every result written once at the width the kernel's
             interface moves it (fp32 4 B, 8-bit codes 1 B) -- SURVEY.md 8(d)'s per-tensor accounting applied to what this build moves
  survey     the same under SURVEY.md 8(d)'s convention 4 B x (in + out elements) of the LayerQ boundary tensors
Operands rotate over several buffer sets (> 256 MiB together): in the step nothing is cache-resident, a block moves ~1 GB.
"""
import torch

from . import kernels as K

B, M = 8, 3999
NB, NH = 128, 512          # bottleneck / hidden channels of the TCN blocks
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
# Dense peaks of the instruction a kernel EXECUTES (MI355X_MICROARCH.md, chip-level table; never the 2:1-sparsity figures): a kernel that
# computes an fp32-grade product as k exact bf16 partial products is priced on its k-fold ISSUED flops against the bf16 peak, not on
# its algorithmic flops against the fp32 peak (that basis allows fractions above 1).
PEAK_TFLOPS = {"bf16": 2500.0, "i8": 5000.0, "f32": 157.3}
N = B * M


# Vector-issue floor (round 6; VERDICT r05 next #1c): a wave64 fp32 VALU instruction occupies its SIMD for 2.3-2.7 cycles once two or
# more waves share it (tools/ubench/pk_rate.hip, profiles/r06_pk_rate.txt: v_fma / v_mul / v_add_f32 alone; packed forms take 4.1-4.4
# for two elements) -- 2.5 cycles at the 2.4 GHz property clock is the rate a kernel's SQ_INSTS_VALU count is priced at.
VALU_CYCLES_PER_INST, SHADER_GHZ = 2.5, 2.4
VALU_PEAK_PER_US = SHADER_GHZ * 1e3 / VALU_CYCLES_PER_INST       # wave instructions per SIMD and microsecond


def valu_counts():
    """{kernel symbol: SQ_INSTS_VALU per SIMD and launch} from the newest committed counter table (profiles/r*_sq_counters.txt, written by
    tools/r06_pmc.sh: the launches of THIS file's cases under rocprofv3 --pmc)"""
    import glob
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    for path in sorted(glob.glob(os.path.join(root, "r*_sq_counters.txt")), reverse=True):
        out = {}
        for line in open(path):
            m = re.match(r"(k_\w+(?:<[^>]*>)?)\s+n=.*?VALU/SIMD\s+(\d+)", line)
            if m:
                out[m.group(1)] = (float(m.group(2)), os.path.join("profiles", os.path.basename(path)))
        if out:
            return out
    return {}


# case name -> the instantiation the counter table lists it under
VALU_ALIAS = {"k_dwq_bwd<3, GA, GB>": "k_dwq_bwd<3, true, true, 1>", "k_qgemm<0>": "k_qgemm<0, 3>", "k_qgemm<1>": "k_qgemm<1, 3>",
              "k_qwgrad_group": "k_qwgrad_group<3>", "k_gnq_apply_t": "k_gnq_apply_t<4>", "k_ewq_bwd": "k_ewq_bwd<true>",
              "k_actq_bwd": "k_actq_bwd<4, true, false, false>", "k_axpby": "k_axpby<4>"}


def priced(flops, products, dtype, nbytes, us, valu=None):
    """`roofline` numbers of ONE launch: flops = algorithmic flop, products = partial products issued per algorithmic product on the
    `dtype` matrix (or, "f32", vector) pipe, nbytes = algorithmic HBM bytes, us = measured launch duration, valu = (vector instructions
    per SIMD and launch from the PMC table, its file) or None.
    floor = max(issued flops / that dtype's dense peak, bytes / 8 TB/s, vector instructions x 2.5 cycles); `bound` names the largest
    floor ("mfma" | "hbm" | "valu"), `achieved` / `peak` / `unit` are that resource's, frac = floor / measured time = achieved / peak
    <= 1 for anything physical; the other resources' fractions ride beside it."""
    sec = us * 1e-6
    issued = flops * products
    t_m = issued / (PEAK_TFLOPS[dtype] * 1e12) if flops > 0 else 0.0
    t_h = nbytes / (HBM_PEAK_GBS * 1e9)
    gbps, tf = nbytes / sec / 1e9, issued / sec / 1e12
    o = {"hbm_frac": round(gbps / HBM_PEAK_GBS, 4), "algorithmic_GBps": round(gbps, 1), "floor_us": round(max(t_m, t_h) * 1e6, 2)}
    if flops > 0:
        o.update(mfma_dtype=dtype, products_per_term=products, issued_TFLOPs=round(tf, 1), issued_frac=round(tf / PEAK_TFLOPS[dtype], 4))
    t_v = 0.0
    if valu is not None:
        t_v = valu[0] / VALU_PEAK_PER_US * 1e-6
        o.update(valu_insts_per_simd=round(valu[0]), valu_floor_us=round(t_v * 1e6, 2), valu_frac=round(t_v / sec, 4), valu_source=valu[1])
        o["floor_us"] = round(max(t_m, t_h, t_v) * 1e6, 2)
    if t_v > t_m and t_v > t_h:
        o.update(bound="valu", achieved=round(valu[0] / us, 1), peak=VALU_PEAK_PER_US, unit="wave-instructions/us/SIMD", frac=round(t_v / sec, 4))
    elif t_m > t_h:
        o.update(bound="mfma", achieved=round(tf, 1), peak=PEAK_TFLOPS[dtype], unit="TFLOP/s", frac=round(tf / PEAK_TFLOPS[dtype], 4))
    else:
        o.update(bound="hbm", achieved=round(gbps, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbps / HBM_PEAK_GBS, 4))
    return o


# partial products issued per algorithmic product, and on which pipe (csrc/teacher.hip: 3 x 3 bf16 split, 6 leading terms; csrc/qgemm.hip:
# fp32 gradient in three exact bf16 pieces x one exact plane of 8-bit codes; forward: codes x codes, one bf16 product)
ISSUED = {"k_tgemm_k128": ("bf16", 6), "k_tgemm2<0>": ("bf16", 6), "k_tgemm2<1>": ("bf16", 6), "k_qgemm<1>": ("bf16", 3), "k_qwgrad2": ("bf16", 3), "k_qwgrad_group": ("bf16", 3),
          "k_qgemm<0>": ("bf16", 1), "k_gemm_x3_wq_multi": ("bf16", 3),
          "k_lstm_fwd_st<128>": ("f32", 1), "k_gemm_x3": ("bf16", 3), "k_qgemm<3>": ("bf16", 6), "k_attn_long_fwd_x3<64>": ("bf16", 6),
          "k_attn_long_fwd_c<64>": ("bf16", 3)}


def _act(C, dev):
    return K.empty_act((B, C, M), dev).normal_()


def _codes(C, dev):
    return K.empty_codes((B, C, M), dev).random_(0, 256)


def build(dev, sets=3):
    """-> list of dicts (kernel, label, bound, launches, flops, rd, wr, survey, fn(i): launch on operand set i % sets)"""
    n = N
    lo, hi = torch.tensor([-2.0], device=dev), torch.tensor([2.5], device=dev)
    slope = torch.tensor([0.25], device=dev)
    cases = []

    def case(kernel, label, bound, launches, rd, wr, survey, fn, flops=0.0, group=False):
        cases.append(dict(kernel=kernel, label=label, bound=bound, launches=launches, flops=flops, rd=float(rd), wr=float(wr),
                          bytes=float(rd + wr), survey=float(survey), fn=fn, group=group))

    R = range(sets)
    h_, acc_, y_ = [_act(NB, dev) for _ in R], [_act(NB, dev) for _ in R], [_act(NH, dev) for _ in R]
    gz_b, gz_b2, gz_h = [_act(NB, dev) for _ in R], [_act(NB, dev) for _ in R], [_act(NH, dev) for _ in R]
    z_h, z_b, z_b2 = [_act(NH, dev) for _ in R], [_act(NB, dev) for _ in R], [_act(NB, dev) for _ in R]
    xc_b, xc_b2, xc_h = [_codes(NB, dev) for _ in R], [_codes(NB, dev) for _ in R], [_codes(NH, dev) for _ in R]

    # ---- fused teacher (csrc/teacher.hip): T1 conv1+PReLU+stats, depthwise, T3 GN-prologue res|skip GEMM + residuals
    w1, w3 = K.split3_planes(torch.randn(NH, NB, device=dev) * 0.1), K.split3_planes(torch.randn(2 * NB, NH, device=dev) * 0.05)
    b1, b3 = torch.randn(NH, device=dev), torch.randn(2 * NB, device=dev)
    st = K.tstat_buffer(3, B, dev)
    K.tstats(y_[0], st[0])
    ga, be = torch.ones(NH, device=dev), torch.zeros(NH, device=dev)
    fl = 2.0 * NH * NB * n
    case("k_tgemm_k128", "teacher T1: 1x1 conv 128->512 + PReLU + GroupNorm statistics (weights in registers, two workgroups per CU)", "mfma", 24, 4.0 * NB * n, 4.0 * NH * n, 4.0 * (NB + NH) * n,
         lambda i: K.tgemm(w1, h_[i % sets], b1, act=K.ACT_PRELU, slope=slope, stats_out=st[1]), flops=fl)
    case("k_tgemm2<1>", "teacher T3: GroupNorm-apply + res|skip 1x1 convs 512->256 + residual adds", "mfma", 24, 4.0 * (NH + 2 * NB) * n, 4.0 * 2 * NB * n,
         4.0 * (NH + 2 * NB) * n,
         lambda i: K.tgemm(w3, y_[i % sets], b3, pro=1, pro_stats=st[0], pro_gamma=ga, pro_beta=be, pro_eps=1e-8, M1=NB, r1=h_[i % sets], r2=acc_[i % sets]),
         flops=2 * fl)
    w_dw, b_dw = torch.randn(NH, 1, 3, device=dev), torch.randn(NH, device=dev) * 0.1
    case("k_tdw", "teacher GN-apply + depthwise + PReLU + statistics, C=512", "hbm", 24, 4.0 * NH * n, 4.0 * NH * n, 8.0 * NH * n,
         lambda i: K.tdw(y_[i % sets], st[0], ga, be, 1e-8, w_dw, b_dw, slope, st[2], 4, 4))

    # ---- student q-GEMMs on 8-bit codes (csrc/qgemm.hip)
    ones = lambda c: torch.ones(c, 1, 1, device=dev) * 0.2
    wc_up = K.wq_codes(torch.randn(NH, NB, 1, device=dev) * 0.05, -ones(NH), ones(NH))
    wr_, ws_ = (K.wq_codes(torch.randn(NB, NH, 1, device=dev) * 0.05, -ones(NB), ones(NB)) for _ in range(2))
    pc = K.WCodes()
    pc.Co, pc.Ci = 2 * NB, NH
    pc.idx, pc.idxT = torch.cat([wr_.idx, ws_.idx], 0).contiguous(), torch.cat([wr_.idxT, ws_.idxT], 1).contiguous()
    pc.dw, pc.rw = torch.cat([wr_.dw, ws_.dw]), torch.cat([wr_.rw, ws_.rw])
    bu, bd, bd2 = torch.randn(NH, device=dev), torch.randn(NB, device=dev), torch.randn(NB, device=dev)
    gw_up, gw_pair = torch.zeros(NH, NB, device=dev), torch.zeros(2 * NB, NH, device=dev)
    # round 5: the weight gradients of the step's 50 quantized 1x1 convolutions run as TWO grouped launches of 25 layers (k_qwgrad_group:
    # 32 teams of 8 workgroups over the (layer, tile group, 64-frame stage) work list, slab reduction in part order, no atomics); the case
    # is one such launch -- 12 TCN blocks (conv1 128 -> 512 + the res | skip pair) + one 512 -> 128 layer, every layer on its own operands
    wq_jobs = []
    for bi in range(12):       # (own operands per layer, 1.4 GB: inside one launch nothing may be served from the 256 MB Infinity Cache twice)
        wq_jobs.append((_act(NH, dev), None, _codes(NB, dev), torch.zeros(NH, NB, device=dev)))
        wq_jobs.append((_act(NB, dev), _act(NB, dev), _codes(NH, dev), torch.zeros(2 * NB, NH, device=dev)))
    wq_jobs.append((gz_b[0], None, xc_h[0], torch.zeros(NB, NH, device=dev)))
    wq_queue = K.WgradQueue()

    def grouped(i):
        for a, b, c, gw_ in wq_jobs:
            wq_queue.push(a, b, c, lo, hi, gw_)
        wq_queue.flush()
    g_rd = 12 * ((4.0 * NH + NB) * n + (8.0 * NB + NH) * n) + (4.0 * NB + NH) * n
    g_wr = 4.0 * (12 * 3 + 1) * NH * NB
    case("k_qwgrad_group", "student weight gradients of 25 quantized 1x1 convs in one launch (12 x (128->512 + res|skip pair) + 512->128; fp32 gz x u8 codes)",
         "hbm", 2, g_rd, g_wr, 4.0 * (12 * ((NH + NB) + (NH + 2 * NB)) + (NH + NB)) * n, grouped, flops=(12 * 3 + 1) * fl)
    case("k_qgemm<1>", "student dgrad of 128->512 (int8 W^T x fp32 gz)", "hbm", 24, 4.0 * NH * n, 4.0 * NB * n, 4.0 * (NH + NB) * n,
         lambda i: K.qpw_bwd_x(gz_h[i % sets], wc_up), flops=fl)
    case("k_qgemm<1>", "student dgrad of the res|skip pair (K = 128+128 -> 512)", "hbm", 24, 8.0 * NB * n, 4.0 * NH * n, 4.0 * (NH + 2 * NB) * n,
         lambda i: K.qpw_bwd_x2(gz_b[i % sets], gz_b2[i % sets], pc), flops=2 * fl)
    stq = K.new_stats("qpw", B, NH, M, dev)
    case("k_qgemm<0>", "student fwd 128->512 + PReLU + fake-quant + gLN statistics (codes in; fp32 z + codes out)", "hbm", 24, 1.0 * NB * n, 5.0 * NH * n,
         4.0 * (NH + NB) * n, lambda i: K.qpw_fwdq(xc_b[i % sets], wc_up, bu, None, lo, hi, NH, 1, slope, (lo, hi), stats=stq), flops=fl)
    # round 5: the residual AddQ and the skip-sum AddQ behind the two outputs run in this GEMM's epilogue (+ their other operands' codes in,
    # + their sum codes out; SURVEY convention: + 12 B per element and AddQ, the figure the separate k_ewq_fwd launches carried)
    case("k_qgemm<0>", "student fwd res|skip pair 512->128+128 + fake-quant + the two AddQ behind it (codes in; fp32 z + codes + sum codes out)", "hbm", 24,
         1.0 * (NH + 2 * NB) * n, 12.0 * NB * n, 4.0 * (NH + 2 * NB) * n + 2 * 12.0 * NB * n * 47 / 48,     # (47 fused AddQ in 24 launches: the first block has no skip sum yet)
         lambda i: K.qpw_fwdq(xc_h[i % sets], pc, bd, bd2, lo, hi, NB, 0, None, (lo, hi), (lo, hi),
                              adds=((xc_b[i % sets], lo, hi, lo, hi), (xc_b2[i % sets], lo, hi, lo, hi))), flops=2 * fl)

    # ---- codes-only streaming layers (csrc/fused_q.hip)
    gm_, bt_ = torch.rand(NH, device=dev) + 0.5, torch.randn(NH, device=dev) * 0.1
    gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
    pgacc, pgacc2 = torch.zeros_like(gacc), torch.zeros_like(gacc)
    gg, gb2, gbb, gw_dw = (torch.zeros(NH, device=dev), torch.zeros(NH, device=dev), torch.zeros(NH, device=dev), torch.zeros(NH, 1, 3, device=dev))
    std = K.new_stats("dwq", B, NH, M, dev)
    K.dwq_fwd(xc_h[0], lo, hi, w_dw, b_dw, 4, 4, 1, slope, lo, hi, False, stats=std)
    _, _, mr = K.gnq_fwd(xc_h[0], lo, hi, gm_, bt_, 1e-8, lo, hi, False)
    case("k_gnq_apply_t", "gLN + fake-quant forward by a per-row code table, statistics from the producer, C=512 (codes in / out)", "hbm", 49, 1.0 * NH * n, 1.0 * NH * n, 8.0 * NH * n,
         lambda i: K.gnq_fwd(xc_h[i % sets], lo, hi, gm_, bt_, 1e-8, lo, hi, False, stats=std))
    case("k_dwq_fwd<3>", "depthwise + PReLU + fake-quant + gLN statistics forward, C=512 (codes in / out)", "hbm", 24, 1.0 * NH * n, 1.0 * NH * n, 8.0 * NH * n,
         lambda i: K.dwq_fwd(xc_h[i % sets], lo, hi, w_dw, b_dw, 4, 4, 1, slope, lo, hi, False, stats=std))
    case("k_ewq_fwd", "AddQ forward as a launch of its own, C=128 (codes + codes -> codes): the adds without a pair GEMM in front", "hbm", 2, 2.0 * NB * n, 1.0 * NB * n, 12.0 * NB * n,
         lambda i: K.ewq_fwd(xc_b[i % sets], lo, hi, xc_b2[i % sets], lo, hi, None, 1.0, 0, None, lo, hi, False))
    # round 5: the depthwise backward takes the apply pass of the GroupNormQ behind it (on load) and the rows pass of the one in front
    # (on the gx it produces): g + its own input codes + that GroupNorm's input codes in, gx out
    _, ws_rows = K.gnq_bwd_rows(xc_h[0], lo, hi, gz_h[0], gm_, bt_, mr, lo, hi, gacc)
    gacc_b = torch.zeros_like(gacc)

    def dwq_bwd_gn(i):
        after = dict(gamma=gm_, beta=bt_, mean_rstd=mr, ws=ws_rows, qmin=lo, qmax=hi, ggamma=gg, gbeta=gb2)
        before = dict(xc0=xc_h[(i + 1) % sets], qmin0=lo, qmax0=hi, gamma=gm_, beta=bt_, mean_rstd=mr, gacc=gacc_b)
        return K.dwq_bwd(xc_h[i % sets], lo, hi, w_dw, b_dw, gz_h[i % sets], 4, 4, 1, slope, lo, hi, gacc, gbb, gw_dw, after=after, before=before)
    case("k_dwq_bwd<3, GA, GB>", "depthwise + PReLU + fake-quant backward + the apply pass of the gLN behind it + the rows pass of the gLN in front, "
         "C=512 (fp32 g + 2 x codes in, fp32 gx out)", "hbm", 24, 6.0 * NH * n, 4.0 * NH * n, 28.0 * NH * n, dwq_bwd_gn)
    pgb_a, pgb_b = torch.zeros(NB, device=dev), torch.zeros(NB, device=dev)
    case("k_ewq_bwd", "AddQ backward with BOTH operands' conv output-quantizer backward fused (residual add), C=128", "hbm", 24, 14.0 * NB * n, 8.0 * NB * n,
         20.0 * NB * n, lambda i: K.ewq_bwd_p(xc_b[i % sets], lo, hi, xc_b2[i % sets], lo, hi, 1.0, gz_b[i % sets], 0, None, lo, hi, gacc, NB,
                                               prod_a=(z_b[i % sets], 0, None, pgacc, pgb_a), prod_b=(z_b2[i % sets], 0, None, pgacc2, pgb_b)))
    # round 5: the 23 AddQ of the skip sum run their backward as ONE launch (k_ewq_chain_bwd): the gradient stays in registers from the top
    # level to the bottom; per level two code words + the skip conv's z in, that conv's gz out (every level on buffers of its own: 0.9 GB,
    # nothing may be served from the Infinity Cache twice)
    NL = 23
    ch_levels = [dict(ac=_codes(NB, dev), amin=lo, amax=hi, bc=_codes(NB, dev), bmin=lo, bmax=hi, qmin=lo, qmax=hi,
                      gacc=torch.zeros_like(gacc), prod_b=(_act(NB, dev), 0, None, torch.zeros_like(pgacc), torch.zeros(NB, device=dev))) for _ in range(NL)]
    case("k_ewq_chain_bwd", "backward of the skip sum's 23 chained AddQ in one launch, each with its conv's output-quantizer backward, C=128", "hbm", 1,
         (NL * 6.0 + 4.0) * NB * n, (NL * 4.0 + 4.0) * NB * n, NL * 16.0 * NB * n, lambda i: K.add_chain_bwd(ch_levels, gz_b[i % sets]))
    gb_b = torch.zeros(NB, device=dev)
    case("k_actq_bwd", "activation fake-quant backward (bottleneck / mask convs), C=128", "hbm", 4, 8.0 * NB * n, 4.0 * NB * n, 12.0 * NB * n,
         lambda i: K.actq_bwd(z_b[i % sets], gz_b[i % sets], 0, None, 2, lo, hi, gacc, gbias=gb_b, C=NB))
    case("k_axpby", "gradient sum at a fork (decoder / encoder side; the residual forks are summed in the dgrad epilogue), C=128", "hbm", 4, 8.0 * NB * n, 4.0 * NB * n, 12.0 * NB * n,
         lambda i: K.axpby(gz_b[i % sets], gz_b2[i % sets], 1.0))
    # ---- what is left of the two-pass gLN backward (round 5): the rows pass of the gLN behind the depthwise layer (its apply pass runs
    # inside k_dwq_bwd), the apply pass of the gLN in front of it (its rows pass runs there) with conv1's epilogue backward fused, and
    # the one gLN of the bottleneck with both passes of its own (two launches, timed as a whole; never picked as `roofline`)
    case("k_gnq_bwd_rows", "gLN+fq backward, rows pass alone (row sums + range partials; codes + fp32 g in), C=512", "hbm", 25, 5.0 * NH * n, 16.0 * B * NH, 8.0 * NH * n,
         lambda i: K.gnq_bwd_rows(xc_h[i % sets], lo, hi, gz_h[i % sets], gm_, bt_, mr, lo, hi, gacc))
    case("k_gnq_bwd_apply<true>", "gLN+fq backward, apply pass alone with the producer conv's STE/PReLU/range/bias backward fused, C=512", "hbm", 24,
         9.0 * NH * n, 4.0 * NH * n, 12.0 * NH * n,
         lambda i: K.gnq_bwd_apply(xc_h[i % sets], lo, hi, gz_h[i % sets], gm_, bt_, mr, lo, hi, ws_rows, gg, gb2, producer=(z_h[i % sets], 1, slope, pgacc, gbb)))
    case("k_gnq_bwd_rows+apply<false>", "gLN+fq backward (2 passes, plain: the bottleneck's gLN), C=512", "hbm", 1, 10.0 * NH * n, 4.0 * NH * n, 16.0 * NH * n,
         lambda i: K.gnq_bwd(xc_h[i % sets], lo, hi, gz_h[i % sets], gm_, bt_, mr, lo, hi, gacc, gg, gb2), group=True)
    return cases


def step_traffic_bytes(cases):
    """algorithmic HBM bytes of ONE step summed over the launch list above (the TCN stack: every kernel x its launches per step), plus the
    passes outside the stack, in bytes per N_H = 8 x 512 x 3999 positions (the [16, 512, M] tensors of the masking product / decoder /
    residual block count twice):
      forward  ~55 B: student encoder z + its quantizer 10, MulQ on codes 5, two decoder launches on codes 4, residual encoder 8, its
               Sub 12, teacher encoder 4 + masking product formed inside the decoder 12;
      backward ~140 B: decoder dgrad 8 + wgrad from codes 2 (twice: main and residual), Sub backward 26 and its negation 16, residual encoder
               dgrad 8 + wgrad 8, decoder dgrad with the fork's addend 16, MulQ backward (+ mask conv epilogue) 31, encoder-side quantizer
               backward + fork sum 24;
    and the 7 x 4 B x 5.13 M of clip + Adam.  (Round 1 / early round 2 counted 14 x 4 B here: the un-fused chain really moved ~300 B.)"""
    stack = sum(c["launches"] * c["bytes"] for c in cases)
    outside = 195.0 * NH * N + 7 * 4.0 * 5.13e6
    return stack + outside


def time_case(case, iters=24, warm=3):
    """average launch duration in ms, HIP events on the stream the kernels are launched on (torch's current stream); the launches are
    recorded into one hipGraph first (a Python launch costs ~15 us: it would hide the short kernels)"""
    for i in range(warm):
        case["fn"](i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            case["fn"](i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters)


def summarize(cases, times_ms):
    """per-kernel (one kernel symbol = one entry, launch-weighted over its shapes), sorted by time per step"""
    groups = {}
    for c, ms in zip(cases, times_ms):
        g = groups.setdefault(c["kernel"], dict(kernel=c["kernel"], bound=c["bound"], group=c["group"], launches=0, ms_step=0.0, flops=0.0,
                                                 bytes=0.0, survey=0.0, shapes=[]))
        g["launches"] += c["launches"]
        g["ms_step"] += c["launches"] * ms
        g["flops"] += c["launches"] * c["flops"]
        g["bytes"] += c["launches"] * c["bytes"]
        g["survey"] += c["launches"] * c["survey"]
        g["shapes"].append({"label": c["label"], "launches_per_step": c["launches"], "launch_us": round(ms * 1e3, 2),
                            "algorithmic_MB": round(c["bytes"] / 1e6, 1)})
    return sorted(groups.values(), key=lambda g: -g["ms_step"])


def roofline_object(g, traffic=None):
    """the `roofline` JSON object of one kernel: launch-weighted averages over its shapes, priced by priced()"""
    n = g["launches"]
    ms = g["ms_step"] / n
    dtype, prod = ISSUED.get(g["kernel"], ("f32", 1))
    out = {"kernel": g["kernel"], "launch_us": round(ms * 1e3, 2), "launches_per_step": n,
           "ms_per_step": round(g["ms_step"], 3), "algorithmic_bytes_per_launch": round(g["bytes"] / n),
           "survey_convention_GBps": round(g["survey"] / n / (ms * 1e-3) / 1e9, 1)}
    vc = valu_counts()
    valu = vc.get(VALU_ALIAS.get(g["kernel"], g["kernel"])) if len(g["shapes"]) == 1 else None      # (a count belongs to ONE launch shape)
    out.update(priced(g["flops"] / n, prod, dtype, g["bytes"] / n, ms * 1e3, valu))
    out["traffic"] = traffic
    out["shapes"] = g["shapes"]
    if g["group"]:
        out["group_of_launches"] = True
    return out


def build_other(dev, sets=2):
    """the kernels named by the `roofline` objects of cfg 3 / 4 / 5 (bench.py) at their live shapes, for the same PMC passes
    (tools/roofline_probe.py --set other): same dict layout as build()"""
    cases = []

    def case(kernel, label, bound, launches, rd, wr, fn, flops=0.0):
        cases.append(dict(kernel=kernel, label=label, bound=bound, launches=launches, flops=flops, rd=float(rd), wr=float(wr), bytes=float(rd + wr),
                          survey=float(rd + wr), fn=fn, group=False))
    R = range(sets)
    # cfg 3: BiLSTM recurrence of the intra-chunk path, S = 250 steps x 194 sequences, H = 128 (saves its gate activations for the BPTT)
    S, Bq, H = 250, 194, 128
    pre = [torch.randn(S, Bq, 8 * H, device=dev) * 0.1 for _ in R]
    whh, bhh = torch.randn(2, 4 * H, H, device=dev) * 0.05, torch.zeros(2, 4 * H, device=dev)
    n = S * Bq
    case("k_lstm_fwd_st<128>", "cfg 3: BiLSTM recurrence 250 steps x 194 sequences (both directions), saving gates and cell states", "mfma", 24,
         4.0 * n * 8 * H, 4.0 * n * (2 * H + 8 * H + 4 * H), lambda i: K.lstm_fwd(pre[i % sets], whh, bhh, S, Bq, H), flops=2.0 * 2 * 4 * H * H * n)
    # cfg 4: weight gradient of the student's coded feed-forward linear, 8 500 rows (250 x 34 chunks) x 256 -> 1024
    rows, Ci, Co = 8500, 256, 1024
    gz = [torch.randn(rows, Co, device=dev) for _ in R]
    xc = [torch.randint(0, 256, (rows, Ci), device=dev, dtype=torch.uint8) for _ in R]
    lo, hi = torch.tensor([-1.0], device=dev), torch.tensor([1.0], device=dev)
    gw = torch.zeros(Co, Ci, device=dev)
    wc4 = K.wq_codes(torch.randn(Co, Ci, 1, device=dev) * 0.05, -torch.ones(Co, 1, 1, device=dev) * 0.2, torch.ones(Co, 1, 1, device=dev) * 0.2)
    case("k_gemm_x3", "cfg 4: coded data gradient, dgrad 8500 x 1024 -> 256 (gz fp32, weight codes int8)", "mfma", 96,
         4.0 * rows * Co + Co * Ci, 4.0 * rows * Ci, lambda i: K.qrow_bwd_x(gz[i % sets], wc4), flops=2.0 * rows * Ci * Co)
    # cfg 5: six-product pointwise GEMM over the frames of the level-0 rewrite conv, 4 x (144 -> 96) x 110250
    Bh, Kk, Cq, Mh = 4, 144, 96, 110250
    f = [K.empty_act((Bh, Kk, Mh), dev).normal_() for _ in R]
    w, b = torch.randn(Cq, Kk, 1, device=dev) * 0.1, torch.zeros(Cq, device=dev)
    case("k_qgemm<3>", "cfg 5: frame GEMM of the level-0 rewrite conv 4 x (48 x 3 -> 96) x 110250, six bf16 products", "mfma", 67,
         4.0 * Bh * Kk * Mh, 4.0 * Bh * Cq * Mh, lambda i: K.pwconv_fwd(f[i % sets], w, b, six=True), flops=2.0 * Bh * Cq * Kk * Mh)
    # cfg 5: streaming attention, spectrogram branch self-attention 4 x 8 heads x 3448^2 x 64: float operands (teacher), coded (student)
    Ba, nh, L, hd = 4, 8, 3448, 64
    E = nh * hd
    q, k, v = ([torch.randn(Ba, L, E, device=dev) * 0.3 for _ in R] for _ in range(3))
    qc, kc, vc = ([torch.randint(0, 256, (Ba, L, E), device=dev, dtype=torch.uint8) for _ in R] for _ in range(3))
    rng = [(torch.tensor([-0.9], device=dev), torch.tensor([0.8], device=dev)) for _ in range(3)]
    na = Ba * L * E
    fa = 4.0 * L * L * hd * Ba * nh
    case("k_attn_long_fwd_x3<64>", "cfg 5: attention forward, float operands split in bf16 pieces (q, k, v read once, o + stats written)", "mfma", 10,
         4.0 * 3 * na, 4.0 * na + 8.0 * Ba * nh * L, lambda i: K.attn_long_fwd(q[i % sets], k[i % sets], v[i % sets], nh, True), flops=fa)
    case("k_attn_long_fwd_c<64>", "cfg 5: attention forward on the u8 codes of q, k, v", "mfma", 10,
         3.0 * na, 4.0 * na + 8.0 * Ba * nh * L, lambda i: K.attn_long_fwd_c(qc[i % sets], kc[i % sets], vc[i % sets], rng, nh, True), flops=fa)
    return cases
