"""Two-stream investigation of docs/history/DESIGN_rounds_1-5.md 9.  Runs the DIAGNOSTIC twin of the library
(`make -C fqss_amd/csrc diag`: k_mulq_bwd with the round-2 "branchy" bias sums, a shadow sum in select form inside the same wave,
per-lane partials and per-wave placement / clock records) as the victim on the main stream while a background runs on a second
stream, and reports for every launch that differs from the quiet launch: which workgroup / wave / lanes, whether the shadow sum
of the SAME wave differs too (then the inputs differed, not the accumulation), by how many terms, where the wave ran (XCC, CU,
SIMD) and how long it took against the median wave.

    FQSS_LIB=fqss_amd/csrc/diag/libfqss_diag.so python tools/diag_streams.py [rounds]      BG=teacher|tgemm1|tgemm0|tdw|mm|none
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd import _lib, kernels as K          # noqa: E402
from fqss_amd.data import synth_batch              # noqa: E402
from fqss_amd.runtime import KDTrainStep           # noqa: E402
from fqss_amd.smoke import build_pair              # noqa: E402


def main(rounds=40):
    dev = "cuda"
    lib = _lib.load()
    assert hasattr(lib, "fqss_diag_set"), "run with FQSS_LIB=<repo>/fqss_amd/csrc/diag/libfqss_diag.so (make -C fqss_amd/csrc diag)"
    lib.fqss_diag_set.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    B, S, C, M = 8, 2, 512, 3999
    x, tgt = synth_batch(8, 32000, seed=0, device=dev)
    model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)
    step = KDTrainStep(model, fmodel, lr=0.0)
    step(x, tgt)
    lo, hi = torch.tensor([-1.0], device=dev), torch.tensor([1.5], device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    codes2 = K.empty_codes((B * S, C, M), dev); codes2.random_(0, 256, generator=g)
    codes1 = K.empty_codes((B, C, M), dev); codes1.random_(0, 256, generator=g)
    act2 = K.empty_act((B * S, C, M), dev); act2.normal_(generator=g)
    pz2 = K.empty_act((B, S * C, M), dev); pz2.normal_(generator=g)
    gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
    pga = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
    nwg = 4 * 512
    lane = torch.zeros(nwg, 256, 4, device=dev)
    wave = torch.zeros(nwg, 4, 4, dtype=torch.int64, device=dev)
    assert lib.fqss_diag_set(lane.data_ptr(), wave.data_ptr()) == 0

    def victim():
        pb = torch.zeros(S * C, device=dev)
        gm, gf = K.mulq_bwd(codes2.view(B, S, C, M), lo, hi, codes1, lo, hi, act2.view(B, S, C, M), lo, hi, gacc,
                            prod=(pz2, K.ACT_RELU, None, pga, pb))
        return pb, gm, gf

    BG = os.environ.get("BG", "teacher")
    A_, B_ = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
    h128 = K.empty_act((B, 128, M), dev); h128.normal_(generator=g)
    h512 = K.empty_act((B, 512, M), dev); h512.normal_(generator=g)
    blk = fmodel.masker.TCN[0]
    sb = blk.shared_block
    step.teacher(x)
    pl = step.teacher._planes
    st = K.tstat_buffer(3, B, dev)
    K.tstats(h512, st[0])

    def background():
        if BG == "teacher":
            for _ in range(2):
                step.teacher(x)
        elif BG == "tgemm0":     # T1: 128 -> 512 + PReLU + statistics (no GroupNorm prologue: 61,440 B of LDS)
            for _ in range(40):
                K.tgemm(pl["blocks"][0][0], h128, sb[0].bias, act=K.ACT_PRELU, slope=sb[1].weight, stats_out=st[1])
        elif BG == "tgemm1":     # T3: GroupNorm prologue + 512 -> 256 (65,536 B of LDS)
            for _ in range(25):
                K.tgemm(pl["blocks"][0][1], h512, pl["blocks"][0][2], pro=1, pro_stats=st[0], pro_gamma=sb[5].weight, pro_beta=sb[5].bias,
                        pro_eps=sb[5].eps, M1=128, r1=h128, r2=None)
        elif BG == "tdw":
            for _ in range(40):
                K.tdw(h512, st[0], sb[2].weight, sb[2].bias, sb[2].eps, sb[3].weight, sb[3].bias, sb[4].weight, st[2], sb[3].dilation[0],
                      sb[3].padding[0])
        elif BG == "mm":
            for _ in range(6):
                torch.mm(A_, B_)

    refs = []
    for _ in range(3):
        pb, gm, gf = victim()
        torch.cuda.synchronize()
        refs.append((pb.clone(), gm.clone(), lane.clone(), wave.clone()))
    for r in refs[1:]:
        print("quiet launches: lanes equal", bool(torch.equal(r[2], refs[0][2])), "| gmask equal", bool(torch.equal(r[1], refs[0][1])),
              "| bias max rel", float((r[0] - refs[0][0]).abs().max() / refs[0][0].abs().max()))
    ref_pb, ref_gm, ref_lane, ref_wave = refs[0]
    print("quiet: branchy sum == shadow select sum in every lane:", bool(torch.equal(ref_lane[..., 0:2], ref_lane[..., 2:4])))
    dur0 = (ref_wave[..., 2] - ref_wave[..., 1]).float()
    print("quiet wave clocks: median %.0f  p99 %.0f  max %.0f" % (dur0.median(), dur0.flatten().kthvalue(int(dur0.numel() * 0.99))[0], dur0.max()))

    side = torch.cuda.Stream()
    nbad = 0
    for it in range(rounds):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            background()
        pb, gm, gf = victim()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        d = lane != ref_lane
        if not d.any() and torch.equal(gm, ref_gm):
            continue
        nbad += 1
        dur = (wave[..., 2] - wave[..., 1]).float()
        idx = d.any(-1).nonzero()
        wgs = sorted({int(i[0]) for i in idx})
        print(f"round {it}: {int(d.any(-1).sum())} lanes differ in {len(wgs)} workgroups; gmask equal {bool(torch.equal(gm, ref_gm))}; "
              f"wave clocks median {dur.median():.0f} max {dur.max():.0f}")
        for wg in wgs[:6]:
            for wv in range(4):
                ls = [int(i[1]) - 64 * wv for i in idx if int(i[0]) == wg and 64 * wv <= int(i[1]) < 64 * wv + 64]
                if not ls:
                    continue
                t = 64 * wv + ls[0]
                hw = int(wave[wg, wv, 0])
                print(f"   wg {wg} (x {wg % 4}, channel {wg // 4}) wave {wv} lanes {ls}: branchy {lane[wg, t, :2].tolist()} shadow {lane[wg, t, 2:].tolist()} "
                      f"quiet {ref_lane[wg, t, :2].tolist()} | shadow differs too: {bool((lane[wg, t, 2:] != ref_lane[wg, t, 2:]).any())} | "
                      f"xcc {hw >> 32} hw_id {hw & 0xffffffff:#x} (cu {(hw >> 8) & 15} sh {(hw >> 12) & 1} se {(hw >> 13) & 7} simd {(hw >> 4) & 3} wave {hw & 15}) "
                      f"clocks {int(dur[wg, wv])} (quiet {int(dur0[wg, wv])})")
    print(f"BG={BG}: {nbad} of {rounds} rounds differ")
    assert lib.fqss_diag_set(None, None) == 0


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
