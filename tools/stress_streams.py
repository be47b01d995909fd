"""Kernels of the waveform side (and a few others) launched on the main stream WHILE a second stream runs the float teacher (or another
background, BG=teacher|mm|conv|ola|mulq|dwq): every launch must return what it returns alone.  Prints, per case, in how many of 12 rounds it
did not.  SKIP=name[,name] replaces teacher kernels by cached results (to find an aggressor), DIAG=case prints where a case went wrong.
History: a four-frames-per-lane decoder kernel (k_ola_convtr4, removed) passed every test alone and failed here in 8-12 rounds of 12 -- a few
hundred outputs per launch off by about one product term whenever its workgroups were scheduled between the teacher's k_tgemm workgroups;
cause not found; the one-frame-per-lane kernel with the same operand modes passes (docs/history/DESIGN_rounds_1-5.md 9).
    python tools/stress_streams.py [case ...]"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd import kernels as K
from fqss_amd.data import synth_batch
from fqss_amd.runtime import KDTrainStep
from fqss_amd.smoke import build_pair


def main(argv=()):
    dev = "cuda"
    x, tgt = synth_batch(8, 32000, seed=0, device=dev)
    model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)
    step = KDTrainStep(model, fmodel, lr=0.0)
    step(x, tgt)
    B, S, C, M, T = 8, 2, 512, 3999, 32000
    lo, hi = torch.tensor([-1.0], device=dev), torch.tensor([1.5], device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    codes2 = K.empty_codes((B * S, C, M), dev); codes2.random_(0, 256, generator=g)
    codes1 = K.empty_codes((B, C, M), dev); codes1.random_(0, 256, generator=g)
    act2 = K.empty_act((B * S, C, M), dev); act2.normal_(generator=g)
    act1 = K.empty_act((B, C, M), dev); act1.normal_(generator=g)
    sig2 = torch.randn(B * S, 1, T, device=dev, generator=g)
    sig_enc = torch.randn(B, 2, T, device=dev, generator=g)
    w_dec = torch.randn(C, 1, 16, device=dev, generator=g) * 0.1
    w_enc = torch.randn(C, 2, 16, device=dev, generator=g) * 0.1
    gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
    def gw(fn):
        o = torch.zeros(C, 1, 16, device=dev); fn(o); return o
    w_dw, b_dw = torch.randn(C, 1, 3, device=dev, generator=g), torch.randn(C, device=dev, generator=g) * 0.1
    slope = torch.tensor([0.25], device=dev)
    gbb, gw_dw = torch.zeros(C, device=dev), torch.zeros(C, 1, 3, device=dev)
    pz2 = K.empty_act((B, S * C, M), dev); pz2.normal_(generator=g)
    pga = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
    # platform probe (tools/ubench/longsum.hip, built on the spot): per-lane accumulators that live as long as the kernel, nothing shared
    import ctypes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tools", "ubench", "liblongsum.so")       # built by __graft_entry__.build() (make -C tools/ubench)
    if not os.path.exists(so):
        raise RuntimeError(f"{so} missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
    lib = ctypes.CDLL(so)
    lib.probe_longsum.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

    def probe(blocks=2048, rounds=1):
        o = torch.empty(blocks * 256 * 3, device=dev)
        lib.probe_longsum(xbig.data_ptr(), o.data_ptr(), xbig.numel(), blocks, rounds, torch.cuda.current_stream().cuda_stream)
        return o
    xbig = torch.randn(32 * 1024 * 1024, device=dev, generator=g)
    CASES = {
        "dwq_bwd": lambda: K.dwq_bwd(codes1, lo, hi, w_dw, b_dw, act1, 4, 4, 1, slope, lo, hi, gacc, gbb, gw_dw),
        "ola_f32_ragged": lambda: K.ola_convtr_fwd(act2[..., :3997], w_dec, 8),

        "ola_q": lambda: K.ola_convtr_fwd_q(codes2, lo, hi, w_dec, 8),
        "ola_f32": lambda: K.ola_convtr_fwd(act2, w_dec, 8),
        "ola_mul": lambda: K.ola_convtr_mul_fwd(act2.view(B, S, C, M), act1, w_dec, 8),
        "mulq_fwd": lambda: K.mulq_fwd(codes2.view(B, S, C, M), lo, hi, codes1, lo, hi, lo, hi, False)[1],
        "mulq_bwd": lambda: K.mulq_bwd(codes2.view(B, S, C, M), lo, hi, codes1, lo, hi, act2.view(B, S, C, M), lo, hi, gacc)[0],
        "mulq_bwd_bias": lambda: (lambda pb: (K.mulq_bwd(codes2.view(B, S, C, M), lo, hi, codes1, lo, hi, act2.view(B, S, C, M), lo, hi, gacc,
                                                         prod=(pz2, K.ACT_RELU, None, pga, pb)), pb)[1])(torch.zeros(S * C, device=dev)),
        "mulq_bwd_prod_out": lambda: K.mulq_bwd(codes2.view(B, S, C, M), lo, hi, codes1, lo, hi, act2.view(B, S, C, M), lo, hi, gacc,
                                                prod=(pz2, K.ACT_RELU, None, pga, torch.zeros(S * C, device=dev)))[0],
        "conv4_ci1": lambda: K.frames_conv_fwd(sig2, w_dec.view(C, 1, 16), 8),
        "conv4_ci1_add": lambda: K.frames_conv_fwd(sig2, w_dec.view(C, 1, 16), 8, add=act2),
        "conv_ci2": lambda: K.frames_conv_fwd(sig_enc, w_enc, 8),
        "wgrad_f32": lambda: gw(lambda o: K.frames_wgrad(act2, sig2, o, 8)),
        "wgrad_q": lambda: gw(lambda o: K.frames_wgrad1_q(codes2, lo, hi, sig2, o, 8)),
    }
    BG = os.environ.get("BG", "teacher")
    A_, B_ = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
    xt = x.unsqueeze(1) if x.dim() == 2 else x
    wt_ = fmodel.encoder.weight.detach()
    def background():
        if BG == "teacher":
            for _ in range(2): step.teacher(x)
        elif BG == "mm":
            for _ in range(6): torch.mm(A_, B_)
        elif BG == "conv":
            for _ in range(40): K.frames_conv_fwd(xt, wt_, 8)
        elif BG == "ola":
            for _ in range(40): K.ola_convtr_mul_fwd(act2.view(B, S, C, M), act1, w_dec, 8)
        elif BG == "mulq":
            for _ in range(40): K.mulq_fwd(codes2.view(B, S, C, M), lo, hi, codes1, lo, hi, lo, hi, False)
        elif BG == "dwq":
            for _ in range(20): K.dwq_bwd(codes1, lo, hi, w_dw, b_dw, act1, 4, 4, 1, slope, lo, hi, gacc, gbb, gw_dw)
    SKIP = os.environ.get("SKIP", "")
    if SKIP:
        for nm in SKIP.split(","):
            real = getattr(K, nm)
            cache = {}
            def mk(real, nm):
                def f(*a, **k):
                    key = (nm, len(cache.setdefault(nm, [])) if False else tuple(getattr(t, "shape", None) for t in a[:2]))
                    if key not in cache:
                        cache[key] = real(*a, **k)
                    return cache[key]
                return f
            setattr(K, nm, mk(real, nm))
        step.teacher(x); torch.cuda.synchronize()      # fill the caches
    if probe is not None:
        CASES["probe_longsum"] = lambda: probe(2048, 1)
        CASES["probe_longsum_x4"] = lambda: probe(512, 4)
    names = list(argv) or list(CASES)
    w_dec0, act20, sig20, codes20 = w_dec.clone(), act2.clone(), sig2.clone(), codes2.clone()
    ref = {k: CASES[k]().clone() for k in names}
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    bad = {k: 0 for k in names}
    for it in range(12):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            background()
        outs = {k: CASES[k]() for k in names}
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for k in names:
            if k.startswith("wgrad") or k == "mulq_bwd_bias":
                ok = torch.allclose(outs[k], ref[k], rtol=1e-4, atol=1e-4 * float(ref[k].abs().max()))
            else:
                ok = torch.equal(outs[k], ref[k])
            bad[k] += 0 if ok else 1
    print({k: v for k, v in bad.items()})
    torch.cuda.synchronize()
    again = {k: CASES[k]() for k in names}
    torch.cuda.synchronize()
    print("alone again == ref:", {k: bool(torch.equal(again[k], ref[k])) for k in names if not k.startswith("wgrad")})
    print("w_dec unchanged:", torch.equal(w_dec, w_dec0), "act2", torch.equal(act2, act20), "sig2", torch.equal(sig2, sig20), "codes2", torch.equal(codes2, codes20))
    if os.environ.get("DIAG"):
        k = os.environ["DIAG"]
        for it in range(4):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                background()
            o = CASES[k]()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if k == "mulq_bwd_bias":
                d = (o - ref[k])
                rel = d.abs() / ref[k].abs().max()
                big = (rel > 1e-5).nonzero().flatten()
                print(k, "channels off", big.numel(), "max rel", float(rel.max()), "first idx", big[:12].tolist(), "d", d[big[:6]].tolist(), "ref", ref[k][big[:6]].tolist())
                continue
            badm = (o != ref[k])
            nb = int(badm.sum())
            if nb:
                idx = badm.nonzero()
                flat = (o.reshape(-1) != ref[k].reshape(-1)).nonzero().flatten()
                d = (flat[1:] - flat[:-1])
                print(k, "bad", nb, "first", idx[:6].tolist(), "flat gaps histogram", torch.unique(d, return_counts=True)[0][:8].tolist(), torch.unique(d, return_counts=True)[1][:8].tolist())
                print("   values", o[badm][:6].tolist(), "ref", ref[k][badm][:6].tolist())
    return bad


if __name__ == "__main__":
    main(sys.argv[1:])
