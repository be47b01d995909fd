#!/usr/bin/env python3
"""Convergence fixture (north_star "SI-SDR within 0.1 dB of the reference"): the REAL reference
(/root/reference, imported through tools/ref_shim.py -- build container only) trains the tiny ConvTasNetQ of tiny_step.npz with
the KD step of mysystem.py:124-151 (Adam 1e-3, clip 5.0) over a STREAM of never-repeating seeded batches
(fqss_amd.data.synth_batch_2band: two spectrally distinct speakers), N_STEPS steps, under several CPU configurations that only
change the fp32 summation order (threads 1 / 8, mkldnn on / off).  Stored: per-step loss and mean student SI-SDR of every run,
and the reference's OWN spread between those runs over the last 50 steps -- the floor any other backend is compared against.
    python tools/make_goldens_long.py            -> tests/golden/tiny_train_long.npz
    python tools/make_goldens_long.py dptnet     -> dpt_train_long.npz
Round 4: the same gate at the REAL model size and for the two remaining families --
    python tools/make_goldens_long.py cfg1       -> cfg1_train_long.npz   FULL-SIZE ConvTasNetQ (5.1 M parameters, cfg1_fill weights),
                                                    B = 2, T = 8000, 300 steps, six CPU configurations
    python tools/make_goldens_long.py cfg1_lr1e-4 -> cfg1_train_long_lr1e-4.npz   (round 5) the same at lr 1e-4, four CPU configurations
    python tools/make_goldens_long.py sepformer  -> sep_train_long.npz    tiny SepformerQ of sep_tiny_step.npz, B = 1 (the speechbrain
                                                    env's per-sample objective == the asteroid objective at B = 1), Adam 1.5e-4, clip 5
    python tools/make_goldens_long.py htdemucs   -> hd_train_long.npz     tiny HTDemucsQ of hd_tiny_step.npz, solver.py:333-366 l1 + SDR-weighted
                                                    l1 KD, Adam 3e-4, NO clipping (htdemucs.yaml:77-84), stereo two-stem stream"""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import make_goldens_htdemucs as MH  # noqa: E402  (FIRST: it hands the shim's demucs.spec its spectro / ispectro before htdemucsq.py binds them)
import make_goldens as MG  # noqa: E402  (installs the shim, imports the reference)

import torch  # noqa: E402

from fqss_amd.data import synth_batch_2band, synth_stems  # noqa: E402  (data generators only: no kernels)

N_STEPS, B, T, SEED0 = 400, 4, 1600, 5000
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def build(which):
    """(student, teacher, lr) from the initial weights of the model family's tiny step fixture"""
    torch.manual_seed(0)
    if which == "convtasnet":
        g = np.load(os.path.join(GOLD, "tiny_step.npz"))
        kw = dict(n_spks=2, kernel_size=16, stride=8, n_filters=32, bn_chan=16, hid_chan=32, n_blocks=2, n_repeats=1)
        model, lr = MG.ConvTasNetQ(**kw), 1e-3
    elif which == "cfg1":
        model = MG.ConvTasNetQ(n_spks=2, kernel_size=16, stride=8)
        fmodel = copy.deepcopy(model)
        model = MG.quantize_model(model, MG.QCFG)
        MG.cfg1_fill(fmodel, "T."); MG.cfg1_fill(model, "S.")
        model.train(); fmodel.eval()
        return model, fmodel, 1e-3
    elif which == "sepformer":
        import make_goldens_sepformer as MS
        g = np.load(os.path.join(GOLD, "sep_tiny_step.npz"))
        model = MS.RS.SepformerQ(**MS.TINY_KW)
        model.masker = MS.RS.MaskGenerator(MS.TINY_KW["n_spks"], MS.TINY_KW["n_filters"], n_repeats=MS.TINY_KW["n_repeats"],
                                           n_heads=MS.TINY_KW["n_heads"], chunk_size=MS.TINY_KW["chunk_size"], n_ffn=MS.TINY_FFN)
        lr = MS.LR
    elif which == "htdemucs":
        g = np.load(os.path.join(GOLD, "hd_tiny_step.npz"))
        model = MH.HTDemucsQ(**MH.TINY_KW)
        fmodel = copy.deepcopy(model)
        model = MH.quantize_model(model, MH.QCFG)
        model.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd0.")})
        fmodel.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fsd.")})
        model.train(); fmodel.eval()
        return model, fmodel, 3e-4
    else:
        import make_goldens_dptnet as MD
        g = np.load(os.path.join(GOLD, "dpt_tiny_step.npz"))
        model, lr = MD.RD.DPTNetQ(**MD.TINY_KW), 4e-4
    fmodel = copy.deepcopy(model)
    model = MG.quantize_model(model, MG.QCFG)
    model.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd0.")})
    fmodel.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fsd.")})
    model.train(); fmodel.eval()
    return model, fmodel, lr


def run(threads, mkldnn, which="convtasnet", N_STEPS=N_STEPS, B=B, T=T, lr_override=None):
    torch.set_num_threads(threads)
    model, fmodel, lr = build(which)
    if lr_override is not None:
        lr = lr_override
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    loss_t, sdr_t, tsdr_t = [], [], []
    with torch.backends.mkldnn.flags(enabled=mkldnn):
        for step in range(N_STEPS):
            opt.zero_grad()
            if which == "htdemucs":
                mix, src = synth_stems(B, 2, 2, T, seed=SEED0 + step)
                est, fest, w, task, kd, loss = MH.kd_step(model, fmodel, mix, src)
                loss.backward()                          # htdemucs.yaml: no gradient clipping
                opt.step()
                with torch.no_grad():
                    sdrq, sdrt = MH.new_sdr(src, est.detach()).mean(), MH.new_sdr(src, fest).mean()
                loss_t.append(float(loss)); sdr_t.append(float(sdrq)); tsdr_t.append(float(sdrt))
            else:
                x, tgt = synth_batch_2band(B, T, seed=SEED0 + step)
                est, fest, w, kd, task, loss, sdrs, sdrqs = MG.common_step(model, fmodel, x, tgt)
                loss.backward()
                torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
                opt.step()
                loss_t.append(float(loss)); sdr_t.append(float(-sdrqs.mean())); tsdr_t.append(float(-sdrs.mean()))
            if step % 50 == 0 or step == N_STEPS - 1:
                print(f"threads {threads} mkldnn {mkldnn} step {step} loss {float(loss):.3f} si-sdr {sdr_t[-1]:.3f} (teacher {tsdr_t[-1]:.3f})", flush=True)
    return np.array(loss_t, np.float32), np.array(sdr_t, np.float32), np.array(tsdr_t, np.float32)


def main(which="convtasnet", n_steps=N_STEPS, batch=B, samples=T, fname="tiny_train_long.npz", variants=None, lr=None):
    variants = variants or [(1, True), (8, True), (1, False), (4, False)]
    d = dict(n_steps=np.int64(n_steps), batch=np.int64(batch), samples=np.int64(samples), seed0=np.int64(SEED0),
             variants=np.array([f"threads={t},mkldnn={m}" for t, m in variants]))
    if lr is not None:
        d["lr"] = np.float64(lr)
    L, S = [], []
    for t, m in variants:
        lo, sd, ts = run(t, m, which, n_steps, batch, samples, lr)
        L.append(lo); S.append(sd)
        d["teacher_sisdr"] = ts
    d["loss"], d["sisdr"] = np.stack(L), np.stack(S)
    tail = d["sisdr"][:, -50:].mean(1)
    d["tail_sisdr"] = tail
    d["spread_db"] = np.float32(tail.max() - tail.min())
    d["tail_loss"] = d["loss"][:, -50:].mean(1)
    print("last-50 mean SI-SDR per variant", tail, "spread", d["spread_db"], "first-50", d["sisdr"][:, :50].mean(1))
    np.savez_compressed(os.path.join(GOLD, fname), **d)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dptnet":      # reduced length: the reference's LSTM / attention layers are slow on the CPU
        main("dptnet", 160, 2, 400, "dpt_train_long.npz")
    elif len(sys.argv) > 1 and sys.argv[1] == "cfg1":      # ~1 s per reference step on 8 cores: three configurations, 300 steps each
        main("cfg1", 300, 2, 8000, "cfg1_train_long.npz", variants=[(8, True), (8, False), (4, True), (2, True), (6, True), (4, False)])
    elif len(sys.argv) > 1 and sys.argv[1] == "cfg1_lr1e-4":
        # round 5 (VERDICT r04 next #5d): the same full-size model at lr 1e-4, a regime in which the quantized student's SI-SDR keeps
        # RISING through step 300 (at the env's 1e-3 it drifts down once every quantizer is live): the "within 0.1 dB" claim where it can
        # be resolved.  Four configurations (threads 8 / 4, oneDNN on / off)
        main("cfg1", 300, 2, 8000, "cfg1_train_long_lr1e-4.npz", variants=[(8, True), (8, False), (4, True), (4, False)], lr=1e-4)
    elif len(sys.argv) > 1 and sys.argv[1] == "sepformer":
        main("sepformer", 200, 1, 800, "sep_train_long.npz", variants=[(8, True), (1, True), (4, False)])
    elif len(sys.argv) > 1 and sys.argv[1] == "htdemucs":
        main("htdemucs", 150, 2, 2100, "hd_train_long.npz", variants=[(8, True), (1, True), (4, False)])
    else:
        main()
