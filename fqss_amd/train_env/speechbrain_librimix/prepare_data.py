"""Data side of the speechbrain env on MI355X.

Reference: train_env/speechbrain_librimix/prepare_data.py:14-230 (`prepare_librimix` lists `<data_folder>/wav16k/min/<set>/{mix_clean |
mix_both, s1, s2[, s3], noise}` into `<save_folder>/libri{2,3}mix_<set>.csv`) and speechbrain_librimix_trainer.py:482-573 (`dataio_prep`: a
DynamicItemDataset per CSV whose items are the whole utterances, each resampled by torchaudio on the CPU) + :40-97, 262-326 (train-time
augmentation inside `compute_forward`: speed perturbation per source, re-mixing, WHAM noise, a random `training_signal_len` cut).

Here the CSVs have the same names and columns (speechbrain and the reference's tools read them), `SbLibriMix.batch()` reads the clips of a
batch on the host and does everything else on the device on the caller's stream -- the 16 -> 8 kHz resampling of all clips in one launch,
speed perturbation as one more polyphase resampling per source (speechbrain's SpeedPerturb IS a resampler: `Resample(orig_freq,
orig_freq * speed // 100)`), the re-mix, the cut -- so that fqss_amd.loader.Prefetcher can run it a batch ahead of the step.
speechbrain (0.5.14: SpeedPerturb, PaddedBatch) is third party and absent: its published behaviour is restated, parity unpinned."""
import csv
import os

import numpy as np
import torch

from ... import kernels as K
from ..asteroid_librimix.librimix_dataset import read_wav

SETS = ("train-360", "dev", "test")


def prepare_librimix(datapath, savepath, n_spks=2, skip_prep=False, librimix_addnoise=False, version="wav16k/min/", set_types=SETS):
    """write `<savepath>/libri<n>mix_<set>.csv` for every set (reference prepare_data.py:14-140: same file names, same columns)"""
    if skip_prep:
        return
    if "Libri" not in datapath:
        raise ValueError("Unsupported Dataset")
    if n_spks not in (2, 3):
        raise ValueError("Unsupported Number of Speakers")
    assert f"Libri{n_spks}Mix" in datapath, "Inconsistent number of speakers and datapath"
    os.makedirs(savepath, exist_ok=True)
    keys = ["mix"] + [f"s{i + 1}" for i in range(n_spks)] + ["noise"]
    cols = ["ID", "duration"] + [f"{k}_wav{suffix}" for k in keys for suffix in ("", "_format", "_opts")]
    for set_type in set_types:
        root = os.path.join(datapath, version, set_type)
        folders = {"mix": "mix_both/" if librimix_addnoise else "mix_clean/", "noise": "noise/", **{f"s{i + 1}": f"s{i + 1}/" for i in range(n_spks)}}
        if not os.path.isdir(os.path.join(root, folders["mix"])):
            continue                     # a set that is not on disk (the reference would raise; a smoke tree has train + dev only)
        with open(os.path.join(savepath, f"libri{n_spks}mix_{set_type}.csv"), "w") as f:
            wr = csv.DictWriter(f, fieldnames=cols)
            wr.writeheader()
            for i, name in enumerate(os.listdir(os.path.join(root, folders["mix"]))):
                row = {"ID": i, "duration": 1.0}
                for k in keys:
                    row.update({f"{k}_wav": os.path.join(root, folders[k]) + name, f"{k}_wav_format": "wav", f"{k}_wav_opts": None})
                wr.writerow(row)


class SbLibriMix:
    """items of one `libri<n>mix_<set>.csv`: whole utterances (mixture, sources[, noise]), resampled `sample_rate -> sample_rate * resample`.

    train=True applies `compute_forward`'s augmentation (:49-73) to the batch: per source one speed drawn from `speeds` (percent) and a
    resampling by it, sources cut to the shortest, mixture = their sum (+ noise, cut to the shorter), then a random window of
    `training_signal_len` samples (:312-326).  Draws come from the global torch RNG in the reference's order (one `torch.rand` +
    `torch.randint` per source for the speed -- speechbrain's SpeedPerturb.forward --, one `torch.randint` for the cut)."""

    def __init__(self, csv_path, n_src=2, sample_rate=16000, resample=0.5, noisy=False, train=False, speeds=(95, 100, 105),
                 use_speedperturb=True, perturb_prob=1.0, limit_training_signal_len=True, training_signal_len=32000, device="cuda",
                 data_root=None):
        with open(csv_path) as f:
            rows = list(csv.DictReader(f))
        fix = (lambda p: p.replace("$data_root", data_root)) if data_root else (lambda p: p)
        self.mix = [fix(r["mix_wav"]) for r in rows]
        self.src = [[fix(r[f"s{i + 1}_wav"]) for r in rows] for i in range(n_src)]
        self.noise = [fix(r["noise_wav"]) for r in rows] if noisy else None
        self.n_src, self.sample_rate, self.new_rate = n_src, int(sample_rate), int(sample_rate * resample)
        self.train, self.speeds, self.use_speedperturb, self.perturb_prob = train, tuple(speeds), use_speedperturb, float(perturb_prob)
        self.cut = int(training_signal_len) if (train and limit_training_signal_len) else None
        self.device = device

    def __len__(self):
        return len(self.mix)

    def stage_elems(self, batch_size):
        return 0                                  # utterances of any length: uploaded from pageable memory by the reader thread

    def batch(self, indices, stage=None):
        """(mixture [B, 1, T], sources [B, n_src, T]) on the device.  Utterances of a batch are zero-padded to the longest (speechbrain's
        PaddedBatch), sources cut to the mixture's length."""
        B, S = len(indices), self.n_src
        clips = []
        for i in indices:
            clips.append([read_wav(self.mix[i])] + [read_wav(s[i]) for s in self.src] + ([read_wav(self.noise[i])] if self.noise else []))
        L = max(len(c[0]) for c in clips)
        host = np.zeros((B, len(clips[0]), L), dtype=np.float32)
        for b, cl in enumerate(clips):
            for k, c in enumerate(cl):
                n = min(len(c), L)
                host[b, k, :n] = c[:n]
        x = torch.from_numpy(host).to(self.device)
        if self.new_rate != self.sample_rate:
            x = K.resample(x, self.sample_rate, self.new_rate)
        mix, tgt = x[:, :1], x[:, 1:1 + S]
        noise = x[:, 1 + S] if self.noise else None
        if self.train:
            if self.use_speedperturb:
                outs = []
                for s in range(S):                # independently on each source, the whole batch at one speed (:269-281)
                    y = tgt[:, s]
                    if not (float(torch.rand(1)) > self.perturb_prob):
                        speed = self.speeds[int(torch.randint(len(self.speeds), (1,))[0])]
                        if speed != 100:
                            y = K.resample(y.contiguous(), self.new_rate, self.new_rate * speed // 100)
                    outs.append(y)
                n = min(o.shape[-1] for o in outs)
                tgt = torch.stack([o[:, :n] for o in outs], dim=1)
                mix = tgt.sum(1, keepdim=True)
                if noise is not None:             # :57-67: the shorter of the two lengths
                    n = min(n, noise.shape[-1])
                    mix, tgt = mix[..., :n] + noise[:, None, :n], tgt[..., :n]
            if self.cut is not None:
                start = int(torch.randint(0, 1 + max(0, mix.shape[-1] - self.cut), (1,)))
                mix, tgt = mix[..., start:start + self.cut], tgt[..., start:start + self.cut]
        return mix.contiguous(), tgt.contiguous()
