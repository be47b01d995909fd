"""LibriMix dataset of the asteroid env on MI355X.

Mirrors train_env/asteroid_librimix/librimix_dataset.py:26-170 of the reference -- same constructor arguments, CSV selection per
task, dropping of utterances shorter than `segment`, random segment start, item = (mixture [1, T'], sources [n_src, T']) -- with the
arithmetic on the device: the 16 -> 8 kHz resampling of every clip (reference: torchaudio.transforms.Resample per clip on the CPU,
:54, 111-165) is ONE batched polyphase-FIR launch (fqss_resample_fir), the SNR augmentation (:139-153, process.py:77-103) the
batched device mixer (fqss_snr_mix).  `batch(indices)` assembles a whole batch that way; `__getitem__` is the same path for one item.

File reading: the reference uses soundfile (third party, absent); LibriMix clips are 16-bit PCM WAV, which the standard library's
`wave` reads -- anything else raises.  torchaudio's resampler is third party too: its published kernel (sinc x Hann, width 6,
rolloff 0.99) is restated in fqss_amd.kernels.sinc_resample_taps; parity is pinned to the oracle's restatement, not to torchaudio."""
import os
import random
import wave

import numpy as np
import pandas as pd
import torch

from ... import kernels as K
from ...process import generate_mix_noise
from ..train_utils import augmentation_2mix, augmentation_3mix


def read_wav(path, start=0, stop=None):
    """float32 samples in [-1, 1) of a mono PCM WAV (soundfile.read(path, dtype='float32', start, stop) for the LibriMix files)"""
    with wave.open(path, "rb") as w:
        if w.getnchannels() != 1 or w.getcomptype() != "NONE" or w.getsampwidth() not in (2, 4):
            raise NotImplementedError(f"{path}: only mono 16 / 32-bit PCM WAV is read without soundfile")
        n = w.getnframes()
        stop = n if stop is None else min(stop, n)
        w.setpos(start)
        raw = w.readframes(stop - start)
        if w.getsampwidth() == 2:
            return np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        return (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)


class LibriMix:
    dataset_name = "LibriMix"

    def __init__(self, csv_dir, task="sep_clean", sample_rate=16000, resample=1.0, n_src=2, segment=3, augmentation_cfg=None,
                 device="cuda"):
        self.csv_dir, self.task, self.resample, self.augmentation_cfg, self.device = csv_dir, task, resample, augmentation_cfg, device
        pick = {"enh_single": "single", "enh_both": "both", "sep_clean": "clean", "sep_noisy": "both"}[task]
        md_file = [f for f in os.listdir(csv_dir) if pick in f][0]
        self.csv_path = os.path.join(csv_dir, md_file)
        if task == "enh_both":
            md_clean = [f for f in os.listdir(csv_dir) if "clean" in f][0]
            self.df_clean = pd.read_csv(os.path.join(csv_dir, md_clean))
        self.segment, self.sample_rate = segment, sample_rate
        self.df = pd.read_csv(self.csv_path)
        if segment is not None:
            max_len = len(self.df)
            self.seg_len = int(segment * sample_rate)
            self.df = self.df[self.df["length"] >= self.seg_len]
            print(f"Drop {max_len - len(self.df)} utterances from {max_len} (shorter than {segment} seconds)")
        else:
            self.seg_len = None
        self.n_src = n_src
        # the columns a batch needs, as plain lists: `df.iloc[idx]` builds a Series per item (~100 us), which a 13 ms step notices
        cols = ["length", "mixture_path"] + [f"source_{i + 1}_path" for i in range(n_src) if f"source_{i + 1}_path" in self.df] + \
               (["noise_path"] if "noise_path" in self.df else [])
        self._col = {k: self.df[k].tolist() for k in cols}
        self._clean_mix = self.df_clean["mixture_path"].tolist() if task == "enh_both" else None

    def __len__(self):
        return len(self.df)

    # ---- host side: which clips, which segment -------------------------------------------------------------------------
    def _read_item(self, idx):
        col = self._col
        if self.seg_len is not None:
            start = random.randint(0, int(col["length"][idx]) - self.seg_len)
            stop = start + self.seg_len
        else:
            start, stop = 0, None
        noise = read_wav(col["noise_path"][idx], start, stop) if self.task in ("enh_single", "sep_noisy") else None
        if self.task == "enh_both":
            sources = [read_wav(self._clean_mix[idx], start, stop)]
        else:
            sources = [read_wav(col[f"source_{i + 1}_path"][idx], start, stop) for i in range(self.n_src)]
        augment = bool(self.augmentation_cfg) and np.random.uniform() < self.augmentation_cfg.get("prob", 1)
        mixture = None if augment else read_wav(col["mixture_path"][idx], start, stop)
        # the item's SNR draws come right behind its probability draw, as inside the reference's __getitem__ (:139-153 calling
        # train_utils.py:30-52): prob_i, snr_i (two for a 3-mix), noise_snr_i -- so a seeded np.random stream is consumed in the
        # reference's order for any batch size
        return sources, noise, mixture, augment, (self._draw_snrs() if augment else None)

    def _draw_snrs(self):
        cfg, n = self.augmentation_cfg, (1 if self.task == "enh_both" else self.n_src)
        if cfg.get("distribution") != "uniform":
            raise AssertionError("Augmentation is not supoorted!")
        lo, hi = cfg.get("param0"), cfg.get("param1")
        if self.task not in ("enh_single", "sep_clean", "sep_noisy") or (self.task != "enh_single" and n not in (2, 3)):
            raise AssertionError("Augmetation is not supported!")
        d = [np.random.uniform(low=lo, high=hi)]
        if self.task != "enh_single" and n == 3:
            d.append(np.random.uniform(low=lo, high=hi))
        if self.task == "sep_noisy":
            d.append(np.random.uniform(low=6, high=18))
        return d

    # ---- device side: resample + mix -----------------------------------------------------------------------------------
    def stage_elems(self, batch_size):
        """float32 elements of the pinned staging buffer a batch of `batch_size` items needs (0: variable-length items, no staging)"""
        per_item = (1 if self.task == "enh_both" else self.n_src) + 2          # sources + noise + mixture
        return 0 if self.seg_len is None else batch_size * per_item * self.seg_len

    def _to_device(self, clips, stage=None):
        n, L = len(clips), len(clips[0])
        if stage is not None and stage.numel() >= n * L:
            # through the caller's pinned buffer: the host->device copy is asynchronous on the current stream (a prefetching reader's
            # side stream), the buffer is the caller's to keep alive until that copy has run
            host = stage[:n * L].view(n, L)
            hv = host.numpy()
            for i, c in enumerate(clips):
                hv[i] = c
            x = host.to(self.device, non_blocking=True)
        else:
            x = torch.from_numpy(np.stack(clips)).to(self.device)
        if self.resample != 1:
            x = K.resample(x, self.sample_rate, int(self.resample * self.sample_rate))
        return x

    def _augment(self, sources, noise, draws):
        """sources [B, n_src, T], noise [B, T] or None, draws = the items' SNR draws (_read_item) -> mixtures [B, T] (:139-153),
        handed to the batched device kernel as [B] tensors"""
        from ...process import generate_2mix_snr, generate_3mix_snr
        n = sources.shape[1]
        col = lambda j: torch.tensor([d[j] for d in draws], dtype=torch.float32, device=sources.device)
        if self.task == "enh_single":
            return generate_2mix_snr(sources[:, 0], noise, col(0))
        if n == 2:
            mix = generate_2mix_snr(sources[:, 0], sources[:, 1], col(0))
        else:
            mix = generate_3mix_snr(sources[:, 0], sources[:, 1], sources[:, 2], col(0), col(1))
        if self.task == "sep_noisy":
            mix = generate_mix_noise(mix, noise, col(-1))
        return mix

    def batch(self, indices, stage=None):
        """(mixtures [B, 1, T'], sources [B, n_src, T']) on the device: every clip of the batch is resampled by one launch, enqueued
        on the calling thread's current stream; stage: an optional pinned float32 buffer (>= stage_elems(B)) for the upload"""
        items = [self._read_item(i) for i in indices]
        if len({len(it[0][0]) for it in items}) != 1:
            raise ValueError("batch(): items of different lengths (segment=None): use one item per batch")
        B, ns = len(items), len(items[0][0])
        clips = [s for it in items for s in it[0]]
        has_noise = items[0][1] is not None
        if has_noise:
            clips += [it[1] for it in items]
        plain = [i for i, it in enumerate(items) if not it[3]]
        clips += [items[i][2] for i in plain]
        x = self._to_device(clips, stage)
        sources = x[:B * ns].reshape(B, ns, -1)
        noise = x[B * ns:B * ns + B] if has_noise else None
        rest = x[B * ns + (B if has_noise else 0):]
        aug = [i for i, it in enumerate(items) if it[3]]
        if not aug:                                   # every mixture comes from its file: no gather, no index tensors
            return rest.unsqueeze(1), sources
        if not plain:
            mixture = self._augment(sources, noise, [it[4] for it in items])
        else:
            mixture = torch.empty(B, sources.shape[-1], device=x.device)
            mixture[plain] = rest
            mixture[aug] = self._augment(sources[aug], noise[aug] if has_noise else None, [items[i][4] for i in aug])
        return mixture.unsqueeze(1), sources

    def __getitem__(self, idx):
        mixture, sources = self.batch([idx])
        return mixture[0], sources[0]

    def get_infos(self):
        """asteroid's dataset card (librimix_dataset.py:239-261)"""
        infos = {"dataset": self.dataset_name}
        if self.task == "enh_single":
            infos["task"], infos["licenses"] = "enhancement", ["librispeech_license", "wham_noise_license"]
        elif self.task == "enh_both":
            infos["task"], infos["licenses"] = "enhancement", ["librispeech_license", "wham_noise_license"]
        elif self.task == "sep_clean":
            infos["task"], infos["licenses"] = "sep_clean", ["librispeech_license"]
        else:
            infos["task"], infos["licenses"] = "sep_noisy", ["librispeech_license", "wham_noise_license"]
        return infos
