"""A generated LibriMix tree for the data-side tests (no dataset ships with the repo): PCM16 WAV clips of band-split noise (speaker 1 low
band, speaker 2 high band: separable), the metadata CSVs of the asteroid env (`<dir>/mixture_<set>_mix_clean.csv`: mixture_ID,
mixture_path, source_1_path, source_2_path, length -- librimix_dataset.py:26-92) and the folder layout the speechbrain env / val.py walk
(`wav16k/min/<set>/{mix_clean, s1, s2}` -- prepare_data.py:56-75, val.py:28-57)."""
import os
import wave

import numpy as np
import pandas as pd


def write_wav(path, sig, rate=16000):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes((np.clip(sig, -1, 1 - 2 ** -15) * 32768).astype("<i2").tobytes())


def make_librimix_tree(root, n_train=12, n_dev=4, seconds=(1.2, 2.5), rate=16000, seed=0, short=None, name="Libri2Mix"):
    """-> dict(train_dir, valid_dir, data_folder, test_dir); `short`: length (s) of ONE extra train clip below `segment` (dropped by the
    asteroid dataset, served as an odd-shaped batch by the speechbrain one)"""
    rs = np.random.RandomState(seed)
    lp = np.array([1, 2, 3, 4, 5, 4, 3, 2, 1], dtype=np.float64) / 25.0
    hp = lp * np.array([1, -1] * 4 + [1])
    out = {"data_folder": os.path.join(str(root), name)}
    for set_type, n in (("train-360", n_train), ("dev", n_dev), ("test", n_dev)):
        base = os.path.join(out["data_folder"], "wav16k", "min", set_type)
        for d in ("mix_clean", "s1", "s2", "metadata"):
            os.makedirs(os.path.join(base, d), exist_ok=True)
        rows = []
        lens = [int(rate * rs.uniform(*seconds)) for _ in range(n)]
        if short is not None and set_type == "train-360":
            lens.append(int(rate * short))
        for i, L in enumerate(lens):
            s1 = np.convolve(0.25 * rs.randn(L + 8), lp, mode="valid")
            s2 = np.convolve(0.25 * rs.randn(L + 8), hp, mode="valid")
            name_i = f"utt{i:03d}.wav"
            for d, sig in (("s1", s1), ("s2", s2), ("mix_clean", s1 + s2)):
                write_wav(os.path.join(base, d, name_i), sig, rate)
            rows.append({"mixture_ID": f"utt{i:03d}", "mixture_path": os.path.join(base, "mix_clean", name_i),
                         "source_1_path": os.path.join(base, "s1", name_i), "source_2_path": os.path.join(base, "s2", name_i), "length": L})
        pd.DataFrame(rows).to_csv(os.path.join(base, "metadata", f"mixture_{set_type}_mix_clean.csv"), index=False)
        out[{"train-360": "train_dir", "dev": "valid_dir", "test": "test_meta"}[set_type]] = os.path.join(base, "metadata")
        if set_type == "test":
            out["test_dir"] = base
    return out
