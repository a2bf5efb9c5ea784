"""The training-time batch reader of BASELINE config 4: the `latent_diffusion` branch of the reference's
BEAT dataset object and its collate function (dm/dataload.py:250-271 `__getitem__`, :287-308 `latdiff_long_collate_fn_v1`).

The cache is an LMDB environment (`lmdb.open(path, readonly=True, lock=False, readahead=False, meminit=False)`, dataload.py:129) whose
values are pyarrow-serialised 7-tuples `(motion, attr, emo_label, audio, audio_con, audio_emo, audio_sty)` under the keys "00000",
"00001", ... (`"{:005}".format(index)`).  Neither `lmdb` nor a BEAT cache ships with the reference tree (and `pyarrow.deserialize` is
gone from current pyarrow), so the environment and the deserialiser are injectable: `LatentDiffusionCache(env=..., deserialize=...)`
- tests drive it with an in-memory environment; with the real packages installed, `LatentDiffusionCache.open(path)` wires them up.
The batch dictionary is what `GestureTrainer.forward_losses` reads (`ld_motion`, `ld_audio_con/emo/sty`, `ld_attr`)."""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Dict, List, Optional

import numpy as np
import torch


class LatentDiffusionCache(torch.utils.data.Dataset):
    """`BEAT(data_type="latent_diffusion")` reduced to what its `__getitem__` / `__len__` do (dataload.py:212-215,250-271)."""

    def __init__(self, env, deserialize: Callable[[bytes], tuple]):
        self.env, self.deserialize = env, deserialize

    @classmethod
    def open(cls, cache_path, loader_type: str = "train") -> "LatentDiffusionCache":
        try:
            import lmdb
            import pyarrow as pa
        except ImportError as e:   # pragma: no cover - neither package is in the build image
            raise ImportError("the BEAT latent-diffusion cache needs `lmdb` and a pyarrow with `deserialize` (the reference pins both)") from e
        cache_path = Path(cache_path)
        if loader_type == "val":                                                  # dataload.py:123-125
            cache_path = cache_path.parent / cache_path.name.replace("_300", "_300_val")
        assert (cache_path / "data.mdb").is_file(), f"LMDB file not found at {cache_path / 'data.mdb'}, please create it first."
        if not hasattr(pa, "deserialize"):   # pragma: no cover
            raise ImportError("this pyarrow has no `deserialize`; the cache was written with pyarrow.serialize (dataload.py:253)")
        return cls(lmdb.open(str(cache_path), readonly=True, lock=False, readahead=False, meminit=False), pa.deserialize)

    def __len__(self) -> int:
        with self.env.begin() as txn:
            return int(txn.stat()["entries"])

    def __getitem__(self, index: int) -> Dict[str, object]:
        with self.env.begin(write=False) as txn:
            sample = txn.get("{:005}".format(index).encode("ascii"))
        if sample is None:
            raise IndexError(index)
        sample = self.deserialize(sample)
        assert len(sample) == 7, f"Latent diffusion should have seven samples, got {len(sample)}"
        s_motion, s_attr, s_emo_label, s_audio, s_audio_con, s_audio_emo, s_audio_sty = sample
        return {"ld_motion": torch.from_numpy(np.copy(s_motion)).float(), "ld_audio": s_audio,
                "ld_audio_con": torch.from_numpy(np.copy(s_audio_con)).float(), "ld_audio_emo": torch.from_numpy(np.copy(s_audio_emo)).float(),
                "ld_audio_sty": torch.from_numpy(np.copy(s_audio_sty)).float(), "ld_emo_label": torch.from_numpy(np.copy(s_emo_label)).long(),
                "ld_attr": s_attr}


_STACKED = ("ld_motion", "ld_audio_con", "ld_audio_emo", "ld_audio_sty", "ld_emo_label")   # per-sample tensors of one shape each


def latdiff_long_collate_fn_v1(batch: List[Dict[str, object]]) -> Dict[str, object]:
    """The collate contract of dm/dataload.py:287-308 (the dict keys ARE the interface of the training iteration): fixed-shape entries stacked along a new batch
    axis, the raw waveforms right-padded with zeros to the longest of the batch (`ld_audio`, with the true lengths in `ld_audio_length`), attributes as a list."""
    n = len(batch)
    out: Dict[str, object] = {key: torch.stack([sample[key] for sample in batch]) for key in _STACKED}
    lengths = np.fromiter((sample["ld_audio"].shape[0] for sample in batch), dtype=np.int64, count=n)
    waves = torch.zeros(n, int(lengths.max()), *np.shape(batch[0]["ld_audio"])[1:], dtype=torch.from_numpy(np.asarray(batch[0]["ld_audio"])).dtype)
    for row, sample in enumerate(batch):
        waves[row, : lengths[row]] = torch.from_numpy(np.array(sample["ld_audio"]))
    if out["ld_motion"].shape[0] != waves.shape[0]:
        raise ValueError(f"collate: {out['ld_motion'].shape[0]} motions but {waves.shape[0]} waveforms")
    out.update(ld_audio=waves, ld_audio_length=torch.from_numpy(lengths), ld_attr=[sample["ld_attr"] for sample in batch])
    return out


def make_loader(dataset, batch_size: int, rank: int = 0, world: int = 1, shuffle: bool = True, seed: int = 0, num_workers: int = 0):
    """One DataLoader per data-parallel rank over a disjoint shard of the cache (DistributedSampler), the reference's collate."""
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed, drop_last=True)
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, collate_fn=latdiff_long_collate_fn_v1,
                                       num_workers=num_workers, drop_last=True)
