"""Device-resident segment feed.

The reference slices 128-frame segments out of a pickled ``{utt: [T, M]}`` dict in 4 DataLoader
worker processes and copies every batch host->device (data_utils.py:43-57, solver.py:57-68,82).
At >3e4 segments/s that feed is the bottleneck, while a whole normalised mel corpus fits the
288 GB of one MI355X many times over.  ``DeviceSegmentFeed`` uploads the corpus once as one
``[sum_T, M]`` tensor and produces each batch by an index gather on the device, handing the engine
the same ``[B, M, T]`` view over ``[B, T, M]`` memory that ``CollateFn`` produces (strides
(T*M, 1, M)) — the first-layer kernels read that layout in place.
"""
import json
import pickle

import numpy as np
import torch


class DeviceSegmentFeed:
    def __init__(self, data, indexes, segment_size, batch_size, device, shuffle=True, seed=0):
        """data: {utt_id: float32 [T, M]}; indexes: [[utt_id, t], ...] (the reference's sample index JSON)."""
        self.segment_size, self.batch_size, self.shuffle = int(segment_size), int(batch_size), shuffle
        offs, chunks, pos = {}, [], 0
        for k, v in data.items():
            v = np.asarray(v, dtype=np.float32)
            offs[k] = pos
            pos += v.shape[0]
            chunks.append(v)
        self.corpus = torch.from_numpy(np.concatenate(chunks, axis=0)).to(device)          # [sum_T, M], resident
        starts = [offs[u] + int(t) for u, t in indexes]
        self.starts = torch.tensor(starts, dtype=torch.long, device=device)
        self.frames = torch.arange(self.segment_size, device=device)
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        self._perm, self._cursor = None, 0

    @classmethod
    def from_files(cls, pickle_path, sample_index_path, segment_size, batch_size, device, **kw):
        with open(pickle_path, "rb") as f:
            data = pickle.load(f)
        with open(sample_index_path, "r") as f:
            indexes = json.load(f)
        return cls(data, indexes, segment_size, batch_size, device, **kw)

    def __len__(self):
        return (self.starts.numel() + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        return self

    def __next__(self):
        """Infinite iterator (utils.py:28-35 infinite_iter over a shuffling DataLoader); the last batch of an
        epoch may be short (drop_last is ignored by the reference, data_utils.py:24-27)."""
        n = self.starts.numel()
        if self._perm is None or self._cursor >= n:
            self._perm = (torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)).to(self.starts.device)
            self._cursor = 0
        sel = self._perm[self._cursor:self._cursor + self.batch_size]
        self._cursor += self.batch_size
        rows = self.starts[sel][:, None] + self.frames[None, :]       # [B, T] frame indices
        seg = self.corpus[rows]                                        # [B, T, M] gather in HBM
        return seg.transpose(1, 2)                                     # [B, M, T] view, strides (T*M, 1, M)
