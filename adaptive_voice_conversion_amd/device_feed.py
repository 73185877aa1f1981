"""Device-resident segment feed.

The reference slices 128-frame segments out of a pickled ``{utt: [T, M]}`` dict in 4 DataLoader
worker processes and copies every batch host->device (data_utils.py:43-57, solver.py:57-68,82).
At >3e4 segments/s that feed is the bottleneck, while a whole normalised mel corpus fits the
288 GB of one MI355X many times over.  ``DeviceSegmentFeed`` uploads the corpus once as one
``[sum_T, M]`` tensor; a batch is ONE launch of the library's gather kernel
(``avc_gather_segments``, csrc/rowops.hip): it reads the B x T corpus rows coalesced along the mel axis and
writes the ``[B, M, T]`` tensor with the time axis contiguous -- the values ``CollateFn`` produces
(data_utils.py:14-22), in the layout every first-layer loader of the engine reads with unit stride (the
reference's collate result is a transposed *view*, strides (T*M, 1, M); the engine accepts both).

Data parallelism: all ranks draw the SAME permutation per epoch (same seed) and rank r takes elements
r, r+W, r+2W, ... of it, so the shards are disjoint and of equal size (the < W left-over samples of an
epoch are dropped; every rank must run the same number of steps for the gradient all-reduce).
"""
import ctypes
import json
import pickle

import numpy as np
import torch

from . import _lib


class DeviceSegmentFeed:
    def __init__(self, data, indexes, segment_size, batch_size, device, shuffle=True, seed=0, rank=0, world_size=1, lib=None):
        """data: {utt_id: float32 [T, M]}; indexes: [[utt_id, t], ...] (the reference's sample index JSON)."""
        self.segment_size, self.batch_size, self.shuffle = int(segment_size), int(batch_size), shuffle
        self.rank, self.world = int(rank), int(world_size)
        if not (0 <= self.rank < self.world):
            raise ValueError("rank must be in [0, world_size)")
        self.lib = lib if lib is not None else _lib.load()
        offs, chunks, pos = {}, [], 0
        for k, v in data.items():
            v = np.asarray(v, dtype=np.float32)
            offs[k] = pos
            pos += v.shape[0]
            chunks.append(v)
        self.corpus = torch.from_numpy(np.concatenate(chunks, axis=0)).to(device).contiguous()   # [sum_T, M], resident
        self.n_mels = int(self.corpus.shape[1])
        starts = [offs[u] + int(t) for u, t in indexes]
        if any(s + self.segment_size > pos for s in starts):
            raise ValueError("a segment runs past the end of the corpus")
        self.starts = torch.tensor(starts, dtype=torch.long, device=self.corpus.device)
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        self._perm, self._cursor = None, 0

    @classmethod
    def from_files(cls, pickle_path, sample_index_path, segment_size, batch_size, device, **kw):
        with open(pickle_path, "rb") as f:
            data = pickle.load(f)
        with open(sample_index_path, "r") as f:
            indexes = json.load(f)
        return cls(data, indexes, segment_size, batch_size, device, **kw)

    def shard_size(self):
        return self.starts.numel() // self.world if self.world > 1 else self.starts.numel()

    def __len__(self):
        return (self.shard_size() + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        return self

    def _new_epoch(self):
        n = self.starts.numel()
        perm = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        if self.world > 1:
            perm = perm[: (n // self.world) * self.world][self.rank::self.world]
        self._perm = perm.to(self.starts.device)
        self._cursor = 0

    def gather(self, sel):
        """[len(sel), M, T] contiguous batch for the sample indices ``sel`` (a device int64 tensor)."""
        st = self.starts[sel].contiguous()
        B, M, T = int(st.numel()), self.n_mels, self.segment_size
        out = torch.empty(B, M, T, dtype=torch.float32, device=self.corpus.device)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        cuda = self.corpus.is_cuda
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.corpus.device).cuda_stream) if cuda else None
        with (torch.cuda.device(self.corpus.device) if cuda else _null()):
            rc = self.lib.avc_gather_segments(P(self.corpus), self.corpus.shape[0], M, P(st), B, T, P(out), stream)
        if rc != 0:
            raise RuntimeError(f"avc_gather_segments failed: {rc}")
        return out

    def __next__(self):
        """Infinite iterator (utils.py:28-35 infinite_iter over a shuffling DataLoader); the last batch of an
        epoch may be short (drop_last is ignored by the reference, data_utils.py:24-27)."""
        if self._perm is None or self._cursor >= self._perm.numel():
            self._new_epoch()
        sel = self._perm[self._cursor:self._cursor + self.batch_size]
        self._cursor += self.batch_size
        return self.gather(sel)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
