"""Host-side segment feed behind the reference's data API (reference: data_utils.py:10-57 -- ``PickleDataset``, ``CollateFn``,
``get_data_loader``; on-disk formats unchanged: pickle ``{utt: [T, M] float32}`` + JSON index ``[[utt, t], ...]``).

The reference slices one segment per ``__getitem__`` in four worker processes, stacks the list through ``np.array`` and pickles the batch
back to the trainer.  At this engine's step time (>= 40 k segments/s) that path cannot keep up, and the product feed is the device-resident one
(``device_feed.DeviceSegmentFeed``).  This module is the HOST path kept for drop-in use, rebuilt around one idea: a batch is ONE vectorised
gather.  The corpus is concatenated once into a single ``[sum_T, M]`` array; a batch of B index entries becomes a ``[B, T]`` matrix of row
numbers and one ``np.take`` writes it straight into a pinned ``[B, T, M]`` staging buffer (two of them, alternating, so that the previous
batch's asynchronous host-to-device copy may still be in flight).  What the trainer receives is the reference's collate result: the
``[B, M, T]`` VIEW of that buffer (strides ``(T*M, 1, M)``, data_utils.py:14-16), which the engine's first-layer loaders read in place.
"""
import json
import pickle

import numpy as np
import torch


class PickleDataset:
    """``dataset[i]`` is the i-th index entry's ``[segment_size, M]`` slice, as in the reference (data_utils.py:51-54); ``gather`` is the
    batched form the loader uses."""

    def __init__(self, pickle_path, sample_index_path, segment_size):
        with open(pickle_path, "rb") as f:
            data = pickle.load(f)
        with open(sample_index_path, "r") as f:
            index = json.load(f)
        self._build(data, index, segment_size)

    @classmethod
    def from_memory(cls, data, index, segment_size):
        self = cls.__new__(cls)
        self._build(data, index, segment_size)
        return self

    def _build(self, data, index, segment_size):
        self.segment_size = int(segment_size)
        names = list(data)
        first = np.zeros(len(names) + 1, dtype=np.int64)
        for k, n in enumerate(names):
            first[k + 1] = first[k] + data[n].shape[0]
        self.corpus = np.concatenate([np.asarray(data[n], dtype=np.float32) for n in names], axis=0)   # [sum_T, M]
        where = {n: int(first[k]) for k, n in enumerate(names)}
        self.rows = np.asarray([where[u] + int(t) for u, t in index], dtype=np.int64)   # first corpus row of every index entry
        ends = np.asarray([where[u] + data[u].shape[0] for u, _ in index], dtype=np.int64)
        if (self.rows + self.segment_size > ends).any():
            raise ValueError("an index entry runs past the end of its utterance")
        self.indexes = index

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        r = int(self.rows[i])
        return self.corpus[r:r + self.segment_size]

    def gather(self, entries, out):
        """segments of the index entries ``entries`` -> ``out`` (a ``[len(entries), segment_size, M]`` float32 array), one ``np.take``"""
        rows = self.rows[np.asarray(entries, dtype=np.int64)][:, None] + np.arange(self.segment_size, dtype=np.int64)[None, :]
        np.take(self.corpus, rows, axis=0, out=out)
        return out


class CollateFn:
    """list of ``[T, M]`` segments -> the ``[B, fs*M, T/fs]`` view the reference hands to the model (data_utils.py:10-22)."""

    def __init__(self, frame_size):
        self.frame_size = int(frame_size)

    def make_frames(self, tensor):
        b, t, m = tensor.shape
        return tensor.reshape(b, t // self.frame_size, self.frame_size * m).permute(0, 2, 1)

    def __call__(self, items):
        buf = torch.empty((len(items),) + tuple(items[0].shape), dtype=torch.float32)
        dst = buf.numpy()
        for k, seg in enumerate(items):   # (rows land in the batch buffer directly: no list -> np.array copy)
            dst[k] = seg
        return self.make_frames(buf)


class HostSegmentLoader:
    """What ``get_data_loader`` returns for a ``PickleDataset``: iterating it yields one epoch of collated batches -- ``ceil(N / B)`` of them,
    the last one short (the reference ignores ``drop_last``, data_utils.py:24-27) -- in a fresh permutation per epoch when ``shuffle``."""

    def __init__(self, dataset, batch_size, frame_size, shuffle, seed=0):
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), bool(shuffle)
        self.collate = CollateFn(frame_size)
        self._rng = np.random.RandomState(seed)
        m = dataset.corpus.shape[1]
        pin = torch.cuda.is_available()
        self._staging = [torch.empty((self.batch_size, dataset.segment_size, m), dtype=torch.float32, pin_memory=pin) for _ in range(2)]
        self._turn = 0

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = self._rng.permutation(n) if self.shuffle else np.arange(n)
        for b0 in range(0, n, self.batch_size):
            entries = order[b0:b0 + self.batch_size]
            buf = self._staging[self._turn][:len(entries)]
            self._turn ^= 1
            self.dataset.gather(entries, buf.numpy())
            yield self.collate.make_frames(buf)


def get_data_loader(dataset, batch_size, frame_size, shuffle=True, num_workers=4, drop_last=False):
    """Reference signature (data_utils.py:24-27).  ``num_workers`` / ``drop_last`` are accepted and unused, as ``drop_last`` is there: the
    vectorised gather runs in the caller's thread.  Any other map-style dataset gets a stock ``DataLoader`` with this module's collate."""
    if isinstance(dataset, PickleDataset):
        return HostSegmentLoader(dataset, batch_size, frame_size, shuffle)
    from torch.utils.data import DataLoader
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, collate_fn=CollateFn(frame_size))
