"""Host-side data feed with the reference's on-disk formats (data_utils.py:10-57):
pickle {utt: [T, M] float32} + JSON index [[utt, t], ...].  The collate view
keeps the reference's memory layout: a [B, M, T] view of a [B, T, M] buffer
(strides (T*M, 1, M)); the engine's first-layer loaders take those strides as is.
"""
import json
import pickle

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


class CollateFn:
    def __init__(self, frame_size):
        self.frame_size = frame_size

    def make_frames(self, tensor):
        fs = self.frame_size
        out = tensor.view(tensor.size(0), tensor.size(1) // fs, fs * tensor.size(2))
        return out.transpose(1, 2)

    def __call__(self, items):
        return self.make_frames(torch.from_numpy(np.array(items)))


class PickleDataset(Dataset):
    def __init__(self, pickle_path, sample_index_path, segment_size):
        with open(pickle_path, "rb") as f:
            self.data = pickle.load(f)
        with open(sample_index_path, "r") as f:
            self.indexes = json.load(f)
        self.segment_size = segment_size

    def __getitem__(self, ind):
        utt_id, t = self.indexes[ind]
        return self.data[utt_id][t:t + self.segment_size]

    def __len__(self):
        return len(self.indexes)


def get_data_loader(dataset, batch_size, frame_size, shuffle=True, num_workers=4, drop_last=False):
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers,
                      collate_fn=CollateFn(frame_size=frame_size), pin_memory=torch.cuda.is_available())
