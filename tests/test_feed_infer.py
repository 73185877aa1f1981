"""Callers on either side of the hot path: device-resident segment feed and the
Inferencer front, on the CPU simulator build (kind='emu') and on the GPU."""
import types

import numpy as np
import pytest
import torch

from adaptive_voice_conversion_amd.data_utils import CollateFn
from adaptive_voice_conversion_amd.device_feed import DeviceSegmentFeed
from adaptive_voice_conversion_amd.inference import Inferencer
from oracle import avc_oracle as O
from tests.emu_util import KINDS, backend


def test_device_feed_equals_reference_collate():
    rng = np.random.RandomState(0)
    data = {f"u{i}": rng.randn(40 + 7 * i, 16).astype(np.float32) for i in range(5)}
    indexes = [[f"u{i % 5}", int(rng.randint(0, 40 + 7 * (i % 5) - 24))] for i in range(23)]
    feed = DeviceSegmentFeed(data, indexes, segment_size=24, batch_size=8, device="cpu", shuffle=False)
    got = [next(feed) for _ in range(len(feed))]
    assert [g.shape[0] for g in got] == [8, 8, 7]               # short last batch kept (drop_last ignored)
    collate = CollateFn(frame_size=1)
    for bi, g in enumerate(got):
        items = [data[u][t:t + 24] for u, t in indexes[bi * 8:(bi + 1) * 8]]   # PickleDataset.__getitem__ (data_utils.py:51-54)
        ref = collate(items)
        assert g.shape == ref.shape and g.stride() == ref.stride()             # the [B,M,T] view over [B,T,M] memory
        assert torch.equal(g, ref)
    assert next(feed).shape[0] == 8                               # wraps around like infinite_iter


@pytest.mark.parametrize("kind", KINDS)
def test_inferencer_matches_oracle_and_buckets(kind, tmp_path):
    lib, dev = backend(kind)
    cfg = O.tiny_config()
    sd = O.make_state_dict(cfg, 7)
    torch.save(sd, tmp_path / "m.ckpt")
    attr = {"mean": np.linspace(-1, 1, 16).astype(np.float32), "std": np.linspace(0.5, 2, 16).astype(np.float32)}
    import pickle
    with open(tmp_path / "attr.pkl", "wb") as f:
        pickle.dump(attr, f)
    args = types.SimpleNamespace(model=str(tmp_path / "m.ckpt"), attr=str(tmp_path / "attr.pkl"))
    inf = Inferencer(cfg, args, lib=lib if kind == "emu" else None)
    g = torch.Generator().manual_seed(1)
    pairs = [(torch.randn(T, 16, generator=g), torch.randn(Tc, 16, generator=g)) for T, Tc in ((37, 19), (24, 24), (37, 19), (31, 40))]
    outs = inf.convert_batch(pairs)
    for (s, t), o in zip(pairs, outs):
        ref = O.ae_inference(s.t()[None], t.t()[None], sd, cfg)[0].t()
        assert o.shape == ref.shape
        torch.testing.assert_close(o, ref, rtol=1e-4, atol=2e-5)
    wav, mel = inf.inference_one_utterance(pairs[0][0], pairs[0][1])
    assert wav is None
    ref = O.ae_inference(pairs[0][0].t()[None], pairs[0][1].t()[None], sd, cfg)[0].t().numpy() * attr["std"] + attr["mean"]
    np.testing.assert_allclose(mel, ref, rtol=1e-4, atol=5e-5)
