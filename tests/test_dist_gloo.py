"""Data-parallel path with world_size 2 on CPU/gloo: each rank runs
the engine (CPU lane-level simulation build) on its shard, ONE all-reduce of the
flat gradient buffer, fused clip+Adam with the 1/W mean folded in.  The result
must equal the single-process step on the global batch (InstanceNorm has no
batch coupling, so batch sharding is exact up to fp32 summation order)."""
import os
import socket
import types

import pytest
import torch
import torch.multiprocessing as mp

from oracle import avc_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, wire="fp32", compute=None, buckets=3):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adaptive_voice_conversion_amd.solver import Solver
    from tests.emu_util import backend
    lib, _ = backend("emu")
    cfg = O.tiny_config()
    cfg["allreduce_dtype"] = wire
    cfg["allreduce_buckets"] = buckets
    if compute:
        cfg["compute_dtype"] = compute
    sd = O.make_state_dict(cfg, 4)
    B = 4
    x, eps = O.make_inputs(cfg, B, 32, 4)
    per = B // world
    args = types.SimpleNamespace(store_model_path=os.path.join(out_dir, "model"), load_model=False, data_dir=None, logdir=out_dir)
    s = Solver(cfg, args, lib=lib)
    s.model.load_state_dict(sd)
    sl = slice(rank * per, (rank + 1) * per)
    metas = [s.ae_step(x[sl].contiguous(), 1.0, eps=eps[sl].contiguous()) for _ in range(2)]
    s.save_model()   # every rank calls it (as Solver.train does); only rank 0 writes, atomically, and all wait
    assert os.path.exists(os.path.join(out_dir, "model.ckpt"))
    torch.save({"params": s.model.flat_parameters().clone(), "metas": metas,
                "eps_draw": s._draw_eps(2, 3, 4, torch.device("cpu"))}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_step_equals_global_batch_step(tmp_path):
    world = 2
    from tests.emu_util import emu_lib
    emu_lib()   # (built here once if stale, not by both ranks)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["params"], r1["params"]), "replicas diverged"
    assert not torch.equal(r0["eps_draw"], r1["eps_draw"]), "ranks must draw different reparameterisation noise"
    ck = torch.load(tmp_path / "model.ckpt")
    assert len(ck) == len(O.param_spec(O.tiny_config())) and not list(tmp_path.glob("*.tmp.*"))
    assert r0["metas"][0]["grad_norm"] == pytest.approx(r1["metas"][0]["grad_norm"], rel=1e-6)  # same global norm on both
    # single process, global batch
    from adaptive_voice_conversion_amd.solver import Solver
    from tests.emu_util import backend
    lib, _ = backend("emu")
    cfg = O.tiny_config()
    sd = O.make_state_dict(cfg, 4)
    x, eps = O.make_inputs(cfg, 4, 32, 4)
    args = types.SimpleNamespace(store_model_path=None, load_model=False, data_dir=None, logdir=str(tmp_path))
    s = Solver(cfg, args, lib=lib)
    s.model.load_state_dict(sd)
    m0 = s.ae_step(x, 1.0, eps=eps)
    assert r0["metas"][0]["grad_norm"] == pytest.approx(m0["grad_norm"], rel=1e-5)
    # per-rank losses are shard means; their average is the global mean
    assert 0.5 * (r0["metas"][0]["loss_rec"] + r1["metas"][0]["loss_rec"]) == pytest.approx(m0["loss_rec"], rel=1e-5)
    s.ae_step(x, 1.0, eps=eps)
    single = s.model.flat_parameters()
    diff = (single - r0["params"]).abs()
    # after 2 steps: identical except a handful of ~0-gradient elements (Adam's sign-like first step)
    assert (diff > 2e-6).float().mean().item() < 5e-3
    assert diff.max().item() <= 4.2 * cfg["optimizer"]["lr"]
    # `allreduce_buckets: 2` (decoder | both encoders as one range) and `1` (the whole buffer) reduce the same numbers: with the fp32 wire the
    # replicas end bit-identical to the three-bucket run
    for nb in (2, 1):
        dn = tmp_path / f"buckets{nb}"
        dn.mkdir()
        mp.spawn(_worker, args=(world, _free_port(), str(dn), "fp32", None, nb), nprocs=world, join=True)
        assert torch.equal(torch.load(dn / "rank0.pt")["params"], r0["params"]), nb
    # BASELINE configs[2]'s wire format (`allreduce_dtype: bf16`, the default under `compute_dtype: bf16`): the flat gradient
    # buffer is all-reduced as bf16.  Replicas must stay bit-identical (every rank receives the same sum) and the step must stay
    # within bf16 rounding of the fp32-wire step.
    d = tmp_path / "bf16"
    d.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), str(d), "bf16"), nprocs=world, join=True)
    b0, b1 = torch.load(d / "rank0.pt"), torch.load(d / "rank1.pt")
    assert torch.equal(b0["params"], b1["params"]), "replicas diverged"
    assert not torch.equal(b0["params"], r0["params"])                        # the wire format really changed
    assert b0["metas"][0]["grad_norm"] == pytest.approx(r0["metas"][0]["grad_norm"], rel=5e-3)
    lr = cfg["optimizer"]["lr"]
    diff = (b0["params"] - r0["params"]).abs()
    assert diff.max().item() <= 4.2 * lr and (diff > 0.05 * lr).float().mean().item() < 0.2
    # BASELINE configs[2] as a whole: the bf16 STORAGE engine (compute_dtype "bf16") on every rank + the bf16 wire it defaults to.
    # Same contract: bit-identical replicas, a global gradient norm within bf16 distance of the fp32 run's.
    d2 = tmp_path / "bf16_storage"
    d2.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), str(d2), "bf16", "bf16"), nprocs=world, join=True)
    c0, c1 = torch.load(d2 / "rank0.pt"), torch.load(d2 / "rank1.pt")
    assert torch.equal(c0["params"], c1["params"]), "replicas diverged"
    assert c0["metas"][0]["grad_norm"] == pytest.approx(r0["metas"][0]["grad_norm"], rel=5e-2)
    assert c0["metas"][0]["loss_rec"] == pytest.approx(r0["metas"][0]["loss_rec"], rel=2e-2)
    assert (c0["params"] - r0["params"]).abs().max().item() <= 4.2 * lr
