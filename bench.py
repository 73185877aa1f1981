#!/usr/bin/env python
"""Benchmark of the AdaIN-VC train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" = one full ``Solver.ae_step`` (forward, L1+KL loss, backward, RCCL
all-reduce of the flat gradient buffer when N > 1, fused clip + Adam/amsgrad)
over one batch of synthetic N(0,1) 80-mel x 128-frame segments already resident
in HBM.  Workload = BASELINE.json configs[1]: batch 256 per GPU, fp32 (weak
scaling: per-GPU batch fixed, global batch = 256*N).  Rank 0 prints ONE JSON
line.  After the timed region (never inside it) two extra legs run on rank 0 at
N = 1: a per-kernel-class HIP-event profile (-> "roofline") and the CPU oracle
timed on the host cores (-> "cpu_baseline").
"""
import argparse
import ctypes
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)


def stock_config(n_mels):
    from adaptive_voice_conversion_amd.config import default_config
    return default_config(n_mels)


def profile_classes(solver, x, eps, steps):
    """Per-kernel-class totals from HIP events recorded on the launch stream."""
    from adaptive_voice_conversion_amd import _lib
    lib = _lib.load()
    n = lib.avc_prof_nclass()
    ms, launches = (ctypes.c_double * n)(), (ctypes.c_long * n)()
    flops, nbytes = (ctypes.c_double * n)(), (ctypes.c_double * n)()
    torch.cuda.synchronize()
    lib.avc_set_single_stream(1)   # one stream: event brackets then measure each kernel class in isolation
    lib.avc_prof_begin()
    for _ in range(steps):
        solver.ae_step(x, 1.0, eps=eps, sync=False)
    torch.cuda.synchronize()
    lib.avc_prof_end(ms, launches, flops, nbytes)
    lib.avc_set_single_stream(0)
    out = {}
    for i in range(n):
        if launches[i]:
            out[lib.avc_prof_class_name(i).decode()] = dict(
                ms_per_step=ms[i] / steps, launches_per_step=launches[i] / steps, avg_us=1e3 * ms[i] / launches[i],
                tflops=(flops[i] / (ms[i] * 1e-3) / 1e12) if flops[i] else None,
                gbs=(nbytes[i] / (ms[i] * 1e-3) / 1e9) if nbytes[i] else None,
                flops_per_launch=flops[i] / launches[i], bytes_per_launch=nbytes[i] / launches[i])
    return out


def instnorm_dominant_shape(B, C, T, launches=50):
    """IN/AdaIN/ReLU forward + backward at the dominant shape [B, C, T] of the step: `launches`
    back-to-back launches between two HIP events on the launch stream (no per-launch bracket)."""
    from adaptive_voice_conversion_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = 8  # rotate buffers: 8 x 3 x 16.8 MB > 256 MiB Infinity Cache at the bench shape
    ys = [torch.randn(B, C, T, device=dev) for _ in range(nb)]
    outs = [torch.empty_like(y) for y in ys]
    gs = [torch.randn_like(y) for y in ys]
    cond = torch.randn(B, 2 * C, device=dev)
    mean, rstd = torch.empty(B * C, device=dev), torch.empty(B * C, device=dev)
    dcond = torch.zeros(B, 2 * C, device=dev)

    def fwd(k):
        lib.avc_instnorm_fwd(P(ys[k]), B, C, T, P(cond), 2 * C, 0, 1, None, 0, 0, P(outs[k]), P(mean), P(rstd), st)

    def bwd(k):
        lib.avc_instnorm_bwd(P(gs[k]), P(ys[k]), P(mean), P(rstd), B, C, T, P(cond), 2 * C, 0, 1, P(outs[k]), P(dcond), 2 * C, 0, st)

    res = {}
    for name, fn, passes in (("fwd", fwd, 2), ("bwd", bwd, 3)):
        for k in range(nb):
            fn(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(launches):
            fn(i % nb)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / launches
        res[name] = dict(avg_launch_us=us, bytes_per_launch=passes * 4.0 * B * C * T, gbs=passes * 4.0 * B * C * T / us / 1e3)
    return res


def pmc_traffic(kernel_prefix, grid):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled on gfx950,
    MI355X_MICROARCH.md §HBM; KB units) — None when no profile is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_summary.json")
    try:
        d = json.load(open(path))
        for name, grids in d.items():
            if name.startswith(kernel_prefix) and str(grid) in grids:
                c = grids[str(grid)]
                return (2.0 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
    except Exception:
        pass
    return None


def cpu_baseline(n_mels, T, budget_s=12.0):
    """The oracle's train step (same ATen CPU ops as the reference) on the host cores, bounded sample."""
    from oracle import avc_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    cfg = O.stock_config(n_mels)
    Bc = 128  # best CPU batch in BASELINE.md
    sd = O.make_state_dict(cfg, 0)
    x, eps = O.make_inputs(cfg, Bc, T, 0)
    opt = O.make_opt(sd, cfg)
    tw = time.perf_counter()
    O.ae_step(x, eps, sd, opt, cfg, 1.0)  # warm-up
    print(f"[bench] cpu_baseline warm-up step: {time.perf_counter() - tw:.2f}s on {cores} threads", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    n = 0
    while True:
        O.ae_step(x, eps, sd, opt, cfg, 1.0)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 20:
            break
    return dict(value=Bc * n / el, unit="mel-segments/sec", cores=cores, kind="port",
                sample=f"{n} oracle train steps (+1 warm-up) at batch {Bc}, 80x{T} segments, torch CPU fp32, {cores} threads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config 2: 256)")
    ap.add_argument("--mels", type=int, default=80)
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--mode", choices=("train", "infer"), default="train",
                    help="train = BASELINE configs[1]/[4]-style train step (the headline metric); infer = configs[3] one-shot conversion")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 = the headline (BASELINE configs[1]); bf16 = configs[2]'s compute mode (bf16 matrix products, "
                         "fp32 master weights / optimizer state) -- a separate, non-headline measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--presleep-ms", type=float, default=0.0,
                    help="tracing aid: park the GPU this long before every timed step so that the (tracer-slowed) host has "
                         "the whole step enqueued when it starts; the reported time is then meaningless")
    ap.add_argument("--profile-json", default=None, help="write the per-kernel-class table here")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if a.gpus != world and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    from adaptive_voice_conversion_amd.solver import Solver
    cfg = stock_config(a.mels)
    if a.dtype == "bf16":
        cfg["compute_dtype"] = "bf16"
    torch.manual_seed(0)
    args = types.SimpleNamespace(store_model_path=None, load_model=False, data_dir=None, logdir="/tmp/avc_bench_log")
    solver = Solver(cfg, args)
    B, T = a.batch, a.frames
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)   # each rank: its own shard of the global batch
    x = torch.randn(B, a.mels, T, generator=g).to(dev)
    plan, _ = solver.model._plan(B, T, T, dev)
    eps = torch.randn(B, cfg["ContentEncoder"]["c_out"], plan.latent_len, generator=g).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.mode == "infer":
        # BASELINE configs[3]: batched one-shot conversion AE.inference(x, x_cond) (model.py:387-391), forward only
        xc = torch.randn(B, a.mels, T, generator=g).to(dev)
        model = solver.model
        plan_i, ws_i = model._plan(B, T, T, dev)
        for _ in range(a.warmup):
            plan_i.forward(model.flat_parameters(), x, xc, None, ws_i)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            plan_i.forward(model.flat_parameters(), x, xc, None, ws_i)
        barrier()
        el = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"metric": "utterances/sec one-shot conversion (AE.inference)", "value": world * B * a.steps / el,
                              "unit": "utterances/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                              "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": a.dtype, "data": "synthetic",
                              "config": {"workload": f"BASELINE configs[3]: AE.inference, {a.mels}-mel x {T}-frame source and target, batch {B}/GPU, "
                                                     + ("fp32" if a.dtype == "f32" else "bf16 matrix products (fp32 storage)")}}),
                  flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    for _ in range(a.warmup):
        solver.ae_step(x, 1.0, eps=eps, sync=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if a.presleep_ms > 0:
            torch.cuda._sleep(int(a.presleep_ms * 2.0e6))
        solver.ae_step(x, 1.0, eps=eps, sync=False)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    meta = solver.ae_step(x, 1.0, eps=eps, sync=True)
    if not all(v == v and abs(v) < 1e6 for v in meta.values()):
        raise SystemExit(f"non-finite training state: {meta}")

    if rank == 0:
        print(f"[bench] timed region: {a.steps} steps in {elapsed:.3f}s", file=sys.stderr, flush=True)
        value = world * B * a.steps / elapsed
        out = {
            "metric": "mel-segments/sec (80x128) train step", "value": value, "unit": "mel-segments/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[1]: recon+KL train step (fwd, loss, bwd, clip, Adam-amsgrad), "
                                    f"{a.mels}-mel x {T}-frame segments, batch {B}/GPU, fp32") if a.dtype == "f32" else
                                   (f"BASELINE configs[2]'s compute mode on {world} GPU(s): same train step, conv/Linear operands bf16 "
                                    f"(fp32 accumulate, fp32 master weights and optimizer state), {a.mels}-mel x {T}-frame "
                                    f"segments, batch {B}/GPU -- NOT the headline metric"),
                       "global_batch": world * B, "segment": [a.mels, T], "parallelism": f"dp{world}",
                       "final_losses": meta},
        }
        if world == 1 and not a.no_profile:
            prof = profile_classes(solver, x, eps, steps=3)
            dom = max((k for k in prof if prof[k]["tflops"]), key=lambda k: prof[k]["ms_per_step"])
            d = prof[dom]
            peak = PEAK_FP32_MFMA_TFLOPS if a.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": d["tflops"], "peak": peak,
                               "unit": "TFLOP/s", "frac": d["tflops"] / peak, "traffic": None,
                               "avg_launch_us": d["avg_us"], "flops_per_launch": d["flops_per_launch"],
                               "ms_per_step": d["ms_per_step"]}
            ib = [prof[k] for k in ("instnorm_fwd", "instnorm_bwd") if k in prof]
            if ib:
                C = cfg["ContentEncoder"]["c_h"]
                dom_s = instnorm_dominant_shape(B, C, T)
                bts = dom_s["fwd"]["bytes_per_launch"] + dom_s["bwd"]["bytes_per_launch"]
                us = dom_s["fwd"]["avg_launch_us"] + dom_s["bwd"]["avg_launch_us"]
                gbs = bts / us / 1e3
                tot_b = sum(p["bytes_per_launch"] * p["launches_per_step"] for p in ib)
                tot_ms = sum(p["ms_per_step"] for p in ib)
                tf = pmc_traffic("instnorm_fwd_kernel<32, 1>", B * C * 32)
                tb = pmc_traffic("instnorm_bwd_kernel<32, 1>", B * C * 32)
                out["roofline_instnorm"] = {
                    "kernel": f"instnorm_fwd + instnorm_bwd (IN/AdaIN/ReLU) at the dominant shape [{B},{C},{T}]",
                    "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                    "traffic": (tf + tb) if (tf and tb and B == 256 and T == 128) else None,
                    "algorithmic_bytes": bts, "fwd": dom_s["fwd"], "bwd": dom_s["bwd"],
                    "all_shapes_per_step": {"gbs": tot_b / (tot_ms * 1e-3) / 1e9, "ms": tot_ms, "algorithmic_bytes": tot_b,
                                            "note": "26+26 launches of all T_l, each bracketed by its own event pair"}}
            out["kernel_classes"] = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()
                                         if kk in ("ms_per_step", "launches_per_step", "avg_us", "tflops", "gbs")}
                                     for k, v in prof.items()}
            if a.profile_json:
                with open(a.profile_json, "w") as f:
                    json.dump(prof, f, indent=1)
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(a.mels, T)
            except Exception as e:  # never lose the GPU measurement to the baseline leg
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
