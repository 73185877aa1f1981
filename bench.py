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

N > 1: one process per GPU over RCCL.  Either the caller launches the ranks
(``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or plain
``python bench.py --gpus N`` re-executes itself under torch.distributed.run with N
ranks on 127.0.0.1.  The JSON line reports the world size the process group
actually had and every rank's step time.
"""
import argparse
import ctypes
import json
import os
import sys
import time
import types

import numpy as np

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)
# algorithmic GFLOP of one train step per segment (fwd + dgrad + wgrad minus the bank-conv input gradient)
TRAIN_GFLOP_PER_SEG = {(80, 128): 1.767, (80, 1024): 14.108}   # 3 x forward minus the bank's input gradient (2 x bank forward); T = 1024: 3 x 5.206 - 2 x 0.755
FWD_GFLOP_PER_SEG = {(80, 128): 0.6519, (80, 1024): 5.206}      # SURVEY 8d (one speaker encoder + one content encoder + decoder = AE.forward = AE.inference)


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
    plan = solver.model._plan(x.shape[0], x.shape[2], x.shape[2], x.device)[0]
    was_single = bool(plan.tuning.get("single_stream", 0))
    plan.set_single_stream(True)   # one stream: event brackets then measure each kernel class in isolation
    lib.avc_prof_begin()
    for _ in range(steps):
        solver.ae_step(x, 1.0, sync=False)   # (eps drawn inside the step, model.py:383)
    torch.cuda.synchronize()
    lib.avc_prof_end(ms, launches, flops, nbytes)
    plan.set_single_stream(was_single)
    out = {}
    for i in range(n):
        if launches[i]:
            out[lib.avc_prof_class_name(i).decode()] = dict(
                ms_per_step=ms[i] / steps, launches_per_step=launches[i] / steps, avg_us=1e3 * ms[i] / launches[i],
                tflops=(flops[i] / (ms[i] * 1e-3) / 1e12) if flops[i] else None,
                gbs=(nbytes[i] / (ms[i] * 1e-3) / 1e9) if nbytes[i] else None,
                flops_per_launch=flops[i] / launches[i], bytes_per_launch=nbytes[i] / launches[i])
    return out


def instnorm_dominant_shape(B, C, T, launches=50, pairs=False):
    """IN/AdaIN/ReLU forward + backward at the dominant shape [B, C, T] of the step: `launches`
    back-to-back launches between two HIP events on the launch stream (no per-launch bracket).
    pairs: the bf16 pair-row kernels of compute_dtype "bf16s" (2 bytes per element)."""
    from adaptive_voice_conversion_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = 16 if pairs else 8  # rotate buffers: 8 x 3 x 16.8 MB > 256 MiB Infinity Cache at the bench shape
    if pairs:   # dword tensors [B][C/2][T]: any bit pattern of finite bf16 pairs will do
        ys = [torch.randn(B, C, T, device=dev).to(torch.bfloat16).view(torch.int32).view(B, C // 2, T) for _ in range(nb)]
        outs = [torch.empty_like(y) for y in ys]
        gs = [torch.randn(B, C, T, device=dev).to(torch.bfloat16).view(torch.int32).view(B, C // 2, T) for _ in range(nb)]
    else:
        ys = [torch.randn(B, C, T, device=dev) for _ in range(nb)]
        outs = [torch.empty_like(y) for y in ys]
        gs = [torch.randn_like(y) for y in ys]
    cond = torch.randn(B, 2 * C, device=dev)
    mean, rstd = torch.empty(B * C, device=dev), torch.empty(B * C, device=dev)
    dcond = torch.zeros(B, 2 * C, device=dev)

    def fwd(k, st=st):
        if pairs:
            lib.avc_instnorm_fwd_pairs(P(ys[k]), B, C, T, P(cond), 2 * C, 0, 1, None, 0, 0, 0, P(outs[k]), P(mean), P(rstd), st)
        else:
            lib.avc_instnorm_fwd(P(ys[k]), B, C, T, P(cond), 2 * C, 0, 1, None, 0, 0, P(outs[k]), P(mean), P(rstd), st)

    def bwd(k, st=st):
        if pairs:
            lib.avc_instnorm_bwd_pairs(P(gs[k]), P(ys[k]), P(mean), P(rstd), B, C, T, P(cond), 2 * C, 0, 1, 0, P(outs[k]), P(dcond), 2 * C, 0, st)
        else:
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
        esz = 2.0 if pairs else 4.0
        res[name] = dict(avg_launch_us=us, bytes_per_launch=passes * esz * B * C * T, gbs=passes * esz * B * C * T / us / 1e3, issue="host loop")
    # The host loop above issues one launch per ctypes call (~5 us each): for a kernel of 5-8 us that is the host's launch rate, not
    # the kernel.  The same launches captured ONCE into a HIP graph and replayed are issued by the GPU's command processor back to
    # back; the events bracket the replay.  The faster of the two is reported (with which one it was).
    try:
        side = torch.cuda.Stream()
        for name, fn, passes in (("fwd", fwd, 2), ("bwd", bwd, 3)):
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                st_side = ctypes.c_void_p(side.cuda_stream)
                with torch.cuda.graph(graph, stream=side):
                    for i in range(launches):
                        fn(i % nb, st_side)
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(2):
                graph.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / launches
            res[name]["host_loop_us"] = res[name]["avg_launch_us"]
            res[name]["graph_replay_us"] = us
            if us < res[name]["avg_launch_us"]:
                bpl = res[name]["bytes_per_launch"]
                res[name].update(avg_launch_us=us, gbs=bpl / us / 1e3, issue="HIP graph replay of the same launches")
            del graph
    except Exception as e:   # (capture unsupported: the host-loop numbers stand)
        res["graph_error"] = repr(e)[:200]
    return res


def instnorm_all_shapes(plan, reps=8, pairs=False):
    """Every InstanceNorm / AdaIN launch of ONE train step (the plan's own site list: shapes, affine or not; every second site with an
    identity residual join, as the blocks have), forward in network order then backward in reverse, issued back to back `reps` times
    between ONE pair of HIP events on the launch stream -- no per-launch bracket (an event pair around a 5-us kernel measures the
    bracket: the in-library class profile does that and is reported beside this).  Bytes: the algorithmic 2 N (forward) / 3 N (backward)
    per launch of BASELINE.md (the residual read is not counted)."""
    from adaptive_voice_conversion_amd import _lib
    from adaptive_voice_conversion_amd._lib import ReluSite
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sites = []
    for i in range(plan.lib.avc_plan_num_relu_sites(plan.h)):
        r = ReluSite()
        plan.lib.avc_plan_relu_site(plan.h, i, ctypes.byref(r))
        if r.kind != 0:
            sites.append((r.B, r.C, r.T, r.cond_off >= 0))
    if not sites:
        return None
    esz = 2.0 if pairs else 4.0
    bufs = {}
    def buf(shape, k):   # (one buffer set per distinct shape and role; values are irrelevant to the timing but finite)
        key = (shape, k)
        if key not in bufs:
            B, C, T = shape
            if pairs:
                bufs[key] = torch.randn(B, C, T, device=dev).to(torch.bfloat16).view(torch.int32).view(B, C // 2, T)
            else:
                bufs[key] = torch.randn(B, C, T, device=dev)
        return bufs[key]
    Bm, Cm = max(s_[0] for s_ in sites), max(s_[1] for s_ in sites)
    cond = torch.randn(Bm, 2 * Cm, device=dev)
    dcond = torch.zeros(Bm, 2 * Cm, device=dev)
    mean, rstd = torch.empty(Bm * Cm, device=dev), torch.empty(Bm * Cm, device=dev)
    def fwd(i, s_, stream):
        B, C, T, aff = s_
        y, o = buf((B, C, T), "y%d" % (i % 3)), buf((B, C, T), "o%d" % (i % 3))
        res = buf((B, C, T), "r") if (i & 1) else None
        c = cond if aff else None
        if pairs:
            return lib.avc_instnorm_fwd_pairs(P(y), B, C, T, P(c), 2 * Cm, 0, 1, P(res), 1 if res is not None else 0, T if res is not None else 0, 0, P(o), P(mean), P(rstd), stream)
        return lib.avc_instnorm_fwd(P(y), B, C, T, P(c), 2 * Cm, 0, 1, P(res), 1 if res is not None else 0, T if res is not None else 0, P(o), P(mean), P(rstd), stream)
    def bwd(i, s_, stream):
        B, C, T, aff = s_
        g, y, o = buf((B, C, T), "g%d" % (i % 3)), buf((B, C, T), "y%d" % (i % 3)), buf((B, C, T), "o%d" % (i % 3))
        c = cond if aff else None
        if pairs:
            return lib.avc_instnorm_bwd_pairs(P(g), P(y), P(mean), P(rstd), B, C, T, P(c), 2 * Cm, 0, 1, 0, P(o), P(dcond if aff else None), 2 * Cm, 0, stream)
        return lib.avc_instnorm_bwd(P(g), P(y), P(mean), P(rstd), B, C, T, P(c), 2 * Cm, 0, 1, P(o), P(dcond if aff else None), 2 * Cm, 0, stream)
    def sequence(stream):
        for i, s_ in enumerate(sites):
            assert fwd(i, s_, stream) == 0
        for i, s_ in reversed(list(enumerate(sites))):
            assert bwd(i, s_, stream) == 0
    nbytes = sum((2 + 3) * esz * B * C * T for (B, C, T, _) in sites)
    sequence(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        sequence(st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out = {"launches": 2 * len(sites), "algorithmic_bytes": nbytes, "ms": ms, "gbs": nbytes / (ms * 1e-3) / 1e9, "issue": "host loop",
           "shapes": sorted({(B, C, T) for (B, C, T, _) in sites}, reverse=True),
           "note": "the step's IN launches (plan site list) back to back between one event pair, mean of %d repetitions" % reps}
    try:   # the same sequence replayed from a HIP graph (no host launches); the faster is reported, with which it was
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                sequence(ctypes.c_void_p(side.cuda_stream))
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(2):
            graph.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        gms = e0.elapsed_time(e1) / reps
        out["host_loop_ms"], out["graph_replay_ms"] = ms, gms
        if gms < ms:
            out.update(ms=gms, gbs=nbytes / (gms * 1e-3) / 1e9, issue="HIP graph replay of the same launches")
        del graph
    except Exception as e:
        out["graph_error"] = repr(e)[:200]
    return out


PMC_SUMMARIES = ("r06_pmc_fetch_write_summary.json", "r05_pmc_fetch_write_summary.json")


def build_fingerprint():
    """sha256 (16 hex) over the kernel sources this library was built from: identifies the BUILD a PMC summary belongs to
    (the GPU box has no .git; scripts/pmc_summary.py stores the same fingerprint in the summary's "_meta")."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "adaptive_voice_conversion_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "build.sh"), os.path.join(ROOT, "include", "avc_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


_PMC_NOTE = [None]


def _pmc_load():
    """The committed PMC summary -- only if it was taken on THIS build (fingerprint match); a stale one is refused."""
    fp = build_fingerprint()
    for fn in PMC_SUMMARIES:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except Exception:
            continue
        meta = d.get("_meta", {})
        if meta.get("build_fingerprint") != fp:
            _PMC_NOTE[0] = (f"profiles/{fn} was taken on build {meta.get('build_fingerprint')} (git {meta.get('git_head')}), this library is build {fp}: "
                            "refused (traffic = null) -- re-run the PMC passes of scripts/gpu_r5_final.sh on this build")
            continue
        return d, f"profiles/{fn} (build {fp}, git {meta.get('git_head')})"
    if _PMC_NOTE[0] is None:
        _PMC_NOTE[0] = "no PMC summary under profiles/ for this build"
    return None, None


def pmc_traffic(kernel_prefix, grid=None):
    """HBM bytes per launch of ONE kernel instance (name prefix + grid size) from the rocprofv3 PMC passes committed
    under profiles/ (separate --pmc runs of this same command; FETCH_SIZE doubled on gfx950 as
    MI355X_MICROARCH.md §HBM prescribes, KB units).  Returns (bytes, source file) or (None, None).  It is a
    replayed profile of the same build, not a counter read of THIS run -- the line says so in "traffic_source"."""
    d, src = _pmc_load()
    if d is None:
        return None, None
    for name, grids in d.items():
        if name.startswith(kernel_prefix):
            c = grids.get(str(grid))
            if c and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return (2.0 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024.0, src
    return None, None


def pmc_class_traffic(cls):
    """Launch-weighted mean HBM bytes per launch over every kernel instance of a kernel CLASS of the profile
    (the same population bench.py's per-class event brackets average over)."""
    match = {"conv_wgrad": lambda n: n.startswith("conv_wgrad_kernel"),
             "conv_fwd": lambda n: n.startswith("conv_gemm_kernel") and ", true, " not in n.split("(")[0][:40],
             "conv_dgrad": lambda n: n.startswith("conv_gemm_kernel") and ", true, " in n.split("(")[0][:40]}.get(cls)
    d, src = _pmc_load()
    if d is None or match is None:
        return None, None
    tot = launches = 0.0
    for name, grids in d.items():
        if not match(name):
            continue
        for c in grids.values():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                n = c["FETCH_SIZE"]["launches"]
                tot += n * (2.0 * c["FETCH_SIZE"]["mean_per_launch"] + c["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
                launches += n
    return (tot / launches, src) if launches else (None, None)


SQ_SUMMARIES = ("r06_sq_step.json", "r05_sq_step.json")
_SQ_NOTE = [None]


def sq_mfma_busy():
    """MFMA utilisation per kernel class from the committed SQ-counter passes over THIS command in its single-stream mode
    (scripts/gpu_r5a.sh + scripts/sq_step_summary.py: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE, summed over every
    dispatch of the class): mfma_busy = matrix-pipe busy cycles / (4 SIMDs x 256 CUs) / cycles the kernels were on the chip.  Only a
    summary taken on THIS build (source fingerprint) is replayed -- like `traffic`, it is a profile of the same build, not a counter
    read of this run.  Returns ({class: {...}}, source) or (None, None)."""
    fp = build_fingerprint()
    for fn in SQ_SUMMARIES:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except Exception:
            continue
        meta = d.get("_meta", {})
        if meta.get("build_fingerprint") != fp:
            _SQ_NOTE[0] = (f"profiles/{fn} was taken on build {meta.get('build_fingerprint')} (git {meta.get('git_head')}), this library is build {fp}: "
                           "refused (mfma_busy = null) -- re-run scripts/gpu_r5a.sh on this build")
            continue
        out = {}
        for k, v in d.get("classes", {}).items():
            dd = v.get("derived", {})
            if "mfma_busy" in dd:
                out[k] = {"mfma_busy": dd["mfma_busy"], "launches_profiled": v.get("launches"),
                          "wave_cycles_waiting_on_waitcnt_or_barrier": dd.get("frac_wave_cycles_waitcnt_or_barrier")}
        return out, f"profiles/{fn} (build {fp}, git {meta.get('git_head')})"
    if _SQ_NOTE[0] is None:
        _SQ_NOTE[0] = "no SQ summary under profiles/ for this build"
    return None, None


def _cpu_busy(interval=0.3):
    """Busy fraction of every logical CPU over `interval` seconds (/proc/stat deltas); {} where that file is not readable."""
    def snap():
        out = {}
        try:
            for line in open("/proc/stat"):
                if line.startswith("cpu") and line[3].isdigit():
                    f = line.split()
                    v = [int(x) for x in f[1:9]]
                    out[int(f[0][3:])] = (sum(v), v[3] + v[4])   # (total, idle + iowait)
        except (OSError, ValueError, IndexError):
            return {}
        return out
    a = snap()
    time.sleep(interval)
    b = snap()
    busy = {}
    for c in a:
        if c in b and b[c][0] > a[c][0]:
            busy[c] = 1.0 - (b[c][1] - a[c][1]) / (b[c][0] - a[c][0])
    return busy


def _pick_cpus(n):
    """n logical CPUs for a pinned CPU-baseline point: distinct physical cores of ONE NUMA node -- the node whose cores are idlest right
    now (the GPU boxes are shared hosts: other tenants' jobs sit on some nodes and not on others, and that, not the pinning, was the
    box-to-box spread of rounds 3-4: 429 vs 612 segments/s), and inside it the idlest cores first."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None

    def parse(txt):
        out = []
        for part in txt.strip().split(","):
            if not part:
                continue
            lo, _, hi = part.partition("-")
            out.extend(range(int(lo), int(hi or lo) + 1))
        return out
    nodes = []
    try:
        for d in sorted(os.listdir("/sys/devices/system/node")):
            if d.startswith("node") and d[4:].isdigit():
                nodes.append([c for c in parse(open(f"/sys/devices/system/node/{d}/cpulist").read()) if c in allowed])
    except OSError:
        pass
    nodes = [x for x in nodes if x]
    if not nodes:
        nodes = [allowed]
    busy = _cpu_busy()

    def cores_of(cpus):   # physical cores of a node: (busy of the core = max over its hardware threads, first logical CPU)
        seen, out = {}, []
        for c in cpus:
            try:
                sibs = parse(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read())
            except OSError:
                sibs = [c]
            key = min(sibs)
            if key in seen:
                continue
            seen[key] = True
            out.append((max(busy.get(x, 0.0) for x in sibs), c))
        return out
    per_node = [cores_of(cpus) for cpus in nodes]
    # a node qualifies if it has n physical cores; the idlest n cores of each candidate decide
    cand = [sorted(cs)[:n] for cs in per_node if len(cs) >= n]
    if cand:
        best = min(cand, key=lambda cs: sum(b for b, _ in cs))
        return sorted(c for _, c in best)
    pick = [c for cs in per_node for _, c in sorted(cs)]   # no node is large enough: idlest cores across nodes
    return sorted((pick + [c for c in allowed if c not in pick])[:n])


def cpu_baseline_worker(n_mels, T, point):
    """Child process of ``cpu_baseline``: ONE (batch, threads) point, pinned to `threads` physical cores of one NUMA node (the parent
    sets OMP_PROC_BIND / OMP_PLACES); prints one JSON line per timed step until done or until the parent's deadline kills it."""
    import statistics
    Bc, threads, n_timed = point
    try:
        cpus = sorted(os.sched_getaffinity(0))   # (the OpenMP runtime may already have bound THIS thread to its first place)
    except AttributeError:
        cpus = []
    cpus = [int(c) for c in os.environ.get("AVC_CPU_PIN", "").split(",") if c] or cpus
    from oracle import avc_oracle as O
    cores = os.cpu_count() or 1
    cfg = O.stock_config(n_mels)
    o = cfg["optimizer"]
    O.use_aten_ops(True)   # the reference's own op choices (F.pad reflect, F.instance_norm, ...): its cost profile
    torch.set_num_threads(threads)
    sd = O.make_state_dict(cfg, 0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=o["lr"], betas=(o["beta1"], o["beta2"]), amsgrad=o["amsgrad"],
                           weight_decay=o["weight_decay"])
    x, eps = O.make_inputs(cfg, Bc, T, 0)

    def step():   # solver.py:81-97 around the oracle's forward
        mu, ls, emb, dec = O.ae_forward(x, eps, params, cfg)
        loss_rec, loss_kl = O.losses(x, mu, ls, dec)
        loss = cfg["lambda"]["lambda_rec"] * loss_rec + 1.0 * loss_kl
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), max_norm=o["grad_norm"])
        opt.step()

    step()  # warm-up: allocator, oneDNN primitive caches, optimizer state
    step()  # ... and one step on the warm state (the first timed step of round 4's runs was the lone outlier of every point)
    ts = []
    for _ in range(n_timed):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
        srt = sorted(ts)
        q25, q75 = srt[(len(srt) - 1) // 4], srt[(3 * (len(srt) - 1) + 3) // 4]
        print(json.dumps({"B": Bc, "threads": threads, "steps": len(ts), "median_step_s": statistics.median(ts), "min_step_s": min(ts), "max_step_s": max(ts),
                          "q25_step_s": q25, "q75_step_s": q75,
                          "seg_per_s": Bc / statistics.median(ts), "host_cores": cores, "pinned_cpus": len(cpus) if cpus else 0}), flush=True)


def cpu_baseline(n_mels, T, budget_s=45.0):
    """BASELINE.md §2 protocol on the host cores of this box with the oracle's forward (the same ATen CPU ops the
    reference issues; the reference itself cannot travel to the GPU box -> kind "port"): real in-place
    ``torch.optim.Adam(amsgrad, weight_decay)`` + ``clip_grad_norm_`` around it (solver.py:75-77, 81-97).  Every (batch, threads) point is
    its own child process PINNED to `threads` physical cores of one NUMA node (sched_setaffinity + OMP_PROC_BIND=close): unpinned, the same
    point moved by 2x between runs on a 256-core host (round 3).  The two points that won every sweep (B = 128 and 256 at 16 threads) run
    first with 2 warm-ups + 7 timed steps (median, min-max reported); the other (batch, threads) points follow with 3 steps each while the
    HARD wall-clock budget lasts.  Best median segments/s among the points with >= 5 timed steps is the value."""
    import select
    import subprocess
    t0 = time.perf_counter()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    best_t = 16 if 16 <= cores else max(1, cores)
    plan = [(128, best_t, 7), (256, best_t, 7)]
    for c in (32, 8, 64):
        if c <= cores and c != best_t:
            plan.append((128, c, 3))
    plan += [(4, min(32, cores), 5), (32, min(32, cores), 3), (256, min(32, cores), 3)]
    env = dict(os.environ)
    env.update(OMP_PROC_BIND="close", OMP_PLACES="cores", KMP_AFFINITY="granularity=fine,compact,1,0")
    pts = {}
    for (Bc, th, n) in plan:
        left = budget_s - (time.perf_counter() - t0)
        if left <= 2.0:
            break
        e = dict(env)
        e["OMP_NUM_THREADS"] = str(th)
        cpus = _pick_cpus(th)
        e["AVC_CPU_PIN"] = ",".join(str(c) for c in (cpus or []))
        pin = (lambda cpus=cpus: os.sched_setaffinity(0, cpus)) if cpus else None   # the child STARTS inside the mask: every OpenMP place lies in it
        proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--cpu-point", f"{Bc},{th},{n}",
                                 "--mels", str(n_mels), "--frames", str(T)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, bufsize=1, env=e,
                                preexec_fn=pin)
        try:
            while True:
                left = budget_s - (time.perf_counter() - t0)
                if left <= 0:
                    break
                r, _, _ = select.select([proc.stdout], [], [], left)
                if not r:
                    break
                line = proc.stdout.readline()
                if not line:
                    break
                try:
                    d = json.loads(line)
                    pts[(d["B"], d["threads"])] = d
                except Exception:
                    continue
        finally:
            proc.kill()
            proc.wait()
    if not pts:
        raise RuntimeError(f"no CPU measurement finished within {budget_s:.0f} s")
    full = [d for d in pts.values() if d["steps"] >= 5]   # median of >= 5 timed steps
    best = max(full or pts.values(), key=lambda d: d["seg_per_s"])
    print(f"[bench] cpu_baseline: {len(pts)} points in {time.perf_counter() - t0:.1f}s, best {best}", file=sys.stderr, flush=True)
    return dict(value=best["seg_per_s"], unit="mel-segments/sec", cores=best["threads"], kind="port", host_cores=best["host_cores"],
                spread={"min_seg_per_s": best["B"] / best["max_step_s"], "max_seg_per_s": best["B"] / best["min_step_s"],
                        "rel": (best["max_step_s"] - best["min_step_s"]) / best["median_step_s"],
                        "iqr_rel": (best["q75_step_s"] - best["q25_step_s"]) / best["median_step_s"]},
                pinning=f"each point: own process, sched_setaffinity to {best['threads']} physical cores of one NUMA node, OMP_PROC_BIND=close",
                sample=(f"oracle forward issuing the reference's ATen ops (F.pad reflect, conv1d, F.instance_norm, avg_pool1d(ceil), interpolate) + autograd backward + in-place torch.optim.Adam(amsgrad, L2) + clip_grad_norm_, {n_mels}x{T} segments, "
                        f"torch CPU fp32; (batch, threads) points under a {budget_s:.0f} s wall-clock budget: "
                        f"{len(pts)} points finished (2 warm-ups + median of the timed steps each); best = B {best['B']}, "
                        f"{best['threads']} threads, median of {best['steps']} steps"),
                points=[{"B": d["B"], "threads": d["threads"], "steps": d["steps"], "seg_per_s": round(d["seg_per_s"], 2),
                         "min_seg_per_s": round(d["B"] / d["max_step_s"], 2), "max_seg_per_s": round(d["B"] / d["min_step_s"], 2)}
                        for d in sorted(pts.values(), key=lambda d: (d["B"], d["threads"]))])


def ragged_bench(a, dev):
    """one-shot conversion of N (source, target) utterance pairs of DIFFERENT lengths -- the real inference
    traffic (the reference converts one utterance per call, inference.py:62-70).  Lengths: uniform in [100, 600] frames (1.25 - 7.5 s
    at the stock 12.5 ms hop), seeded.  ONE ragged launch set (avc_forward_ragged) beside the round-2 path (one uniform B=1 plan
    per distinct shape, four streams) and beside one utterance per call."""
    import types
    from adaptive_voice_conversion_amd.inference import Inferencer
    cfg = stock_config(a.mels)
    torch.manual_seed(0)
    inf = Inferencer(cfg, types.SimpleNamespace(model=None, attr=None))
    n = a.batch if a.batch != 256 else 32
    rng = np.random.RandomState(3)
    T, Tc = rng.randint(100, 601, size=n), rng.randint(100, 601, size=n)
    g = torch.Generator().manual_seed(1)
    pairs = [(torch.randn(int(t), a.mels, generator=g).to(dev), torch.randn(int(c), a.mels, generator=g).to(dev)) for t, c in zip(T, Tc)]
    xs, cs = [p_[0] for p_ in pairs], [p_[1] for p_ in pairs]

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    # cold path: real traffic almost never repeats a tuple of lengths -- every call below sees a NEW tuple (the same utterances in another
    # order: the plan cache misses, a plan is created, its tables built and uploaded), next to the cache-hit figure
    perm_state = [0]

    def cold():
        perm_state[0] += 1
        k = perm_state[0] % n
        inf.model.inference_ragged(xs[k:] + xs[:k], cs[k:] + cs[:k])
    with torch.no_grad():
        t_rag = timed(lambda: inf.model.inference_ragged(xs, cs), a.steps, a.warmup)
        t_cold = timed(cold, max(3, min(a.steps, n - 2)), 1)
        inf.model.set_plan_cache_size(inference=2 * n)
        t_bkt = timed(lambda: inf._convert_batch_bucketed(pairs, 4), max(2, a.steps // 4), 1)
        t_one = timed(lambda: [inf.model.inference(x.t()[None], c.t()[None]) for x, c in pairs], max(2, a.steps // 4), 1)
    frames = int(T.sum())
    gflop = 0.6519 / 128 * (T.sum() * (0.2600 + 0.1324) / 0.6519 + Tc.sum() * 0.2594 / 0.6519)   # per-frame forward work of the three networks
    print(json.dumps({"metric": f"utterances/sec, one-shot conversion of {n} pairs of different lengths (100-600 frames) in one ragged launch set",
                      "value": n / t_rag, "unit": "utterances/sec", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * t_rag,
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "SURVEY 8f-1 (BASELINE configs[3] generalised to real utterance lengths): AE.inference over "
                                             f"{n} (source, target) pairs, {frames} source frames in total", "pairs": n, "source_frames": frames,
                                 "ragged_ms": 1e3 * t_rag, "ragged_cold_ms (a NEW tuple of lengths every call: plan creation + tables included)": 1e3 * t_cold,
                                 "compute": inf.model.last_ragged_compute,
                                 "bucketed_4_streams_ms (round 2: a B=1 plan per distinct shape; includes the result download)": 1e3 * t_bkt,
                                 "one_call_per_utterance_ms (the reference's loop)": 1e3 * t_one,
                                 "algorithmic_tflops_ragged": gflop / t_rag / 1e3}}), flush=True)


def dsp_bench(a, dev):
    """Mel <-> waveform DSP of one utterance (preprocess/tacotron/utils.py:34-87, :89-109 with the stock hyper-parameters:
    24 kHz, n_fft 2048, hop 300, window 1200, 512 mels, 100 Griffin-Lim iterations).  A "step" = melspectrogram2wav of one
    utterance, features resident in HBM, the waveform left in HBM (trim's two indices are the only host read)."""
    from adaptive_voice_conversion_amd.dsp import Hyperparams, MelDSP
    hp = Hyperparams
    n = int(a.seconds * hp.sr)
    t = np.arange(n) / hp.sr
    ph = 2 * np.pi * np.cumsum(120 + 30 * np.sin(2 * np.pi * 0.7 * t)) / hp.sr
    y = (0.3 * np.clip(np.sin(2 * np.pi * 1.3 * t), 0, None) ** 2 * sum(np.sin(k * ph) / k for k in range(1, 12))
         + 0.002 * np.random.RandomState(0).randn(n)).astype(np.float32)
    dsp = MelDSP(hp, device=dev)
    yd = torch.from_numpy(y).to(dev)
    mel, _ = dsp.get_spectrograms_device(yd, do_trim=False)
    T = mel.shape[0]

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    t_front = timed(lambda: dsp.get_spectrograms_device(yd, do_trim=False), a.steps, a.warmup)
    t_back = timed(lambda: dsp.melspectrogram2wav(mel, do_trim=False), a.steps, a.warmup)
    nb = a.batch if a.batch != 256 else 8      # (--batch defaults to the train step's 256)
    mels = [mel] * nb
    t_batch = timed(lambda: dsp.melspectrogram2wav_batch(mels, do_trim=False), max(1, a.steps // 2), 1)
    F2, K = hp.n_fft + 2, hp.win_length
    flops = (2 * hp.n_iter + 1) * 2.0 * F2 * K * T + 2.0 * (hp.n_fft // 2 + 1) * hp.n_mels * T
    out = {"metric": f"utterances/sec mel -> waveform (melspectrogram2wav, {hp.n_iter} Griffin-Lim iterations, {a.seconds:g} s of {hp.sr} Hz audio = {T} frames)",
           "value": 1.0 / t_back, "unit": "utterances/sec", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * t_back,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "SURVEY §8f row 4 (not a BASELINE.json config): mel <-> waveform DSP of one utterance, stock hyper-parameters",
                      "frames": T, "front_end_ms (get_spectrograms, no trim)": 1e3 * t_front,
                      f"batched: {nb} equally long utterances per call (melspectrogram2wav_batch)": {
                          "ms_per_call": 1e3 * t_batch, "utterances_per_sec": nb / t_batch, "gemm_tflops": nb * flops / t_batch / 1e12}},
           "roofline": {"kernel": "conv_gemm 1x1 (the STFT / iSTFT GEMMs against the windowed DFT bases)", "bound": "mfma",
                        "achieved": flops / t_back / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / t_back / 1e12 / 157.3,
                        "traffic": None,
                        "note": "algorithmic GEMM flops of the whole call over its wall time (row kernels and launch gaps included)"}}
    if not a.no_cpu_baseline:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from oracle import dsp_oracle as D   # the CPU restatement (numpy float64, pocketfft), timed on this host
        melh = mel.cpu().numpy()
        t0 = time.perf_counter()
        ref = D.melspectrogram2wav(melh, D.Hyperparams, do_trim=False)
        tc = time.perf_counter() - t0
        wav = dsp.melspectrogram2wav(melh, do_trim=False)
        out["cpu_baseline"] = {"value": 1.0 / tc, "unit": "utterances/sec", "cores": 1, "kind": "port",
                               "sample": f"oracle/dsp_oracle.py melspectrogram2wav (numpy rfft / irfft per iteration), the same {T}-frame utterance, one run"}
        out["config"]["waveform rel-L2 vs the float64 oracle after 100 iterations"] = float(np.linalg.norm(wav - ref) / np.linalg.norm(ref))
    print(json.dumps(out), flush=True)


def config2_bf16_record(a, dev, g, x):
    """The per-GPU share of BASELINE configs[2] (bf16, 256 segments per GPU) on ONE GPU: compute_dtype "bf16" = the bf16 storage engine
    (DESIGN 3.4).  Same harness as the headline: `a.warmup` untimed steps, `a.steps` timed ones between synchronisations."""
    from adaptive_voice_conversion_amd.solver import Solver
    cfg = stock_config(a.mels)
    cfg["compute_dtype"] = "bf16s"
    B, T = a.batch, a.frames
    torch.manual_seed(0)
    args = types.SimpleNamespace(store_model_path=None, load_model=False, data_dir=None, logdir="/tmp/avc_bench_log", tuning={})
    solver = Solver(cfg, args)
    plan, _ = solver.model._plan(B, T, T, dev)
    eps = torch.randn(B, cfg["ContentEncoder"]["c_out"], plan.latent_len, generator=g).to(dev)
    for _ in range(a.warmup):
        solver.ae_step(x, 1.0, sync=False)   # (eps drawn inside the step, model.py:383)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        solver.ae_step(x, 1.0, sync=False)   # (eps drawn inside the step, model.py:383)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    meta = solver.ae_step(x, 1.0, eps=eps, sync=True)
    rec = {"workload": f"BASELINE.json configs[2], one GPU's share: batch {B} x {a.mels}-mel x {T}, bf16 storage engine "
                       "(bf16 activations / weight images, fp32 accumulation, statistics, master weights and optimizer state), recon+KL train step",
           "baseline_config_index": 2, "dtype": "bf16", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "pair_storage": bool(getattr(plan, "pair_storage", False)),
           "ms_per_step": 1e3 * el / a.steps, "value": B * a.steps / el, "unit": "mel-segments/sec", "final_losses": meta,
           "note": "the 8-GPU data-parallel run of this configuration is the driver's (bench.py --gpus 8 --dtype bf16)"}
    if not all(v == v and abs(v) < 1e6 for v in meta.values()):
        rec["error"] = "non-finite training state"
        return rec
    if not a.no_profile:
        prof = profile_classes(solver, x, eps, steps=3)
        dom = max((k for k in prof if prof[k]["tflops"]), key=lambda k: prof[k]["ms_per_step"])
        d = prof[dom]
        whole = TRAIN_GFLOP_PER_SEG.get((a.mels, T), 0.0) * B / 1e3 / (el / a.steps)
        rec["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": d["tflops"], "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": d["tflops"] / PEAK_BF16_MFMA_TFLOPS, "traffic": None, "avg_launch_us": d["avg_us"],
                           "ms_per_step": d["ms_per_step"], "whole_step": {"tflops": whole, "frac": whole / PEAK_BF16_MFMA_TFLOPS}}
        C = cfg["ContentEncoder"]["c_h"]
        dom_s = instnorm_dominant_shape(B, C, T, pairs=True)
        bts = dom_s["fwd"]["bytes_per_launch"] + dom_s["bwd"]["bytes_per_launch"]
        us = dom_s["fwd"]["avg_launch_us"] + dom_s["bwd"]["avg_launch_us"]
        rec["roofline_instnorm"] = {"kernel": f"instnorm_fwd/bwd_pairs (bf16 pair rows) at [{B},{C},{T}]", "bound": "hbm", "achieved": bts / us / 1e3,
                                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": bts / us / 1e3 / PEAK_HBM_GBS, "algorithmic_bytes": bts,
                                    "fwd": dom_s["fwd"], "bwd": dom_s["bwd"]}
        rec["kernel_classes"] = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()
                                     if kk in ("ms_per_step", "launches_per_step", "avg_us", "tflops", "gbs")} for k, v in prof.items()}
    return rec


def config2_in_child(a):
    """config2_bf16_record in a fresh process of the same interpreter (same seeds, same harness); the record says so."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config2-worker", "--steps", str(a.steps), "--warmup", str(a.warmup), "--batch", str(a.batch),
           "--mels", str(a.mels), "--frames", str(a.frames), "--no-cpu-baseline", "--no-config2"] + (["--no-profile"] if a.no_profile else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"config2 worker rc={r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    rec = json.loads(lines[-1])
    rec["process"] = "child process of this run (its own HIP context): python bench.py --config2-worker"
    return rec


def extra_config_in_child(a, name):
    """Driver-visible numbers for the other single-GPU BASELINE configs: bench.py ITSELF in a fresh process with that
    config's arguments (same harness: `warmup` untimed passes, `steps` timed ones between synchronisations), compacted into a sub-record.
    config3_infer_b1024 = BASELINE configs[3] (inference B = 1024); config4_t1024_b64 = configs[4] (T = 1024, B = 64 train step: the
    "HBM-bound InstanceNorm regime" of BASELINE.json -- its roofline_instnorm is taken at [64, 128, 1024])."""
    import subprocess
    argv = {"config3_infer_b1024": ["--mode", "infer", "--batch", "1024", "--frames", "128"],
            "config4_t1024_b64": ["--batch", "64", "--frames", "1024"]}[name]
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(a.steps), "--warmup", str(a.warmup), "--mels", str(a.mels),
           "--no-cpu-baseline", "--no-config2"] + argv + (["--no-profile"] if a.no_profile else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"{name} worker rc={r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    d = json.loads(lines[-1])
    rec = {"workload": d["config"]["workload"], "baseline_config_index": d["config"].get("baseline_config_index"), "dtype": d["dtype"],
           "n_gpus": 1, "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"],
           "process": "child process of this run (its own HIP context): python bench.py " + " ".join(argv)}
    for k in ("roofline", "roofline_instnorm", "kernel_classes"):
        if k in d:
            rec[k] = d[k]
    if "final_losses" in d["config"]:
        rec["final_losses"] = d["config"]["final_losses"]
    return rec


def workload_label(a, world):
    """Which BASELINE.json config (if any) the arguments correspond to, and a metric string that names the real shape."""
    prec = {"f32": "fp32", "bf16r": "bf16 matrix products on fp32 storage: operands rounded as they enter the matrix core (fp32 accumulate, fp32 master "
                                    "weights and optimizer state)",
            "bf16": "bf16 matrix products AND bf16 storage of activations / activation gradients (fp32 accumulate and statistics, fp32 master "
                    "weights and optimizer state)",
            "f32x3": "fp32 storage and results; the big conv / weight-gradient products from three bf16 terms per operand on the bf16 matrix core "
                     "(fp32-level accuracy, opt-in; DESIGN 3.5) -- NOT the headline precision path"}[a.dtype]
    if a.mode == "infer":
        idx = 3 if (a.mels == 80 and a.frames == 128 and a.batch == 1024) else None
        what = f"AE.inference one-shot conversion, {a.mels}-mel x {a.frames}-frame source and target, batch {a.batch}/GPU, {prec}"
        metric = f"utterances/sec one-shot conversion (AE.inference, {a.mels}x{a.frames})"
    else:
        idx = None
        if a.mels == 80 and a.frames == 128 and a.batch == 256:
            idx = {"f32": 1, "bf16": 2, "bf16r": 2}.get(a.dtype)
        elif a.mels == 80 and a.frames == 1024 and a.batch == 64 and a.dtype == "f32":
            idx = 4
        what = (f"recon+KL train step (fwd, loss, bwd, {'RCCL all-reduce, ' if world > 1 else ''}clip, Adam-amsgrad), "
                f"{a.mels}-mel x {a.frames}-frame segments, batch {a.batch}/GPU, {prec}")
        metric = f"mel-segments/sec ({a.mels}x{a.frames}) train step"
    tag = f"BASELINE configs[{idx}]" if idx is not None else "not a BASELINE.json config (shape given on the command line)"
    if idx == 2:
        tag += f" (its per-GPU workload on {world} GPU(s); configs[2] itself is 8 GPUs, global batch 2048)"
    return metric, f"{tag}: {what}", idx


def bind_rank_to_cores(local_rank, local_world):
    """N ranks x ~200 kernel launches per 2.5 - 6 ms step share one host: give every rank its own contiguous slice of the cores this
    process may run on (on a two-socket node contiguous core numbers are one NUMA node, so a rank's launch thread, its RCCL proxy thread and
    their memory stay on one socket) instead of letting N launch threads migrate over all of them.  Best effort: no-op where
    sched_setaffinity is unavailable or the slice would be empty."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // max(local_world, 1)
        if per < 1:
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return mine
    except (AttributeError, OSError):
        return None


def relaunch_ranks(a):
    """``python bench.py --gpus N`` without a launcher: start N ranks (one per GPU) on this node."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] launching {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config 2: 256)")
    ap.add_argument("--mels", type=int, default=80)
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--mode", choices=("train", "infer", "dsp", "ragged"), default="train",
                    help="train = BASELINE configs[1]/[4]-style train step (the headline metric); infer = configs[3] one-shot conversion; "
                         "dsp = the mel <-> waveform back end of a conversion (SURVEY §8f row 4; not a BASELINE.json config)")
    ap.add_argument("--seconds", type=float, default=5.0, help="--mode dsp: length of the synthetic utterance")
    ap.add_argument("--dtype", choices=("f32", "bf16", "bf16s", "bf16r", "f32x3"), default="f32",
                    help="f32 = the headline (BASELINE configs[1]); bf16 = configs[2]'s compute mode (bf16 matrix products, "
                         "fp32 master weights / optimizer state) -- a separate, non-headline measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-config2", action="store_true", help="skip the config2_bf16 / config3_infer_b1024 / config4_t1024_b64 sub-records of the default (configs[1]) line")
    ap.add_argument("--single-stream", action="store_true", help="profiling aid: every kernel on the caller's stream")
    ap.add_argument("--feed", action="store_true", help="draw every batch from an HBM-resident synthetic corpus through the "
                                                        "device-side gather kernel (DeviceSegmentFeed) inside the timed loop")
    ap.add_argument("--presleep-ms", type=float, default=0.0,
                    help="tracing aid: park the GPU this long before every timed step so that the (tracer-slowed) host has "
                         "the whole step enqueued when it starts; the reported time is then meaningless")
    ap.add_argument("--profile-json", default=None, help="write the per-kernel-class table here")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus N > 1: nccl (= RCCL over xGMI, the real thing) or gloo (CUDA tensors staged through the "
                         "host: lets N ranks SHARE one GPU, which executes the multi-rank code path on a 1-GPU box; not a scaling number)")
    ap.add_argument("--tune", action="append", default=[], metavar="NAME=VALUE",
                    help="avc_tuning field captured by the plans (A/B measurements), e.g. wgrad_batch=1, kg_wgs=0")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--config2-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-point", default="128,16,7", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_worker:
        return cpu_baseline_worker(a.mels, a.frames, tuple(int(v) for v in a.cpu_point.split(",")))

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_ranks(a)   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    if a.dist_backend == "gloo":
        local = local % torch.cuda.device_count()   # ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        bind_rank_to_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.dist_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        world = dist.get_world_size()   # what the RCCL process group actually has
    if a.gpus != world and rank == 0:
        print(f"warning: --gpus {a.gpus} but the process group has {world} rank(s)", file=sys.stderr)

    if a.mode == "dsp":
        return dsp_bench(a, dev)
    if a.mode == "ragged":
        return ragged_bench(a, dev)

    from adaptive_voice_conversion_amd import _lib
    from adaptive_voice_conversion_amd.solver import Solver
    tuning = {}
    if a.single_stream:
        tuning["single_stream"] = 1
    for kv in a.tune:
        k, v = kv.split("=")
        tuning[k] = int(v)
    try:
        _lib.make_tuning(_lib.load(), tuning)
    except KeyError as e:
        raise SystemExit(str(e))
    cfg = stock_config(a.mels)
    if a.dtype == "bf16s":
        a.dtype = "bf16"
    if a.dtype == "bf16":    # configs[2] on the bf16 storage engine (bf16 channel-pair tensors)
        cfg["compute_dtype"] = "bf16s"
        cfg["inference_compute_dtype"] = "bf16s"   # (--mode infer --dtype bf16 measures the pair-storage engine; AE's default for inference is "bf16r")
    if a.dtype == "bf16r":   # ... on fp32 storage with operand rounding (round 2's bf16 mode)
        cfg["compute_dtype"] = "bf16r"
    if a.dtype == "f32x3":   # opt-in: fp32-accurate products from three bf16 terms on the bf16 matrix core
        cfg["compute_dtype"] = "fp32x3"
    torch.manual_seed(0)
    args = types.SimpleNamespace(store_model_path=None, load_model=False, data_dir=None, logdir="/tmp/avc_bench_log", tuning=tuning)
    solver = Solver(cfg, args)
    B, T = a.batch, a.frames
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)   # each rank: its own shard of the global batch
    x = torch.randn(B, a.mels, T, generator=g).to(dev)
    if a.config2_worker:   # child of the default run: the config2_bf16 sub-record in a process (HIP context, hardware queues) of its own
        del solver
        print(json.dumps(config2_bf16_record(a, dev, g, x)), flush=True)
        return
    metric, workload, cfg_idx = workload_label(a, world)
    dtype_label = a.dtype
    if any(kv.split("=")[0] == "conv_x3" and int(kv.split("=")[1]) for kv in a.tune):   # say so: not the exact-fp32 products
        dtype_label = a.dtype + " (opt-in conv_x3: the big k=5 conv products as 3 bf16 terms x 6 bf16 MFMAs, fp32 accumulate; fp32-level accuracy)"
        workload += " -- NOT the headline precision path: split-bf16 conv products (DESIGN 3.5)"
        cfg_idx = None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def rank_times(elapsed):
        """max over ranks (the contract's number) + every rank's own time"""
        if world == 1:
            return elapsed, [elapsed]
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        ts = [float(v.item()) for v in allt]
        return max(ts), ts

    if a.mode == "infer":
        # BASELINE configs[3]: batched one-shot conversion AE.inference(x, x_cond) (model.py:387-391), forward only
        xc = torch.randn(B, a.mels, T, generator=g).to(dev)
        model = solver.model
        plan_i, ws_i = model._plan(B, T, T, dev, "inference")
        for _ in range(a.warmup):
            plan_i.forward(model.flat_parameters(), x, xc, None, ws_i)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            plan_i.forward(model.flat_parameters(), x, xc, None, ws_i)
        barrier()
        el, per_rank = rank_times(time.perf_counter() - t0)
        if rank == 0:
            rec = {"metric": metric, "value": world * B * a.steps / el,
                   "unit": "utterances/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                   "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": dtype_label, "data": "synthetic",
                   "config": {"workload": workload, "baseline_config_index": cfg_idx, "world_size": world,
                              "per_rank_ms_per_step": [1e3 * t / a.steps for t in per_rank]}}
            if (a.mels, T) in FWD_GFLOP_PER_SEG and world == 1:
                peak = {"f32": PEAK_FP32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16r": PEAK_BF16_MFMA_TFLOPS, "f32x3": PEAK_BF16_MFMA_TFLOPS / 6.0}[a.dtype]
                tf = FWD_GFLOP_PER_SEG[(a.mels, T)] * B / 1e3 / (el / a.steps)
                rec["roofline"] = {"kernel": "whole forward pass (conv_gemm launches: every Conv1d / Linear of AE.inference)", "bound": "mfma", "achieved": tf,
                                   "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
                                   "algorithmic_gflop_per_pass": FWD_GFLOP_PER_SEG[(a.mels, T)] * B}
                if not a.no_profile:
                    try:   # the InstanceNorm / AdaIN rows of a forward pass that still run as row kernels, at this batch's dominant shape (forward only)
                        dom_s = instnorm_dominant_shape(B, cfg["ContentEncoder"]["c_h"], T, pairs=False)
                        f_ = dom_s["fwd"]
                        rec["roofline_instnorm"] = {"kernel": f"instnorm_fwd (IN/AdaIN/ReLU) at [{B},{cfg['ContentEncoder']['c_h']},{T}]", "bound": "hbm",
                                                    "achieved": f_["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": f_["gbs"] / PEAK_HBM_GBS,
                                                    "algorithmic_bytes": f_["bytes_per_launch"], "fwd": f_}
                    except Exception as e:
                        rec["roofline_instnorm"] = {"error": repr(e)[:200]}
            print(json.dumps(rec), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    plan, _ = solver.model._plan(B, T, T, dev)
    eps = torch.randn(B, cfg["ContentEncoder"]["c_out"], plan.latent_len, generator=g).to(dev)
    feed = None
    if a.feed:   # HBM-resident synthetic corpus (2048 utterances' worth of frames), per-rank disjoint shards
        from adaptive_voice_conversion_amd.device_feed import DeviceSegmentFeed
        gc = torch.Generator(device="cpu").manual_seed(99)
        rows = 2048 * (T + 64)
        data = {"corpus": torch.randn(rows, a.mels, generator=gc).numpy()}
        n_idx = 64 * B * max(world, 1)
        idx = [["corpus", int(t)] for t in torch.randint(0, rows - T, (n_idx,), generator=gc).tolist()]
        feed = DeviceSegmentFeed(data, idx, T, B, dev, shuffle=True, seed=0, rank=rank, world_size=world)

    def one_step():   # the reparameterisation noise (model.py:383 `normal_`) is drawn INSIDE the step, as the reference does
        xb = next(feed) if feed is not None else x
        solver.ae_step(xb, 1.0, sync=False)

    for _ in range(a.warmup):
        one_step()
    barrier()
    if world > 1:   # how much of the gradient all-reduce the schedule did NOT hide (event pair around the compute stream's wait)
        solver.measure_allreduce = True
        solver._ar_events.clear()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if a.presleep_ms > 0:
            torch.cuda._sleep(int(a.presleep_ms * 2.0e6))
        one_step()
    barrier()
    elapsed, per_rank = rank_times(time.perf_counter() - t0)
    exposed_ar = solver.exposed_allreduce_ms() if world > 1 else None
    solver.measure_allreduce = False
    meta = solver.ae_step(x, 1.0, eps=eps, sync=True)
    if not all(v == v and abs(v) < 1e6 for v in meta.values()):
        raise SystemExit(f"non-finite training state: {meta}")

    if rank == 0:
        print(f"[bench] timed region: {a.steps} steps in {elapsed:.3f}s on {world} rank(s)", file=sys.stderr, flush=True)
        value = world * B * a.steps / elapsed
        out = {
            "metric": metric, "value": value, "unit": "mel-segments/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_label, "data": "synthetic",
            "config": {"workload": workload + (" + device-side segment gather from an HBM-resident corpus inside the timed loop" if feed else ""),
                       "baseline_config_index": cfg_idx, "global_batch": world * B, "segment": [a.mels, T], "parallelism": f"dp{world}",
                       "world_size": world, "dist_backend": (a.dist_backend if world > 1 else None), "per_rank_ms_per_step": [1e3 * t / a.steps for t in per_rank],
                       "exposed_allreduce_ms": exposed_ar, "allreduce_buckets": (int(cfg.get("allreduce_buckets", 3)) if world > 1 else None),
                       "tuning": a.tune or None,
                       "allreduce": ("decoder range on a communication stream under the encoders' backward, encoders' range after it; "
                                     + ("RCCL via torch.distributed nccl" if a.dist_backend == "nccl" else f"torch.distributed {a.dist_backend} (staged through the host: executes the path, not a scaling number)")) if world > 1 else None,
                       "final_losses": meta},
        }
        if world == 1 and not a.no_profile:
            prof = profile_classes(solver, x, eps, steps=3)
            # the weight gradient's reduce launches belong to its class when classes are ranked (they carry no FLOPs of their own)
            red_ms = prof.get("slab_reduce", {}).get("ms_per_step", 0.0)
            dom = max((k for k in prof if prof[k]["tflops"]), key=lambda k: prof[k]["ms_per_step"] + (red_ms if k == "conv_wgrad" else 0.0))
            d = prof[dom]
            # f32x3: six 8-pass bf16 MFMAs per 16 reduction steps -> the matrix pipe's ceiling for these products is the bf16 peak / 6
            peak = {"f32": PEAK_FP32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16r": PEAK_BF16_MFMA_TFLOPS, "f32x3": PEAK_BF16_MFMA_TFLOPS / 6.0}[a.dtype]
            traffic, tsrc = pmc_class_traffic(dom) if (cfg_idx == 1) else (None, None)
            sq, sq_src = sq_mfma_busy() if (cfg_idx == 1 and a.dtype == "f32") else (None, None)
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": d["tflops"], "peak": peak,
                               "unit": "TFLOP/s", "frac": d["tflops"] / peak, "traffic": traffic,
                               "traffic_source": (f"{tsrc}: launch-weighted mean HBM bytes per launch over the kernel instances of this class "
                                                  "(2 x FETCH_SIZE + WRITE_SIZE), separate rocprofv3 --pmc passes of this command on the same "
                                                  "build (replayed, not a counter read of this run)") if tsrc else _PMC_NOTE[0],
                               "avg_launch_us": d["avg_us"], "flops_per_launch": d["flops_per_launch"],
                               "ms_per_step": d["ms_per_step"],
                               "with_reduce_launches": ({"ms_per_step": d["ms_per_step"] + red_ms,
                                                         "tflops": d["tflops"] * d["ms_per_step"] / (d["ms_per_step"] + red_ms)} if dom == "conv_wgrad" else None),
                               "mfma_classes": {k: {"tflops": prof[k]["tflops"], "frac": prof[k]["tflops"] / peak, "ms_per_step": prof[k]["ms_per_step"],
                                                    "mfma_busy": (sq or {}).get(k, {}).get("mfma_busy") if (cfg_idx == 1 and a.dtype == "f32") else None}
                                                for k in prof if prof[k]["tflops"]},
                               "mfma_busy": (sq or {}).get(dom, {}).get("mfma_busy") if (cfg_idx == 1 and a.dtype == "f32") else None,
                               "mfma_busy_source": ((sq_src + ": SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) / (GRBM_GUI_ACTIVE / 8 XCDs), summed "
                                                     "over every dispatch of the class in the single-stream step (replayed, not a counter read of this run)")
                                                    if sq_src else _SQ_NOTE[0]) if (cfg_idx == 1 and a.dtype == "f32") else None,
                               "whole_step": {"algorithmic_tflop_per_step": TRAIN_GFLOP_PER_SEG.get((a.mels, T), 0.0) * B / 1e3,
                                              "tflops": (TRAIN_GFLOP_PER_SEG[(a.mels, T)] * B / 1e3 / (elapsed / a.steps)) if (a.mels, T) in TRAIN_GFLOP_PER_SEG else None,
                                              "frac": (TRAIN_GFLOP_PER_SEG[(a.mels, T)] * B / 1e3 / (elapsed / a.steps) / peak) if (a.mels, T) in TRAIN_GFLOP_PER_SEG else None}}
            ib = [prof[k] for k in ("instnorm_fwd", "instnorm_bwd") if k in prof]
            if ib:
                C = cfg["ContentEncoder"]["c_h"]
                dom_s = instnorm_dominant_shape(B, C, T, pairs=(a.dtype == "bf16"))
                bts = dom_s["fwd"]["bytes_per_launch"] + dom_s["bwd"]["bytes_per_launch"]
                us = dom_s["fwd"]["avg_launch_us"] + dom_s["bwd"]["avg_launch_us"]
                gbs = bts / us / 1e3
                tot_b = sum(p["bytes_per_launch"] * p["launches_per_step"] for p in ib)
                tot_ms = sum(p["ms_per_step"] for p in ib)
                tf, s1 = pmc_traffic("instnorm_fwd_kernel<32, 1>", B * C * 32)
                tb, s2 = pmc_traffic("instnorm_bwd_kernel<32, 1>", B * C * 32)
                have = bool(tf and tb and B == 256 and T == 128 and a.dtype != "bf16")
                out["roofline_instnorm"] = {
                    "kernel": f"instnorm_fwd + instnorm_bwd (IN/AdaIN/ReLU) at the dominant shape [{B},{C},{T}]",
                    "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                    "traffic": (tf + tb) if have else None,
                    "traffic_source": (f"{s1}: rocprofv3 --pmc passes of this command on the same kernels (replayed, not a counter read of this run)") if have else None,
                    "algorithmic_bytes": bts, "fwd": dom_s["fwd"], "bwd": dom_s["bwd"],
                    "all_shapes_per_step": {"gbs": tot_b / (tot_ms * 1e-3) / 1e9, "ms": tot_ms, "algorithmic_bytes": tot_b,
                                            "note": "all IN launches of one step (every T_l), each bracketed by its own event pair (the bracket of a 5-us kernel includes event overhead)"}}
                try:
                    plan_, _ws = solver.model._plan(B, T, T, dev)
                    out["roofline_instnorm"]["all_shapes_back_to_back"] = instnorm_all_shapes(plan_, pairs=(a.dtype == "bf16" and getattr(plan_, "pair_storage", False)))
                except Exception as e:   # never lose the bench line to a diagnostic
                    out["roofline_instnorm"]["all_shapes_back_to_back"] = {"error": repr(e)[:200]}
            out["kernel_classes"] = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()
                                         if kk in ("ms_per_step", "launches_per_step", "avg_us", "tflops", "gbs")}
                                     for k, v in prof.items()}
            if a.profile_json:
                with open(a.profile_json, "w") as f:
                    json.dump(prof, f, indent=1)
        if world == 1 and cfg_idx == 1 and a.dtype == "f32" and not a.no_config2 and not feed:
            # BASELINE configs[2] is "8 x MI355X data-parallel, global batch 2048, bf16": its per-GPU half -- the bf16 STORAGE engine at 256
            # segments per GPU -- is measured here with the same harness (same warm-up / step counts, same timing brackets), so that the
            # driver's default run carries a driver-timed bf16 number.  `value` above stays the exact-fp32 step.
            # ... in a CHILD process: a second plan created in a process that already drove another plan's helper streams lands its own streams
            # on whatever hardware queues the round-robin has reached, and can come out 1.8 - 2.4x slower (measured in round 5,
            # scripts/two_plans_probe.py: 4.9 - 6.1 vs 2.59 ms; the lottery of DESIGN appendix (1b)(ii)) -- the sub-record must not depend on it.
            try:
                out["config2_bf16"] = config2_in_child(a)
            except Exception as e:   # never lose the headline line to the sub-record
                out["config2_bf16"] = {"error": repr(e)[:300]}
            for name in ("config3_infer_b1024", "config4_t1024_b64"):   # the other single-GPU BASELINE configs, driver-timed (~10 s each)
                try:
                    out[name] = extra_config_in_child(a, name)
                except Exception as e:
                    out[name] = {"error": repr(e)[:300]}
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
