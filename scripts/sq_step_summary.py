"""Per-kernel and per-class SQ counter summary of a whole bench step (rocprofv3 --pmc passes over `bench.py --single-stream`).

usage: sq_step_summary.py OUT.json PASS_DIR [PASS_DIR ...]

Every pass directory holds one *counter_collection.csv (one row per dispatch and counter, with the dispatch's start / end
timestamps).  Dispatches are classified by kernel name and by their position in the step: conv_gemm launches in front of the
step's loss kernel are forward convolutions, those behind it input gradients.  Derived per kernel / class:

  mfma_busy        = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs) / window,  window = GRBM_GUI_ACTIVE / XCDs   (cycles the
                     matrix pipes of the chip were busy over the cycles the kernel was on the chip)
  mfma_busy_wave   = the same over the mean wave lifetime (SQ_WAVE_CYCLES x 4 / SQ_WAVES)

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES is cycles summed
over SIMDs; GRBM_GUI_ACTIVE is cycles summed over the 8 XCDs (MI355X_MICROARCH.md, rocprofv3 section)."""
import collections
import csv
import glob
import json
import os
import sys

CUS, SIMDS, XCDS = 256, 4, 8


def short(name):
    n = name.replace("void ", "")
    i = n.find("(")
    return n[:i] if i > 0 else n


def load(path):
    """dispatch id -> (name, grid, start, end, {counter: value})"""
    disp = {}
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d = int(r["Dispatch_Id"])
            e = disp.setdefault(d, [short(r["Kernel_Name"]), r.get("Grid_Size", ""), int(r.get("Start_Timestamp") or 0), int(r.get("End_Timestamp") or 0), {}])
            e[4][r["Counter_Name"]] = e[4].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return disp


def classify(disp):
    """class per dispatch id; a step = the dispatches between two optimizer launches"""
    cls = {}
    seen_loss = False
    for d in sorted(disp):
        n = disp[d][0]
        if n.startswith("loss_") or "loss_kernel" in n:
            seen_loss = True
        if "clip_adam" in n:
            seen_loss = False
        if n.startswith("conv_gemm_kernel") or n.startswith("conv_x3_kernel"):
            cls[d] = "conv_dgrad" if seen_loss else "conv_fwd"
        elif n.startswith("conv_wgrad_kernel"):
            cls[d] = "conv_wgrad"
        elif n.startswith("wgrad_reduce"):
            cls[d] = "slab_reduce"
        elif n.startswith("instnorm") or n.startswith("rag_instnorm"):
            cls[d] = "instnorm"
        elif n.startswith("dense_stack"):
            cls[d] = "dense"
        else:
            cls[d] = "other"
    return cls


def derive(c, n):
    out = {}
    win = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if busy is not None and win > 0:
        out["mfma_busy"] = busy / (SIMDS * CUS) / win
        out["window_cycles_per_launch"] = win / n
    if busy is not None and c.get("SQ_WAVES") and c.get("SQ_WAVE_CYCLES"):
        life = 4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]
        out["wave_lifetime_cycles"] = life
        out["mfma_busy_cycles_per_simd_per_launch"] = busy / (SIMDS * CUS) / n
        # (only meaningful for a launch that is ONE resident round of workgroups)
    if c.get("SQ_WAVE_CYCLES"):
        for k, lab in (("SQ_WAIT_ANY", "frac_wave_cycles_waitcnt_or_barrier"), ("SQ_WAIT_INST_ANY", "frac_wave_cycles_issue_stall"),
                       ("SQ_ACTIVE_INST_ANY", "frac_wave_cycles_issuing")):
            if k in c:
                out[lab] = c[k] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_WAVES"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM", "SQ_INSTS_LDS"):
            if k in c:
                out[k.lower() + "_per_wave"] = c[k] / c["SQ_WAVES"]
    return out


def main():
    out_path, passes = sys.argv[1], sys.argv[2:]
    per_kernel = collections.defaultdict(lambda: {"launches": 0, "ns": 0, "counters": collections.defaultdict(float)})
    per_class = collections.defaultdict(lambda: {"launches": 0, "ns": 0, "counters": collections.defaultdict(float)})
    npass = 0
    for p in passes:
        disp = load(p)
        if not disp:
            continue
        npass += 1
        cls = classify(disp)
        for d, (name, grid, t0, t1, ctr) in disp.items():
            for key, table in (((name, grid, cls[d]), per_kernel), (cls[d], per_class)):
                e = table[key]
                for k, v in ctr.items():
                    e["counters"][k] += v
                e["counters"]["_n_" + "+".join(sorted(ctr))] += 1   # launches seen by THIS counter set
                if npass == 1:
                    e["launches"] += 1
                    e["ns"] += t1 - t0
    def pack(e):
        c = dict(e["counters"])
        nsets = {k: v for k, v in c.items() if k.startswith("_n_")}
        c = {k: v for k, v in c.items() if not k.startswith("_n_")}
        n = max(nsets.values()) if nsets else 1
        return {"launches": int(n), "mean_ns": e["ns"] / max(e["launches"], 1), "counters_sum": c, "derived": derive(c, n)}
    res = {"_what": "rocprofv3 --pmc passes over bench.py --single-stream (BASELINE configs[1], B = 256, fp32), summed over every dispatch of the "
                    "profiled steps; see scripts/sq_step_summary.py for the derivations",
           "classes": {k: pack(v) for k, v in sorted(per_class.items())},
           "kernels": [dict(name=k[0], grid_threads=k[1], cls=k[2], **pack(v)) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ns"])]}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from bench import build_fingerprint
        res["_meta"] = {"build_fingerprint": build_fingerprint(), "git_head": os.environ.get("AVC_GIT_HEAD", "unknown")}
    except Exception as e:  # pragma: no cover
        res["_meta"] = {"error": str(e)}
    json.dump(res, open(out_path, "w"), indent=1)
    for k, v in res["classes"].items():
        d = v["derived"]
        print(f"{k:12s} launches={v['launches']:5d} mfma_busy={d.get('mfma_busy', float('nan')):.3f}")


if __name__ == "__main__":
    main()
