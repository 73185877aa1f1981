"""Worker of tests/test_dist_nccl.py: one rank of a torch.distributed job that runs Solver.ae_step through the
all-reduce branch (RCCL when the backend is nccl) and saves what it ended up with."""
import os
import sys
import types

import torch


def main():
    backend, out_dir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local = local % torch.cuda.device_count()   # (gloo on a 1-GPU box: every rank drives cuda:0)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from adaptive_voice_conversion_amd.solver import Solver
    from oracle import avc_oracle as O
    cfg = O.stock_config(80)
    cfg["allreduce_world1"] = True     # a 1-rank group still walks the collective branch (it is what this worker tests)
    sd = O.make_state_dict(cfg, 4)
    B = 4 * world
    x, eps = O.make_inputs(cfg, B, 128, 4)
    per = B // world
    args = types.SimpleNamespace(store_model_path=os.path.join(out_dir, "ckpt"), load_model=False, data_dir=None, logdir=out_dir)
    s = Solver(cfg, args)
    s.model.load_state_dict(sd)
    sl = slice(rank * per, (rank + 1) * per)
    metas, grads1 = [], None
    for it in range(steps):
        metas.append(s.ae_step(x[sl].contiguous().to(dev), 1.0, eps=eps[sl].contiguous().to(dev)))
        if it == 0:
            grads1 = s.model.flat_grads().detach().cpu().clone()   # the all-reduced SUM of the ranks' gradients of step 1 (the 1 / W lives in the optimizer)
    e1 = s._draw_eps(2, 4, 4, dev).cpu()       # per-rank noise stream
    s.save_model()                              # rank 0 writes, everyone waits
    torch.save({"params": s.model.flat_parameters().cpu(), "metas": metas, "eps_draw": e1, "grads_step1": grads1, "world": world,
                "comm_stream": s._comm_stream is not None}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
