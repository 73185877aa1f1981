"""Training driver with the reference's surface (solver.py:16-118): ``Solver(config,
args)``, ``ae_step(data, lambda_kl)``, ``train(n)``, ``save_model`` /
``load_model`` with the same ``<path>.ckpt`` / ``<path>.opt`` files.

``ae_step`` is ONE engine pass: forward, L1+KL loss, backward, (RCCL all-reduce
of the flat gradient buffer when torch.distributed is initialised), fused
clip+Adam — no torch op on the hot path and no host sync unless the caller asks
for Python floats.
"""
import os

import torch
import yaml

from .data_utils import PickleDataset, get_data_loader
from .model import AE
from .optim import FusedClipAdam
from .utils import Logger, cc, infinite_iter, local_device


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class Solver(object):
    def __init__(self, config, args, lib=None):
        self.config = config
        self.args = args
        self._lib = lib
        self.logger = Logger(getattr(args, "logdir", "./log"))
        if getattr(args, "data_dir", None):
            self.get_data_loaders()
        self.build_model()
        if getattr(args, "store_model_path", None):
            self.save_config()
        if getattr(args, "load_model", False):
            self.load_model()

    # ---- checkpoints (solver.py:39-55) ---------------------------------------
    # The reference is single-process.  Under data parallelism the replicas are identical, so only rank 0
    # writes -- to a temporary file that is renamed into place (a reader never sees a truncated
    # checkpoint) -- and every rank waits for it.
    @staticmethod
    def _rank():
        d = _dist()
        return d.get_rank() if d is not None else 0

    @staticmethod
    def _atomic_save(obj, path):
        tmp = f"{path}.tmp.{os.getpid()}"
        torch.save(obj, tmp)
        os.replace(tmp, path)

    def save_model(self, iteration=None):
        if self._rank() == 0:
            self._atomic_save(self.model.state_dict(), f"{self.args.store_model_path}.ckpt")
            self._atomic_save(self.opt.state_dict(), f"{self.args.store_model_path}.opt")
        d = _dist()
        if d is not None and d.get_world_size() > 1:
            d.barrier()

    def save_config(self):
        if self._rank() != 0:
            return
        for suffix, obj in ((".config.yaml", self.config), (".args.yaml", vars(self.args))):
            path = f"{self.args.store_model_path}{suffix}"
            tmp = f"{path}.tmp.{os.getpid()}"
            with open(tmp, "w") as f:
                yaml.dump(obj, f)
            os.replace(tmp, path)

    def load_model(self):
        """solver.py:50-54 loads the model AND the optimizer state, unconditionally: a missing ``.opt`` raises, as it does there.
        ``args.load_opt = False`` opts out (a bare ``.ckpt`` such as the published vctk_model.ckpt: fresh optimizer state)."""
        dev = self.model.flat_parameters().device
        self.model.load_state_dict(torch.load(f"{self.args.load_model_path}.ckpt", map_location=dev))
        if getattr(self.args, "load_opt", True):
            self.opt.load_state_dict(torch.load(f"{self.args.load_model_path}.opt", map_location=dev))

    # ---- data (solver.py:57-68) ----------------------------------------------
    def get_data_loaders(self):
        d = self.args.data_dir
        if getattr(self.args, "device_feed", False):
            from .device_feed import DeviceSegmentFeed
            rank, world = 0, 1
            dd = _dist()
            if dd is not None:
                rank, world = dd.get_rank(), dd.get_world_size()
            self.train_iter = DeviceSegmentFeed.from_files(
                os.path.join(d, f"{self.args.train_set}.pkl"), os.path.join(d, self.args.train_index_file),
                self.config["data_loader"]["segment_size"], self.config["data_loader"]["batch_size"], local_device(),
                shuffle=self.config["data_loader"]["shuffle"], seed=0, rank=rank, world_size=world,   # disjoint shards of ONE permutation
                lib=self._lib)
            return
        self.train_dataset = PickleDataset(os.path.join(d, f"{self.args.train_set}.pkl"),
                                           os.path.join(d, self.args.train_index_file),
                                           segment_size=self.config["data_loader"]["segment_size"])
        self.train_loader = get_data_loader(self.train_dataset, frame_size=self.config["data_loader"]["frame_size"],
                                            batch_size=self.config["data_loader"]["batch_size"],
                                            shuffle=self.config["data_loader"]["shuffle"], num_workers=4, drop_last=False)
        self.train_iter = infinite_iter(self.train_loader)

    # ---- model + optimizer (solver.py:70-79) -----------------------------------
    def build_model(self):
        if self.config["data_loader"].get("frame_size", 1) != 1:
            raise NotImplementedError("data_loader.frame_size != 1")
        tuning = getattr(self.args, "tuning", None)
        self.model = cc(AE(self.config, lib=self._lib, tuning=tuning)) if self._lib is None else AE(self.config, lib=self._lib, tuning=tuning)
        o = self.config["optimizer"]
        self.opt = FusedClipAdam(self.model, lr=o["lr"], betas=(o["beta1"], o["beta2"]), amsgrad=o["amsgrad"],
                                 weight_decay=o["weight_decay"], lib=self._lib)
        d = _dist()
        if d is not None and d.get_world_size() > 1:  # identical replicas: rank 0's init wins
            d.broadcast(self.model.flat_parameters(), src=0)
            self.model.weights_changed()   # (a collective writes the flat buffer without touching the views' version counters)
        # reparameterisation noise (model.py:383): every rank draws from its OWN generator stream
        self._eps_gen = None
        self._comm_stream = None
        self._wire = None   # persistent bf16 gradient bucket (allreduce_dtype: bf16)
        self.measure_allreduce = False
        self._ar_events = []

    # ---- one training step (solver.py:81-97) -------------------------------------
    def _draw_eps(self, B, C, Tb, device):
        if self._eps_gen is None or self._eps_gen.device != device:
            self._eps_gen = torch.Generator(device=device)
            self._eps_gen.manual_seed(torch.initial_seed() + 7919 * self._rank())
        return torch.randn(B, C, Tb, device=device, dtype=torch.float32, generator=self._eps_gen)

    def _allreduce_grads(self, d, plan, grads):
        """SUM over ranks of the flat gradient buffer (RCCL over xGMI when the backend is "nccl"); the 1/W of the
        mean is folded into the optimizer kernel.  Three buckets in the order avc_backward finishes them: the decoder's range
        and the speaker encoder's range are reduced on a communication stream under the rest of the backward pass
        (avc_plan_stream_wait_grads), the content encoder's range once the whole backward is done."""
        from . import _lib
        # compute_dtype bf16 (BASELINE configs[2]: "bf16 compute, fp32 master and optimizer state"): the bucket travels as
        # bf16 too -- 9.8 MB instead of 19.6 MB per step over xGMI; `allreduce_dtype: fp32` in the config keeps
        # the wire fp32.  The sum of W bf16-rounded gradients carries a relative error of ~2^-9 per element, the same order as
        # the bf16 matrix products that produced them.
        wire_bf16 = self.config.get("allreduce_dtype", "bf16" if str(self.config.get("compute_dtype", "fp32")).lower() .startswith(("bf16", "bfloat16")) else "fp32") == "bf16"

        def reduce(seg, wire):
            if wire is None:
                d.all_reduce(seg)
                return
            wire.copy_(seg)            # persistent bf16 bucket: no per-step allocation
            d.all_reduce(wire)
            seg.copy_(wire)
        parts = [plan.param_range(k) for k in (_lib.GRADS_DECODER, _lib.GRADS_SPEAKER, _lib.GRADS_CONTENT)]
        # `allreduce_buckets` (config; default 3): 3 = decoder | speaker encoder | content encoder, the first two under the backward pass;
        # 2 = decoder under the backward pass, the two encoders (one contiguous range: the head of the flat buffer) after it -- one
        # collective launch fewer on a step this short; 1 = the whole buffer after the backward pass (no overlap)
        nb = int(self.config.get("allreduce_buckets", 3))
        early = [(_lib.GRADS_DECODER, parts[0]), (_lib.GRADS_SPEAKER, parts[1])]
        late = [parts[2]]
        if nb <= 2:
            lo = min(parts[1][0], parts[2][0])
            hi = max(parts[1][0] + parts[1][1], parts[2][0] + parts[2][1])
            if hi - lo != parts[1][1] + parts[2][1]:   # (the two encoders are not adjacent in this plan: keep them apart)
                late = [parts[1], parts[2]]
            else:
                late = [(lo, hi - lo)]
            early = early[:1]
        if nb <= 1:
            early, late = [], [(0, grads.numel())]
        wire = None
        if wire_bf16:
            if self._wire is None or self._wire.device != grads.device or self._wire.numel() != grads.numel():
                self._wire = torch.empty(grads.numel(), dtype=torch.bfloat16, device=grads.device)
            wire = self._wire
        w = (lambda o, n: wire[o:o + n]) if wire is not None else (lambda o, n: None)
        if not grads.is_cuda:
            for _, (o, n) in early:
                reduce(grads[o:o + n], w(o, n))
            for (o, n) in late:
                reduce(grads[o:o + n], w(o, n))
            return
        if self._comm_stream is None or self._comm_stream.device != grads.device:
            self._comm_stream = torch.cuda.Stream(device=grads.device)
        cs, main = self._comm_stream, torch.cuda.current_stream(grads.device)
        # Buckets in the order the backward pass finishes them (avc_backward: decoder -> speaker encoder's branch -> content encoder's
        # longer branch): the early ones are reduced on the communication stream UNDER the rest of the backward, only the late one(s)
        # start after it.
        for k, (o, n) in early:
            if not plan.stream_wait_grads(k, cs):
                cs.wait_stream(main)               # a plan without helper streams / events: order behind the whole backward
            with torch.cuda.stream(cs):
                reduce(grads[o:o + n], w(o, n))
        cs.wait_stream(main)                       # the whole backward (avc_backward joins its helper streams into main)
        with torch.cuda.stream(cs):
            for (o, n) in late:
                reduce(grads[o:o + n], w(o, n))
        # the time the compute stream WAITS here is the all-reduce the schedule could not hide: measured when asked for
        # (Solver.measure_allreduce = True; bench.py --gpus N reports the mean as config.exposed_allreduce_ms)
        if getattr(self, "measure_allreduce", False):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            main.wait_stream(cs)
            e1.record(main)
            self._ar_events.append((e0, e1))
        else:
            main.wait_stream(cs)

    def exposed_allreduce_ms(self, last=None):
        """Mean time per step the compute stream waited for the gradient all-reduce (needs measure_allreduce = True; synchronises)."""
        ev = self._ar_events[-last:] if last else self._ar_events
        if not ev:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    def ae_step(self, data, lambda_kl, eps=None, sync=True):
        model = self.model
        flat = model.flat_parameters()
        x = data if data.device == flat.device else data.to(flat.device, non_blocking=True)
        if x.dtype != torch.float32:
            x = x.float()
        B, _, T = x.shape
        plan, ws = model._plan(B, T, T, x.device)
        if eps is None:
            eps = self._draw_eps(B, model._c_lat, plan.latent_len, x.device)  # model.py:383
        grads = model.flat_grads()
        # the weight images in `ws` are current iff the last thing that touched the parameters was OUR optimizer step followed by
        # pack_weights into this very workspace: torch's version counters (of the flat buffer and of every parameter view) see every
        # other in-place change -- load_state_dict, a torch optimizer, user code.  (In-place edits through `.data` bypass the counters,
        # as they bypass autograd's own checks: call AE.weights_changed() after those.)
        # Fail-safe side: the promise is made ONLY here, behind our own optimizer step + pack_weights, and AE voids it on _alias /
        # _apply / load_state_dict / weights_changed(); `repack_every_step: true` in the config gives it up altogether, and
        # `verify_packed_weights: true` (debug: one host sync per step) compares a checksum of the parameters with the one taken when
        # the images were packed and raises on a silent edit.
        pk = getattr(self, "_packed", None)
        packed = (pk is not None and pk[0] is plan and pk[1] is ws and pk[2] is flat and pk[3] == model.weights_version()
                  and not self.config.get("repack_every_step", False))
        if packed and self.config.get("verify_packed_weights", False):
            if self._param_fingerprint(flat) != pk[4]:
                raise RuntimeError("the parameters changed behind the version counters since the weight images were packed "
                                   "(in-place edit through .data, a foreign kernel, a collective): call AE.weights_changed() after such writes")
        plan.forward(flat, x, None, eps, ws, weights_packed=packed)
        plan.loss(x, self.config["lambda"]["lambda_rec"], ws)
        plan.backward(flat, x, None, eps, grads, ws, lambda_kl=float(lambda_kl))
        prescale = 1.0
        d = _dist()
        if d is not None and (d.get_world_size() > 1 or self.config.get("allreduce_world1", False)):
            # (world_size 1: nothing to reduce -- and no bf16 round trip of the gradients; `allreduce_world1: true` keeps the
            # collective path for tests that exercise it with one rank)
            self._allreduce_grads(d, plan, grads)
            prescale = 1.0 / d.get_world_size()
        gnorm = self.opt.step(self.config["optimizer"]["grad_norm"], grad_prescale=prescale)
        plan.pack_weights(flat, ws)    # the next step opens with its first convolution (engine.Plan.forward, weights_packed)
        self._packed = (plan, ws, flat, model.weights_version(),
                        self._param_fingerprint(flat) if self.config.get("verify_packed_weights", False) else None)
        losses = plan.view(ws, "losses", (2,))
        if not sync:
            return {"loss_rec": losses[0], "loss_kl": losses[1], "grad_norm": gnorm[0]}
        vals = torch.cat([losses, gnorm]).tolist()  # one D2H sync (the reference does .item() x2, solver.py:94-95)
        return {"loss_rec": vals[0], "loss_kl": vals[1], "grad_norm": vals[2]}

    @staticmethod
    def _param_fingerprint(flat):
        """Position-dependent fingerprint of the flat parameter buffer (`verify_packed_weights`): the BITS of every element, multiplied by an odd
        per-position weight, summed modulo 2^64.  A plain sum misses sum-preserving edits -- two swapped weights, a permuted row."""
        bits = flat.view(torch.int32).to(torch.int64)
        pos = torch.arange(bits.numel(), device=bits.device, dtype=torch.int64)
        return int(((bits + 0x9E3779B9) * (2 * pos + 1)).sum().item())

    def kl_weight(self, iteration):
        """solver.py:101-104 linear annealing."""
        lam = self.config["lambda"]["lambda_kl"]
        if iteration >= self.config["annealing_iters"]:
            return lam
        return lam * (iteration + 1) / self.config["annealing_iters"]

    def train(self, n_iterations):
        for iteration in range(n_iterations):
            lambda_kl = self.kl_weight(iteration)
            data = next(self.train_iter)
            meta = self.ae_step(data, lambda_kl)
            if iteration % self.args.summary_steps == 0:
                self.logger.scalars_summary(f"{self.args.tag}/ae_train", meta, iteration)
            print(f"AE:[{iteration + 1}/{n_iterations}], loss_rec={meta['loss_rec']:.2f}, "
                  f"loss_kl={meta['loss_kl']:.2f}, lambda={lambda_kl:.1e}     ", end="\r")
            if (iteration + 1) % self.args.save_steps == 0 or iteration + 1 == n_iterations:
                self.save_model(iteration=iteration)
                print()
