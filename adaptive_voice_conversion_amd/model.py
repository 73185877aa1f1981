"""Drop-in ``AE`` module (reference surface: model.py:373-395) backed by libavc_hip.so.

What is kept identical to the reference: constructor signature ``AE(config)``,
sub-module / parameter names (the 166-key ``state_dict`` of SURVEY.md §8b),
registration order (so ``parameters()`` feeds torch optimizers in the same
order), default initialisation (same nn.Conv1d / nn.Linear constructors in the
same order => same tensors under the same ``torch.manual_seed``), and the
methods ``forward / inference / get_speaker_embeddings``.

What is different: no torch op computes anything.  All parameters alias ONE flat
fp32 buffer (and all gradients another) laid out as the C plan dictates, and
forward/backward are single calls into the gfx950 engine.
"""
import torch
import torch.nn as nn

from .engine import Plan, cfg_from_dict


def _bank_kernels(c):
    return list(range(c["bank_scale"], c["bank_size"] + 1, c["bank_scale"]))


class _ParamHolder(nn.Module):
    """A network of the reference reduced to its parameter containers."""

    def extra_repr(self):
        return "parameters only; compute runs in libavc_hip.so"


class SpeakerEncoder(_ParamHolder):
    """Parameter layout of model.py:209-235."""

    def __init__(self, c_in, c_h, c_out, kernel_size, bank_size, bank_scale, c_bank, n_conv_blocks, n_dense_blocks,
                 subsample, act, dropout_rate):
        super().__init__()
        ks = range(bank_scale, bank_size + 1, bank_scale)
        self.conv_bank = nn.ModuleList(nn.Conv1d(c_in, c_bank, kernel_size=k) for k in ks)
        self.in_conv_layer = nn.Conv1d(c_bank * (bank_size // bank_scale) + c_in, c_h, kernel_size=1)
        self.first_conv_layers = nn.ModuleList(nn.Conv1d(c_h, c_h, kernel_size=kernel_size) for _ in range(n_conv_blocks))
        self.second_conv_layers = nn.ModuleList(
            nn.Conv1d(c_h, c_h, kernel_size=kernel_size, stride=s) for s, _ in zip(subsample, range(n_conv_blocks)))
        self.first_dense_layers = nn.ModuleList(nn.Linear(c_h, c_h) for _ in range(n_dense_blocks))
        self.second_dense_layers = nn.ModuleList(nn.Linear(c_h, c_h) for _ in range(n_dense_blocks))
        self.output_layer = nn.Linear(c_h, c_out)


class ContentEncoder(_ParamHolder):
    """Parameter layout of model.py:279-299."""

    def __init__(self, c_in, c_h, c_out, kernel_size, bank_size, bank_scale, c_bank, n_conv_blocks, subsample, act,
                 dropout_rate):
        super().__init__()
        ks = range(bank_scale, bank_size + 1, bank_scale)
        self.conv_bank = nn.ModuleList(nn.Conv1d(c_in, c_bank, kernel_size=k) for k in ks)
        self.in_conv_layer = nn.Conv1d(c_bank * (bank_size // bank_scale) + c_in, c_h, kernel_size=1)
        self.first_conv_layers = nn.ModuleList(nn.Conv1d(c_h, c_h, kernel_size=kernel_size) for _ in range(n_conv_blocks))
        self.second_conv_layers = nn.ModuleList(
            nn.Conv1d(c_h, c_h, kernel_size=kernel_size, stride=s) for s, _ in zip(subsample, range(n_conv_blocks)))
        self.mean_layer = nn.Conv1d(c_h, c_out, kernel_size=1)
        self.std_layer = nn.Conv1d(c_h, c_out, kernel_size=1)


class Decoder(_ParamHolder):
    """Parameter layout of model.py:325-345 (sn=False only)."""

    def __init__(self, c_in, c_cond, c_h, c_out, kernel_size, n_conv_blocks, upsample, act, sn, dropout_rate):
        super().__init__()
        self.in_conv_layer = nn.Conv1d(c_in, c_h, kernel_size=1)
        self.first_conv_layers = nn.ModuleList(nn.Conv1d(c_h, c_h, kernel_size=kernel_size) for _ in range(n_conv_blocks))
        self.second_conv_layers = nn.ModuleList(
            nn.Conv1d(c_h, c_h * up, kernel_size=kernel_size) for _, up in zip(range(n_conv_blocks), upsample))
        self.conv_affine_layers = nn.ModuleList(nn.Linear(c_cond, c_h * 2) for _ in range(n_conv_blocks * 2))
        self.out_conv_layer = nn.Conv1d(c_h, c_out, kernel_size=1)


class _Token:
    """Liveness marker of one autograd forward: while it is alive and not yet consumed, the workspace it
    was run in holds activations a pending backward() needs."""
    __slots__ = ("done", "__weakref__")

    def __init__(self):
        self.done = False


class _Entry:
    """One cached (plan, workspace)."""
    __slots__ = ("plan", "ws", "pending")

    def __init__(self, plan, ws):
        self.plan, self.ws, self.pending = plan, ws, None

    def busy(self):
        t = self.pending() if self.pending is not None else None
        return t is not None and not t.done


class _PlanCache:
    """Bounded LRU of launch plans + workspaces keyed by (mode, B, T, T_cond, device).  An evicted plan is
    destroyed (avc_plan_destroy releases its helper streams / events) and its workspace goes back to
    torch's caching allocator -- real inference traffic sees a new (T, T') for almost every utterance."""

    def __init__(self, capacity):
        from collections import OrderedDict
        self.capacity = dict(capacity)
        self.d = {m: OrderedDict() for m in self.capacity}
        self.zombies = []   # evicted while a pending backward still needed them: closed as soon as that backward ran (or its graph died)

    def _retire(self, e):
        if e.busy():
            self.zombies.append(e)   # the plan (streams, events) and its workspace stay alive for the pending backward ...
        else:
            e.plan.close()
            e.ws = None

    def sweep(self):
        """... and are released here, on the next cache access after that backward consumed them."""
        live = []
        for e in self.zombies:
            if e.busy():
                live.append(e)
            else:
                e.plan.close()
                e.ws = None
        self.zombies = live

    def get(self, mode, key, factory):
        if self.zombies:
            self.sweep()
        lru = self.d[mode]
        hit = lru.get(key)
        if hit is None:
            hit = lru[key] = factory()
            while len(lru) > self.capacity[mode]:
                _, old = lru.popitem(last=False)
                self._retire(old)
        else:
            lru.move_to_end(key)
        return hit

    def clear(self):
        for lru in self.d.values():
            for e in lru.values():
                self._retire(e)
            lru.clear()
        self.sweep()

    def __len__(self):
        return sum(len(v) for v in self.d.values())


class _AEFunction(torch.autograd.Function):
    """Autograd seam for drop-in use with arbitrary torch losses/optimizers.

    The engine saves its activations in the plan's workspace.  Each autograd forward marks the workspace
    it ran in as pending until its backward() has consumed it; a second autograd forward of the same shape
    that arrives meanwhile (micro-batches, an extra evaluation pass with grad enabled) gets a private
    workspace instead of overwriting the saved activations.  no_grad / inference / embedding calls run in
    separate forward-only plans and never touch a training workspace."""

    @staticmethod
    def forward(ctx, ae, x, eps, *params):
        entry = ae._entry("train", x.shape[0], x.shape[2], x.shape[2], x.device)
        plan, ws = entry.plan, entry.ws
        if entry.busy():
            ws = torch.zeros(plan.workspace_floats, dtype=torch.float32, device=x.device)   # private to this forward
        token = _Token()
        if ws is entry.ws:
            import weakref
            entry.pending = weakref.ref(token)
        plan.forward(ae._flat, x, None, eps, ws)
        ctx.ae, ctx.plan, ctx.ws, ctx.token = ae, plan, ws, token
        ctx.save_for_backward(x, eps)
        muls, emb, dec = ae._outputs(plan, ws)
        C = muls.shape[1] // 2
        return muls[:, :C].clone(), muls[:, C:].clone(), emb.clone(), dec.clone()

    @staticmethod
    def backward(ctx, d_mu, d_ls, d_emb, d_dec):
        ae, plan, ws = ctx.ae, ctx.plan, ctx.ws
        if ctx.token.done:
            raise RuntimeError("AE: backward through the same forward twice (the engine keeps one set of saved activations)")
        if plan.h is None:
            raise RuntimeError("AE: the launch plan of this forward was released (model moved / cache cleared) before backward()")
        x, eps = ctx.saved_tensors
        B = x.shape[0]
        C, Tb = ae._c_lat, plan.latent_len
        d_muls = torch.zeros(B, 2 * C, Tb, device=x.device, dtype=torch.float32)
        if d_mu is not None:
            d_muls[:, :C] = d_mu
        if d_ls is not None:
            d_muls[:, C:] = d_ls
        d_dec = (torch.zeros(B, ae._n_mels, plan.out_len, device=x.device) if d_dec is None else d_dec).contiguous().float()
        d_emb = None if d_emb is None else d_emb.contiguous().float()
        g = torch.empty_like(ae._flat)
        plan.backward(ae._flat, x, None, eps, g, ws, d_dec=d_dec, d_muls=d_muls, d_emb=d_emb, lambda_kl=0.0)
        ctx.token.done = True
        grads = tuple(g[off:off + n].view(shape) for off, n, shape in ae._layout)
        return (None, None, None) + grads


class AE(nn.Module):
    """model.py:373-395."""

    def __init__(self, config, lib=None, compute_dtype=None, tuning=None):
        """``compute_dtype`` ("fp32" default | "bf16" (bf16 storage engine) | "bf16r" (fp32 storage, bf16 operands) | "fp32x3"; also read
        from ``config["compute_dtype"]``; engine.Plan documents the modes) is an
        extension over the reference: the precision of the conv / Linear matrix products (engine.Plan).
        ``tuning``: {avc_tuning field: value} captured by every plan of this module (A/B measurements, tests)."""
        super().__init__()
        self.config = config
        self._lib = lib
        self._tuning = dict(tuning or {})
        self.compute_dtype = compute_dtype or (config.get("compute_dtype") if isinstance(config, dict) else None) or "fp32"
        cfg_from_dict(config)  # validates (raises on sn / lrelu / dropout)
        self.speaker_encoder = SpeakerEncoder(**config["SpeakerEncoder"])
        self.content_encoder = ContentEncoder(**config["ContentEncoder"])
        self.decoder = Decoder(**config["Decoder"])
        self._n_mels = int(config["ContentEncoder"]["c_in"])
        self._c_lat = int(config["ContentEncoder"]["c_out"])
        self._c_emb = int(config["SpeakerEncoder"]["c_out"])
        # flat layout: state_dict order, every tensor 16-byte aligned (must equal the C plan's, checked in _plan)
        self._layout, off = [], 0
        for p in self.parameters():
            self._layout.append((off, p.numel(), tuple(p.shape)))
            off += (p.numel() + 3) // 4 * 4
        self._flat = torch.zeros(off, dtype=torch.float32)
        with torch.no_grad():
            for (o, n, _), p in zip(self._layout, self.parameters()):
                self._flat[o:o + n] = p.detach().reshape(-1)
        self._gflat = None
        self._bump = 0
        self._alias()
        # train: the regular batch + the short last batch of an epoch; inference / speaker: a few recent shapes
        self._plans = _PlanCache({"train": 2, "inference": 8, "speaker": 4})
        self._ragged = {}   # (lengths, device) -> (RaggedPlan, None), a few most recent
        self._ragged_ws = None   # the one workspace they share
        self.last_ragged_compute = None

    # ---- flat storage ------------------------------------------------------
    def _alias(self):
        self._bump = getattr(self, "_bump", 0) + 1   # the parameters may point at new storage: any "weights already packed" promise is void
        for (o, n, shape), p in zip(self._layout, self.parameters()):
            p.data = self._flat[o:o + n].view(shape)
            if self._gflat is not None:
                p.grad = self._gflat[o:o + n].view(shape)

    def _apply(self, fn, recurse=True):
        new = fn(self._flat)
        if new.dtype != torch.float32:
            raise NotImplementedError("the engine stores fp32 master parameters")
        self._flat = new.contiguous()
        if self._gflat is not None:
            self._gflat = fn(self._gflat).contiguous()
        self._alias()
        self._plans.clear()
        for plan, _ in getattr(self, "_ragged", {}).values():
            plan.close()
        self._ragged = {}
        self._ragged_ws = None
        return self

    def flat_parameters(self):
        return self._flat

    def weights_version(self):
        """Changes whenever PyTorch saw an in-place write to the parameters (flat buffer or any parameter view): what
        Solver.ae_step compares to decide whether the weight images it packed behind its last optimizer step are still current."""
        return (self._flat._version, self._bump, sum(p._version for p in self.parameters()))

    def weights_changed(self):
        """Tell the engine that the parameters were modified behind PyTorch's back (in-place through `.data`, a foreign kernel)."""
        self._bump += 1

    def flat_grads(self):
        if self._gflat is None or self._gflat.device != self._flat.device:
            self._gflat = torch.zeros_like(self._flat)
            self._alias()
        return self._gflat

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)  # copies into the aliased views
        self.weights_changed()   # (the version counters see the copies too; this does not depend on it)
        return out

    # ---- plans ---------------------------------------------------------------
    def _entry(self, mode, B, T, Tc, device):
        key = (int(B), int(T), int(Tc), str(device))

        def make():
            plan = Plan(self.config, B, T, Tc, lib=self._lib, compute_dtype=self._compute_for(mode), mode=mode, device=device, tuning=self._tuning)
            if [(o, n) for o, n, _ in plan.param_info] != [(o, n) for o, n, _ in self._layout]:
                raise RuntimeError("flat parameter layout of the C plan differs from the module's")
            return _Entry(plan, torch.zeros(plan.workspace_floats, dtype=torch.float32, device=device))
        return self._plans.get(mode, key, make)

    def _compute_for(self, mode):
        """ONE rounding model per compute mode and use: under ``compute_dtype: bf16`` the TRAINING plans run the bf16 pair-storage engine
        (BASELINE configs[2]); every forward-only plan -- ``AE.inference``, ``get_speaker_embeddings``, no-grad forwards -- rounds the
        operands of the matrix products to bf16 on fp32 storage ("bf16r"), which is what the ragged plan of ``inference_ragged`` /
        ``Inferencer.convert_batch`` runs and what the uniform plan falls back to at lengths outside the pair kernels anyway: the same
        utterance converts to the same numbers (to fp32 summation order) whichever entry point it comes through.
        ``inference_compute_dtype`` in the config overrides it (e.g. "bf16s": pair storage for uniform inference batches, faster, its own
        rounding points)."""
        cd = str(self.compute_dtype).lower()
        if mode != "train" and cd in ("bf16", "bfloat16"):
            return (self.config.get("inference_compute_dtype") if isinstance(self.config, dict) else None) or "bf16r"
        return self.compute_dtype

    def _plan(self, B, T, Tc, device, mode="train"):
        e = self._entry(mode, B, T, Tc, device)
        return e.plan, e.ws

    def set_plan_cache_size(self, train=None, inference=None, speaker=None):
        """How many (shape, device) launch plans + workspaces each mode keeps (LRU)."""
        for k, v in (("train", train), ("inference", inference), ("speaker", speaker)):
            if v is not None:
                self._plans.capacity[k] = max(1, int(v))

    def _outputs(self, plan, ws):
        B = plan.B
        muls = plan.view(ws, "muls", (B, 2 * self._c_lat, plan.latent_len))
        emb = plan.view(ws, "emb", (B, self._c_emb))
        dec = plan.view(ws, "dec", (B, self._n_mels, plan.out_len))
        return muls, emb, dec

    def _check_device(self, x):
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but parameters on {self._flat.device}")

    @staticmethod
    def _prep(x):
        if x.dtype != torch.float32:
            x = x.float()
        return x

    # ---- reference surface ---------------------------------------------------
    def forward(self, x, eps=None):
        """model.py:380-385; ``eps`` may be injected for parity tests."""
        x = self._prep(x)
        self._check_device(x)
        B, T = x.shape[0], x.shape[2]
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        mode = "train" if grad else "inference"
        if eps is None:
            Tb = self._plan(B, T, T, x.device, mode)[0].latent_len
            eps = torch.randn(B, self._c_lat, Tb, device=x.device, dtype=torch.float32)
        eps = eps.contiguous()
        if grad:
            return _AEFunction.apply(self, x, eps, *self.parameters())
        plan, ws = self._plan(B, T, T, x.device, "inference")   # forward-only workspace: never a training one
        plan.forward(self._flat, x, None, eps, ws)
        muls, emb, dec = self._outputs(plan, ws)
        C = self._c_lat
        return muls[:, :C].clone(), muls[:, C:].clone(), emb.clone(), dec.clone()

    def inference(self, x, x_cond):
        """model.py:387-391: decoder(mu(x), speaker(x_cond)); lengths may differ."""
        x, x_cond = self._prep(x), self._prep(x_cond)
        self._check_device(x)
        plan, ws = self._plan(x.shape[0], x.shape[2], x_cond.shape[2], x.device, "inference")
        plan.forward(self._flat, x, x_cond, None, ws)
        return self._outputs(plan, ws)[2].clone()

    def inference_ragged(self, xs, x_conds):
        """Batched ``inference`` over utterances of DIFFERENT lengths in ONE launch set (engine.RaggedPlan; the reference converts
        one utterance per call, inference.py:62-70).  xs / x_conds: lists of [T_b, M] / [T'_b, M] tensors (frames as rows -- what
        ``utt_make_frames`` views); returns the list of converted [M, T''_b] tensors, result b == inference(x_b, x_cond_b)."""
        from .engine import RaggedPlan
        dev = self._flat.device
        T, Tc = tuple(int(x.shape[0]) for x in xs), tuple(int(x.shape[0]) for x in x_conds)
        key = (T, Tc, str(dev))
        hit = self._ragged.get(key)
        if hit is None:
            # compute_dtype "bf16" -> "bf16r" here: the pair-STORAGE engine takes uniform shapes only; ragged plans round the operands of
            # the matrix products to bf16 on fp32 storage (engine.RaggedPlan).  The mode that ran is reported in `last_ragged_compute`.
            plan = RaggedPlan(self.config, T, Tc, lib=self._lib, compute_dtype="bf16r" if str(self.compute_dtype).lower().startswith(("bf16", "bfloat16")) else "fp32", device=dev,
                              tuning=self._tuning)
            if [(o, n) for o, n, _ in plan.param_info] != [(o, n) for o, n, _ in self._layout]:
                raise RuntimeError("flat parameter layout of the C plan differs from the module's")
            hit = self._ragged[key] = (plan, None)
            while len(self._ragged) > 4:
                old = self._ragged.pop(next(iter(self._ragged)))
                old[0].close()
        else:
            self._ragged[key] = self._ragged.pop(key)   # most recently used last
        plan = hit[0]
        # ONE pooled workspace for all ragged plans (real traffic almost never repeats a tuple of lengths: a fresh multi-hundred-MB
        # torch.zeros per call was most of the cold-path cost): every region a forward pass reads it has written before, in that pass
        ws = self._ragged_ws
        if ws is None or ws.device != dev or ws.numel() < plan.workspace_floats:
            ws = self._ragged_ws = torch.zeros(int(plan.workspace_floats * 1.25) + 1024, dtype=torch.float32, device=dev)
        self.last_ragged_compute = plan.compute_dtype
        x = torch.cat([self._prep(t).to(dev) for t in xs]).contiguous()
        xc = torch.cat([self._prep(t).to(dev) for t in x_conds]).contiguous()
        plan.forward(self._flat, x, xc, ws)
        return [o.clone() for o in plan.outputs(ws)]

    def get_speaker_embeddings(self, x):
        """model.py:393-395: only the speaker encoder runs (a speaker-only plan)."""
        x = self._prep(x)
        self._check_device(x)
        plan, ws = self._plan(x.shape[0], x.shape[2], x.shape[2], x.device, "speaker")
        plan.forward(self._flat, x, x, None, ws)
        return plan.view(ws, "emb", (x.shape[0], self._c_emb)).clone()
