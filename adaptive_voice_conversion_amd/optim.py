"""Fused global-norm clip + Adam(amsgrad, coupled L2) over the flat buffers.

Replaces ``torch.nn.utils.clip_grad_norm_`` + ``torch.optim.Adam.step`` of
solver.py:75-77,:91-93 (≈1,700 tiny per-tensor ops in the stock loop, SURVEY
§2.3) with two kernels.  ``state_dict()`` / ``load_state_dict()`` use
torch.optim.Adam's on-disk format so ``<path>.opt`` files stay interchangeable
(solver.py:42,54).
"""
import contextlib
import ctypes

import torch

from . import _lib


class FusedClipAdam:
    def __init__(self, model, lr, betas, amsgrad, weight_decay, eps=1e-8, lib=None):
        self.model = model
        self.lib = lib if lib is not None else (model._lib if model._lib is not None else _lib.load())
        self.lr, self.betas, self.amsgrad, self.weight_decay, self.eps = float(lr), tuple(betas), bool(amsgrad), float(weight_decay), float(eps)
        self.step_count = 0
        self._init_state()

    def _init_state(self):
        flat = self.model.flat_parameters()
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.vmax = torch.zeros_like(flat)
        n = self.lib.avc_clip_adam_ws_floats(flat.numel())
        self.ws = torch.zeros(n + 64, device=flat.device, dtype=torch.float32)
        self.gnorm = torch.zeros(1, device=flat.device, dtype=torch.float32)

    def step(self, max_norm, grad_prescale=1.0, write_clipped=False):
        """clip_grad_norm_(max_norm) then Adam; returns the (device) total norm tensor."""
        flat, g = self.model.flat_parameters(), self.model.flat_grads()
        if self.m.device != flat.device:
            self._init_state()
        self.step_count += 1
        stream = ctypes.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream) if flat.is_cuda else None
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        with (torch.cuda.device(flat.device) if flat.is_cuda else contextlib.nullcontext()):   # launch on the buffers' device
            rc = self.lib.avc_clip_adam_step(P(flat), P(g), P(self.m), P(self.v), P(self.vmax), flat.numel(), self.step_count,
                                             self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                             int(self.amsgrad), float(max_norm), float(grad_prescale), int(write_clipped),
                                             P(self.ws), P(self.gnorm), stream)
        if rc != 0:
            raise RuntimeError(f"avc_clip_adam_step failed: {rc}")
        return self.gnorm

    def zero_grad(self):
        pass  # the engine overwrites every gradient each backward

    # ---- torch.optim.Adam-compatible serialisation -------------------------
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (o, n, shape) in enumerate(self.model._layout):
                st = {"step": torch.tensor(float(self.step_count)),
                      "exp_avg": self.m[o:o + n].view(shape).clone(),
                      "exp_avg_sq": self.v[o:o + n].view(shape).clone()}
                if self.amsgrad:
                    st["max_exp_avg_sq"] = self.vmax[o:o + n].view(shape).clone()
                state[i] = st
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": self.amsgrad, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(self.model._layout)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
        self.weight_decay, self.amsgrad = float(g["weight_decay"]), bool(g["amsgrad"])
        self.step_count = 0
        for i, (o, n, shape) in enumerate(self.model._layout):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.step_count = int(float(st["step"]))
            self.m[o:o + n] = st["exp_avg"].reshape(-1).to(self.m.device)
            self.v[o:o + n] = st["exp_avg_sq"].reshape(-1).to(self.m.device)
            if "max_exp_avg_sq" in st:
                self.vmax[o:o + n] = st["max_exp_avg_sq"].reshape(-1).to(self.m.device)
