"""Thin Python handle on the C-ABI launch plan (include/avc_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every kernel is
launched by libavc_hip.so.
"""
import contextlib
import ctypes

import torch

from . import _lib
from ._lib import DecoderCfg, EncoderCfg, ModelCfg


ACTS = {"relu": 0, "lrelu": 1}   # model.py:93-99 get_act: nn.ReLU / nn.LeakyReLU (slope 0.01)
_warned = set()


def _warn_once(key, msg):
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _act_code(name, c):
    """model.py:93-99 get_act: 'relu' -> ReLU, 'lrelu' -> LeakyReLU, ANYTHING ELSE -> ReLU as well (the reference's fallback branch).
    The same here, with a warning: a config with e.g. act: 'elu' trains as ReLU in the reference too."""
    act = c.get("act", "relu")
    if act not in ACTS:
        _warn_once(("act", name, str(act)), f"{name}.act={act!r}: the reference's get_act (model.py:93-99) maps every string other than "
                                            "'relu' / 'lrelu' to nn.ReLU(); so does this engine.")
        return ACTS["relu"]
    return ACTS[act]


def _check_common(name, c):
    if float(c.get("dropout_rate", 0)) != 0.0:
        raise NotImplementedError(f"{name}.dropout_rate={c['dropout_rate']}: only 0 (config.yaml default) is implemented")


def cfg_from_dict(config) -> ModelCfg:
    """config.yaml:1-36 -> avc_model_cfg.  Unsupported options fail loudly."""
    m = ModelCfg()
    for key, dst, dense in (("SpeakerEncoder", m.spk, True), ("ContentEncoder", m.enc, False)):
        c = config[key]
        _check_common(key, c)
        for f in ("c_in", "c_h", "c_out", "kernel_size", "bank_size", "bank_scale", "c_bank", "n_conv_blocks"):
            setattr(dst, f, int(c[f]))
        dst.n_dense_blocks = int(c["n_dense_blocks"]) if dense else 0
        dst.act = _act_code(key, c)
        if dst.n_conv_blocks > _lib.MAX_BLOCKS:
            raise NotImplementedError("more than 8 conv blocks")
        for i, s in enumerate(list(c["subsample"])[: dst.n_conv_blocks]):
            dst.subsample[i] = int(s)
    d = config["Decoder"]
    _check_common("Decoder", d)
    if d.get("sn", False):
        raise NotImplementedError("Decoder.sn=True (spectral norm) is not implemented; config.yaml default is False")
    for f in ("c_in", "c_cond", "c_h", "c_out", "kernel_size", "n_conv_blocks"):
        setattr(m.dec, f, int(d[f]))
    m.dec.act = _act_code("Decoder", d)
    for i, s in enumerate(list(d["upsample"])[: m.dec.n_conv_blocks]):
        m.dec.upsample[i] = int(s)
    return m


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def _on(t):
    """Make the tensor's device the current HIP device for the duration of a C-ABI call: the library
    launches on (and its plans create helper streams on) whatever device is current."""
    return torch.cuda.device(t.device) if (t is not None and t.is_cuda) else contextlib.nullcontext()


def unpack_pairs(p):
    """bf16 pair tensor (int32 [B, C/2, T]: channel 2p in the low half, 2p + 1 in the high half) -> fp32 [B, C, T]"""
    lo = (p & 0xFFFF).to(torch.int16).view(torch.bfloat16).to(torch.float32)
    hi = ((p >> 16) & 0xFFFF).to(torch.int16).view(torch.bfloat16).to(torch.float32)
    return torch.stack((lo, hi), dim=2).reshape(p.shape[0], 2 * p.shape[1], p.shape[2])


def unpack_planar(p):
    """natural bf16 rows (int32 [B, C, T/2]: frames 2i, 2i + 1 per dword) -> fp32 [B, C, T]"""
    lo = (p & 0xFFFF).to(torch.int16).view(torch.bfloat16).to(torch.float32)
    hi = ((p >> 16) & 0xFFFF).to(torch.int16).view(torch.bfloat16).to(torch.float32)
    return torch.stack((lo, hi), dim=3).reshape(p.shape[0], p.shape[1], 2 * p.shape[2])


class Plan:
    """One (B, T, T_cond) launch plan.  ``lib`` defaults to the gfx950 library;
    tests may inject the CPU lane-level simulation build instead."""

    COMPUTE = {"fp32": 0, "float32": 0, "f32": 0, "fp32x3": 0, "f32x3": 0, "bf16": 3, "bfloat16": 3, "bf16s": 3, "bf16_storage": 3,
               "bf16r": 1, "bf16_operands": 1}

    def __init__(self, config, B, T, T_cond=None, lib=None, compute_dtype="fp32", mode="train", device=None, tuning=None):
        """compute_dtype: "fp32" (default, the reference's precision);
        "bf16" = BASELINE config 3's precision on the bf16 STORAGE engine (AVC_PLAN_BF16S: activations and activation gradients are bf16
        channel-pair tensors in HBM and LDS, v_mfma_f32_32x32x16_bf16 products, fp32 accumulation / statistics / parameters / optimizer);
        shapes the pair kernels do not take (odd channel counts, frame counts that are not multiples of 4 at some level) fall back to
        "bf16r" -- ``plan.compute_dtype`` says which one the plan runs; "bf16s" = the storage engine or an error;
        "bf16r" = fp32 storage, conv / Linear operands rounded to bf16 as they enter the matrix core (round 2's bf16 mode);
        "fp32x3" = fp32-accurate products from three bf16 terms per operand on the bf16 matrix core for the big k = 5 convs and the
        whole-chunk weight gradients (opt-in; csrc/conv_x3.hip), exact fp32 everywhere else.
        mode: "train" (forward + loss + backward), "inference" (forward only: the workspace holds no gradient,
        slab or dy buffers) or "speaker" (only the speaker encoder runs, AE.get_speaker_embeddings).
        device: the plan's helper streams are created on it (default: the current device).
        tuning: {avc_tuning field: value} overrides of the launch heuristics / diagnostic switches the plan captures
        (A/B measurements and tests; include/avc_hip.h).  The library has no process-wide knobs."""
        self.lib = lib if lib is not None else _lib.load()
        self.cfg = cfg_from_dict(config)
        self.B, self.T, self.T_cond = int(B), int(T), int(T_cond or T)
        self.mode = mode
        flags = {"train": 0, "inference": _lib.PLAN_INFERENCE, "speaker": _lib.PLAN_INFERENCE | _lib.PLAN_SPEAKER_ONLY}[mode]
        h = ctypes.c_void_p()
        dev = torch.device(device) if device is not None else None
        key = str(compute_dtype).lower()
        if key not in self.COMPUTE:
            raise ValueError(f"compute_dtype must be one of {sorted(self.COMPUTE)}, got {compute_dtype!r}")
        x3 = key in ("fp32x3", "f32x3")
        if x3:
            flags |= _lib.PLAN_X3
        bh = self.COMPUTE[key] == 3
        strict = key in ("bf16s", "bf16_storage")
        self.tuning = dict(tuning or {})
        tun = _lib.make_tuning(self.lib, self.tuning)
        with (torch.cuda.device(dev) if (dev is not None and dev.type == "cuda") else contextlib.nullcontext()):
            rc = self.lib.avc_plan_create_tuned(ctypes.byref(self.cfg), self.B, self.T, self.T_cond, flags | (_lib.PLAN_BF16S if bh else 0),
                                                ctypes.byref(tun), ctypes.byref(h))
            if rc == _lib.ERR_PAIR_SHAPE and bh and not strict:
                # a shape outside the pair kernels (and nothing else: every other failure is reported): the operand-rounding bf16 mode takes
                # any shape.  The numerics (and the speed) of "bf16" then differ between shapes of the same model -- say so, once per shape class.
                _warn_once(("bf16r", self.T % 4, self.T_cond % 4),
                           f"compute_dtype 'bf16': B={self.B}, T={self.T}, T_cond={self.T_cond} is outside the bf16 storage engine "
                           f"({self.lib.avc_last_error().decode()}); this plan runs 'bf16r' (fp32 storage, operands rounded to bf16) instead. "
                           "Pass compute_dtype='bf16s' to make this an error.")
                bh = False
                rc = self.lib.avc_plan_create_tuned(ctypes.byref(self.cfg), self.B, self.T, self.T_cond, flags, ctypes.byref(tun), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(self.lib.avc_last_error().decode())
        self.h = h
        operand_bf16 = (self.COMPUTE[key] in (1, 3)) and not bh
        self.compute_dtype = "fp32x3" if x3 else ("bf16" if bh else ("bf16r" if operand_bf16 else "fp32"))
        self.pair_storage = bh
        if not bh and self.lib.avc_plan_set_compute_dtype(h, 1 if operand_bf16 else 0) != 0:
            raise RuntimeError(self.lib.avc_last_error().decode())
        self.num_params = self.lib.avc_plan_num_params(h)
        self.param_floats = self.lib.avc_plan_param_floats(h)
        self.workspace_floats = self.lib.avc_plan_workspace_floats(h)
        self.out_len = self.lib.avc_plan_out_len(h)
        self.latent_len = self.lib.avc_plan_latent_len(h)
        self.param_info = []
        for i in range(self.num_params):
            off, n, dims = ctypes.c_long(), ctypes.c_long(), (ctypes.c_int * 3)()
            self.lib.avc_plan_param_info(h, i, ctypes.byref(off), ctypes.byref(n), ctypes.byref(dims))
            self.param_info.append((off.value, n.value, tuple(d for d in dims if d > 0)))

    def close(self):
        """avc_plan_destroy: releases the plan's helper streams / events (the workspace is the caller's)."""
        if getattr(self, "h", None):
            self.lib.avc_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def param_range(self, part):
        """(offset, numel) of a part of the flat parameter / gradient buffer (_lib.GRADS_*)."""
        off, n = ctypes.c_long(), ctypes.c_long()
        if self.lib.avc_plan_param_range(self.h, int(part), ctypes.byref(off), ctypes.byref(n)) != 0:
            raise ValueError(part)
        return off.value, n.value

    def stream_wait_grads(self, part, stream):
        """Make ``stream`` (a torch.cuda.Stream) wait until that part of the gradients of the last
        ``backward`` call is final (data-parallel overlap).  Returns False when the plan has no helper
        streams / events (nothing was ordered: the caller must wait for the stream backward ran on)."""
        rc = self.lib.avc_plan_stream_wait_grads(self.h, int(part), ctypes.c_void_p(stream.cuda_stream))
        if rc == -9:
            return False
        self._chk(rc)
        return True

    def set_single_stream(self, on):
        """Profiling aid: every kernel of this plan on the caller's stream."""
        self._chk(self.lib.avc_plan_set_single_stream(self.h, int(bool(on))))

    def buffer(self, name):
        off = self.lib.avc_plan_buffer(self.h, name.encode())
        if off < 0:
            raise KeyError(name)
        return off

    def view(self, ws, name, shape):
        off = self.buffer(name)
        n = 1
        for s in shape:
            n *= s
        return ws[off:off + n].view(*shape)

    def relu_masks(self, ws):
        """0/1 masks of every ReLU of the last forward, in the reference's call order, computed
        from the engine's own saved tensors (diagnostics / branch-matched gradient checks)."""
        import ctypes as C
        from ._lib import ReluSite
        out = []
        for i in range(self.lib.avc_plan_num_relu_sites(self.h)):
            s = ReluSite()
            self.lib.avc_plan_relu_site(self.h, i, C.byref(s))
            B, Cc, T = s.B, s.C, s.T
            if s.kind == 0:
                if s.storage == 1:
                    act = unpack_pairs(torch.as_strided(ws.view(torch.int32), (B, Cc // 2, T), (s.sb, s.sc, s.st), s.act_off))
                else:
                    act = torch.as_strided(ws, (B, Cc, T), (s.sb, s.sc, s.st), s.act_off)
                m = act > 0
                if T == 1 and s.st == 0:
                    m = m.reshape(B, Cc)
            else:
                if s.storage == 1:
                    y = unpack_pairs(ws.view(torch.int32)[s.y_off:s.y_off + B * (Cc // 2) * T].view(B, Cc // 2, T))
                elif s.storage == 2:
                    y = unpack_planar(ws.view(torch.int32)[s.y_off:s.y_off + B * Cc * (T // 2)].view(B, Cc, T // 2))
                else:
                    y = ws[s.y_off:s.y_off + B * Cc * T].view(B, Cc, T)
                mean = ws[s.stat_off:s.stat_off + B * Cc].view(B, Cc, 1)
                rstd = ws[s.stat_off + B * Cc:s.stat_off + 2 * B * Cc].view(B, Cc, 1)
                xh = ((y - mean) * rstd).double()      # the same two fp32 roundings as the kernel
                if s.cond_off >= 0:
                    cond = torch.as_strided(ws, (B, 2 * Cc), (s.cond_sb, 1), s.cond_off).double()
                    w = xh * cond[:, Cc:, None] + cond[:, :Cc, None]   # exact in fp64 -> same sign as the fp32 fma
                else:
                    w = xh
                m = w > 0
            out.append(m)
        return out

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(f"libavc: {self.lib.avc_last_error().decode()}")

    def forward(self, params, x, x_cond, eps, ws, weights_packed=False):
        """weights_packed: the weight images in ``ws`` are current (``pack_weights(params, ws)`` ran after the last change of ``params``):
        the pass then opens with its first convolution instead of the pack launch."""
        xc = x if x_cond is None else x_cond
        with _on(ws):
            self._chk(self.lib.avc_forward_ex(self.h, _ptr(params), _ptr(x), x.stride(0), x.stride(1), x.stride(2), _ptr(xc),
                                              xc.stride(0), xc.stride(1), xc.stride(2), _ptr(eps), _ptr(ws),
                                              _lib.FWD_WEIGHTS_PACKED if weights_packed else 0, _stream(ws)))

    def pack_weights(self, params, ws):
        """Every weight tensor -> the plan's LDS-image order, ONE launch (a training loop calls this right behind its optimizer step)."""
        with _on(ws):
            self._chk(self.lib.avc_plan_pack_weights(self.h, _ptr(params), _ptr(ws), _stream(ws)))

    def loss(self, x, lambda_rec, ws):
        with _on(ws):
            self._chk(self.lib.avc_loss(self.h, _ptr(x), x.stride(0), x.stride(1), x.stride(2), float(lambda_rec), _ptr(ws), _stream(ws)))

    def backward(self, params, x, x_cond, eps, grads, ws, d_dec=None, d_muls=None, d_emb=None, lambda_kl=0.0):
        xc = x if x_cond is None else x_cond
        with _on(ws):
            self._chk(self.lib.avc_backward(self.h, _ptr(params), _ptr(x), x.stride(0), x.stride(1), x.stride(2), _ptr(xc),
                                            xc.stride(0), xc.stride(1), xc.stride(2), _ptr(eps), _ptr(d_dec), _ptr(d_muls),
                                            _ptr(d_emb), float(lambda_kl), _ptr(grads), _ptr(ws), _stream(ws)))


class RaggedPlan:
    """Launch plan of ONE forward-only pass over B (source, target) utterance pairs of DIFFERENT lengths
    (avc_plan_create_ragged): the batched form of ``Inferencer.inference_one_utterance`` (inference.py:54-70).  Nothing is
    padded; result b equals ``AE.inference(x_b, x_cond_b)`` (model.py:387-391)."""

    def __init__(self, config, T, T_cond=None, lib=None, compute_dtype="fp32", device=None, tuning=None):
        self.lib = lib if lib is not None else _lib.load()
        self.cfg = cfg_from_dict(config)
        self.T = [int(t) for t in T]
        self.T_cond = [int(t) for t in (T_cond if T_cond is not None else T)]
        if len(self.T) != len(self.T_cond) or not self.T:
            raise ValueError("T and T_cond must be equally long, non-empty lists")
        self.B = len(self.T)
        key = str(compute_dtype).lower()
        if key not in ("fp32", "float32", "f32", "bf16", "bfloat16", "bf16r", "bf16_operands"):
            raise ValueError("ragged plans compute in fp32 or bf16 (operand rounding: the pair-storage engine takes uniform shapes)")
        h = ctypes.c_void_p()
        tun = _lib.make_tuning(self.lib, tuning)
        arr = ctypes.c_int * self.B
        dev = torch.device(device) if device is not None else None
        with (torch.cuda.device(dev) if (dev is not None and dev.type == "cuda") else contextlib.nullcontext()):
            rc = self.lib.avc_plan_create_ragged(ctypes.byref(self.cfg), self.B, arr(*self.T), arr(*self.T_cond), ctypes.byref(tun), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(self.lib.avc_last_error().decode())
        self.h = h
        bf = key in ("bf16", "bfloat16", "bf16r", "bf16_operands")
        self.compute_dtype = "bf16r" if bf else "fp32"
        if self.lib.avc_plan_set_compute_dtype(h, 1 if bf else 0) != 0:
            raise RuntimeError(self.lib.avc_last_error().decode())
        self.param_floats = self.lib.avc_plan_param_floats(h)
        self.workspace_floats = self.lib.avc_plan_workspace_floats(h)
        self.num_params = self.lib.avc_plan_num_params(h)
        self.param_info = []
        for i in range(self.num_params):
            off, n, dims = ctypes.c_long(), ctypes.c_long(), (ctypes.c_int * 3)()
            self.lib.avc_plan_param_info(h, i, ctypes.byref(off), ctypes.byref(n), ctypes.byref(dims))
            self.param_info.append((off.value, n.value, tuple(d for d in dims if d > 0)))
        lens, offs = (ctypes.c_int * self.B)(), (ctypes.c_long * self.B)()
        if self.lib.avc_plan_ragged_out(h, lens, offs) != 0:
            raise RuntimeError(self.lib.avc_last_error().decode())
        self.out_len, self.out_off = list(lens), list(offs)
        self.n_mels = int(self.cfg.enc.c_in)

    close = Plan.close
    __del__ = Plan.__del__
    _chk = Plan._chk

    def forward(self, params, x, x_cond, ws):
        """x / x_cond: the utterances back to back as rows of frames, [sum T, M] contiguous fp32 (x_cond None = x)."""
        if x.dim() != 2 or x.shape[0] != sum(self.T) or x.shape[1] != self.n_mels or not x.is_contiguous():
            raise ValueError(f"x must be a contiguous [{sum(self.T)}, {self.n_mels}] tensor")
        if x_cond is not None and (x_cond.dim() != 2 or x_cond.shape[0] != sum(self.T_cond) or x_cond.shape[1] != self.n_mels or not x_cond.is_contiguous()):
            raise ValueError(f"x_cond must be a contiguous [{sum(self.T_cond)}, {self.n_mels}] tensor")
        with _on(ws):
            self._chk(self.lib.avc_forward_ragged(self.h, _ptr(params), _ptr(x), _ptr(x_cond), _ptr(ws), _stream(ws)))

    def outputs(self, ws):
        """list of [M, out_len[b]] views of the converted utterances in the workspace"""
        return [ws[o:o + self.n_mels * n].view(self.n_mels, n) for o, n in zip(self.out_off, self.out_len)]
