"""ctypes binding of libavc_hip.so (the C ABI of include/avc_hip.h).

The product path has NO fallback: if the gfx950 extension is missing or fails
to load, importing code gets a RuntimeError telling how to build it.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libavc_hip.so")
_lib = None

MAX_BLOCKS = 8
PLAN_INFERENCE, PLAN_SPEAKER_ONLY = 1, 2
GRADS_ALL, GRADS_DECODER, GRADS_ENCODERS, GRADS_SPEAKER, GRADS_CONTENT = 0, 1, 2, 3, 4


class EncoderCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("c_in", "c_h", "c_out", "kernel_size", "bank_size", "bank_scale", "c_bank",
                                            "n_conv_blocks", "n_dense_blocks")] + [("subsample", ctypes.c_int * MAX_BLOCKS), ("act", ctypes.c_int)]


class DecoderCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("c_in", "c_cond", "c_h", "c_out", "kernel_size", "n_conv_blocks")] + \
               [("upsample", ctypes.c_int * MAX_BLOCKS), ("act", ctypes.c_int)]


class ModelCfg(ctypes.Structure):
    _fields_ = [("spk", EncoderCfg), ("enc", EncoderCfg), ("dec", DecoderCfg)]


class ReluSite(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("kind", "B", "C", "T")] + \
               [(n, ctypes.c_long) for n in ("act_off", "sb", "sc", "st", "y_off", "stat_off", "cond_off", "cond_sb", "storage")]


class Tuning(ctypes.Structure):
    """avc_tuning (include/avc_hip.h): launch heuristics / diagnostic switches a plan captures at creation."""
    _fields_ = [(n, ctypes.c_int) for n in ("struct_size", "single_stream", "dec_split_min", "conv_x3", "wgrad_x3", "dgrad_par", "bank_switch",
                                            "conv_ck5", "wgrad_batch", "wgrad_batch_wgs", "wgrad_target_wgs", "conv_ablation", "wgrad_ablation",
                                            "op_compute_dtype", "side_prio", "tile12_wgs")] + \
               [(n, ctypes.c_long) for n in ("wgrad_batch_units", "tile_thr11", "tile_thr21", "ck16_wgs", "ck32_wgs", "kg_wgs", "conv_min_lds", "in_pairs_nv", "bh_ck5", "wgrad_cw8", "dec_wgrad_flush", "dec_wgrad_wgs", "conv_walk", "conv_walk_min", "conv_in_fuse", "dbg_streams")]


PLAN_X3 = 4
PLAN_RAGGED = 8
PLAN_BF16S = 16
FWD_WEIGHTS_PACKED = 1   # avc_forward_ex: the caller packed the weight images behind its optimizer step (avc_plan_pack_weights)
ERR_PAIR_SHAPE = -12   # avc_plan_create*: the shape is outside the bf16 pair kernels (odd channel count / frames not a multiple of 4)
c_void_p, c_long, c_int, c_float = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float


def make_tuning(lib, overrides=None):
    """Library defaults (avc_tuning_init) with ``overrides`` ({field: value}) applied."""
    t = Tuning()
    lib.avc_tuning_init(ctypes.byref(t))
    for k, v in (overrides or {}).items():
        if not hasattr(t, k) or k == "struct_size":
            raise KeyError(f"unknown tuning field {k!r}")
        setattr(t, k, int(v))
    return t


def declare(lib):
    """Attach prototypes (so that 64-bit strides/pointers are marshalled correctly)."""
    lib.avc_version.restype = c_int
    lib.avc_last_error.restype = ctypes.c_char_p
    lib.avc_plan_create.argtypes = [ctypes.POINTER(ModelCfg), c_int, c_int, c_int, ctypes.POINTER(c_void_p)]
    lib.avc_plan_create_ex.argtypes = [ctypes.POINTER(ModelCfg), c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]
    lib.avc_plan_flags.argtypes = [c_void_p]
    if hasattr(lib, "avc_plan_side_priority"):   # (absent from pre-round-6 builds loaded through AVC_HIP_LIB for same-box A/B runs)
        lib.avc_plan_side_priority.argtypes = [c_void_p]
    lib.avc_plan_param_range.argtypes = [c_void_p, c_int, ctypes.POINTER(c_long), ctypes.POINTER(c_long)]
    lib.avc_plan_stream_wait_grads.argtypes = [c_void_p, c_int, c_void_p]
    lib.avc_set_tuning.argtypes = [ctypes.c_char_p, c_int]
    lib.avc_tuning_init.argtypes = [ctypes.POINTER(Tuning)]
    lib.avc_tuning_init.restype = None
    lib.avc_get_op_tuning.argtypes = [ctypes.POINTER(Tuning)]
    lib.avc_get_op_tuning.restype = None
    lib.avc_plan_create_tuned.argtypes = [ctypes.POINTER(ModelCfg), c_int, c_int, c_int, c_int, ctypes.POINTER(Tuning), ctypes.POINTER(c_void_p)]
    lib.avc_plan_set_single_stream.argtypes = [c_void_p, c_int]
    lib.avc_plan_create_ragged.argtypes = [ctypes.POINTER(ModelCfg), c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(Tuning),
                                           ctypes.POINTER(c_void_p)]
    lib.avc_plan_ragged_out.argtypes = [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_long)]
    lib.avc_forward_ragged.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_gather_segments.argtypes = [c_void_p, c_long, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.avc_plan_destroy.argtypes = [c_void_p]
    lib.avc_plan_destroy.restype = None
    lib.avc_plan_num_params.argtypes = [c_void_p]
    lib.avc_plan_param_floats.argtypes = [c_void_p]
    lib.avc_plan_param_floats.restype = c_long
    lib.avc_plan_param_info.argtypes = [c_void_p, c_int, ctypes.POINTER(c_long), ctypes.POINTER(c_long), ctypes.POINTER(c_int * 3)]
    lib.avc_plan_workspace_floats.argtypes = [c_void_p]
    lib.avc_plan_workspace_floats.restype = c_long
    lib.avc_plan_buffer.argtypes = [c_void_p, ctypes.c_char_p]
    lib.avc_plan_buffer.restype = c_long
    lib.avc_plan_num_relu_sites.argtypes = [c_void_p]
    lib.avc_plan_relu_site.argtypes = [c_void_p, c_int, ctypes.POINTER(ReluSite)]
    lib.avc_plan_set_compute_dtype.argtypes = [c_void_p, c_int]
    lib.avc_plan_compute_dtype.argtypes = [c_void_p]
    lib.avc_plan_out_len.argtypes = [c_void_p]
    lib.avc_plan_latent_len.argtypes = [c_void_p]
    lib.avc_forward.argtypes = [c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_void_p, c_long, c_long, c_int,
                                c_void_p, c_void_p, c_void_p]
    lib.avc_forward_ex.argtypes = [c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_void_p, c_long, c_long, c_int,
                                   c_void_p, c_void_p, c_int, c_void_p]
    lib.avc_plan_pack_weights.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_loss.argtypes = [c_void_p, c_void_p, c_long, c_long, c_int, c_float, c_void_p, c_void_p]
    lib.avc_backward.argtypes = [c_void_p, c_void_p, c_void_p, c_long, c_long, c_int, c_void_p, c_long, c_long, c_int,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]
    lib.avc_clip_adam_ws_floats.argtypes = [c_long]
    lib.avc_clip_adam_ws_floats.restype = c_long
    lib.avc_clip_adam_step.argtypes = [c_void_p] * 5 + [c_long, c_int] + [c_float] * 5 + [c_int, c_float, c_float, c_int,
                                                                                        c_void_p, c_void_p, c_void_p]
    # op-level entry points
    lib.avc_packed_weight_floats.argtypes = [c_int] * 4
    lib.avc_packed_weight_floats.restype = c_long
    lib.avc_packed_weight_floats_x3.argtypes = [c_int] * 4
    lib.avc_packed_weight_floats_x3.restype = c_long
    lib.avc_pack_weight_x3.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.avc_pack_weight.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.avc_conv1d_fwd.argtypes = [c_void_p, c_long, c_long, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                   c_int, c_int, c_void_p, c_long, c_long, c_int, c_int, c_void_p, c_int, c_long, c_long,
                                   c_int, c_int, c_void_p, c_int, c_void_p]
    lib.avc_conv1d_dgrad.argtypes = [c_void_p, c_long, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                     c_int, c_int, c_void_p, c_long, c_long, c_int, c_void_p, c_int, c_long, c_long, c_int,
                                     c_int, c_void_p, c_void_p, c_int, c_void_p]
    lib.avc_conv1d_wgrad_ws_floats.argtypes = [c_int] * 5
    lib.avc_conv1d_wgrad_ws_floats.restype = c_long
    lib.avc_conv1d_wgrad.argtypes = [c_void_p, c_long, c_long, c_int, c_void_p, c_long, c_long, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_instnorm_fwd.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_instnorm_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_int,
                                     c_int, c_void_p, c_void_p, c_long, c_int, c_void_p]
    lib.avc_conv1d_in_fwd.argtypes = [c_void_p, c_long, c_long, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_conv1d_dgrad_in_bwd.argtypes = [c_void_p, c_long, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                            c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int,
                                            c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p]
    # bf16 pair rows
    lib.avc_instnorm_fwd_pairs.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                           c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_instnorm_bwd_pairs.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_int, c_int, c_int,
                                           c_void_p, c_void_p, c_long, c_int, c_void_p]
    lib.avc_to_pairs.argtypes = [c_void_p, c_long, c_long, c_long, c_int, c_int, c_int, c_void_p, c_void_p]
    # mel <-> waveform DSP
    lib.avc_dsp_num_frames.argtypes = [c_long, c_int]
    lib.avc_dsp_basis_floats.argtypes = [c_int, c_int, c_int]
    lib.avc_dsp_basis_floats.restype = c_long
    lib.avc_dsp_basis_scratch_floats.argtypes = [c_int, c_int]
    lib.avc_dsp_basis_scratch_floats.restype = c_long
    lib.avc_dsp_make_basis.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.avc_dsp_stft.argtypes = [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_dsp_istft.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_dsp_griffin_lim_ws_floats.argtypes = [c_int, c_int, c_int, c_int]
    lib.avc_dsp_griffin_lim_ws_floats.restype = c_long
    lib.avc_dsp_griffin_lim.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_dsp_stft_batch.argtypes = [c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_dsp_istft_batch.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_dsp_griffin_lim_batch.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_dsp_griffin_lim_ragged.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_dsp_magnitude.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.avc_dsp_db_normalize.argtypes = [c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p]
    lib.avc_dsp_denormalize_amp.argtypes = [c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p]
    lib.avc_dsp_preemphasis.argtypes = [c_void_p, c_long, c_float, c_void_p, c_void_p]
    lib.avc_dsp_deemphasis.argtypes = [c_void_p, c_long, c_float, c_void_p, c_void_p]
    lib.avc_dsp_frame_power.argtypes = [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p]
    lib.avc_prof_end.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
    lib.avc_prof_class_name.argtypes = [c_int]
    lib.avc_prof_class_name.restype = ctypes.c_char_p
    return lib


def load():
    """The gfx950 library, or a loud failure (never a CPU fallback)."""
    global _lib, LIB_PATH
    if _lib is None:
        LIB_PATH = os.environ.get("AVC_HIP_LIB", LIB_PATH)   # A/B measurements of alternative builds of the same sources (scripts/)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or adaptive_voice_conversion_amd/csrc/build.sh)")
        # torch first: its wheel bundles its own libamdhip64 (soname .so.7, looked up as "libamdhip64.so" through torch/lib's RPATH); if this
        # library pulled /opt/rocm's copy in before torch, the process would hold TWO HIP runtimes and every launch on a torch stream would
        # fail with hipErrorNoDevice (100).  With torch's copy already mapped, the .so.7 dependency below resolves to it.
        import torch  # noqa: F401
        try:
            _lib = declare(ctypes.CDLL(LIB_PATH))
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    return _lib
