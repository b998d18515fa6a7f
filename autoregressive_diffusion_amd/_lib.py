"""ctypes binding of liboniris_hip.so (the C-ABI declared in include/oniris.h).

The product path has NO fallback: if the shared library is missing or a symbol cannot be resolved, importing
this module raises.  Build it with `make -j8` at the repo root (or `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

import torch  # noqa: F401  -- MUST come first: binds liboniris_hip.so to the HIP/RCCL runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("ONIRIS_LIB_NAME", "liboniris_hip.so"))   # (diagnostic builds: `make stamp`)

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: build the HIP extension first (make -j8 at the repo root); "
                      "there is no CPU/PyTorch fallback for the denoiser kernels")

lib = C.CDLL(LIB_PATH)

c_void_p, c_int, c_float, c_int32, c_int64, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_int32, C.c_int64, C.c_size_t


class WeightDesc(C.Structure):
    _fields_ = [("w", c_void_p), ("grad", c_void_p), ("wf", c_void_p), ("wb", c_void_p), ("dwp", c_void_p), ("dws", c_void_p),
                ("cout", c_int32), ("cin", c_int32), ("taps", c_int32), ("kt", c_int32),
                ("CoutP", c_int32), ("CinP", c_int32), ("CoutPb", c_int32), ("CinPb", c_int32),
                ("row_start", c_int32), ("perm3", c_int32), ("gain", c_float), ("nsplit_cap", c_int32),
                ("nsplit", c_void_p), ("tile_start", c_int32), ("pad_", c_int32)]


class ConvArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("ctx", c_void_p), ("w_own", c_void_p), ("w_ctx", c_void_p), ("out", c_void_p),
                ("coef_own", c_void_p), ("coef_ctx", c_void_p),
                ("B", c_int32), ("S", c_int32), ("T", c_int32), ("H", c_int32), ("W", c_int32),
                ("Cin", c_int32), ("CinP", c_int32), ("Cout", c_int32), ("CoutP", c_int32), ("taps", c_int32),
                ("ctx_bstride", c_int32), ("ctx_T", c_int32), ("coff0", c_int32), ("coff1", c_int32),
                ("ctx_fill", c_float), ("epi", c_int32),
                ("res", c_void_p), ("escale", c_void_p), ("emb_gain", c_void_p), ("out2", c_void_p),
                ("ta", c_float), ("tb", c_float), ("clip", c_float), ("ctx_out", c_void_p),
                ("big_tile", c_int32), ("escale_pitch", c_int32),
                ("splitk_ws", c_void_p), ("splitk_ws_bytes", C.c_size_t), ("clip_flag", c_void_p),
                ("ctx_prod", c_void_p), ("ctx_prod_mode", c_int32), ("x_split", c_int32), ("x2", c_void_p), ("act_out", c_void_p),
                ("cat_w1", c_float), ("cat_w2", c_float)]


class WgradArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("dwp", c_void_p), ("scale", c_void_p),
                ("B", c_int32), ("T", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("CinP", c_int32),
                ("Cout", c_int32), ("CoutP", c_int32), ("taps", c_int32),
                ("xb_stride", c_int32), ("x_T", c_int32), ("coff", c_int32), ("fill", c_float),
                ("nsplit_cap", c_int32), ("taps_total", c_int32), ("tap0", c_int32), ("pad_", c_int32),
                ("nsplit_out", c_void_p)]


class AttnArgs(C.Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("qt", c_void_p), ("kt", c_void_p),
                ("vt", c_void_p), ("out", c_void_p), ("lse", c_void_p),
                ("kv_num", c_void_p), ("kv_idx", c_void_p), ("q_num", c_void_p), ("q_idx", c_void_p),
                ("tab_cols", c_int32), ("qtab_cols", c_int32),
                ("B", c_int32), ("heads", c_int32), ("Lq", c_int32), ("Lk", c_int32), ("C", c_int32),
                ("mask_mode", c_int32), ("P", c_int32), ("T", c_int32), ("tab_block", c_int32),
                ("dout", c_void_p), ("doutt", c_void_p), ("delta", c_void_p),
                ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p),
                ("dkv_part", c_void_p), ("dkv_chunks", c_int32), ("dkv_item_keys", c_int32),
                ("sched", c_void_p), ("sched_wgs", c_int32), ("sched_slots", c_int32), ("v_bstride", c_int64),
                ("k_bstride", c_int64), ("split_ws", c_void_p), ("kv_splits", c_int32), ("frame_kernel", c_int32)]


class AttnF32Args(C.Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("out", c_void_p), ("lse", c_void_p),
                ("dout", c_void_p), ("delta", c_void_p), ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p),
                ("BH", c_int32), ("Lq", c_int32), ("Lk", c_int32), ("D", c_int32), ("mask_mode", c_int32), ("P", c_int32),
                ("T", c_int32), ("q_frame_off", c_int32), ("scale", c_float), ("pad_", c_int32)]


EPI_NONE, EPI_EMB_SILU, EPI_MPSUM = 0, 1, 2

# every symbol include/oniris.h declares (tests/test_abi.py checks the list against the header)
_SIGS = {
    "oniris_last_error": (C.c_char_p, []),
    "oniris_abi_version": (c_int, []),
    "oniris_struct_sizes": (c_int, [c_void_p]),
    "oniris_profile_arm": (c_int, [c_void_p, c_void_p]),
    "oniris_profile_disarm": (c_int, []),
    "oniris_set_cu_reserve": (c_int, [c_int]),
    "oniris_set_ew_nt_bytes": (c_int64, [c_int64]),
    "oniris_frame_attn_bwd": (c_int, [c_void_p, c_void_p]),
    "oniris_frame_attn_qkv_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "oniris_frame_attn_qkv_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "oniris_conv_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oniris_wgrad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oniris_attn_f32_fwd": (c_int, [c_void_p, c_void_p]),
    "oniris_attn_f32_bwd": (c_int, [c_void_p, c_void_p]),
    "oniris_census": (c_int, [c_int]),
    "oniris_census_read": (c_int64, [C.c_char_p, c_int64]),
    "oniris_train_mask": (c_int, [c_int, c_int, c_void_p, c_void_p, C.POINTER(c_int)]),
    "oniris_infer_mask": (c_int, [c_int, c_int, c_void_p, c_void_p, C.POINTER(c_int)]),
    "oniris_mask_transpose": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "oniris_attn_schedule": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_int]),
    "oniris_weight_prep": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "oniris_weight_bwd": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "oniris_adamw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_float,
                             c_float, c_int, c_float, c_void_p]),
    "oniris_adamw_clip_ema": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float,
                                      c_float, c_float, c_int, c_float, c_void_p, c_float, c_void_p, c_float, c_void_p,
                                      c_float, c_void_p]),
    "oniris_sqnorm": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "oniris_dart_input": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_float, c_void_p, c_int, c_void_p]),
    "oniris_dart_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_float, c_void_p]),
    "oniris_dart_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "oniris_loss_tail": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "oniris_qkv_norm_hd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                   c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oniris_qkv_norm_hd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "oniris_rope_hd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_void_p]),
    "oniris_conv_fwd": (c_int, [C.POINTER(ConvArgs), c_void_p]),
    "oniris_conv_wgrad": (c_int, [C.POINTER(WgradArgs), c_void_p]),
    "oniris_conv_wgrad_group": (c_int, [C.POINTER(WgradArgs), c_int, c_void_p]),
    "oniris_gconv_bwd_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_int, c_int, c_int64, c_void_p]),
    "oniris_gconv_bwd_fused": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_float, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "oniris_act_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float,
                               c_int, c_int, c_int, c_int, c_void_p]),
    "oniris_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                               c_float, c_float, c_int, c_float, c_void_p]),
    "oniris_mpsum_mask": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    "oniris_emb_silu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p]),
    "oniris_mpsum_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_void_p]),
    "oniris_resample": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "oniris_resample_filter": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float,
                                       c_void_p]),
    "oniris_qkv_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "oniris_qkv_norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "oniris_qkv_norm_rope": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                     c_int, c_void_p]),
    "oniris_qkv_norm_rope_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_int, c_int, c_int, c_void_p]),
    "oniris_qkv_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                c_int, c_int, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "oniris_precond_out": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                   c_void_p]),
    "oniris_sampler_update": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float,
                                      c_void_p, c_int, c_float, c_void_p]),
    "oniris_embed_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_int, c_void_p]),
    "oniris_gates": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "oniris_gates_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "oniris_emb_scale": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "oniris_emb_scale_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "oniris_embed_pre": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p]),
    "oniris_embed_post": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p]),
    "oniris_embed_post_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p]),
    "oniris_rope": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                            c_int, c_int, c_int, c_int64, c_int64, c_void_p]),
    "oniris_qkv_norm_rope_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int64, c_int, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "oniris_attn_fwd": (c_int, [C.POINTER(AttnArgs), c_void_p]),
    "oniris_attn_bwd_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_void_p]),
    "oniris_attn_bwd_dq": (c_int, [C.POINTER(AttnArgs), c_void_p]),
    "oniris_attn_bwd_dkv": (c_int, [C.POINTER(AttnArgs), c_void_p]),
}
EXPORTED = sorted(_SIGS)

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here == missing symbol == loud failure
    _fn.restype = _res
    _fn.argtypes = _args


if os.environ.get("ONIRIS_HOST_TIMING") == "2":          # diagnostic: wall time spent inside every C-ABI call
    import time as _time
    call_stats = {}

    def _timed(name, fn):
        def w(*a):
            t0 = _time.perf_counter()
            r = fn(*a)
            st = call_stats.setdefault(name, [0, 0.0])
            st[0] += 1; st[1] += _time.perf_counter() - t0
            return r
        return w

    class _TimedLib:
        pass
    _tl = _TimedLib()
    for _name in _SIGS:
        setattr(_tl, _name, _timed(_name, getattr(lib, _name)))
    lib = _tl

_sz = (c_int32 * 4)()
lib.oniris_struct_sizes(_sz)
if list(_sz) != [C.sizeof(WeightDesc), C.sizeof(ConvArgs), C.sizeof(WgradArgs), C.sizeof(AttnArgs)]:
    raise ImportError(f"struct layout mismatch between _lib.py and liboniris_hip.so: {list(_sz)}")


class OnirisError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = lib.oniris_last_error()
        raise OnirisError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
