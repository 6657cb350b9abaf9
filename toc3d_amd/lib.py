"""ctypes binding of ``libtoc3d_gfx950.so`` (the C ABI declared in ``include/toc3d.h``).

There is deliberately no fallback: if the shared library is missing the import of any compute entry
point raises, so a GPU box can never silently run a CPU / eager path.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("TOC3D_LIB", "libtoc3d_gfx950.so"))       # TOC3D_LIB: a development build beside the shipped one (library A/B runs)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "toc3d.h")

ABI_VERSION = 8                 # == TOC3D_ABI_VERSION of include/toc3d.h (tests/test_cpu_abi.py cross-checks)
F32, BF16, F32X3, F32X6, F32X3W, F32X3P, F32X3WO, F32X3WA = 0, 1, 2, 3, 4, 5, 6, 7          # F32X3: linear layers only -- f32 buffers, products as three bf16 MFMAs (include/toc3d.h)
EPI_BIAS, EPI_RESIDUAL, EPI_SWIGLU, EPI_GELU, EPI_SWIGLU_STATS, EPI_RESIDUAL_LN, EPI_RESIDUAL_STATS, EPI_SWIGLU_STATS_LN, EPI_CONV3X3, EPI_QKV_ROPE = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
# q scale of the pre-rotated attention path (toc3d_linear_qkv_rope -> toc3d_window_attention_rot, head_dim 64): head_dim^-0.5 (eva_vit.py:104-109) times log2(e) --
# that kernel's softmax is exp2-based (include/toc3d.h), so the conversion factor rides on the multiply the epilogue does anyway
ATTN_ROT_Q_SCALE = 64 ** -0.5 * 1.4426950408889634
NO_FUSED = (None, 0, None, 0, None, 0, 0.0, None, 0, None)     # the ten extra arguments of toc3d_linear_fused for epilogues 0-3
# GEMM tile variants (mod 100; + 100 / 200 / 300 select the XCD order at run time) the product library carries: every variant the autotuner may pick or a
# shipped table names (csrc/gemm_kernels.h launch_epi).
PRODUCT_VARIANTS = (1, 8, 9, 10, 13, 14, 15, 16, 17, 19, 22, 24, 26, 27, 28, 29, 30, 33, 45, 47, 49, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66)


def has_variant(v: int) -> bool:
    return (v % 100) in PRODUCT_VARIANTS

_P, _I64, _I, _F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float

# name -> argtypes, in header order.  'p' pointer, 'l' int64, 'i' int, 'f' float
_SIGS = {
    "toc3d_linear": "iiplplpplpllppllllp",
    "toc3d_linear_ex": "iiiplplpplpllppllllp",
    "toc3d_linear_fused": "iiiplplpplpllppllll" + "plplplf" + "pl" + "p" + "p",
    "toc3d_linear_fused_ws": "iiiplplpplpllppllll" + "plplplf" + "pl" + "p" + "pl" + "p",
    "toc3d_pack_swiglu_lnfold": "ippppppllpppllp",
    "toc3d_pack_weight_lnfold": "ippppllpllppp",
    "toc3d_conv3x3_nhwc": "iiplplpplllllpp",
    "toc3d_pack_weight": "ipllpllp",
    "toc3d_x3_planes": "plplllp",
    "toc3d_pack_swiglu": "ippppllppllp",
    "toc3d_im2col_patches": "ippllllllp",
    "toc3d_normalize_images": "plllppipllp",
    "toc3d_im2col_patches_u8": "iplllppipllllp",
    "toc3d_abs_pos_bicubic": "pllpllp",
    "toc3d_layernorm_rows": "iplppppfplllp",
    "toc3d_layernorm_act": "iplppfplllp",
    "toc3d_window_map_dense": "llllppppp",
    "toc3d_window_attention": "iplplppppppllllpplpfp",
    "toc3d_window_attention_pf": "iplplppppppllllpplpf" + "lppl" + "p",
    "toc3d_rank_desc": "pllpp",
    "toc3d_window_topk": "plllllppppppppppppp",
    "toc3d_linear_qkv_rope": "iiplplppllllpplfp",
    "toc3d_window_attention_rot": "iplplppppppllllp" + "lppl" + "p",
    "toc3d_gather_merge_ln": "iplppppllllppfpplp",
    "toc3d_gather_merge_ln_ex": "iplppppllllppfppllp",
    "toc3d_gather_merge_ln_split": "iplppppllllppfppll" + "pll" + "p",
    "toc3d_scatter_update": "plpplllpppppp",
    "toc3d_rebase_layernorm_rows": "iplpppllppppfpllp",
    "toc3d_pack_motion_weights": "p" * 24 + "p",
    "toc3d_motion_queries": "pllppppippllpp",
    "toc3d_collapse_query_scorer": "ppppplllfppp",
    "toc3d_score_tokens": "plpppplllpppp",
    "toc3d_gumbel_noise": "plLpp",
    "toc3d_gumbel_from_bits": "plpp",
    "toc3d_global_mean_half": "ipllllp",
    "toc3d_score_head": "ipllppplpppp",
    "toc3d_nhwc_to_nchw": "pplllp",
    "toc3d_im2col_3x3": "ipplllllp",
    "toc3d_head_frustum_inputs": "ippppllllllllplplpp",
    "toc3d_relu_inplace": "iplp",
    "toc3d_nchw_to_rows": "ippllllp",
    "toc3d_mln_apply": "ipppllpplp",
    "toc3d_se_gate": "ppplp",
    "toc3d_memory_pre_update": "pppppppppplllllip",
    "toc3d_memory_scores": "pllpp",
    "toc3d_memory_post_update": "ppppppppppppplpppllllllp",
    "toc3d_copy_bytes": "pplp",
    "toc3d_copy_segments": "lpppp",
    "toc3d_plan_create": "p",
    "toc3d_plan_destroy": "p",
    "toc3d_plan_begin": "p",
    "toc3d_plan_wait": "pll",
    "toc3d_plan_end": "pi",
    "toc3d_plan_run": "pp",
}
_CT = {"p": _P, "l": _I64, "i": _I, "f": _F, "L": ctypes.c_uint64}

_lib = None


def header_text() -> str:
    """include/toc3d.h without comments."""
    return re.sub(r"/\*.*?\*/", "", open(HEADER_PATH).read(), flags=re.S)


def header_functions():
    """Names of all functions declared in include/toc3d.h (used by the symbol-export test)."""
    return sorted(set(re.findall(r"\b(toc3d_\w+)\s*\(", header_text())))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C toc3d_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    try:
        lib.toc3d_abi_version.restype = _I
        abi = lib.toc3d_abi_version()
    except AttributeError:
        abi = None
    if abi != ABI_VERSION:
        # checked BEFORE any symbol is bound: an older build (TOC3D_LIB) fails here with a message, not with a bare
        # AttributeError on the first entry point it lacks
        raise RuntimeError(f"{LIB_PATH} reports ABI version {abi}, this package binds ABI {ABI_VERSION} (include/toc3d.h): "
                           "rebuild the library (`make -C toc3d_amd/csrc`)")
    lib.toc3d_last_error.restype = ctypes.c_char_p
    lib.toc3d_attn_rot_q_scale.restype = _F
    lib.toc3d_attn_rot_q_scale.argtypes = [_I64]
    lib.toc3d_motion_weights_floats.restype = _I64
    lib.toc3d_window_topk_rows.restype = _I64
    lib.toc3d_window_topk_rows.argtypes = [_I64] * 5
    lib.toc3d_linear_splitk_workspace_bytes.restype = _I64
    lib.toc3d_linear_splitk_workspace_bytes.argtypes = [_I, _I64, _I64]
    lib.toc3d_gather_merge_ln_scratch_bytes.restype = _I64
    lib.toc3d_gather_merge_ln_scratch_bytes.argtypes = [_I64, _I64]
    lib.toc3d_plan_lane_stream.restype = _P
    lib.toc3d_plan_lane_stream.argtypes = [_I64]
    lib.toc3d_plan_num_launches.restype = _I64
    lib.toc3d_plan_num_launches.argtypes = [_P]
    if hasattr(lib, "toc3d_linear_chain_info"):
        lib.toc3d_linear_chain_info.restype = _I        # number of ops (< 0: unknown config), not an error code
        lib.toc3d_linear_chain_info.argtypes = [_I, _P]
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = _I
        fn.argtypes = [_CT[c] for c in sig]
    _lib = lib
    return lib


def _conv(a):
    import torch
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


def call(name: str, *args):
    """Invoke a C-ABI entry point; tensors are passed as device pointers; raises on a non-zero return."""
    lib = load()
    rc = getattr(lib, name)(*[_conv(a) for a in args])
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.toc3d_last_error().decode()}")


def copy_segments(pairs, stream):
    """[(dst tensor, src tensor), ...] (same byte sizes) staged with one toc3d_copy_segments launch per 16 pairs."""
    for i in range(0, len(pairs), 16):
        chunk = pairs[i:i + 16]
        n = len(chunk)
        d = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in chunk])
        s = (ctypes.c_void_p * n)(*[t.data_ptr() for _, t in chunk])
        b = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t, _ in chunk])
        call("toc3d_copy_segments", n, d, s, b, stream)


def prefetch(tensors, workgroups, stream):
    """One toc3d_prefetch launch over up to 8 (contiguous) device tensors."""
    n = len(tensors)
    p = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    b = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t in tensors])
    call("toc3d_prefetch", n, p, b, workgroups, stream)


# Lane of the launch plan being recorded by THIS thread (toc3d_amd/plan.py); None = launch on torch's current stream.  Thread-local
# like the C side's recording flag (csrc/plan.cpp, toc3d_tls_recording): another Python thread that calls a toc3d op while this one
# records must get a real stream, not a lane handle the C side would not recognise for that thread.
_tls = threading.local()


def rec_lane():
    return getattr(_tls, "lane", None)


def set_rec_lane(lane):
    _tls.lane = lane


def stream_ptr():
    """The `stream` argument of a C-ABI call: torch's current HIP stream, or -- while a launch plan is being recorded by this
    thread -- the handle of the current lane (include/toc3d.h, toc3d_plan_lane_stream)."""
    lane = rec_lane()
    if lane is not None:
        return load().toc3d_plan_lane_stream(lane)
    import torch
    return torch.cuda.current_stream().cuda_stream


def recording() -> bool:
    return rec_lane() is not None
