"""CPU: the C-ABI library loads without a GPU, exports every symbol include/toc3d.h declares, the ctypes
signatures agree with the header, and the host-side modules mirror the reference interface."""
import ctypes
import json
import os
import re

import pytest
import torch

import toc3d_amd
from toc3d_amd import configs, lib


def test_library_exports_every_declared_symbol():
    assert os.path.exists(lib.LIB_PATH), "build first: make -C toc3d_amd/csrc"
    dll = ctypes.CDLL(lib.LIB_PATH)
    names = lib.header_functions()
    assert len(names) >= 50
    for n in names:
        assert hasattr(dll, n), f"{n} declared in include/toc3d.h but not exported"
    l = lib.load()
    hdr = int(re.search(r"#define\s+TOC3D_ABI_VERSION\s+(\d+)", open(lib.HEADER_PATH).read()).group(1))
    assert l.toc3d_abi_version() == hdr == lib.ABI_VERSION
    assert l.toc3d_motion_weights_floats() > 500000


def test_ctypes_signatures_match_header():
    txt = lib.header_text()
    sigs = dict(lib._SIGS)
    seen = 0
    for m in re.finditer(r"\bint\s+(toc3d_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        name, args = m.group(1), m.group(2)
        if name == "toc3d_abi_version":
            continue
        if name not in sigs:
            continue                      # helpers with other return types are bound by hand in lib.load()
        sig = ""
        for a in (x.strip() for x in args.split(",")):
            if "*" in a or "toc3d_stream_t" in a or "toc3d_plan_t" in a:
                sig += "p"
            elif a.startswith("int64_t"):
                sig += "l"
            elif a.startswith("uint64_t"):
                sig += "L"
            elif a.startswith("int "):
                sig += "i"
            elif a.startswith("float "):
                sig += "f"
            else:
                raise AssertionError(f"unparsed parameter {a!r} in {name}")
        assert sigs[name] == sig, name
        seen += 1
    assert seen == len(sigs)
    assert lib.load().toc3d_window_topk_rows(6, 20, 50, 16, 128) == 6 * (3 * 129 + 33 + 3 * 65 + 9)


def test_argument_validation_reports_errors_without_gpu():
    l = lib.load()
    rc = l.toc3d_linear(lib.BF16, 0, None, 0, None, 0, None, None, 0, None, 0, 0, None, None, 4, 4, 64, 0, None)
    assert rc == -1 and b"null buffer" in l.toc3d_last_error()
    with pytest.raises(RuntimeError, match="toc3d_rank_desc failed"):
        lib.call("toc3d_rank_desc", None, 1, 10, None, None)


def test_module_surface_matches_reference(golden_dir):
    spec = json.load(open(os.path.join(golden_dir, "state_dict_spec.json")))
    for name in ("toc3d_tiny", "eva_tiny"):
        m = toc3d_amd.build_backbone(configs.get(name))
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == spec[name]
    m = toc3d_amd.build_backbone(configs.get("toc3d_tiny"))
    assert m.pruning_loc == [3, 6, 9] and m.pruning_num_queries == 64          # read by Petr3D (petr3d.py:73-74,123)
    assert set(toc3d_amd.BACKBONES.module_dict) >= {"ToC3DEVAViT", "EVA_ViT"} and "CPFPN" in toc3d_amd.NECKS.module_dict
    with pytest.raises(AssertionError):                                            # toc3d_eva_vit.py:141-142
        toc3d_amd.build_backbone(dict(configs.get("toc3d_tiny"), pruning_loc=[2, 6, 9]))
    with pytest.raises(NotImplementedError):
        toc3d_amd.build_backbone(dict(configs.get("toc3d_tiny"), use_rel_pos=True))
    n = toc3d_amd.build_neck(configs.CPFPN_TINY)
    assert list(n.state_dict()) == ["lateral_convs.0.conv.weight", "lateral_convs.0.conv.bias", "fpn_convs.0.conv.weight", "fpn_convs.0.conv.bias"]


def test_no_cpu_fallback():
    m = toc3d_amd.build_backbone(configs.get("eva_tiny"))
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(2, 3, 320, 800))
    n = toc3d_amd.build_neck(configs.CPFPN_TINY)
    with pytest.raises(RuntimeError, match="no CPU"):
        n([torch.zeros(2, 128, 20, 50)])


def test_launch_plan_api_validates_without_gpu():
    """toc3d_plan_*: handles, lane handles and the recording state machine work (and fail loudly) without a GPU."""
    l = lib.load()
    h = ctypes.c_void_p()
    assert l.toc3d_plan_create(ctypes.addressof(h)) == 0 and h.value
    assert l.toc3d_plan_lane_stream(0) and l.toc3d_plan_lane_stream(3) and not l.toc3d_plan_lane_stream(64)
    assert l.toc3d_plan_num_launches(h.value) == 0
    assert l.toc3d_plan_wait(h.value, 1, 0) == -1 and b"not recording" in l.toc3d_last_error()
    assert l.toc3d_plan_run(h.value, None) == -1 and b"not finalized" in l.toc3d_last_error()
    assert l.toc3d_plan_begin(h.value) == 0
    h2 = ctypes.c_void_p()
    assert l.toc3d_plan_create(ctypes.addressof(h2)) == 0
    assert l.toc3d_plan_begin(h2.value) == -1 and b"already recording" in l.toc3d_last_error()
    assert l.toc3d_plan_wait(h.value, 1, 0) == 0 and l.toc3d_plan_wait(h.value, 1, 99) == -1
    assert l.toc3d_plan_end(h.value, 0) == -1 and b"nothing was recorded" in l.toc3d_last_error()
    assert l.toc3d_plan_destroy(h.value) == 0 and l.toc3d_plan_destroy(h2.value) == 0
    assert l.toc3d_plan_begin(None) == -1


def test_return_type_binds_to_the_plugin_class_when_the_plugin_is_imported():
    """petr3d.py:159 does isinstance(out, ToC3DViTReturnType) with the plugin's class (petr3d.py:17): once that module is in
    sys.modules the backbone returns the plugin's class, no manual re-binding."""
    import sys
    import types
    from toc3d_amd import backbone as bb
    name = "projects.mmdet3d_plugin.models.backbones.toc3d_utils"
    assert name not in sys.modules and bb._return_type() is bb.ToC3DViTReturnType

    class PluginReturnType:                                    # stands in for toc3d_utils.py:10-25 (same constructor signature)
        def __init__(self, img_feats=None, token_masks=None, attn_scores=None, keep_idx=None, drop_idx=None, aux_outputs=None):
            self.img_feats, self.token_masks, self.keep_idx, self.drop_idx = img_feats, token_masks, keep_idx, drop_idx
    mod = types.ModuleType(name)
    mod.ToC3DViTReturnType = PluginReturnType
    sys.modules[name] = mod
    try:
        assert bb._return_type() is PluginReturnType
        out = bb._return_type()({"last_feat": 1}, None, None, keep_idx=None, drop_idx=None, aux_outputs=None)
        assert isinstance(out, PluginReturnType) and out.img_feats == {"last_feat": 1}
    finally:
        del sys.modules[name]


def test_registration_on_an_mmcv_style_registry():
    """The HAVE_MMDET branch of toc3d_amd/registry.py: with mmdet.models.builder importable, the classes are registered on ITS
    registries under the reference's type names (force=True shadows the plugin's own registration) and built through them."""
    import importlib
    import sys
    import types

    class Registry:                                            # the slice of mmcv.utils.Registry the plugin relies on
        def __init__(self, name):
            self.name, self._module_dict = name, {}

        @property
        def module_dict(self):
            return self._module_dict

        def register_module(self, name=None, force=False, module=None):
            def _register(cls):
                key = name or cls.__name__
                if not force and key in self._module_dict:
                    raise KeyError(f"{key} is already registered in {self.name}")
                self._module_dict[key] = cls
                return cls
            return _register(module) if module is not None else _register

        def get(self, key):
            return self._module_dict.get(key)

        def build(self, cfg):
            cfg = dict(cfg)
            return self._module_dict[cfg.pop("type")](**cfg)

    class RefBackbone:                                         # what the plugin registered first
        pass
    fake = {n: types.ModuleType(n) for n in ("mmdet", "mmdet.models", "mmdet.models.builder")}
    fake["mmdet.models.builder"].BACKBONES, fake["mmdet.models.builder"].NECKS = Registry("backbone"), Registry("neck")
    fake["mmdet.models.builder"].BACKBONES.register_module(name="ToC3DEVAViT", module=RefBackbone)
    saved = {n: sys.modules.get(n) for n in fake}
    sys.modules.update(fake)
    from toc3d_amd import registry
    try:
        reg = importlib.reload(registry)
        assert reg.HAVE_MMDET and reg.BACKBONES is fake["mmdet.models.builder"].BACKBONES
        reg.register_all()
        assert reg.BACKBONES.get("ToC3DEVAViT") is toc3d_amd.ToC3DEVAViT and reg.BACKBONES.get("EVA_ViT") is toc3d_amd.EVA_ViT
        assert reg.NECKS.get("CPFPN") is toc3d_amd.CPFPN
        m = reg.build_backbone(configs.get("toc3d_tiny"), precision="fp32")
        assert isinstance(m, toc3d_amd.ToC3DEVAViT) and m.precision == "fp32"
        assert isinstance(reg.build_neck(configs.CPFPN_TINY), toc3d_amd.CPFPN)
    finally:
        for n, v in saved.items():
            if v is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = v
        importlib.reload(registry)
        registry.register_all()
    assert not registry.HAVE_MMDET


def test_copies_of_modules_drop_recorded_plans_and_packed_weights():
    """A LaunchPlan's launches carry baked device pointers of the module that recorded it: plans are not copyable, and copies /
    pickles of the modules start without plans, workspaces and packed weights (they re-pack and re-record)."""
    import copy
    import pickle

    from toc3d_amd.plan import LaunchPlan
    lp = LaunchPlan()
    with pytest.raises(TypeError):
        copy.deepcopy(lp)
    with pytest.raises(TypeError):
        copy.copy(lp)
    with pytest.raises(TypeError):
        pickle.dumps(lp)
    m = toc3d_amd.build_backbone(configs.get("toc3d_tiny"))
    m._plans = {"k": {"launch": {None: {"cplan": lp}}, "x": torch.zeros(2)}}
    m._packed = {"dev": "cuda", "blocks": [lp]}
    m._tuned[(0, 1, 2, 3)] = 16
    for c in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert c._plans == {} and c._packed is None and c._stream_pool == []
        assert c._tuned == {(0, 1, 2, 3): 16} and c.pruning_loc == m.pruning_loc
        assert all(torch.equal(a, b) for a, b in zip(c.state_dict().values(), m.state_dict().values()))
        assert c.blocks[0].attn.q_proj.weight.data_ptr() != m.blocks[0].attn.q_proj.weight.data_ptr()
    n = toc3d_amd.build_neck(configs.CPFPN_TINY)
    n._ws = {(1, 2, 3): {"launch": {7: {"cplan": lp}}}}
    n._packed = {"w": lp}
    c = copy.deepcopy(n)
    assert c._ws == {} and c._packed is None
    # weights arriving through a parent module (mmcv load_checkpoint on the detector) invalidate the neck's packed state too
    parent = torch.nn.Module()
    parent.neck = n
    parent.load_state_dict({"neck." + k: v for k, v in n.state_dict().items()})
    assert n._ws == {} and n._packed is None


def test_recording_lane_is_thread_local():
    """lib.stream_ptr() hands out lane handles only on the thread that records (the C side's recording flag is thread_local)."""
    import threading
    lib.set_rec_lane(2)
    try:
        assert lib.recording() and lib.stream_ptr() == lib.load().toc3d_plan_lane_stream(2)
        seen = {}
        t = threading.Thread(target=lambda: seen.update(rec=lib.recording(), lane=lib.rec_lane()))
        t.start(); t.join()
        assert seen == {"rec": False, "lane": None}
    finally:
        lib.set_rec_lane(None)
    assert not lib.recording()


def test_product_forward_has_no_test_switches():
    """forced_scores / block_hook live in toc3d_amd.testing (a test-only subclass), not in the product forward."""
    import inspect

    from toc3d_amd.backbone import ToC3DEVAViT
    from toc3d_amd.testing import InstrumentedToC3DEVAViT, instrument
    sig = inspect.signature(ToC3DEVAViT.forward)
    assert "forced_scores" not in sig.parameters and "gumbel_noise" in sig.parameters
    m = toc3d_amd.build_backbone(configs.get("toc3d_tiny"))
    assert not hasattr(m, "block_hook") and m._instrumented is False
    with pytest.raises(TypeError, match="test instrument"):
        ToC3DEVAViT.forward(m, torch.zeros(2, 3, 320, 800), forced_scores=[])
    im = instrument(m)
    assert im is m and isinstance(m, InstrumentedToC3DEVAViT) and m._instrumented is True and "forced_scores" in inspect.signature(type(m).forward).parameters


def test_launch_schedule_switches_are_attributes_not_environment():
    """The schedule switches of the product forward (toc3d_amd.backbone.schedule_defaults) are constructor kwargs / attributes with shipped defaults per
    precision; no module of the product package reads an environment variable to pick a code path (TOC3D_LIB, the library file name, is the one
    environment hook: it selects WHICH build is loaded, not what the host code does)."""
    from toc3d_amd.backbone import schedule_defaults
    assert set(schedule_defaults("bf16")) == {"carry_compact", "fold_ffn_ln", "fold_norm2", "gathered_residual", "prefetch_weights", "attn_rot", "side_lanes",
                                              "big_windows_first", "launch_mode", "gather_split", "x3_planes", "x3_attention"}
    d16, d32, dx3 = schedule_defaults("bf16"), schedule_defaults("fp32"), schedule_defaults("fp32x3")
    assert d16["fold_norm2"] and d16["carry_compact"] and d16["attn_rot"] and d16["prefetch_weights"] > 0
    assert not (d32["fold_ffn_ln"] or d32["fold_norm2"] or d32["carry_compact"] or d32["attn_rot"]), "the strict-parity path keeps the reference's sequence"
    assert dx3["fold_ffn_ln"] and dx3["fold_norm2"] and dx3["carry_compact"] and dx3["attn_rot"], "fp32x3: the folds, the carried compact set and (round 6) the pre-rotated attention on (hi, lo) planes"
    assert dx3["x3_planes"] and dx3["x3_attention"] and not (d16["x3_planes"] or d32["x3_planes"] or d16["x3_attention"] or d32["x3_attention"]), "(hi, lo) planes are the fp32x3 path's operand layout only"
    m = toc3d_amd.build_backbone(dict(configs.get("toc3d_tiny"), schedule=dict(fold_norm2=False, side_lanes=False)))
    assert m.fold_norm2 is False and m.side_lanes is False and m.fold_ffn_ln is True
    with pytest.raises(TypeError, match="unknown schedule"):
        toc3d_amd.build_backbone(dict(configs.get("toc3d_tiny"), schedule=dict(ln_self=True)))
    pkg = os.path.dirname(os.path.abspath(toc3d_amd.__file__))
    for fn in sorted(os.listdir(pkg)):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            reads = set(re.findall(r"environ[^\n]*?[\"'](TOC3D_\w+)[\"']", src))
            assert reads <= ({"TOC3D_LIB"} if fn == "lib.py" else set()), (fn, reads)


def test_frame_timeline_cuts_frames_and_counts_idle_time():
    """tools/frame_timeline.py (the tool behind profiles/r03_where_time_goes*.txt): frames are cut at the copy_segments launches, overlapping side-lane
    kernels do not count as idle time, mangled and plain kernel names land in one family, torch / runtime kernels are counted as foreign."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("frame_timeline", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "frame_timeline.py"))
    ft = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ft)
    g1 = "_ZN12_GLOBAL__N_111gemm_kernelIDF16bLi4ELi128ELi128ELi1ELi128ELi2ELi4ELi6ELi0EEEv8GemmArgs"
    g2 = "void (anonymous namespace)::gemm_kernel<__bf16, 5, 64, 64, 2, 128, 2, 2, 1, 0>(GemmArgs)"
    rows = []
    for f in range(3):
        t = f * 10000
        rows += [(t, t + 100, "(anonymous namespace)::copy_segments_kernel(void const*)", "0"), (t + 100, t + 1100, g1, "0"),
                 (t + 300, t + 900, "(anonymous namespace)::motion_queries_kernel<float>(int)", "3"),      # another stream, inside the GEMM: no idle time
                 (t + 1150, t + 2150, g2, "0"),                                                             # 50 ns gap
                 (t + 2150, t + 2200, "void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>>(int)", "0")]
    frames = ft.cut_frames(list(reversed(rows)))
    assert len(frames) == 2                                                                                # the open last frame is dropped
    r = ft.summarize(frames[0])
    assert r["launches"] == 5 and r["span"] == 2200 and r["idle"] == 50 and r["foreign"] == 1
    assert r["families"]["gemm_kernel (all toc3d_linear* launches)"] == [2, 2000]
    assert r["families"]["motion_queries_kernel (side lane)"] == [1, 600]


def test_header_constants_match_the_python_binding():
    """dtype / epilogue codes of include/toc3d.h and their twins in toc3d_amd/lib.py (what the host passes through ctypes) and csrc/common.h (what the library
    compares against) are one table."""
    txt = open(lib.HEADER_PATH).read()
    hdr = {m[1]: int(m[2]) for m in re.finditer(r"#define\s+(TOC3D_(?:DTYPE|EPI)_\w+)\s+(\d+)\b", txt)}
    py = {"TOC3D_DTYPE_F32": lib.F32, "TOC3D_DTYPE_BF16": lib.BF16, "TOC3D_DTYPE_F32X3": lib.F32X3, "TOC3D_DTYPE_F32X6": lib.F32X6,
          "TOC3D_DTYPE_F32X3W": lib.F32X3W, "TOC3D_DTYPE_F32X3P": lib.F32X3P, "TOC3D_DTYPE_F32X3WO": lib.F32X3WO, "TOC3D_DTYPE_F32X3WA": lib.F32X3WA,
          "TOC3D_EPI_BIAS": lib.EPI_BIAS, "TOC3D_EPI_RESIDUAL": lib.EPI_RESIDUAL, "TOC3D_EPI_SWIGLU": lib.EPI_SWIGLU, "TOC3D_EPI_GELU": lib.EPI_GELU,
          "TOC3D_EPI_SWIGLU_STATS": lib.EPI_SWIGLU_STATS, "TOC3D_EPI_RESIDUAL_LN": lib.EPI_RESIDUAL_LN, "TOC3D_EPI_RESIDUAL_STATS": lib.EPI_RESIDUAL_STATS,
          "TOC3D_EPI_SWIGLU_STATS_LN": lib.EPI_SWIGLU_STATS_LN, "TOC3D_EPI_CONV3X3": lib.EPI_CONV3X3, "TOC3D_EPI_QKV_ROPE": lib.EPI_QKV_ROPE}
    for k, v in py.items():
        assert hdr.get(k) == v, (k, hdr.get(k), v)
    common = open(os.path.join(os.path.dirname(os.path.abspath(toc3d_amd.__file__)), "csrc", "common.h")).read()
    enum = dict((m[1], int(m[2])) for m in re.finditer(r"(TOC3D_(?:F32|BF16)\w*)\s*=\s*(\d+)", common))
    for k, v in py.items():
        if k.startswith("TOC3D_DTYPE_"):
            assert enum.get("TOC3D_" + k[len("TOC3D_DTYPE_"):]) == v, (k, enum)
    # the q scale of the pre-rotated attention path is the LIBRARY's convention (exp2-based softmax): the Python constant must be what it reports
    assert abs(lib.load().toc3d_attn_rot_q_scale(64) - lib.ATTN_ROT_Q_SCALE) < 1e-7 * lib.ATTN_ROT_Q_SCALE


def test_backbone_survives_pickles_written_before_gumbel_seed_became_a_property():
    """a whole-module pickle from before the property carries `gumbel_seed` in __dict__ and no `_gumbel_seed`; reading the property then raised through
    nn.Module.__getattr__.  __setstate__ migrates the entry; the class carries defaults."""
    import copy
    import pickle
    m = toc3d_amd.build_backbone(dict(configs.get("toc3d_tiny")))
    m.gumbel_seed = 1234
    st = m.__getstate__()
    st["gumbel_seed"] = st.pop("_gumbel_seed")                 # the old layout
    old = m.__class__.__new__(m.__class__)
    old.__setstate__(st)
    assert old.gumbel_seed == 1234 and "gumbel_seed" not in old.__dict__
    assert pickle.loads(pickle.dumps(m)).gumbel_seed == 1234 and copy.deepcopy(m).gumbel_seed == 1234


def test_unchanged_reference_config_builds_the_parity_grade_precision():
    """A config without a `precision` key (a reference config dropped in unchanged) builds the path that meets the reference's 1e-3 tolerance, not the bf16
    headline path (INTEGRATION.md "Precision"); backbone and neck agree."""
    from toc3d_amd.backbone import DEFAULT_PRECISION
    assert DEFAULT_PRECISION == "fp32x3"
    for name in ("toc3d_tiny", "eva_tiny"):
        assert toc3d_amd.build_backbone(configs.get(name)).precision == "fp32x3"
        assert toc3d_amd.build_backbone(dict(configs.get(name), precision="bf16")).precision == "bf16"
    assert toc3d_amd.build_neck(configs.CPFPN_TINY).precision == "fp32x3"


def test_splitk_workspace_is_sized_from_the_launched_tile_shape():
    """toc3d_linear_splitk_workspace_bytes inferred (BM, BN) from BM * BN, which cannot tell variant 9's 128x64 tile from the 64x128 tile of
    variants 10 / 26 -- the host then counted too few tiles when N % 128 is in 1..64 and accepted a workspace the kernel writes past.  The size now comes from the launch
    table's own tile shape (csrc/gemm_kernels.h, sk_tile_dims): checked here for every split-K tile variant on shapes with ragged N, without a GPU (a host function)."""
    L = lib.load()
    dims = {1: (128, 128), 9: (128, 64), 10: (64, 128), 14: (64, 64), 16: (128, 128), 17: (128, 128), 19: (256, 128), 22: (128, 128), 26: (64, 128), 28: (128, 128),
            29: (128, 128), 55: (96, 128), 56: (96, 128)}
    for v, (bm, bn) in dims.items():
        for split in (2, 3, 4):
            for M, N in ((128, 192), (128, 64), (300, 1024), (6000, 1024), (97, 130)):
                tiles = -(-M // bm) * -(-N // bn)
                want = 65536 + tiles * split * bm * bn * 4
                got = L.toc3d_linear_splitk_workspace_bytes(1000 * split + v, M, N)
                assert got == want, (v, split, M, N, got, want)
    assert L.toc3d_linear_splitk_workspace_bytes(2060, 128, 128) < 0 and L.toc3d_linear_splitk_workspace_bytes(16, 128, 128) < 0      # no split-K form / not a split variant
