"""ctypes binding of ``libbgnn_hip.so`` (C ABI: ``include/bgnn.h``) and per-GPU contexts.

There is deliberately no CPU fallback: if the library is missing or no GPU is visible the
compute entry points raise.  PyTorch is used only as the container for device memory and
streams (``tensor.data_ptr()``, ``torch.cuda.Stream``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, Optional

import numpy as np
import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libbgnn_hip.so")

K_NAMES = ["scan", "stats", "features", "export", "gemm", "attcoef", "aggregate", "heads", "scatter", "fused"]
K_INDEX = {n: i for i, n in enumerate(K_NAMES)}

NODE_FEATURE_IDS = {"depth": 0, "local_mean": 1, "local_std": 2, "gradient_x": 3, "gradient_y": 4,
                    "gradient_magnitude": 5, "curvature": 6, "uncertainty": 7}
EDGE_FEATURE_IDS = {"distance": 0, "depth_difference": 1, "slope": 2}
EF_ZERO = 3

ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_UNSUPPORTED = -1, -2, -3, -4
ABI_VERSION = 6
MATRIX_PATHS = {"exact_f32": 0, "bf16x3": 1, "fp16x3": 2, "bf16": 3}


GNN_TYPES = {"GAT": 0, "GCN": 1, "GraphSAGE": 2, "GIN": 3}      # BGNN_GNN_*


class ModelDesc(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("hidden", C.c_int32), ("num_layers", C.c_int32),
                ("heads", C.c_int32), ("num_classes", C.c_int32), ("edge_dim", C.c_int32),
                ("predict_correction", C.c_int32), ("bn_eps", C.c_float), ("gnn_type", C.c_int32)]


class Tiles(C.Structure):
    _fields_ = [("n_tiles", C.c_int32), ("hw", C.POINTER(C.c_int32)), ("resolution", C.POINTER(C.c_double)),
                ("depth", C.c_void_p), ("mask", C.c_void_p), ("uncertainty", C.c_void_p)]


class GraphOpts(C.Structure):
    _fields_ = [("connectivity", C.c_int32), ("include_self_loops", C.c_int32),
                ("n_node_features", C.c_int32), ("node_features", C.c_int32 * 8),
                ("n_edge_features", C.c_int32), ("edge_features", C.c_int32 * 4)]


class Outputs(C.Structure):
    _fields_ = [("class_logits", C.c_void_p), ("class_probs", C.c_void_p), ("predicted_class", C.c_void_p),
                ("confidence", C.c_void_p), ("correction", C.c_void_p), ("action", C.c_void_p),
                ("needs_review", C.c_void_p), ("auto_correct", C.c_void_p), ("hidden", C.c_void_p)]


class Dropout(C.Structure):
    """bgnn_dropout: the four dropout probabilities of a training-mode forward and the seed of its counter-based draws."""
    _fields_ = [("p_extractor", C.c_float), ("p_attention", C.c_float), ("p_features", C.c_float), ("p_heads", C.c_float),
                ("seed", C.c_uint64)]


# symbol -> (restype, argtypes); every symbol include/bgnn.h declares
_SIGNATURES = {
    "bgnn_abi_version": (C.c_int, []),
    "bgnn_last_error": (C.c_char_p, []),
    "bgnn_build_id": (C.c_char_p, []),
    "bgnn_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bgnn_ctx_destroy": (C.c_int, [C.c_void_p]),
    "bgnn_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "bgnn_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "bgnn_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "bgnn_ctx_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "bgnn_ctx_profile": (C.c_int, [C.c_void_p, C.c_uint32]),
    "bgnn_ctx_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "bgnn_model_weight_count": (C.c_size_t, [C.POINTER(ModelDesc)]),
    "bgnn_model_create": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.POINTER(C.c_float), C.c_size_t,
                                    C.POINTER(C.c_void_p)]),
    "bgnn_model_destroy": (C.c_int, [C.c_void_p]),
    "bgnn_graph_build": (C.c_int, [C.c_void_p, C.POINTER(Tiles), C.POINTER(GraphOpts), C.POINTER(C.c_void_p)]),
    "bgnn_graph_from_edges": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bgnn_graph_destroy": (C.c_int, [C.c_void_p]),
    "bgnn_graph_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "bgnn_graph_export": (C.c_int, [C.c_void_p] + [C.c_void_p] * 8),
    "bgnn_graph_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "bgnn_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.POINTER(Outputs)]),
    "bgnn_feature_extractor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgnn_heads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.POINTER(Outputs)]),
    "bgnn_forward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Outputs)]),
    "bgnn_forward_train_dropout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Dropout), C.c_void_p, C.c_void_p,
                                             C.POINTER(Outputs)]),
    "bgnn_stitch_tiles": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 6 +
                          [C.c_int32] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 4),
    "bgnn_cut_tiles": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                 C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgnn_tile_valid_counts": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_void_p]),
    "bgnn_vr_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_int32, C.c_void_p, C.c_double] +
                       [C.c_void_p] * 5),
    "bgnn_vr_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p]),
    "bgnn_infer_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Tiles), C.POINTER(GraphOpts), C.c_float,
                                   C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None
_lib_lock = threading.Lock()


class BgnnError(RuntimeError):
    pass


def load_library(path: Optional[str] = None):
    """dlopen the HIP library and bind every symbol of include/bgnn.h.  Needs no GPU."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        p = path or os.environ.get("BGNN_LIB", LIB_PATH)
        if not os.path.exists(p):
            raise ImportError(
                f"{p} not found: the HIP library is not built. Run `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(p)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.bgnn_abi_version() != ABI_VERSION:
            raise ImportError(f"{p}: ABI version {lib.bgnn_abi_version()} != {ABI_VERSION}")
        _lib = lib
        return lib


def build_id() -> str:
    """The kernel-source hash the loaded library was built from (``bgnn_build_id``)."""
    return load_library().bgnn_build_id().decode()


def check(rc: int):
    if rc == 0:
        return
    msg = load_library().bgnn_last_error().decode(errors="replace")
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    raise BgnnError(msg)


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Context:
    """One library context per (process, GPU): owns a HIP stream (created through torch so that
    torch events / stream waits can order against it) and the library's device arenas."""

    def __init__(self, device: torch.device):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise BgnnError("no GPU visible: bathymetric_gnn_amd computes only on an MI355X "
                            "(there is no CPU fallback)")
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream(self.device)
        h = C.c_void_p()
        check(self.lib.bgnn_ctx_create(self.device.index, C.c_void_p(self.stream.cuda_stream), C.byref(h)))
        self.handle = h
        self._closers = []                 # run by close() before the context goes: packed models that live on it

    def on_close(self, fn):
        """``fn()`` is called when this context is closed (objects that hold library handles created on it)."""
        self._closers.append(fn)

    def close(self):
        """Release the context (stream-synchronised): first whatever registered through ``on_close`` (packed models), then the
        library context with its arenas.  Graph objects built on it must not be used afterwards.  Idempotent."""
        if getattr(self, "handle", None) is None:
            return
        for fn in reversed(getattr(self, "_closers", [])):
            try:
                fn()
            except Exception:
                pass
        self._closers = []
        h, self.handle = self.handle, None
        self.lib.bgnn_ctx_destroy(h)

    # -- ordering against the caller's current torch stream ------------------------------------
    def begin(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def end(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def synchronize(self):
        check(self.lib.bgnn_ctx_synchronize(self.handle))

    # -- run-time switches (include/bgnn.h: bgnn_ctx_set_option).  The environment is read once, when the
    #    context is created; afterwards only these calls change a switch.
    def set_option(self, name: str, value):
        if name == "matrix_path" and isinstance(value, str):
            value = MATRIX_PATHS[value]
        check(self.lib.bgnn_ctx_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int()
        check(self.lib.bgnn_ctx_get_option(self.handle, name.encode(), C.byref(v)))
        return v.value

    def options(self, **kw):
        """``with ctx.options(matrix_path="bf16x3"): ...`` -- set, run, restore."""
        ctx = self

        class _Scope:
            def __enter__(self_):
                self_.old = {k: ctx.get_option(k) for k in kw}
                for k, v in kw.items():
                    ctx.set_option(k, v)
                return ctx

            def __exit__(self_, *exc):
                for k, v in self_.old.items():
                    ctx.set_option(k, v)
                return False
        return _Scope()

    def profile(self, kernels):
        mask = 0
        for k in kernels:
            mask |= 1 << (K_INDEX[k] if isinstance(k, str) else int(k))
        check(self.lib.bgnn_ctx_profile(self.handle, mask))

    def profile_read(self) -> Dict[str, Dict[str, float]]:
        ms = (C.c_double * len(K_NAMES))()
        n = (C.c_int64 * len(K_NAMES))()
        check(self.lib.bgnn_ctx_profile_read(self.handle, ms, n))
        return {K_NAMES[i]: {"ms": ms[i], "launches": int(n[i])} for i in range(len(K_NAMES))}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts: Dict[int, Context] = {}
_ctx_lock = threading.Lock()


def resolve_device(device=None) -> torch.device:
    if device is None:
        if not torch.cuda.is_available():
            raise BgnnError("no GPU visible: bathymetric_gnn_amd computes only on an MI355X "
                            "(there is no CPU fallback)")
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise BgnnError(f"device {device} unsupported: the hot path runs on the GPU only (no CPU fallback)")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def get_context(device=None) -> Context:
    device = resolve_device(device)
    with _ctx_lock:
        ctx = _contexts.get(device.index)
        if ctx is None:
            ctx = Context(device)
            _contexts[device.index] = ctx
        return ctx


def new_context(device=None) -> Context:
    """An ADDITIONAL library context on ``device`` (own HIP stream, own arenas): batches submitted to different contexts
    overlap on the GPU -- the tail of one small batch's kernels runs beside the head of the next one's.  The caller keeps
    it alive (models only hold weak references to the contexts they are packed on) and may ``close()`` it; ``get_context`` keeps
    returning the device's default context."""
    return Context(resolve_device(device))


def make_graph_opts(connectivity: str, include_self_loops: bool, node_features, edge_features) -> GraphOpts:
    conn = {"4-connected": 4, "8-connected": 8, "16-dilated": 16}.get(connectivity)
    if conn is None:
        raise ValueError(f"Unknown connectivity: {connectivity}")
    o = GraphOpts()
    o.connectivity = conn
    o.include_self_loops = 1 if include_self_loops else 0
    ids = [NODE_FEATURE_IDS[n] for n in node_features if n in NODE_FEATURE_IDS]   # unknown names are skipped
    if len(ids) > 8:
        raise ValueError("at most 8 node features")
    o.n_node_features = len(ids)
    for i, v in enumerate(ids):
        o.node_features[i] = v
    eids = [EDGE_FEATURE_IDS.get(n, EF_ZERO) for n in edge_features]             # unknown names give 0.0
    if not 1 <= len(eids) <= 4:
        raise ValueError("between 1 and 4 edge features are supported")
    o.n_edge_features = len(eids)
    for i, v in enumerate(eids):
        o.edge_features[i] = v
    return o


def make_tiles(hw: np.ndarray, res: np.ndarray, depth: torch.Tensor, mask: torch.Tensor,
               unc: Optional[torch.Tensor]):
    """hw int32 [T,2], res float64 [T,2] (host); depth/mask/unc flat device tensors.
    Returns (Tiles, keepalive)."""
    hw = np.ascontiguousarray(hw, dtype=np.int32)
    res = np.ascontiguousarray(res, dtype=np.float64)
    t = Tiles()
    t.n_tiles = hw.shape[0]
    t.hw = hw.ctypes.data_as(C.POINTER(C.c_int32))
    t.resolution = res.ctypes.data_as(C.POINTER(C.c_double))
    t.depth = depth.data_ptr()
    t.mask = mask.data_ptr()
    t.uncertainty = unc.data_ptr() if unc is not None else None
    return t, (hw, res, depth, mask, unc)
