"""BathymetricGNN -- drop-in for the reference's ``models/gnn.py`` (GAT backbone = the hot path; GCN / GraphSAGE / GIN too).

The module tree and parameter names equal the reference's (and torch_geometric's ``GATConv`` /
``BatchNorm`` wrappers), so ``load_state_dict`` takes a reference checkpoint unchanged and
``model.feature_extractor.mlp[0].in_features`` (read at ``scripts/inference_native.py:147``) works.
The torch parameters are only the weight container: ``forward`` packs them once into the library's
blob (``bgnn_model_create``) and runs the hand-written HIP kernels (``bgnn_forward``).  Inference
(eval) semantics, plus the training-mode FORWARD: batch-statistics BatchNorm and the reference's four dropouts (counter-based draws).
"""
from __future__ import annotations

import ctypes as C
import logging
import math
import weakref
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import runtime as rt
from ..data.graph_construction import GraphData

logger = logging.getLogger(__name__)


class LocalFeatureExtractor(nn.Module):
    """Per-node MLP: Linear, ReLU, Dropout, [hidden blocks], Linear (reference :34-71)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers: int = 2, dropout: float = 0.1):
        super().__init__()
        seq = [nn.Linear(in_channels, hidden_channels), nn.ReLU(), nn.Dropout(dropout)]
        for _ in range(num_layers - 2):
            seq += [nn.Linear(hidden_channels, hidden_channels), nn.ReLU(), nn.Dropout(dropout)]
        seq.append(nn.Linear(hidden_channels, out_channels))
        self.mlp = nn.Sequential(*seq)
        self._owner = None           # weakref to the BathymetricGNN whose packed weights the kernels read

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[N, in] -> [N, out] on the HIP kernels (``bgnn_feature_extractor``): eval semantics (dropout = identity)."""
        return _owner_of(self)._run_submodule("feature_extractor", x)


def _glorot_(t: torch.Tensor):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


class GATConv(nn.Module):
    """Parameter container with torch_geometric ``GATConv``'s names and shapes (edge_dim variant):
    ``lin.weight [H*C, D]``, ``att_src/att_dst/att_edge [1,H,C]``, ``lin_edge.weight [H*C, edge_dim]``,
    ``bias [H*C]`` (concat) or ``[C]``.  Initialised like upstream (glorot / zeros)."""

    def __init__(self, in_channels, out_channels, heads=1, dropout=0.0, edge_dim=None, concat=True):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.dropout, self.edge_dim = concat, dropout, edge_dim
        self.negative_slope = 0.2
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        if edge_dim is not None:
            self.lin_edge = nn.Linear(edge_dim, heads * out_channels, bias=False)
            self.att_edge = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels))
        _glorot_(self.lin.weight); _glorot_(self.att_src); _glorot_(self.att_dst)
        if edge_dim is not None:
            _glorot_(self.lin_edge.weight); _glorot_(self.att_edge)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # older torch_geometric releases store the shared projection as lin_src (+ alias lin_dst)
        src = prefix + "lin_src.weight"
        if src in state_dict and prefix + "lin.weight" not in state_dict:
            state_dict[prefix + "lin.weight"] = state_dict.pop(src)
            state_dict.pop(prefix + "lin_dst.weight", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class BatchNorm(nn.Module):
    """torch_geometric's ``BatchNorm`` wrapper: parameters live under ``.module``."""

    def __init__(self, in_channels, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps=eps, momentum=momentum)


class GCNConv(nn.Module):
    """Parameter container with torch_geometric ``GCNConv``'s names: ``lin.weight [out, in]`` (no bias), ``bias [out]``."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        _glorot_(self.lin.weight)


class SAGEConv(nn.Module):
    """torch_geometric ``SAGEConv`` (mean aggregation, root weight): ``lin_l.{weight,bias}``, ``lin_r.weight``."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)


class GINConv(nn.Module):
    """torch_geometric ``GINConv(nn)`` with eps = 0 (not trained): parameters live under ``nn``."""

    def __init__(self, mlp: nn.Module):
        super().__init__()
        self.nn = mlp


class GNNBackbone(nn.Module):
    """L x (conv, BatchNorm[, ReLU, Dropout]) (reference :74-188).  ``gnn_type='GAT'`` is the accelerated hot path;
    GCN / GraphSAGE / GIN (every layer hidden -> hidden, torch_geometric default arguments) run on plain gather +
    GEMM kernels."""

    def __init__(self, in_channels, hidden_channels, num_layers, gnn_type="GAT", heads=4, dropout=0.1, edge_dim=None):
        super().__init__()
        if gnn_type not in ("GAT", "GCN", "GraphSAGE", "GIN"):
            raise ValueError(f"Unknown GNN type: {gnn_type}")
        self.gnn_type, self.num_layers, self.dropout = gnn_type, num_layers, dropout
        self.convs, self.norms = nn.ModuleList(), nn.ModuleList()
        for i in range(num_layers):
            last = i == num_layers - 1
            if gnn_type == "GAT":
                layer_in = in_channels if i == 0 else hidden_channels * heads
                self.convs.append(GATConv(layer_in, hidden_channels, heads=1 if last else heads, dropout=dropout,
                                          edge_dim=edge_dim, concat=not last))
                self.norms.append(BatchNorm(hidden_channels if last else hidden_channels * heads))
                continue
            layer_in = in_channels if i == 0 else hidden_channels
            if gnn_type == "GCN":
                self.convs.append(GCNConv(layer_in, hidden_channels))
            elif gnn_type == "GraphSAGE":
                self.convs.append(SAGEConv(layer_in, hidden_channels))
            else:
                self.convs.append(GINConv(nn.Sequential(nn.Linear(layer_in, hidden_channels), nn.ReLU(),
                                                        nn.Linear(hidden_channels, hidden_channels))))
            self.norms.append(BatchNorm(hidden_channels))


def _owner_of(sub):
    owner = sub._owner() if sub._owner is not None else None
    if owner is None:
        raise RuntimeError(f"{type(sub).__name__} runs through its BathymetricGNN's packed weights on the GPU; "
                           "a detached sub-module has no compute path (there is no CPU fallback)")
    return owner


class _Head(nn.Module):
    _which = None

    def __init__(self, in_channels, hidden_channels, out_features, dropout, sigmoid=False):
        super().__init__()
        seq = [nn.Linear(in_channels, hidden_channels), nn.ReLU(), nn.Dropout(dropout),
               nn.Linear(hidden_channels, out_features)]
        if sigmoid:
            seq.append(nn.Sigmoid())
        self.mlp = nn.Sequential(*seq)
        self._owner = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Backbone output [N, hidden] -> this head's output (reference :191-260) on the HIP kernels (``bgnn_heads``)."""
        return _owner_of(self)._run_submodule(self._which, x)


class ClassificationHead(_Head):
    _which = "class_logits"

    def __init__(self, in_channels, hidden_channels, num_classes, dropout=0.1):
        super().__init__(in_channels, hidden_channels, num_classes, dropout)


class ConfidenceHead(_Head):
    _which = "confidence"

    def __init__(self, in_channels, hidden_channels, dropout=0.1):
        super().__init__(in_channels, hidden_channels, 1, dropout, sigmoid=True)


class CorrectionHead(_Head):
    _which = "correction"

    def __init__(self, in_channels, hidden_channels, dropout=0.1):
        super().__init__(in_channels, hidden_channels, 1, dropout)


class BathymetricGNN(nn.Module):
    CLASS_SEAFLOOR = 0
    CLASS_FEATURE = 1
    CLASS_NOISE = 2

    def __init__(self, in_channels: int, hidden_channels: int = 64, num_gnn_layers: int = 4, gnn_type: str = "GAT",
                 heads: int = 4, num_classes: int = 3, predict_correction: bool = True, dropout: float = 0.1,
                 edge_dim: Optional[int] = None):
        super().__init__()
        # edge_dim=None (the signature's default, models/gnn.py:93,130,291): GATConv then has no lin_edge / att_edge and the attention
        # logit loses its edge term.  The kernels keep one code path: the folded edge vector V = att_edge . lin_edge is packed as zeros.
        self.gnn_type = gnn_type
        self.predict_correction = predict_correction
        self.num_classes = num_classes
        # The reference's config takes any hidden_channels / heads (config/config.py:43-45).  The kernels exist for hidden 32 / 64 /
        # 128 and power-of-two head counts; any other shape runs zero-padded to the next of those (bgnn_model_create: results
        # unchanged), as long as the padded layer -- heads rounded up to a power of two x hidden rounded up to 32 / 64 / 128 -- stays
        # within 512 columns.  What does not fit is refused HERE, by the constructor the reference's user calls.
        if not 2 <= int(hidden_channels) <= 128:
            raise ValueError(f"hidden_channels={hidden_channels} is not supported by the MI355X kernels (2..128)")
        if gnn_type == "GAT":
            pad_c = 32 if hidden_channels <= 32 else 64 if hidden_channels <= 64 else 128
            pad_h = 1 << max(0, int(heads) - 1).bit_length()
            if int(heads) < 1 or pad_h * pad_c > 512:
                raise ValueError(f"heads={heads} x hidden_channels={hidden_channels} is not supported by the MI355X kernels: a layer is "
                                 f"laid out as {pad_h} heads of {pad_c} channels (next power of two x next of 32 / 64 / 128), which must "
                                 "stay within 512 columns")
        self.in_channels, self.hidden_channels, self.heads = in_channels, hidden_channels, heads
        self.num_gnn_layers, self.edge_dim = num_gnn_layers, edge_dim
        self.feature_extractor = LocalFeatureExtractor(in_channels, hidden_channels, hidden_channels, 2, dropout)
        self.gnn = GNNBackbone(hidden_channels, hidden_channels, num_gnn_layers, gnn_type, heads, dropout, edge_dim)
        self.classification_head = ClassificationHead(hidden_channels, hidden_channels // 2, num_classes, dropout)
        self.confidence_head = ConfidenceHead(hidden_channels, hidden_channels // 2, dropout)
        self.correction_head = CorrectionHead(hidden_channels, hidden_channels // 2, dropout) if predict_correction else None
        self._native = {}            # id(ctx) -> (weakref to ctx, handle): a model does not keep extra contexts alive
        self._native_key = None
        for sub in (self.feature_extractor, self.classification_head, self.confidence_head, self.correction_head):
            if sub is not None:
                object.__setattr__(sub, "_owner", weakref.ref(self))
        logger.info(f"Created BathymetricGNN: {gnn_type} with {num_gnn_layers} layers, {hidden_channels} hidden channels")

    # ---- weights -> library ------------------------------------------------------------------
    def _edge_width(self, graph_edge_dim: Optional[int] = None) -> int:
        """Columns of edge_attr the packed model is built for.  With ``edge_dim`` given it is that (a graph of another width is
        refused by the library, as GATConv's lin_edge would refuse it); with ``edge_dim=None`` GATConv ignores edge_attr whatever
        its width (reference models/gnn.py:93,130), so the width is the GRAPH's and the edge weights packed for it are zeros."""
        if self.edge_dim is not None:
            return int(self.edge_dim)
        return int(graph_edge_dim) if graph_edge_dim else 3

    def _desc(self, graph_edge_dim: Optional[int] = None) -> rt.ModelDesc:
        d = rt.ModelDesc()
        d.in_channels, d.hidden, d.num_layers = self.in_channels, self.hidden_channels, self.num_gnn_layers
        d.heads, d.num_classes, d.edge_dim = self.heads, self.num_classes, self._edge_width(graph_edge_dim)
        d.gnn_type = rt.GNN_TYPES[self.gnn_type]
        d.predict_correction = 1 if self.predict_correction else 0
        d.bn_eps = float(self.gnn.norms[0].module.eps)
        return d

    def pack_weights(self, graph_edge_dim: Optional[int] = None) -> np.ndarray:
        """Flat float32 blob in the order ``bgnn_model_weight_count`` documents (include/bgnn.h)."""
        ed = self._edge_width(graph_edge_dim)
        sd = {k: v.detach().to("cpu", torch.float32).contiguous().numpy().ravel() for k, v in self.state_dict().items()
              if v.dtype.is_floating_point}
        parts = []
        for p in ("feature_extractor.mlp.0", "feature_extractor.mlp.3"):
            parts += [sd[p + ".weight"], sd[p + ".bias"]]
        for l in range(self.num_gnn_layers):
            c, n = f"gnn.convs.{l}.", f"gnn.norms.{l}.module."
            if self.gnn_type == "GAT":
                if self.edge_dim is None:        # no edge term: zero att_edge / lin_edge over however many edge features the graph has
                    att_e, lin_e = np.zeros_like(sd[c + "att_src"]), np.zeros(sd[c + "att_src"].size * ed, np.float32)
                else:
                    att_e, lin_e = sd[c + "att_edge"], sd[c + "lin_edge.weight"]
                parts += [sd[c + "lin.weight"], sd[c + "att_src"], sd[c + "att_dst"], att_e, lin_e, sd[c + "bias"]]
            elif self.gnn_type == "GCN":
                parts += [sd[c + "lin.weight"], sd[c + "bias"]]
            elif self.gnn_type == "GraphSAGE":
                parts += [sd[c + "lin_l.weight"], sd[c + "lin_l.bias"], sd[c + "lin_r.weight"]]
            else:
                parts += [sd[c + "nn.0.weight"], sd[c + "nn.0.bias"], sd[c + "nn.2.weight"], sd[c + "nn.2.bias"]]
            parts += [sd[n + "weight"], sd[n + "bias"], sd[n + "running_mean"], sd[n + "running_var"]]
        heads = ["classification_head", "confidence_head"] + (["correction_head"] if self.predict_correction else [])
        for h in heads:
            parts += [sd[h + ".mlp.0.weight"], sd[h + ".mlp.0.bias"], sd[h + ".mlp.3.weight"], sd[h + ".mlp.3.bias"]]
        return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)

    def _weights_version(self):
        """(storage, version counter) of every parameter and buffer: changes whenever a weight is written or moved.  The LIST of
        tensors is cached -- walking the module tree costs ~0.2 ms, as much as the host side of a whole 50 000-node batch -- beside
        the slots it was read from: every ``_parameters`` / ``_buffers`` / ``_modules`` dict of the tree with its length and the
        objects it held.  A call re-checks those slots by identity (~10 us), so replacing a nested parameter or sub-module
        (``model.gnn.convs[0].lin.weight = nn.Parameter(...)``, ``head.mlp[3] = nn.Linear(...)``, ``register_buffer`` on a child)
        rebuilds the list like ``_apply`` / ``load_state_dict`` / an assignment on the root do.  ``invalidate_native()`` forces it."""
        st = self.__dict__.get("_wv_struct")
        if st is not None:
            for d, k, o in st[0]:
                if d.get(k) is not o:
                    st = None
                    break
            else:
                for d, n in st[1]:
                    if len(d) != n:
                        st = None
                        break
        if st is None:
            slots, sizes, ts = [], [], []
            for m in self.modules():
                for d in (m._parameters, m._buffers, m._modules):
                    sizes.append((d, len(d)))
                    slots += [(d, k, o) for k, o in d.items()]
                ts += [t for d in (m._parameters, m._buffers) for t in d.values() if t is not None]
            st = self.__dict__["_wv_struct"] = (slots, sizes, ts)
        return tuple((p.data_ptr(), p._version) for p in st[2])

    def invalidate_native(self):
        """Forget the cached tensor list and the packed copies: the next call re-reads every weight."""
        self.__dict__.pop("_wv_struct", None)
        self._drop_native()

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_wv_struct", None)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.__dict__.pop("_wv_struct", None)
        return super().load_state_dict(*a, **k)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)):
            self.__dict__.pop("_wv_struct", None)
        super().__setattr__(name, value)

    def _drop_native(self, only_ctx_id=None):
        """Destroy the packed copies (all, or the one on the context with this id -- called when that context closes)."""
        for cid, (wctx, h) in list((self._native or {}).items()):
            if only_ctx_id is not None and cid[0] != only_ctx_id:
                continue
            ctx = wctx()
            try:
                if ctx is not None and (ctx.handle is not None or only_ctx_id is not None):
                    ctx.lib.bgnn_model_destroy(h)
            except Exception:
                pass
            del self._native[cid]
        if only_ctx_id is None:
            self._native, self._native_key = {}, None

    def native(self, ctx: rt.Context, graph_edge_dim: Optional[int] = None):
        """The packed model on ``ctx`` (one per library context -- and, for ``edge_dim=None`` models, per edge_attr width of the
        graphs they meet; rebuilt when a weight changes)."""
        key = self._weights_version()
        if self._native is None or self._native_key != key:
            self._drop_native()
            self._native_key = key
        ed = self._edge_width(graph_edge_dim)
        ent = self._native.get((id(ctx), ed))
        if ent is None:
            blob = self.pack_weights(ed)
            desc = self._desc(ed)
            n = ctx.lib.bgnn_model_weight_count(C.byref(desc))
            if n != blob.size:
                raise ValueError(f"weight blob has {blob.size} floats, library expects {n}")
            h = C.c_void_p()
            rt.check(ctx.lib.bgnn_model_create(ctx.handle, C.byref(desc), blob.ctypes.data_as(C.POINTER(C.c_float)),
                                               blob.size, C.byref(h)))
            ent = self._native[(id(ctx), ed)] = (weakref.ref(ctx), h)
            me, cid = weakref.ref(self), id(ctx)
            ctx.on_close(lambda: me() is not None and me()._drop_native(cid))
        return ent[1]

    def __del__(self):
        try:
            self._drop_native()
        except Exception:
            pass

    # ---- forward -----------------------------------------------------------------------------
    def _device_of(self, data):
        if isinstance(data, GraphData):
            return data.device
        p = next(self.parameters())
        return p.device if p.device.type == "cuda" else None

    def _graph_of(self, data, ctx):
        if isinstance(data, GraphData):
            return data, None
        # a Data built elsewhere: x / edge_index / edge_attr tensors (gnn.py:381-383)
        x = data.x.detach().to(ctx.device, torch.float32).contiguous()
        ei = data.edge_index.detach().to(ctx.device, torch.int64).contiguous()
        ea = getattr(data, "edge_attr", None)
        if ea is None:
            if self.edge_dim is not None or self.gnn_type != "GAT":
                raise NotImplementedError("edge_attr=None is outside the built path")
            # GATConv(edge_dim=None) never looks at edge_attr (reference models/gnn.py:93,130,176): one column of zeros stands in
            ea = torch.zeros((ei.shape[1], 1), dtype=torch.float32, device=ctx.device)
        ea = ea.detach().to(ctx.device, torch.float32).contiguous()
        h = C.c_void_p()
        ctx.begin()
        rt.check(ctx.lib.bgnn_graph_from_edges(ctx.handle, x.shape[0], x.shape[1], rt.ptr(x), ei.shape[1], rt.ptr(ei),
                                               ea.shape[1], rt.ptr(ea), C.byref(h)))
        ctx.end()
        hw = np.zeros((1, 2), np.int32)
        g = GraphData(ctx, h, hw, x.shape[1], ea.shape[1])
        del g.grid_shape
        g._sizes = (x.shape[0], ei.shape[1], np.array([0, x.shape[0]]), np.array([0, ei.shape[1]]))
        return g, (x, ei, ea)

    def _run(self, data, thr_auto: float, thr_review: float, with_flags: bool, want_hidden: bool = False,
             train: bool = False):
        ctx = rt.get_context(self._device_of(data))
        g, keep = self._graph_of(data, ctx)
        if g.num_features != self.in_channels:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({g.num_nodes}x{g.num_features} and "
                               f"{self.in_channels}x{self.hidden_channels})")
        N, dev = g.num_nodes, ctx.device
        out = {
            "class_logits": torch.empty((N, self.num_classes), dtype=torch.float32, device=dev),
            "class_probs": torch.empty((N, self.num_classes), dtype=torch.float32, device=dev),
            "predicted_class": torch.empty((N,), dtype=torch.int64, device=dev),
            "confidence": torch.empty((N,), dtype=torch.float32, device=dev),
        }
        if self.predict_correction:
            out["correction"] = torch.empty((N,), dtype=torch.float32, device=dev)
        extra = {}
        if with_flags:
            extra["action"] = torch.empty((N,), dtype=torch.int64, device=dev)
            extra["needs_review"] = torch.empty((N,), dtype=torch.bool, device=dev)
            extra["auto_correct"] = torch.empty((N,), dtype=torch.bool, device=dev)
        hidden = torch.empty((N, self.hidden_channels), dtype=torch.float32, device=dev) if want_hidden else None
        o = rt.Outputs()
        o.class_logits, o.class_probs = out["class_logits"].data_ptr(), out["class_probs"].data_ptr()
        o.predicted_class, o.confidence = out["predicted_class"].data_ptr(), out["confidence"].data_ptr()
        o.correction = out["correction"].data_ptr() if self.predict_correction else None
        if with_flags:
            o.action, o.needs_review = extra["action"].data_ptr(), extra["needs_review"].data_ptr()
            o.auto_correct = extra["auto_correct"].data_ptr()
        if hidden is not None:
            o.hidden = hidden.data_ptr()
        model_h = self.native(ctx, g.edge_dim)
        if N > 0 and train:
            if N == 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                                 f"torch.Size([1, {self.gnn.norms[0].module.num_features}])")
            widths = [n.module.num_features for n in self.gnn.norms]
            mean = torch.empty(sum(widths), dtype=torch.float32, device=dev)
            var = torch.empty_like(mean)
            dp = self._dropout_spec()
            ctx.begin()
            rt.check(ctx.lib.bgnn_forward_train_dropout(ctx.handle, model_h, g._handle, C.byref(dp) if dp is not None else None,
                                                        rt.ptr(mean), rt.ptr(var), C.byref(o)))
            ctx.end()
            with torch.no_grad():                       # torch.nn.BatchNorm1d's bookkeeping (momentum None = cumulative average)
                off = 0
                for n, w in zip(self.gnn.norms, widths):
                    bn = n.module
                    if bn.track_running_stats and bn.running_mean is not None:
                        bn.num_batches_tracked += 1
                        f = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
                        bn.running_mean.mul_(1.0 - f).add_(mean[off:off + w].to(bn.running_mean.device), alpha=f)
                        bn.running_var.mul_(1.0 - f).add_(var[off:off + w].to(bn.running_var.device), alpha=f)
                    off += w
        elif N > 0:
            ctx.begin()
            rt.check(ctx.lib.bgnn_forward(ctx.handle, model_h, g._handle, C.c_float(thr_auto), C.c_float(thr_review),
                                          C.byref(o)))
            ctx.end()
        out.update(extra)
        if hidden is not None:
            out["hidden"] = hidden
        return out

    def _run_submodule(self, which: str, x: torch.Tensor) -> torch.Tensor:
        """``model.feature_extractor(x)`` / ``model.classification_head(h)`` / ``confidence_head(h)`` /
        ``correction_head(h)`` (reference :386, :392-406) through the C ABI, eval semantics."""
        p = next(self.parameters())
        ctx = rt.get_context(x.device if x.device.type == "cuda" else (p.device if p.device.type == "cuda" else None))
        x = x.detach().to(ctx.device, torch.float32).contiguous()
        width = self.in_channels if which == "feature_extractor" else self.hidden_channels
        if x.dim() != 2 or x.shape[1] != width:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(x.shape)} and {width}x...)")
        N, model_h = x.shape[0], self.native(ctx)
        ctx.begin()
        if which == "feature_extractor":
            out = torch.empty((N, self.hidden_channels), dtype=torch.float32, device=ctx.device)
            rt.check(ctx.lib.bgnn_feature_extractor(ctx.handle, model_h, rt.ptr(x), N, rt.ptr(out)))
        else:
            shape = (N, self.num_classes) if which == "class_logits" else (N,)
            out = torch.empty(shape, dtype=torch.float32, device=ctx.device)
            o = rt.Outputs()
            setattr(o, which, out.data_ptr())
            rt.check(ctx.lib.bgnn_heads(ctx.handle, model_h, rt.ptr(x), N, C.c_float(0.85), C.c_float(0.6), C.byref(o)))
        ctx.end()
        return out

    def _dropout_spec(self):
        """The four dropout probabilities of the reference's training-mode forward (models/gnn.py:57 extractor, :125-132 GATConv
        attention, :186 between the layers, :206 / :229 / :253 heads) and a seed -> ``rt.Dropout``, or None when all are 0.
        The seed is ``self.dropout_seed`` if set, else drawn from torch's default generator (so ``torch.manual_seed`` makes a
        run reproducible); the one used is left in ``self.last_dropout_seed``.  WHICH values are dropped is a counter-based
        draw of this library (include/bgnn.h), not torch's own stream."""
        def one(ps, what):
            ps = sorted(set(float(p) for p in ps))
            if len(ps) > 1:
                raise NotImplementedError(f"training-mode forward: the {what} dropout modules carry different probabilities {ps}")
            return ps[0] if ps else 0.0
        heads = [h for h in (self.classification_head, self.confidence_head, self.correction_head) if h is not None]
        p_ext = one([m.p for m in self.feature_extractor.modules() if isinstance(m, nn.Dropout)], "feature extractor")
        p_att = one([c.dropout for c in self.gnn.convs if isinstance(c, GATConv)], "GATConv")
        p_feat = float(self.gnn.dropout)
        p_head = one([m.p for h in heads for m in h.modules() if isinstance(m, nn.Dropout)], "head")
        for p in (p_ext, p_att, p_feat, p_head):
            if not 0.0 <= p < 1.0:
                raise ValueError(f"dropout probability has to be in [0, 1), but got {p}")
        if p_ext == p_att == p_feat == p_head == 0.0:
            return None
        seed = getattr(self, "dropout_seed", None)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self.last_dropout_seed = int(seed)
        return rt.Dropout(p_ext, p_att, p_feat, p_head, int(seed))

    def forward(self, data) -> Dict[str, torch.Tensor]:
        """class_logits [N,C], class_probs [N,C], predicted_class [N] i64, confidence [N],
        correction [N] (reference :360-408).

        ``eval()``: running statistics, dropout the identity.  ``train()`` (forward only, there is no backward pass
        here): every BatchNorm layer normalises with the statistics of this batch and moves its running statistics, and
        the four dropouts of the reference are active with their modules' probabilities (``bgnn_forward_train_dropout``;
        see ``_dropout_spec`` for the seed)."""
        if not self.training:
            return self._run(data, 0.85, 0.6, with_flags=False)
        return self._run(data, 0.85, 0.6, with_flags=False, train=True)

    def predict(self, data, auto_correct_threshold: float = 0.85, review_threshold: float = 0.6):
        """forward + deployment flags (reference :410-451)."""
        self.eval()
        with torch.no_grad():
            return self._run(data, auto_correct_threshold, review_threshold, with_flags=True)
