"""CPU oracle: BathymetricGNN forward (restates reference ``models/gnn.py`` + the
torch_geometric ``GATConv`` / ``BatchNorm`` / ``Batch`` semantics it calls).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  **Parity unpinned**: the GATConv /
BatchNorm arithmetic lives in ``torch_geometric`` (third party; the reference leaves the
version unpinned -- ``environment.yml:50-52``, ``install.sh:58`` -- and it is not
installable here), and the reference's tests assert no value at this boundary.  This file
restates the published upstream algorithm (torch_geometric 2.3-2.6 ``GATConv`` with
``edge_dim``) and is anchored on the reference's call sites:

* ``models/gnn.py:52-68``   LocalFeatureExtractor: Linear, ReLU, Dropout, Linear (keys mlp.0 / mlp.3)
* ``models/gnn.py:125-132`` GATConv(layer_in, 64, heads=H or 1 (last), dropout, edge_dim, concat=not last);
  every other argument at its upstream default: negative_slope=0.2, add_self_loops=True,
  fill_value='mean', bias=True
* ``models/gnn.py:151-154`` BatchNorm(H*64) / BatchNorm(64) (wraps torch.nn.BatchNorm1d, eps 1e-5)
* ``models/gnn.py:173-188`` conv -> norm -> (ReLU, dropout) except last layer
* ``models/gnn.py:191-260`` heads: Linear(64,32) ReLU Dropout Linear(32,k) [Sigmoid for confidence]
* ``models/gnn.py:386-406`` forward: logits, softmax, argmax, confidence, correction
* ``models/gnn.py:427-449`` predict: action / needs_review / auto_correct

The torch op sequence below is the one torch_geometric issues on a CPU tensor
(index_select / scatter-amax / exp / index_add), with edges kept in the given order and the
N self-loops appended last, so the float32 mode reproduces the summation order of the real
library; the float64 mode is the "truth" used to report absolute error.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import numpy as np
import torch
import torch.nn.functional as F


def _t(v, dtype):
    if isinstance(v, torch.Tensor):
        return v.detach().to("cpu", dtype)
    return torch.as_tensor(np.asarray(v)).to(dtype)


def _lin_weight(sd: Mapping, prefix: str, dtype):
    # newer releases: lin.weight; older: lin_src.weight (lin_dst is an alias of it)
    for k in (prefix + "lin.weight", prefix + "lin_src.weight"):
        if k in sd:
            return _t(sd[k], dtype)
    raise KeyError(prefix + "lin.weight")


def num_layers_of(sd: Mapping) -> int:
    n = 0
    while (f"gnn.norms.{n}.module.weight") in sd:       # one BatchNorm per layer, whatever the convolution
        n += 1
    return n


class CounterDropout:
    """Training-mode dropout with the library's counter-based draws (include/bgnn.h, ``bgnn_dropout``): the reference's four
    dropouts (models/gnn.py:57, :125-132, :186, :206 / :229 / :253) keep a value with probability 1 - p and multiply it by
    1 / (1 - p); WHICH values go is a pure function of (seed, stream, index), restated here in numpy so that the oracle runs
    with the very masks the kernels use (torch's own generator stream cannot be followed: it differs between torch's CPU and
    GPU builds).  hash = splitmix64's finaliser of seed + 0x9E3779B97F4A7C15 (stream + 1) + 0xD1B54A32D192ED03 index."""

    def __init__(self, seed: int, p_extractor: float = 0.0, p_attention: float = 0.0, p_features: float = 0.0, p_heads: float = 0.0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.p_extractor, self.p_attention, self.p_features, self.p_heads = p_extractor, p_attention, p_features, p_heads

    @staticmethod
    def hash32(seed: int, stream: int, index) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.asarray(index, dtype=np.uint64)
            z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.uint64(stream + 1) + np.uint64(0xD1B54A32D192ED03) * idx)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return (z >> np.uint64(32)).astype(np.uint32)

    def _mult(self, p: float, stream: int, index, dtype) -> torch.Tensor:
        """keep / (1 - p) per element, the scale rounded to float32 as the kernels hold it."""
        thr = min(int(float(np.float32(p)) * 4294967296.0), 4294967295)
        keep = self.hash32(self.seed, stream, index) >= np.uint32(thr)
        scale = np.float32(1.0 / (1.0 - float(np.float32(p))))
        return torch.as_tensor(np.where(keep, scale, np.float32(0.0)).astype(np.float32)).to(dtype)

    def elementwise(self, x: torch.Tensor, p: float, stream: int) -> torch.Tensor:
        if p <= 0.0:
            return x
        n, w = x.shape
        return x * self._mult(p, stream, np.arange(n * w, dtype=np.uint64), x.dtype).view(n, w)

    def attention(self, alpha: torch.Tensor, src, dst, layer: int) -> torch.Tensor:
        if self.p_attention <= 0.0:
            return alpha
        E, H = alpha.shape
        with np.errstate(over="ignore"):
            pair = (dst.numpy().astype(np.uint64) << np.uint64(32)) | src.numpy().astype(np.uint64)
            idx = pair[:, None] * np.uint64(H) + np.arange(H, dtype=np.uint64)[None, :]
        return alpha * self._mult(self.p_attention, 16 + layer, idx.reshape(-1), alpha.dtype).view(E, H)


def gat_conv(x, edge_index, edge_attr, sd: Mapping, prefix: str, concat: bool, dtype,
             return_alpha: bool = False, dropout: "CounterDropout" = None, layer: int = 0):
    """One torch_geometric GATConv forward (eval mode; ``dropout``: training mode's ``F.dropout`` on alpha)."""
    N = x.shape[0]
    W = _lin_weight(sd, prefix, dtype)                    # [H*C, D]
    att_src = _t(sd[prefix + "att_src"], dtype)           # [1,H,C]
    att_dst = _t(sd[prefix + "att_dst"], dtype)
    # edge_dim=None (models/gnn.py:93,130): no lin_edge / att_edge in the state dict -> the logit has no edge term
    has_edge = (prefix + "att_edge") in sd
    att_edge = _t(sd[prefix + "att_edge"], dtype) if has_edge else None
    W_e = _t(sd[prefix + "lin_edge.weight"], dtype) if has_edge else None   # [H*C, edge_dim]
    bias = _t(sd[prefix + "bias"], dtype)
    H, C = att_src.shape[1], att_src.shape[2]

    xs = (x @ W.t()).view(N, H, C)
    a_s = (xs * att_src).sum(-1)                          # [N,H]
    a_d = (xs * att_dst).sum(-1)

    src, dst = edge_index[0], edge_index[1]
    # remove_self_loops, then add_self_loops(fill_value='mean')
    keep = src != dst
    src, dst, ea = src[keep], dst[keep], edge_attr[keep]
    loop_sum = torch.zeros(N, ea.shape[1], dtype=dtype).index_add_(0, dst, ea)
    cnt = torch.zeros(N, dtype=dtype).index_add_(0, dst, torch.ones(dst.shape[0], dtype=dtype))
    loop_attr = loop_sum / cnt.clamp(min=1).unsqueeze(-1)
    ar = torch.arange(N, dtype=src.dtype)
    src2 = torch.cat([src, ar]); dst2 = torch.cat([dst, ar]); ea2 = torch.cat([ea, loop_attr], 0)

    if has_edge:
        a_e = ((ea2 @ W_e.t()).view(-1, H, C) * att_edge).sum(-1)      # [E',H]
        e = F.leaky_relu(a_s.index_select(0, src2) + a_d.index_select(0, dst2) + a_e, 0.2)
    else:
        e = F.leaky_relu(a_s.index_select(0, src2) + a_d.index_select(0, dst2), 0.2)
    m = torch.full((N, H), float("-inf"), dtype=dtype)
    m = m.scatter_reduce(0, dst2.unsqueeze(-1).expand(-1, H), e, reduce="amax", include_self=True)
    p = torch.exp(e - m.index_select(0, dst2))
    s = torch.zeros(N, H, dtype=dtype).index_add_(0, dst2, p)
    alpha = p / (s.index_select(0, dst2) + 1e-16)
    if dropout is not None:
        alpha = dropout.attention(alpha, src2, dst2, layer)
    msg = alpha.unsqueeze(-1) * xs.index_select(0, src2)           # [E',H,C]
    out = torch.zeros(N, H, C, dtype=dtype).index_add_(0, dst2, msg)
    out = out.reshape(N, H * C) if concat else out.mean(dim=1)
    out = out + bias
    if return_alpha:
        return out, alpha, src2, dst2
    return out


def batch_norm_eval(x, sd: Mapping, prefix: str, dtype, eps: float = 1e-5):
    w = _t(sd[prefix + "weight"], dtype); b = _t(sd[prefix + "bias"], dtype)
    rm = _t(sd[prefix + "running_mean"], dtype); rv = _t(sd[prefix + "running_var"], dtype)
    return F.batch_norm(x, rm, rv, w, b, False, 0.1, eps)


def batch_norm_train(x, sd: Mapping, prefix: str, dtype, stats: dict, eps: float = 1e-5, momentum: float = 0.1):
    """torch.nn.BatchNorm1d in training mode (what torch_geometric's BatchNorm wraps; reference models/gnn.py:151-154
    with the module in train()): normalise with the batch mean / biased variance; the updated running statistics
    (momentum 0.1, unbiased variance) are returned in ``stats[prefix + 'running_mean' / 'running_var']``."""
    w = _t(sd[prefix + "weight"], dtype); b = _t(sd[prefix + "bias"], dtype)
    rm = _t(sd[prefix + "running_mean"], dtype).clone(); rv = _t(sd[prefix + "running_var"], dtype).clone()
    out = F.batch_norm(x, rm, rv, w, b, True, momentum, eps)
    stats[prefix + "running_mean"], stats[prefix + "running_var"] = rm, rv
    return out


def _mlp2(x, sd, p0, p1, dtype, hidden_mult=None):
    h = F.relu(F.linear(x, _t(sd[p0 + ".weight"], dtype), _t(sd[p0 + ".bias"], dtype)))
    if hidden_mult is not None:                           # nn.Dropout between the two Linears, training mode
        h = h * hidden_mult
    return F.linear(h, _t(sd[p1 + ".weight"], dtype), _t(sd[p1 + ".bias"], dtype))


def gnn_type_of(sd: Mapping) -> str:
    """Backbone type from the state_dict key names (models/gnn.py:120-143)."""
    if "gnn.convs.0.lin_l.weight" in sd:
        return "GraphSAGE"
    if "gnn.convs.0.nn.0.weight" in sd:
        return "GIN"
    if "gnn.convs.0.att_src" in sd:
        return "GAT"
    return "GCN"


def gcn_conv(x, edge_index, sd: Mapping, prefix: str, dtype):
    """torch_geometric ``GCNConv`` with default arguments (improved=False, add_self_loops=True, normalize=True,
    bias=True; parity unpinned like GATConv): x = lin(x); gcn_norm (add_remaining_self_loops with unit weights: the
    N loops go LAST; deg = scatter-add of the weights over the targets; norm = d^-1/2[src] * w * d^-1/2[dst]);
    out = scatter-add(norm * x[src]) + bias."""
    N = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    loops = torch.arange(N, dtype=torch.int64)
    src = torch.cat([src[keep], loops]); dst = torch.cat([dst[keep], loops])
    w = torch.ones(src.shape[0], dtype=dtype)
    deg = torch.zeros(N, dtype=dtype).index_add_(0, dst, w)
    dinv = deg.pow(-0.5)
    dinv[torch.isinf(dinv)] = 0
    norm = dinv[src] * w * dinv[dst]
    xw = F.linear(x, _t(sd[prefix + "lin.weight"], dtype))
    out = torch.zeros_like(xw).index_add_(0, dst, norm.unsqueeze(-1) * xw[src])
    return out + _t(sd[prefix + "bias"], dtype)


def sage_conv(x, edge_index, sd: Mapping, prefix: str, dtype):
    """``SAGEConv`` defaults (aggr='mean', root_weight=True, normalize=False): lin_l(mean_j x_j) + lin_r(x_i); the
    mean is scatter-sum / clamp(count, min=1); explicit self loops in edge_index are ordinary edges."""
    N = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    s = torch.zeros_like(x).index_add_(0, dst, x[src])
    cnt = torch.zeros(N, dtype=dtype).index_add_(0, dst, torch.ones(src.shape[0], dtype=dtype)).clamp(min=1)
    mean = s / cnt.unsqueeze(-1)
    return (F.linear(mean, _t(sd[prefix + "lin_l.weight"], dtype), _t(sd[prefix + "lin_l.bias"], dtype))
            + F.linear(x, _t(sd[prefix + "lin_r.weight"], dtype)))


def gin_conv(x, edge_index, sd: Mapping, prefix: str, dtype):
    """``GINConv(nn)`` defaults (eps=0, train_eps=False): nn(sum_j x_j + (1 + eps) x_i), nn = Linear ReLU Linear."""
    src, dst = edge_index[0], edge_index[1]
    s = torch.zeros_like(x).index_add_(0, dst, x[src])
    s = s + (1.0 + 0.0) * x
    return _mlp2(s, sd, prefix + "nn.0", prefix + "nn.2", dtype)


def backbone(x, edge_index, edge_attr, sd: Mapping, dtype, train_stats: dict = None, dropout: CounterDropout = None):
    """feature extractor + GNN backbone -> [N, hidden].  ``train_stats`` (a dict): training-mode BatchNorm, the updated
    running statistics are left in it; ``dropout``: training mode's active dropouts (CounterDropout)."""
    L = num_layers_of(sd)
    kind = gnn_type_of(sd)
    ext_mult = None
    if dropout is not None and dropout.p_extractor > 0:
        hid = _t(sd["feature_extractor.mlp.0.weight"], dtype).shape[0]
        ext_mult = dropout.elementwise(torch.ones(x.shape[0], hid, dtype=dtype), dropout.p_extractor, 1)
    h = _mlp2(x, sd, "feature_extractor.mlp.0", "feature_extractor.mlp.3", dtype, ext_mult)
    for l in range(L):
        last = l == L - 1
        if kind == "GAT":
            h = gat_conv(h, edge_index, edge_attr, sd, f"gnn.convs.{l}.", concat=not last, dtype=dtype, dropout=dropout, layer=l)
        elif kind == "GCN":
            h = gcn_conv(h, edge_index, sd, f"gnn.convs.{l}.", dtype)
        elif kind == "GraphSAGE":
            h = sage_conv(h, edge_index, sd, f"gnn.convs.{l}.", dtype)
        else:
            h = gin_conv(h, edge_index, sd, f"gnn.convs.{l}.", dtype)
        if train_stats is None:
            h = batch_norm_eval(h, sd, f"gnn.norms.{l}.module.", dtype)
        else:
            h = batch_norm_train(h, sd, f"gnn.norms.{l}.module.", dtype, train_stats)
        if not last:
            h = F.relu(h)
            if dropout is not None:
                h = dropout.elementwise(h, dropout.p_features, 64 + l)
    return h


def forward(sd: Mapping, x, edge_index, edge_attr, dtype=torch.float32, train_stats: dict = None,
            dropout: CounterDropout = None) -> Dict[str, torch.Tensor]:
    """BathymetricGNN.forward (models/gnn.py:360-408): eval mode, or -- with ``train_stats`` a dict -- training mode
    (batch-statistics BatchNorm; ``dropout``: the active dropouts with the library's counter-based masks)."""
    x = _t(x, dtype); edge_attr = _t(edge_attr, dtype)
    edge_index = _t(edge_index, torch.int64)
    with torch.no_grad():
        h = backbone(x, edge_index, edge_attr, sd, dtype, train_stats, dropout)
        hm = [None, None, None]
        if dropout is not None and dropout.p_heads > 0:
            # the kernels hold the three heads' hidden units side by side: classification | confidence | correction
            hh = _t(sd["classification_head.mlp.0.weight"], dtype).shape[0]
            nh = 3 if "correction_head.mlp.0.weight" in sd else 2
            m_all = dropout.elementwise(torch.ones(h.shape[0], nh * hh, dtype=dtype), dropout.p_heads, 2)
            hm = [m_all[:, i * hh:(i + 1) * hh] if i < nh else None for i in range(3)]
        logits = _mlp2(h, sd, "classification_head.mlp.0", "classification_head.mlp.3", dtype, hm[0])
        probs = F.softmax(logits, dim=-1)
        out = {
            "class_logits": logits,
            "class_probs": probs,
            "predicted_class": torch.argmax(probs, dim=-1),
            "confidence": torch.sigmoid(
                _mlp2(h, sd, "confidence_head.mlp.0", "confidence_head.mlp.3", dtype, hm[1])).squeeze(-1),
            "hidden": h,
        }
        if "correction_head.mlp.0.weight" in sd:
            out["correction"] = _mlp2(h, sd, "correction_head.mlp.0", "correction_head.mlp.3", dtype, hm[2]).squeeze(-1)
    return out


def predict(sd: Mapping, x, edge_index, edge_attr, auto_correct_threshold: float = 0.85,
            review_threshold: float = 0.6, dtype=torch.float32):
    """BathymetricGNN.predict (models/gnn.py:410-451)."""
    out = forward(sd, x, edge_index, edge_attr, dtype)
    conf, cls = out["confidence"], out["predicted_class"]
    action = torch.zeros_like(cls)
    action[(cls == 2) & (conf > auto_correct_threshold)] = 1
    action[conf < review_threshold] = 2
    out["action"] = action
    out["needs_review"] = action == 2
    out["auto_correct"] = action == 1
    return out


def process_tile(sd: Mapping, g, auto_correct_threshold=0.85, review_threshold=0.6,
                 dtype=torch.float32):
    """BathymetricPipeline._process_tile (models/pipeline.py:243-314) on an oracle graph:
    returns classification / confidence / correction grids (fill 0.0)."""
    from . import graph_cpu
    shape = g.grid_shape
    out = predict(sd, g.x, g.edge_index, g.edge_attr, auto_correct_threshold, review_threshold, dtype)
    cls = graph_cpu.graph_to_grid(g, out["predicted_class"].float().numpy(), 0.0)
    conf = graph_cpu.graph_to_grid(g, out["confidence"].float().numpy(), 0.0)
    corr = np.zeros(shape, np.float32)
    if "correction" in out:
        nc = graph_cpu.graph_to_grid(g, out["correction"].float().numpy(), 0.0)
        ls = graph_cpu.graph_to_grid(g, g.local_std, 0.0)
        corr = nc * np.maximum(ls, np.float32(0.01))      # config/constants.py:12
    return {"classification": cls, "confidence": conf, "correction": corr}
