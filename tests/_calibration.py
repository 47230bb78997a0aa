"""Test weights whose outputs are not constant.

The seeded synthetic weights (``synthetic.synthetic_state_dict``) give almost node-independent head outputs (logit
spread ~0.005, confidence spread ~0.002): one class, one action everywhere, so class / flag comparisons would be
vacuous.  ``calibrate_heads`` re-centres and stretches the LAST layer of the classification and confidence heads around
the oracle's outputs on the test's own graph: class means become equal (classes mix), the confidence logit gets mean
``conf_centre`` and standard deviation ``conf_std`` (confidence straddles the 0.6 / 0.85 thresholds).  Gains are kept
moderate: they also amplify the float32 noise of the backbone, and the 1e-4 logit bar is absolute.
"""
import numpy as np
import torch

from oracle import gat_cpu


def calibrate_heads(sd, x, edge_index, edge_attr, logit_spread=0.1, conf_lo=0.35, conf_hi=1.9, max_gain=128.0):
    """The last layers are affine, so the calibrated outputs follow from ONE oracle forward: logits' = g (logits - median),
    z' = g_c (z - z_10) + conf_lo with the 10th..90th percentiles of the confidence logit z stretched onto
    [conf_lo, conf_hi] (confidence 0.59 .. 0.87: both thresholds inside); small per-class bias offsets are
    chosen so that every class and 'noise & confident' (action 1) occur."""
    sd = dict(sd)
    out = gat_cpu.forward(sd, x, edge_index, edge_attr)
    lg = out["class_logits"].numpy().astype(np.float64)
    c = np.clip(out["confidence"].numpy().astype(np.float64), 1e-9, 1 - 1e-9)
    z = np.log(c / (1 - c))
    med = np.median(lg, 0)
    spread = np.mean(np.percentile(lg, 90, 0) - np.percentile(lg, 10, 0))
    g_l = float(min(max_gain, logit_spread / max(spread, 1e-12)))
    z10, z90 = np.percentile(z, 10), np.percentile(z, 90)
    g_c = float(min(max_gain, (conf_hi - conf_lo) / max(z90 - z10, 1e-12)))
    bump = np.zeros(lg.shape[1])
    best, sign = -1.0, 1.0
    steps = (-6.0, -4.0, -3.0, -2.0, -1.5, -1.0, -0.5, -0.25, 0.0, 0.25, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0) if lg.shape[1] > 2 else (0.0,)
    for sg in (1.0, -1.0):                               # the confidence logit may be anti-correlated with 'noise'
        zc = sg * g_c * (z - (z10 if sg > 0 else z90)) + conf_lo
        conf2 = 1.0 / (1.0 + np.exp(-zc))
        for b1 in steps:                                 # small per-class bias offsets: pick the most balanced outcome
            for b2 in steps:
                cand = np.zeros(lg.shape[1])
                if lg.shape[1] > 2:
                    cand[1], cand[2] = b1 * logit_spread, b2 * logit_spread
                cls = np.argmax(g_l * (lg - med) + cand, 1)
                act1 = np.mean((cls == 2) & (conf2 > 0.86))
                score = min([np.mean(cls == k) for k in range(lg.shape[1])] + [act1 * 3])
                if score > best:
                    best, bump, sign = score, cand, sg
    z_ref = z10 if sign > 0 else z90
    g_c *= sign
    for head, centre, gain, off in (("classification_head", med, g_l, bump), ("confidence_head", np.array([z_ref]), g_c, np.array([conf_lo]))):
        w = np.asarray(sd[f"{head}.mlp.3.weight"], np.float64); b = np.asarray(sd[f"{head}.mlp.3.bias"], np.float64)
        sd[f"{head}.mlp.3.weight"] = (gain * w).astype(np.float32)
        sd[f"{head}.mlp.3.bias"] = (gain * (b - centre) + off).astype(np.float32)
    return sd


def assert_mixed(ref, min_nodes=400):
    """The oracle's outputs exercise every branch: >= 2 classes (all of them for 3-class models on big graphs) and all
    three `action` values.  Only asserted on graphs large enough for that to be expected."""
    n = int(ref["predicted_class"].shape[0])
    if n < min_nodes:
        return False
    cls = torch.unique(ref["predicted_class"]).numel()
    assert cls >= 2, f"test weights give a single class on {n} nodes: the class comparison would be vacuous"
    if "action" in ref:
        acts = set(torch.unique(ref["action"]).tolist())
        assert acts == {0, 1, 2}, f"test weights give actions {acts} on {n} nodes: the flag comparison would be vacuous"
    return True
