"""The plain-torch half of the model, pinned to the reference itself.

``tests/golden/model_*.npz`` were produced by executing the reference's own ``LocalFeatureExtractor``, the three heads
and ``BathymetricGNN.forward`` / ``.predict`` (``/root/reference/models/gnn.py``, loaded by path in the build container:
``tests/golden/make_golden_model.py``).  The oracle (``oracle/gat_cpu.py``) must reproduce them; the ``-m gpu``
counterpart (``tests/test_gpu_model_golden.py``) holds the HIP kernels to the same vectors.

``model_predict_wiring.npz`` was generated with a per-node linear stand-in for torch_geometric's GATConv: it is a
WIRING fixture (module order, BatchNorm placement, ReLU except after the last layer, softmax / argmax, thresholds,
dtypes) and pins nothing about the GATConv arithmetic, which stays "parity unpinned".
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from conftest import GOLDEN_DIR
from oracle import gat_cpu


def _load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def _close(a, b, tol=2e-6):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def test_oracle_extractor_matches_reference_fixture():
    z = _load("model_extractor.npz")
    for in_ch in (7, 8):
        sd = {k[len(f"in{in_ch}."):]: v for k, v in z.items() if k.startswith(f"in{in_ch}.mlp")}
        y = gat_cpu._mlp2(torch.from_numpy(z[f"in{in_ch}.x"]), sd, "mlp.0", "mlp.3", torch.float32).numpy()
        assert y.shape == z[f"in{in_ch}.out"].shape == (257, 64)
        assert _close(y, z[f"in{in_ch}.out"])
        y64 = gat_cpu._mlp2(torch.from_numpy(z[f"in{in_ch}.x"]).double(), sd, "mlp.0", "mlp.3", torch.float64).numpy()
        assert _close(y64, z[f"in{in_ch}.out"], 1e-5)          # the float64 statement agrees to float32 rounding


def test_oracle_heads_match_reference_fixture():
    z = _load("model_heads.npz")
    h = torch.from_numpy(z["h"])
    lg = gat_cpu._mlp2(h, z, "classification_head.mlp.0", "classification_head.mlp.3", torch.float32)
    cf = torch.sigmoid(gat_cpu._mlp2(h, z, "confidence_head.mlp.0", "confidence_head.mlp.3", torch.float32)).squeeze(-1)
    cr = gat_cpu._mlp2(h, z, "correction_head.mlp.0", "correction_head.mlp.3", torch.float32).squeeze(-1)
    assert _close(lg.numpy(), z["class_logits"]) and _close(cf.numpy(), z["confidence"]) and _close(cr.numpy(), z["correction"])


def test_oracle_forward_wiring_and_predict_flags_match_reference_fixture(monkeypatch):
    """WIRING ONLY (stand-in conv = per-node Linear): extractor -> [conv -> BatchNorm -> ReLU except last] x L -> heads,
    softmax / argmax, predict's strict thresholds with review overriding auto-correct."""
    z = _load("model_predict_wiring.npz")
    sd = {k[3:]: v for k, v in z.items() if k.startswith("sd.")}

    def stand_in(x, edge_index, edge_attr, sd_, prefix, concat, dtype, return_alpha=False, dropout=None, layer=0):
        return F.linear(x, gat_cpu._t(sd_[prefix + "lin.weight"], dtype), gat_cpu._t(sd_[prefix + "lin.bias"], dtype))

    monkeypatch.setattr(gat_cpu, "gat_conv", stand_in)
    monkeypatch.setattr(gat_cpu, "gnn_type_of", lambda sd_: "GAT")
    assert gat_cpu.num_layers_of(sd) == 3
    out = gat_cpu.forward(sd, z["x"], z["edge_index"], z["edge_attr"])
    for k in ("class_logits", "class_probs", "confidence", "correction"):
        assert _close(out[k].numpy(), z["forward." + k], 5e-6), k
    top2 = np.sort(z["forward.class_probs"], axis=1)
    sure = (top2[:, -1] - top2[:, -2]) > 1e-5
    assert out["predicted_class"].dtype == torch.int64 and z["forward.predicted_class"].dtype == np.int64
    assert np.array_equal(out["predicted_class"].numpy()[sure], z["forward.predicted_class"][sure])
    assert len(np.unique(z["forward.predicted_class"])) == 3
    for i in range(4):
        ta, tr = z[f"predict{i}.thresholds"]
        p = gat_cpu.predict(sd, z["x"], z["edge_index"], z["edge_attr"], float(ta), float(tr))
        c = z["forward.confidence"]
        safe = sure & (np.abs(c - ta) > 1e-5) & (np.abs(c - tr) > 1e-5)
        for k in ("action", "needs_review", "auto_correct"):
            assert np.array_equal(p[k].numpy()[safe], z[f"predict{i}.{k}"][safe]), (i, k)
    assert set(np.unique(z["predict0.action"])) == {0, 1, 2}


def test_predict_flag_rule_on_the_fixture_itself():
    """The fixture's flags follow the rule the C kernels implement (models/gnn.py:427-449): action 1 where class == 2 and
    confidence > auto threshold; then action 2 wherever confidence < review threshold (overrides)."""
    z = _load("model_predict_wiring.npz")
    cls, c = z["forward.predicted_class"], z["forward.confidence"]
    for i in range(4):
        ta, tr = z[f"predict{i}.thresholds"]
        a = np.zeros_like(cls)
        a[(cls == 2) & (c > np.float32(ta))] = 1
        a[c < np.float32(tr)] = 2
        assert np.array_equal(a, z[f"predict{i}.action"])
        assert np.array_equal(a == 2, z[f"predict{i}.needs_review"]) and np.array_equal(a == 1, z[f"predict{i}.auto_correct"])
