"""Host-side mirror of the reference API: config tree, module/parameter naming, checkpoint key
compatibility, NativeVRProcessor bookkeeping -- everything that needs no GPU."""
import os

import numpy as np
import pytest
import torch

from bathymetric_gnn_amd import synthetic
from bathymetric_gnn_amd.config import Config, CORRECTION_NORM_FLOOR, CORRECTION_NORM_CAP
from bathymetric_gnn_amd.models import BathymetricGNN


def test_config_defaults_roundtrip_and_asserts(tmp_path):
    c = Config()
    assert (c.tile.tile_size, c.tile.overlap, c.tile.min_valid_ratio) == (1024, 128, 0.1)
    assert c.graph.connectivity == "8-connected" and c.graph.edge_features == ["distance", "depth_difference", "slope"]
    assert (c.model.gnn_type, c.model.gnn_hidden_channels, c.model.gnn_num_layers, c.model.gnn_heads) == ("GAT", 64, 4, 4)
    assert (c.inference.auto_correct_threshold, c.inference.review_threshold) == (0.85, 0.6)
    assert (CORRECTION_NORM_FLOOR, CORRECTION_NORM_CAP) == (0.01, 50.0)
    c.tile.tile_size = 512; c.model.gnn_num_layers = 3; c.training = {"epochs": 7}
    p = tmp_path / "config.yaml"
    c.save(p)
    d = Config.load(p)
    assert d.tile.tile_size == 512 and d.model.gnn_num_layers == 3 and d.training == {"epochs": 7}
    with pytest.raises(AssertionError):
        Config(tile=type(c.tile)(tile_size=100, overlap=80))
    with pytest.raises(AssertionError):
        Config(graph=type(c.graph)(connectivity="6-connected"))


def test_state_dict_keys_match_reference_names():
    m = BathymetricGNN(in_channels=7, edge_dim=3)
    keys = set(m.state_dict().keys())
    ref = set(synthetic.synthetic_state_dict().keys())
    assert keys == ref
    assert m.feature_extractor.mlp[0].in_features == 7          # scripts/inference_native.py:147
    assert (m.CLASS_SEAFLOOR, m.CLASS_FEATURE, m.CLASS_NOISE) == (0, 1, 2)
    assert m.gnn.convs[0].lin.weight.shape == (256, 64) and m.gnn.convs[3].lin.weight.shape == (64, 256)
    assert m.gnn.convs[3].bias.shape == (64,) and m.gnn.norms[0].module.weight.shape == (256,)
    # older torch_geometric checkpoints (lin_src / lin_dst) load too
    m.load_state_dict({k: torch.as_tensor(v) for k, v in synthetic.synthetic_state_dict(legacy_lin_src=True).items()})
    # the other backbones carry torch_geometric's parameter names as well (models/gnn.py:120-143)
    for kind, keys in (("GCN", ["gnn.convs.0.lin.weight", "gnn.convs.0.bias"]),
                       ("GraphSAGE", ["gnn.convs.0.lin_l.weight", "gnn.convs.0.lin_l.bias", "gnn.convs.0.lin_r.weight"]),
                       ("GIN", ["gnn.convs.0.nn.0.weight", "gnn.convs.0.nn.0.bias", "gnn.convs.0.nn.2.weight", "gnn.convs.0.nn.2.bias"])):
        mk = BathymetricGNN(in_channels=7, gnn_type=kind, num_gnn_layers=2)
        sk = set(mk.state_dict())
        assert all(k in sk for k in keys) and "gnn.norms.1.module.running_var" in sk
        assert "gnn.convs.0.lin_r.bias" not in sk
        ref = synthetic.synthetic_state_dict(gnn_type=kind, num_layers=2)
        assert sk == set(ref)
        mk.load_state_dict({k: torch.as_tensor(v) for k, v in ref.items()})
    # edge_dim=None (the reference signature's default): GATConv without edge parameters, like torch_geometric's state dict
    m0 = BathymetricGNN(in_channels=7, gnn_type="GAT", edge_dim=None)
    k0 = set(m0.state_dict())
    assert k0 == {k for k in set(synthetic.synthetic_state_dict()) if "lin_edge" not in k and "att_edge" not in k}
    assert m0.pack_weights().size == m.pack_weights().size            # the library's layout is unchanged: zero edge weights are packed
    with pytest.raises(ValueError):
        BathymetricGNN(in_channels=7, gnn_type="Transformer", edge_dim=3)


def test_pack_weights_order_and_invalidate():
    sd = synthetic.synthetic_state_dict(seed=3)
    m = BathymetricGNN(in_channels=7, edge_dim=3)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    blob = m.pack_weights()
    n0 = sd["feature_extractor.mlp.0.weight"].size
    assert np.array_equal(blob[:n0], sd["feature_extractor.mlp.0.weight"].ravel())
    tail = sd["correction_head.mlp.3.bias"]
    assert np.array_equal(blob[-tail.size:], tail.ravel())
    v0 = m._weights_version()
    with torch.no_grad():
        m.gnn.convs[0].bias.add_(1.0)
    assert m._weights_version() != v0                           # native handle would be rebuilt
    # replacing NESTED parameters / modules / buffers is seen too (the cached tensor list is re-checked slot by slot)
    v1 = m._weights_version()
    assert m._weights_version() == v1
    m.gnn.convs[0].lin.weight = torch.nn.Parameter(torch.zeros_like(m.gnn.convs[0].lin.weight))
    v2 = m._weights_version()
    assert v2 != v1
    m.classification_head.mlp[3] = torch.nn.Linear(m.classification_head.mlp[3].in_features, 3)
    v3 = m._weights_version()
    assert v3 != v2 and np.array_equal(m.pack_weights()[:n0], blob[:n0])
    m.gnn.norms[0].module.register_buffer("running_mean", torch.ones_like(m.gnn.norms[0].module.running_mean))
    v4 = m._weights_version()
    assert v4 != v3
    m.invalidate_native()
    assert m._weights_version() == v4                           # same tensors, re-read


def test_checkpoint_loading_paths(tmp_path, monkeypatch):
    """load_model accepts the trainer's dict layout (training/trainer.py:809-829) and finds the model
    hyper-parameters under 'model_config' (models/pipeline.py:108) -- exercised up to set_model."""
    from bathymetric_gnn_amd.models import pipeline as pl
    sd = {k: torch.as_tensor(v) for k, v in synthetic.synthetic_state_dict(in_channels=8, num_layers=3, seed=1).items()}
    path = tmp_path / "best_model.pt"
    torch.save({"epoch": 3, "model_state_dict": sd, "in_channels": 8, "edge_dim": 3,
                "model_config": {"gnn_hidden_channels": 64, "gnn_num_layers": 3, "gnn_type": "GAT", "gnn_heads": 4,
                                 "num_classes": 3, "predict_correction": True}}, path)
    captured = {}
    p = pl.BathymetricPipeline.__new__(pl.BathymetricPipeline)
    p.config = Config()
    monkeypatch.setattr(pl.BathymetricPipeline, "set_model", lambda self, m: captured.setdefault("m", m))
    p.load_model(path)
    assert captured["m"].in_channels == 8 and captured["m"].num_gnn_layers == 3
    with pytest.raises(FileNotFoundError):
        p.load_model(tmp_path / "missing.pt")
    # a checkpoint that pickles a Python object (the trainer's Config): refused unless unpickling is opted into
    import argparse
    pickled = tmp_path / "with_config_object.pt"
    torch.save({"model_state_dict": sd, "in_channels": 8, "edge_dim": 3,
                "config": argparse.Namespace(model=argparse.Namespace(gnn_num_layers=3))}, pickled)
    captured.clear()
    with pytest.raises(RuntimeError, match="trust_pickle=True"):
        p.load_model(pickled)
    assert "m" not in captured
    p.load_model(pickled, trust_pickle=True)
    assert captured["m"].num_gnn_layers == 3


def test_apply_results_arithmetic():
    from bathymetric_gnn_amd.scripts.inference_native import apply_results
    depth = np.array([[10.0, 11.0], [12.0, 1.0e6]], np.float32)
    unc = np.full((2, 2), 0.5, np.float32)
    cls = np.array([[2, 2], [1, 2]], np.float32)
    conf = np.array([[0.85, 0.5], [0.9, 0.99]], np.float32)
    corr = np.full((2, 2), 0.25, np.float32)
    valid = depth != 1.0e6
    applied = apply_results(depth, unc, cls, conf, corr, valid, 0.85)
    assert applied.tolist() == [[True, False], [False, False]]      # >= threshold, noise, valid (:480-503)
    assert depth[0, 0] == np.float32(9.75) and unc[0, 0] == np.float32(0.5 * (2 - 0.85))
    assert depth[1, 1] == np.float32(1.0e6)


def test_training_mode_dropout_spec_and_counter_hash():
    """forward() in train() mode runs the reference's four dropouts with a counter-based draw (include/bgnn.h, bgnn_dropout):
    the probabilities come from the modules, the seed from torch's generator (or ``dropout_seed``); modules of one place that
    disagree are refused.  The oracle's numpy restatement of the hash is pinned to splitmix64's published first output."""
    import pytest
    import torch
    from bathymetric_gnn_amd.models import BathymetricGNN
    from oracle.gat_cpu import CounterDropout
    # splitmix64, state 0: first output 0xE220A8397B1DCDAF (seed 0, stream 0, index 0 -> z = the golden-ratio increment)
    assert int(CounterDropout.hash32(0, 0, [0])[0]) == 0xE220A839
    h = CounterDropout.hash32(7, 2, np.arange(200000))
    assert abs(float((h < int(0.25 * 2 ** 32)).mean()) - 0.25) < 5e-3
    assert not np.array_equal(h, CounterDropout.hash32(7, 3, np.arange(200000)))
    m = BathymetricGNN(in_channels=7, edge_dim=3)            # reference default dropout 0.1, module starts in train()
    assert m.training
    torch.manual_seed(5)
    a = m._dropout_spec()
    torch.manual_seed(5)
    b = m._dropout_spec()
    assert a.seed == b.seed == m.last_dropout_seed
    assert [round(x, 6) for x in (a.p_extractor, a.p_attention, a.p_features, a.p_heads)] == [0.1] * 4
    m.dropout_seed = 99
    assert m._dropout_spec().seed == 99
    m.gnn.dropout = 0.3
    m.classification_head.mlp[2].p = 0.2
    with pytest.raises(NotImplementedError, match="head"):
        m._dropout_spec()
    m.confidence_head.mlp[2].p = 0.2; m.correction_head.mlp[2].p = 0.2
    s = m._dropout_spec()
    assert round(s.p_features, 6) == 0.3 and round(s.p_heads, 6) == 0.2
    assert BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0)._dropout_spec() is None


def test_model_shapes_that_stay_refused_say_so_in_the_constructor():
    from bathymetric_gnn_amd.models import BathymetricGNN
    for kw, what in ((dict(hidden_channels=160), "hidden_channels=160"), (dict(hidden_channels=1), "hidden_channels=1"),
                     (dict(hidden_channels=96, heads=5), "8 heads of 128"), (dict(hidden_channels=64, heads=9), "16 heads of 64"),
                     (dict(heads=0), "heads=0")):
        with pytest.raises(ValueError, match=what):
            BathymetricGNN(in_channels=7, **kw)
    BathymetricGNN(in_channels=7, hidden_channels=80, heads=4)                            # 4 heads of 128: fits
    BathymetricGNN(in_channels=7, hidden_channels=100, heads=9, gnn_type="GCN")           # heads does not shape a plain backbone
