"""Configuration objects read by the inference hot path.

Field names and defaults follow the reference's dataclass tree (``config/config.py:12-142``) for
the sections the path uses -- ``tile``, ``graph``, ``model``, ``inference`` and ``device`` -- so a
YAML written by the reference's ``Config.save`` loads here unchanged.  Training / synthetic-noise
sections are outside the path: they are carried through as plain dicts, not interpreted.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional

import yaml

_GNN_TYPES = ("GCN", "GAT", "GraphSAGE", "GIN")
_CONNECTIVITIES = ("4-connected", "8-connected")


@dataclass
class TileConfig:                       # config/config.py:12-17
    tile_size: int = 1024
    overlap: int = 128
    min_valid_ratio: float = 0.1


@dataclass
class GraphConfig:                      # config/config.py:20-30
    connectivity: str = "8-connected"
    max_edge_distance: float = 2.0      # never read by the reference either
    include_self_loops: bool = False
    edge_features: List[str] = field(default_factory=lambda: ["distance", "depth_difference", "slope"])


@dataclass
class ModelConfig:                      # config/config.py:33-50
    local_feature_channels: int = 32
    local_feature_layers: int = 3
    local_kernel_size: int = 5
    gnn_type: str = "GAT"
    gnn_hidden_channels: int = 64
    gnn_num_layers: int = 4
    gnn_heads: int = 4
    gnn_dropout: float = 0.1
    num_classes: int = 3
    predict_correction: bool = True


@dataclass
class InferenceConfig:                  # config/config.py:103-114
    auto_correct_threshold: float = 0.85
    review_threshold: float = 0.6
    export_classification: bool = True
    export_confidence: bool = True
    export_correction_magnitude: bool = True
    export_review_priority: bool = True


def _section(cls, data: Optional[Dict[str, Any]]):
    data = dict(data or {})
    known = {f.name for f in dataclasses.fields(cls)}
    return cls(**{k: v for k, v in data.items() if k in known})


@dataclass
class Config:                           # config/config.py:119-142
    tile: TileConfig = field(default_factory=TileConfig)
    graph: GraphConfig = field(default_factory=GraphConfig)
    model: ModelConfig = field(default_factory=ModelConfig)
    inference: InferenceConfig = field(default_factory=InferenceConfig)
    training: Dict[str, Any] = field(default_factory=dict)   # carried, not interpreted
    noise: Dict[str, Any] = field(default_factory=dict)      # carried, not interpreted
    data_dir: Optional[str] = None
    output_dir: Optional[str] = None
    model_path: Optional[str] = None
    device: str = "cuda"
    num_workers: int = 4
    pin_memory: bool = True
    log_level: str = "INFO"
    wandb_project: Optional[str] = None
    wandb_entity: Optional[str] = None

    def __post_init__(self):            # same three checks as config/config.py:215-222
        assert self.tile.tile_size >= self.tile.overlap * 2, "Tile size must be at least 2x overlap"
        assert self.model.gnn_type in _GNN_TYPES, f"Unknown GNN type: {self.model.gnn_type}"
        assert self.graph.connectivity in _CONNECTIVITIES, f"Unknown connectivity: {self.graph.connectivity}"

    def save(self, path) -> None:
        def plain(o):
            if isinstance(o, dict):
                return {k: plain(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [plain(v) for v in o]
            return o
        with open(Path(path), "w") as f:
            yaml.dump(plain(dataclasses.asdict(self)), f, default_flow_style=False)

    @classmethod
    def load(cls, path) -> "Config":
        with open(Path(path), "r") as f:
            data = yaml.safe_load(f) or {}
        top = {f.name for f in dataclasses.fields(cls)} - {"tile", "graph", "model", "inference", "training", "noise"}
        return cls(tile=_section(TileConfig, data.get("tile")), graph=_section(GraphConfig, data.get("graph")),
                   model=_section(ModelConfig, data.get("model")),
                   inference=_section(InferenceConfig, data.get("inference")),
                   training=dict(data.get("training") or {}), noise=dict(data.get("noise") or {}),
                   **{k: data[k] for k in top if k in data})
