"""bathymetric-gnn_amd: MI355X-native hot path of grant-froelich/Bathymetric-GNN.

Grid -> graph construction, the GAT message-passing classifier forward and the
tile-batch inference loop, behind the reference's own Python API
(``data.GraphBuilder``, ``models.BathymetricGNN``, ``models.BathymetricPipeline``,
``scripts.inference_native.NativeVRProcessor``).  All compute runs in hand-written HIP
kernels for gfx950 reached through the C ABI declared in ``include/bgnn.h``; there is no
CPU fallback -- importing works anywhere, computing raises without the library + a GPU.

Submodules mirror the reference's layout: ``config``, ``data``, ``models``, ``scripts``.
"""
__version__ = "0.1.0"

__all__ = ["config", "data", "models", "scripts", "synthetic", "runtime"]


def __getattr__(name):  # lazy submodules: `import bathymetric_gnn_amd as b; b.data.GraphBuilder`
    if name in __all__:
        import importlib
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
