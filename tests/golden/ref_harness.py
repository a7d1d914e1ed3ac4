"""Import harness for the READ-ONLY reference at /root/reference (build container only).

Used ONLY by tests/golden/make_golden.py to generate the committed golden vectors; nothing in the
product, the -m gpu tests, smoke() or bench.py imports this (the reference does not exist on the GPU box).

Recipe: stub the third-party modules the reference imports but that are absent here
(cv2, torchvision, pyshtools, torch_cluster, torch_geometric, skimage, neural_renderer's CUDA ext),
shim `np.int` (network.py:46 needs it under numpy >= 1.24) and never write bytecode into /root/reference.
"""
import sys
import types
import importlib

REF_ROOT = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package so that `import a.b` works
    sys.modules[name] = m
    return m


def import_reference():
    """Returns a dict of the reference modules on the hot path (CPU-importable)."""
    import numpy as np
    sys.dont_write_bytecode = True
    if not hasattr(np, 'int'):
        np.int = int  # noqa: NPY001 - reference uses the removed alias
    for name in ['cv2', 'torchvision', 'torchvision.utils', 'pyshtools', 'torch_cluster',
                 'torch_geometric', 'torch_geometric.nn', 'torch_geometric.utils',
                 'skimage', 'skimage.transform', 'skimage.io', 'neural_renderer',
                 'tensorboardX', 'pytorch_msssim', 'trimesh']:
        if name not in sys.modules:
            _stub(name)
    sys.modules['torch_cluster'].knn_graph = None
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    mods = {}
    for name in ['misc', 'camera', 'sph_harm', 'render', 'data_util', 'network']:
        mods[name] = importlib.import_module(name)
    mods['pytorch_prototyping'] = importlib.import_module('pytorch_prototyping.pytorch_prototyping')
    return mods


if __name__ == '__main__':
    m = import_reference()
    print({k: v.__file__ for k, v in m.items()})
