"""Import harness for the READ-ONLY reference at /root/reference (build container only).

Used ONLY by tests/golden/make_golden.py to generate the committed golden vectors; nothing in the
product, the -m gpu tests, smoke() or bench.py imports this (the reference does not exist on the GPU box).

Recipe: stub the third-party modules the reference imports but that are absent here
(cv2, torchvision, pyshtools, torch_cluster, torch_geometric, skimage, neural_renderer's CUDA ext),
shim `np.int` (network.py:46 needs it under numpy >= 1.24) and never write bytecode into /root/reference.

Provenance guard: the repo ships drop-in modules with the SAME import names as the reference (`network`, `render`,
`camera`, `misc`, `sph_harm`, `dataio`, `neural_renderer`, `pytorch_prototyping`).  `/root/reference/pytorch_prototyping/`
has no `__init__.py` (a namespace package), and a regular package found anywhere on `sys.path` beats a namespace
portion — so plain `import pytorch_prototyping` would pick up the repo's package whenever `relightable-nr_amd/` is on the
path.  Therefore every reference module is resolved HERE, by explicit file location under REF_ROOT, seeded into
`sys.modules` before anything imports it by name, and `assert_reference_modules` verifies that every hot-path module
in `sys.modules` really comes from REF_ROOT.
"""
import importlib
import importlib.machinery
import importlib.util
import os
import sys
import types

REF_ROOT = '/root/reference'
REF_NR_ROOT = os.path.join(REF_ROOT, 'neural_renderer')

# import names that exist both in the reference and in the repo's drop-in layer
SHARED_NAMES = ('misc', 'camera', 'sph_harm', 'render', 'data_util', 'network', 'dataio', 'neural_renderer',
                'pytorch_prototyping', 'gcn_lib')


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package so that `import a.b` works
    sys.modules[name] = m
    return m


def _purge_shared():
    """Forget any module of a shared name that was imported from somewhere else (e.g. the repo's drop-in layer)."""
    for name in list(sys.modules):
        top = name.split('.')[0]
        if top not in SHARED_NAMES:
            continue
        f = getattr(sys.modules[name], '__file__', None)
        if f is not None and not os.path.abspath(f).startswith(REF_ROOT + os.sep):
            del sys.modules[name]


def _seed_namespace_package(name, directory):
    """Register `name` as a package rooted at `directory` (the reference's namespace packages: no __init__.py)."""
    assert os.path.isdir(directory), directory
    spec = importlib.machinery.ModuleSpec(name, None, is_package=True)
    spec.submodule_search_locations = [directory]
    pkg = importlib.util.module_from_spec(spec)
    pkg.__path__ = [directory]
    sys.modules[name] = pkg
    return pkg


def _load_by_path(name, path):
    """Import the file `path` as module `name` (no sys.path lookup involved)."""
    assert path.startswith(REF_ROOT + os.sep) and os.path.isfile(path), path
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def module_origin(mod):
    f = getattr(mod, '__file__', None)
    if f is None:
        p = list(getattr(mod, '__path__', []) or [])
        f = p[0] if p else ''
    return os.path.abspath(f) if f else ''


def assert_reference_modules(mods=None):
    """Every shared-name module in sys.modules (and every module in `mods`) must live under /root/reference."""
    bad = []
    for name, mod in list(sys.modules.items()):
        if name.split('.')[0] in SHARED_NAMES and isinstance(mod, types.ModuleType):
            org = module_origin(mod)
            if org and not org.startswith(REF_ROOT + os.sep):
                bad.append((name, org))
    for name, mod in (mods or {}).items():
        org = module_origin(mod)
        if not org.startswith(REF_ROOT + os.sep):
            bad.append((name, org))
    if bad:
        raise AssertionError('modules not loaded from %s: %r' % (REF_ROOT, bad))


def import_reference():
    """Returns a dict of the reference modules on the hot path (CPU-importable), each loaded by file location."""
    import numpy as np
    sys.dont_write_bytecode = True
    if not hasattr(np, 'int'):
        np.int = int  # noqa: NPY001 - reference uses the removed alias
    for name in ['cv2', 'torchvision', 'torchvision.utils', 'pyshtools', 'torch_cluster',
                 'torch_geometric', 'torch_geometric.nn', 'torch_geometric.utils',
                 'skimage', 'skimage.transform', 'skimage.io',
                 'tensorboardX', 'pytorch_msssim', 'trimesh']:
        if name not in sys.modules:
            _stub(name)
    sys.modules['torch_cluster'].knn_graph = None
    _purge_shared()
    # the reference's `neural_renderer` is only stubbed when the caller did not import the real python package
    if 'neural_renderer' not in sys.modules:
        _stub('neural_renderer')
    # namespace packages of the reference, pinned to their directories
    if 'pytorch_prototyping' not in sys.modules:
        _seed_namespace_package('pytorch_prototyping', os.path.join(REF_ROOT, 'pytorch_prototyping'))
    mods = {}
    mods['pytorch_prototyping'] = sys.modules.get('pytorch_prototyping.pytorch_prototyping') or _load_by_path(
        'pytorch_prototyping.pytorch_prototyping', os.path.join(REF_ROOT, 'pytorch_prototyping', 'pytorch_prototyping.py'))
    sys.modules['pytorch_prototyping'].pytorch_prototyping = mods['pytorch_prototyping']
    if 'gcn_lib' not in sys.modules:
        # gcn_lib has regular __init__ files; resolve it by location as well so that sys.path order is irrelevant
        spec = importlib.util.spec_from_file_location(
            'gcn_lib', os.path.join(REF_ROOT, 'gcn_lib', '__init__.py'),
            submodule_search_locations=[os.path.join(REF_ROOT, 'gcn_lib')])
        pkg = importlib.util.module_from_spec(spec)
        sys.modules['gcn_lib'] = pkg
        spec.loader.exec_module(pkg)
    # order: leaves first — network.py imports all the others by bare name and finds them in sys.modules
    for name in ['misc', 'camera', 'sph_harm', 'data_util', 'render', 'network']:
        mods[name] = sys.modules.get(name) or _load_by_path(name, os.path.join(REF_ROOT, name + '.py'))
    assert_reference_modules(mods)
    return mods


def import_reference_module(name):
    """A further top-level reference module (e.g. `dataio`) by file location."""
    _purge_shared()
    mod = sys.modules.get(name) or _load_by_path(name, os.path.join(REF_ROOT, name + '.py'))
    assert_reference_modules({name: mod})
    return mod


def import_reference_nr():
    """The reference's own `neural_renderer` python package (its CUDA extension modules must already be in
    sys.modules as stand-ins: make_golden.import_all)."""
    _purge_shared()
    if 'neural_renderer' in sys.modules and module_origin(sys.modules['neural_renderer']).startswith(REF_NR_ROOT):
        return sys.modules['neural_renderer']
    pkg_dir = os.path.join(REF_NR_ROOT, 'neural_renderer')
    spec = importlib.util.spec_from_file_location('neural_renderer', os.path.join(pkg_dir, '__init__.py'),
                                                  submodule_search_locations=[pkg_dir])
    pkg = importlib.util.module_from_spec(spec)
    # keep pre-registered stand-ins for the CUDA extension sub-package
    sys.modules['neural_renderer'] = pkg
    spec.loader.exec_module(pkg)
    assert_reference_modules({'neural_renderer': pkg})
    return pkg


if __name__ == '__main__':
    m = import_reference()
    print({k: v.__file__ for k, v in m.items()})
