"""CPU: the C-ABI library loads and exports every symbol include/rnr_hip.h declares (no kernel launches)."""
import os
import re

import pytest


def test_header_symbols_exported():
    from rnr_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, 'include', 'rnr_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(rnr_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), 'librnr_hip.so does not export %s' % name
        assert name in _lib.SIGNATURES, 'python binding lacks a signature for %s' % name
    assert set(_lib.SIGNATURES) <= declared, set(_lib.SIGNATURES) - declared
    assert lib.rnr_abi_version() == 1


def test_product_does_not_import_oracle():
    """The product package must never route through the oracle (or any CPU fallback)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, 'relightable-nr_amd')
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith('.py'):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(dp, fn)


def test_missing_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from rnr_amd import ops
    with pytest.raises(RuntimeError):
        ops.sh_basis(torch.zeros(4, 3), 2)          # CPU tensor -> loud error, not a CPU fallback
