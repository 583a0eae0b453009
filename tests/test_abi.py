"""The C-ABI shared library loads without a GPU and exports every symbol include/gru4rec_hip.h declares.
No compute entry point is called here."""
import ctypes
import os
import re

from gru4rec_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'gru4rec_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(g4r_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_native.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _native.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.g4r_version().startswith(b'gru4rec_hip')


def test_config_struct_size_is_stable():
    assert ctypes.sizeof(_native.G4RConfig) == _native.lib().g4r_sizeof_config() == 184


def test_create_without_gpu_fails_loudly():
    if _native.device_count() > 0:
        return
    import pytest
    with pytest.raises(_native.NativeError):
        _native.Model(n_items=10, layers=[8], batch_size=4, n_sample=0, loss=1, final_act=0, hidden_act=2,
                      embed_mode=0, learning_rate=0.1, sample_store=0, seed=1, device=0, rank=0, nranks=1)
