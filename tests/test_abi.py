"""CPU: the C-ABI library builds, loads, and exports every symbol include/sf_hip.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'sf_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sf_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_loads():
    import __graft_entry__
    __graft_entry__.build()
    from speaker_follower_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    assert _lib.lib.sf_abi_version() == 9
    from speaker_follower_amd import build
    assert _lib.lib.sf_build_id().decode() == build.build_id() == build.lib_build_id()
    assert _lib.lib.sf_workspace_bytes() >= 32 << 20
    assert _lib.lib.sf_status_string(0) == b'ok'
    assert _lib.lib.sf_status_string(4) == b'workspace too small'


def test_every_header_symbol_is_exported_and_bound():
    import ctypes
    from speaker_follower_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), 'libsf_hip.so does not export %s' % s
    assert set(syms) == set(_lib.EXPORTS), set(syms) ^ set(_lib.EXPORTS)


def test_argument_validation_without_gpu():
    """Entry points reject null pointers / bad sizes before touching the device."""
    from speaker_follower_amd import _lib
    assert _lib.lib.sf_linear_fwd(None, 4, None, None, 1, 1, 4, 0, None, 1, None, 0, None) == 1
    assert _lib.lib.sf_reduce_terms(None, None, 0, 0, None, None) == 1


def test_product_fails_loudly_on_cpu_tensors():
    import pytest
    import torch
    from speaker_follower_amd import model
    m = model.SoftDotAttention(16)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(2, 16), torch.zeros(2, 3, 16))
    dec = model.AttnDecoderLSTM(24, 16, 0.5, feature_size=24)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dec(torch.zeros(2, 24), torch.zeros(2, 3, 24), torch.zeros(2, 5, 24), torch.zeros(2, 16),
            torch.zeros(2, 16), torch.zeros(2, 4, 16))
