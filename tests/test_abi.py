"""CPU: the C-ABI library loads and exports every symbol include/fastmot_hip.h declares."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / 'include' / 'fastmot_hip.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(fm_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_symbols():
    syms = declared_symbols()
    assert 'fm_ctx_create' in syms and 'fm_assoc_stage' in syms and len(syms) >= 25


def test_library_exports_every_declared_symbol():
    from fastmot_amd import _lib
    lib = _lib.load()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f'missing exports: {missing}'


def test_diagnostics_are_not_in_the_shipped_library():
    """include/fastmot_hip_diag.h (round 3's bisect apparatus) exists only in -DFM_DIAG builds."""
    from fastmot_amd import _lib, build
    if '-DFM_DIAG' in build.FLAGS:
        return
    lib = _lib.load()
    text = (ROOT / 'include' / 'fastmot_hip_diag.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    diag = sorted(set(re.findall(r'\b(fm_[a-z0-9_]+)\s*\(', text)))
    assert len(diag) == 4
    assert not [s for s in diag if hasattr(lib, s)]


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    from fastmot_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', tmp_path / 'nope.so')
    try:
        _lib.load()
    except RuntimeError as err:
        assert 'no CPU fallback' in str(err)
    else:
        raise AssertionError('load() must fail loudly')


def test_error_string_is_a_c_string():
    from fastmot_amd import _lib
    lib = _lib.load()
    assert isinstance(lib.fm_last_error(), bytes)


def test_layer_table_builder_agrees_with_the_kernels():
    """graph.py decides per layer whether a fused kernel applies; its predicates must be the kernels' own
    (LDS budget of the OSNet chain kernel, channel sets of the fused residual unit)."""
    from fastmot_amd import _lib
    from fastmot_amd.models.graph import Graph
    lib = _lib.load()
    lib.fm_litechain_lds_bytes.restype = ctypes.c_size_t
    for c in (8, 16, 24, 32, 48, 64, 96, 128):
        for h, w in ((64, 32), (32, 16), (16, 8), (9, 40), (128, 64)):
            assert Graph.lightchain_fits(c, h, w) == (lib.fm_litechain_lds_bytes(c, w, h) <= 64 * 1024), (c, h, w)
    assert not Graph.lightchain_fits(20, 64, 32) and not Graph.lightchain_fits(136, 64, 32)
    for c in (32, 64, 96, 128, 256, 512):
        for mid in (16, 32, 64, 128, 256, 512):
            assert Graph.resblock_supported(c, mid) == bool(lib.fm_resblock_supported(c, mid)), (c, mid)
