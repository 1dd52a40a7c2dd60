"""CPU-only checks of the C ABI: the shared library loads, exports every symbol include/laed_hip.h declares,
and validates arguments before touching a device (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "laed_hip.h")
SO = os.path.join(ROOT, "neuraldialog-laed_amd", "laed_amd", "liblaed_hip.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(laed_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 25
    lib = ctypes.CDLL(SO)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    from laed_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_symbols()
    assert _lib.lib.laed_version() >= 100


def test_binding_signatures_match_the_header():
    """Every ctypes signature in laed_amd/_lib.py has the header's parameter count, and each parameter's ctypes kind
    (pointer / int / int64 / uint64 / float / double) is the one the C declaration spells."""
    from laed_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    decls = dict(re.findall(r"\b(laed_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S))

    def kind(param):
        p = " ".join(param.split())
        if "*" in p:
            return ctypes.c_void_p
        if p.startswith("uint64_t"):
            return ctypes.c_uint64
        if p.startswith("int64_t"):
            return ctypes.c_int64
        if p.startswith("int "):
            return ctypes.c_int
        if p.startswith("float "):
            return ctypes.c_float
        if p.startswith("double "):
            return ctypes.c_double
        raise AssertionError("unrecognised parameter %r" % p)

    for name, (args, _res) in _lib._SIGS.items():
        params = [q for q in decls[name].split(",") if q.strip() and q.strip() != "void"]
        assert len(params) == len(args), (name, len(params), len(args))
        for i, (q, a) in enumerate(zip(params, args)):
            want = kind(q)
            if want is ctypes.c_void_p:          # POINTER(c_double) etc. are pointers too
                assert a is ctypes.c_void_p or a is ctypes.c_char_p or hasattr(a, "contents") or issubclass(a, ctypes._Pointer), (name, i)
            else:
                assert a is want, (name, i, q.strip(), a)


def test_argument_validation_without_a_device():
    from laed_amd import _lib
    lib = _lib.lib
    # negative dimension / null pointers / unsupported layout are refused before any launch
    assert lib.laed_sgemm(0, 1, -1, 4, 4, None, 4, None, 4, None, 4, None, 0, None, 0, None) == -1
    assert b"negative" in lib.laed_last_error()
    assert lib.laed_sgemm(0, 1, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, None, 0, None) == -1
    assert b"null" in lib.laed_last_error()
    assert lib.laed_gumbel_latent_fwd(None, None, 10, 65, 1.0, 0, None, None, None, None) == -1
    assert b"k <= 64" in lib.laed_last_error()
    assert lib.laed_gumbel_latent_fwd(None, None, 10, 10, 0.0, 0, None, None, None, None) == -1
    assert lib.laed_gru_fwd(None, None, None, None, None, None, 0, 1, 1, 1, None, None, None, None, 0, None) == -1
    assert lib.laed_adam_step(None, None, None, None, 10, 0, 1e-3, 0.9, 0.999, 1e-8, 1.0, None) == -1
    assert lib.laed_proj_nll_ldl(50000) == 50016 and lib.laed_proj_nll_ldl(10002) == 10016
    assert lib.laed_latent_stats_blocks() > 0
    # empty problems are no-ops
    assert lib.laed_sgemm(0, 1, 0, 4, 4, None, 4, None, 4, None, 4, None, 0, None, 0, None) == 0
    assert lib.laed_embed_gather(None, 0, None, 5, 4, None, 0.0, None, None) == 0


def test_wrappers_refuse_cpu_tensors():
    import pytest
    import torch
    from laed_amd import ops
    from laed_amd._lib import LaedError
    with pytest.raises(LaedError):
        ops.adam_step(torch.zeros(4), torch.zeros(4), torch.zeros(4), torch.zeros(4), 1, 1e-3)
