"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly the entry points ``include/otvae.h`` declares
(no compute calls: there is no GPU here).  Also checks the product never imports the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _header_functions():
    hdr = open(os.path.join(ROOT, "include", "otvae.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(otvae_[a-z0-9_]+)\s*\(", hdr))


@pytest.fixture(scope="module")
def lib_path():
    from ot_vae_lightning_amd import build
    return build.build(verbose=False)


def test_header_matches_python_binding():
    from ot_vae_lightning_amd import _lib
    assert _header_functions() == set(_lib.SIGNATURES)


def _header_prototypes():
    """name -> list of parameter declarations, parsed from include/otvae.h"""
    src = open(os.path.join(ROOT, "include", "otvae.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(otvae_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        params = [p.strip() for p in m.group(2).replace("\n", " ").split(",")]
        protos[m.group(1)] = [] if params in ([""], ["void"]) else params
    return protos


def test_python_binding_matches_header_prototypes():
    """Arity and the pointer / integer / float class of every parameter: a mismatch here is a host-side segfault (or a
    silently shifted argument list) at the first call."""
    import ctypes as C
    from ot_vae_lightning_amd import _lib
    protos = _header_prototypes()
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        params = protos[name]
        assert len(params) == len(argtypes), f"{name}: header has {len(params)} parameters, the binding {len(argtypes)}"
        for i, (decl, at) in enumerate(zip(params, argtypes)):
            is_ptr = "*" in decl
            at_ptr = at in (C.c_void_p, C.c_char_p) or hasattr(at, "contents") or getattr(at, "_type_", None) == "P"
            assert is_ptr == bool(at_ptr), f"{name} parameter {i} ({decl!r}): pointer-ness differs from {at}"
            if not is_ptr:
                is_float = re.match(r"(const\s+)?(float|double)\b", decl) is not None
                assert is_float == (at in (C.c_float, C.c_double)), f"{name} parameter {i} ({decl!r}) vs {at}"
                if is_float:
                    assert (decl.split()[-2] == "double") == (at is C.c_double), f"{name} parameter {i} ({decl!r}) vs {at}"


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in sorted(_header_functions()):
        assert hasattr(lib, name), f"{name} declared in include/otvae.h but not exported"
    lib.otvae_abi_version.restype = ctypes.c_int
    assert lib.otvae_abi_version() == 1


def test_library_contains_gfx950_code_object(lib_path):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", lib_path], capture_output=True, text=True)
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded"
    assert b"v_mfma" in blob or b"mfma" in blob or True  # ISA mnemonics are not stored in the binary; see test below


def test_conv_kernels_use_fp32_mfma():
    """Device assembly of the conv kernels: the implicit GEMMs must issue v_mfma_f32_16x16x4_f32 and must not spill."""
    src = os.path.join(ROOT, "ot_vae_lightning_amd", "csrc", "conv.hip")
    r = subprocess.run(["hipcc", "-x", "hip", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                        "-o", "-", src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("v_mfma_f32_16x16x4_f32") >= 30, "conv kernels lost their MFMA instructions"
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", r.stdout)]
    assert spills and max(spills) == 0, f"register spills in conv kernels: {spills}"


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ot_vae_lightning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "otvae_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ot_vae_lightning_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.load()


def test_cpu_tensors_are_refused():
    import torch
    import ot_vae_lightning_amd as A
    with pytest.raises(RuntimeError):
        A.sinkhorn_log(torch.ones(3) / 3, torch.ones(3) / 3, torch.rand(3, 3))
    layer = A.ConvLayer(2, 4, normalization="batchnorm", activation="relu")
    with pytest.raises(RuntimeError):
        layer(torch.zeros(1, 2, 4, 4))


def test_host_pointers_never_reach_a_kernel():
    """The last gate before every kernel call (``_lib.ptr`` / ``ptr_array``) refuses host tensors: a module left on the CPU whose
    buffers meet GPU samples (``GaussianTransport(...)`` without ``.cuda()``) must raise, not fault the GPU."""
    import torch
    from ot_vae_lightning_amd import _lib
    assert _lib.ptr(None) is None
    with pytest.raises(RuntimeError, match="MI355X kernel"):
        _lib.ptr(torch.zeros(3))
    with pytest.raises(RuntimeError, match="MI355X kernel"):
        _lib.ptr_array([None, torch.zeros(2, dtype=torch.float64)])
    assert _lib.ptr_array([None, None])[0] is None
