"""The C-ABI shared library builds, loads and exports every symbol include/rdc_assembly.h declares
(no compute calls: those need a GPU and live in the -m gpu tests)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib_path():
    from rdcfes_amd import build
    return build.build(verbose=False)


def _declared():
    text = (ROOT / "include" / "rdc_assembly.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rdc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("rdc_ctx_create", "rdc_mesh_upload", "rdc_assemble_pihna", "rdc_assemble_ripf", "rdc_assemble_hcc",
                 "rdc_solid_assemble", "rdc_csr_download", "rdc_field_bind_device"):
        assert must in names


def test_library_exports_every_declared_symbol(lib_path):
    lib = C.CDLL(str(lib_path))
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_python_binding_table_matches_header(lib_path):
    from rdcfes_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    _lib.load()


def test_abi_version_and_struct_sizes(lib_path):
    from rdcfes_amd import _lib, HccParams, PihnaParams, RipfParams, SolidMaterial, SolidParams
    assert _lib.load().rdc_abi_version() == _lib.header_abi_version() == 3
    # POD layouts: all doubles (+ one padded int pair)
    assert C.sizeof(PihnaParams) == 23 * 8
    assert C.sizeof(RipfParams) == 28 * 8 + 8
    assert C.sizeof(HccParams) == 11 * 8
    assert C.sizeof(SolidMaterial) == 6 * 8
    assert C.sizeof(SolidParams) == 3 * 8


def test_no_cpu_fallback_without_device(lib_path):
    """On a machine without a GPU context creation must fail loudly, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rdcfes_amd import AssemblyContext, RdcError
    with pytest.raises(RdcError) as ei:
        AssemblyContext(0)
    assert ei.value.code == 2 and "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
    for p in list((ROOT / "rdcfes_amd").rglob("*.py")) + list((ROOT / "rdcfes_amd" / "csrc").glob("*")) + \
            list((ROOT / "include").glob("*")):
        if p.is_file():
            t = p.read_text(errors="ignore")
            assert "rdc_oracle" not in t and "from oracle" not in t and "import oracle" not in t, p
