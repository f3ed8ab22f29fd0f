"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every symbol that
include/ssd_hip.h declares (no compute without a GPU)."""
import os
import re

from tests.conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ssd_hip.h")).read()
    return sorted(set(re.findall(r"^int\s+(ssd_\w+)\s*\(", txt, flags=re.M)))


def test_library_exports_header_symbols():
    from ssd_amd.hip.lib import build_library, load_library, SIGNATURES, lib_path
    if not os.path.exists(lib_path()):
        build_library()
    lib = load_library()
    syms = header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in ssd_hip.h but not exported"
        assert s in SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(SIGNATURES) == syms
    assert lib.ssd_abi_version() == 1


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under ssd_amd/ (or the ssd alias) may import it."""
    bad = []
    for base in ("ssd_amd", "ssd"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_integration_doc_covers_every_symbol():
    """INTEGRATION.md shows the reference-side binding (or states the role) of every entry point the header declares."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [s for s in header_symbols() if s not in doc]
    assert not missing, missing
