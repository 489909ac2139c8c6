"""CPU-side checks of the drop-in boundary: the shared library loads without a
GPU, exports every symbol include/moc_hip.h declares, and the ctypes mirrors of
the ABI structs have the layout a C compiler gives the header."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "moc_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(moc_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from moc_amd import _lib
    h = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(h, n), f"libmoc_hip.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), "binding table and header disagree"
    assert h.moc_version() == _lib.ABI_VERSION


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md section 4 maps each declared entry point to what it replaces in the reference."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in declared_symbols() if n not in doc and not n.startswith("moc_p2p_")]
    assert not missing, f"INTEGRATION.md does not mention {missing}"
    assert "moc_p2p_" in doc


def test_struct_layout_matches_c_compiler(tmp_path):
    from moc_amd import _lib
    prog = tmp_path / "layout.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "moc_hip.h"\n'
        "int main(void){\n"
        ' printf("%zu %zu %zu %zu %zu\\n", sizeof(moc_batch_t), offsetof(moc_batch_t, row_off_host), offsetof(moc_batch_t, x_off), offsetof(moc_batch_t, C), offsetof(moc_batch_t, cand));\n'
        ' printf("%zu %zu %zu %zu\\n", sizeof(moc_meta_t), offsetof(moc_meta_t, lr), offsetof(moc_meta_t, H), offsetof(moc_meta_t, step));\n'
        ' printf("%zu %zu\\n", sizeof(moc_meta_ws_t), offsetof(moc_meta_ws_t, tile_ws_bytes));\n'
        " return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(prog), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    got = [int(v) for v in out]
    B, M, W = _lib.MocBatch, _lib.MocMeta, _lib.MocMetaWs
    exp = [ctypes.sizeof(B), B.row_off_host.offset, B.x_off.offset, B.C.offset, B.cand.offset,
           ctypes.sizeof(M), M.lr.offset, M.H.offset, M.step.offset,
           ctypes.sizeof(W), W.tile_ws_bytes.offset]
    assert got == exp


def test_host_side_argument_checks_need_no_gpu():
    """Contract violations are reported before anything is launched."""
    from moc_amd import _lib
    h = _lib.lib()
    assert h.moc_bank_bytes(512, 6, _lib.MOC_BF16) == 16 * 3 * 1024
    assert h.moc_bank_bytes(512, 34, _lib.MOC_F32) == 3 * 32 * 1024
    rc = h.moc_prepare_bank(None, None, 512, 2, 6, _lib.MOC_BF16, 0, None, None)
    assert rc == 1 and b"null pointer" in h.moc_last_error()
    b = _lib.MocBatch()
    rc = h.moc_scores(ctypes.byref(b), None, None)
    assert rc == 1 and b"null X/row_off" in h.moc_last_error()
    with pytest.raises(AssertionError):
        _lib.check(rc, "moc_scores")


def test_no_product_module_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under moc_amd/ may import or call it."""
    pat = re.compile(r"(import\s+oracle|from\s+oracle|moc_oracle|oracle[/.])")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "moc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f"{f} references the oracle"


def test_missing_library_fails_loudly(monkeypatch):
    from moc_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmoc_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_cpu_tensors_are_refused():
    import torch
    from moc_amd.patch_selection_classifier import topj_pooling
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        topj_pooling(torch.randn(10, 2), [3])
