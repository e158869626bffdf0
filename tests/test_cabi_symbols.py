"""The C-ABI library builds, loads, and exports every symbol include/sglang_amd.h declares.
No compute calls: this runs without a GPU."""
import ctypes

from sglang_amd import build, native


def test_library_exports_all_declared_symbols():
    path = build.build()
    assert path.exists()
    cdll = native.lib()     # imports torch first so a single HIP runtime is shared (see native.lib)
    names = native.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(cdll, n), f"{n} declared in sglang_amd.h but not exported"


def test_identification_entry_points():
    lib = native.lib()
    assert lib.sgl_amd_abi_version() == 1
    assert lib.sgl_amd_target_arch() == b"gfx950"
    assert lib.sgl_amd_decode_attention_min_chunk() > 0


def test_argument_errors_are_reported_not_thrown():
    # bad hidden size is rejected on the host before any launch (no GPU needed)
    lib = native.lib()
    rc = lib.sgl_amd_rmsnorm(None, None, None, 1, 7, 8, 8, 1e-5, None)
    assert rc == -1
    assert b"hidden" in lib.sgl_amd_last_error()
