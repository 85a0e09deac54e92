"""CPU: the C-ABI library loads and exports every symbol include/stabnet_hip.h declares (no compute calls)."""
import ctypes
import os


def test_library_exports_every_declared_symbol():
    from stabnet_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = [s[0] for s in _lib.declared_symbols()]
    assert len(names) >= 7
    for n in names:
        assert hasattr(L, n), "libstabnet_hip.so does not export " + n
    assert _lib.lib().stabnet_abi_version() >= 1


def test_bad_argument_status_without_gpu():
    from stabnet_amd import _lib
    L = _lib.lib()
    assert L.stabnet_interp_fwd(0, 0, 0, 1, 4, 4, 1, 0, 0) == -1
    assert b"null" in L.stabnet_last_error()
    assert L.stabnet_warp_fwd(1, 1, 0, 4, 4, 1, 4, 4, 0.8, 1, 1, 1, 1, 1, 0, 0) == -1
