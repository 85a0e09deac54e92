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


def test_plan_level_host_logic_without_gpu():
    """Plans are host objects: launch counts, workspace sizes and the training debug offsets need no GPU.  The 720p deploy frame
    is 64 launches with the shipped split-K table: 49 conv launches (incl. the merged shortcut|conv1 ones; three-slice 3x3 layers
    and two-slice prologue layers on the ring kernel reduce inside their workgroups) + 7 split-K reduce launches + stack assembly,
    pool, gap partials + fc_1, fc_2, fc_3, output layer + mesh, sampler + push -- bench.py reports 65: + the copy of the frame into
    the captured graph's input."""
    import ctypes as C
    from stabnet_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    assert L.stabnet_net_create(C.byref(h), 1, 720, 1280, 13, 50, 0) == 0
    n_plan = L.stabnet_net_num_launches(h)
    n_frame = L.stabnet_deploy_frame_launches(h, 4, 4)
    assert n_frame == n_plan + 1                         # stack assembly for the pad step, + sampler; the mesh rides with the output layer
    assert n_frame == 64, n_frame                        # (follows the measured split-K table; update the text above with it)
    assert L.stabnet_net_workspace_bytes(h) > 100 << 20 and abs(L.stabnet_net_flops(h) / 142.94e9 - 1) < 1e-3
    off, cnt = C.c_long(), C.c_long()
    assert L.stabnet_net_train_debug_offset(h, b"argmax", C.byref(off), C.byref(cnt)) == -1      # not a keep_activations plan
    L.stabnet_net_destroy(h)
    t = C.c_void_p()
    assert L.stabnet_net_create(C.byref(t), 2, 64, 96, 13, 50, 1) == 0
    assert L.stabnet_net_train_debug_offset(t, b"argmax", C.byref(off), C.byref(cnt)) == 0 and cnt.value == 2 * 16 * 24 * 64
    assert L.stabnet_net_train_debug_offset(t, b"fcx1", C.byref(off), C.byref(cnt)) == 0 and cnt.value == 2 * 2 * 2048
    assert L.stabnet_net_train_debug_offset(t, b"bn:0", C.byref(off), C.byref(cnt)) == 0 and cnt.value == 2 * 16 * 24 * 64
    assert L.stabnet_net_train_debug_offset(t, b"nonsense", C.byref(off), C.byref(cnt)) == -1
    L.stabnet_net_destroy(t)
