"""GPU: max-inscribed-rectangle crop search (deploy_bundle.py:344-366) -- exact integer result vs the oracle's literal
restatement, incl. the reference's first-found-wins tie-breaking."""
import numpy as np
import pytest
import torch

from oracle import stabnet_oracle as O

pytestmark = pytest.mark.gpu


def _mask(rng, H, W, kind):
    m = np.zeros((H, W), np.int32)
    if kind == "border":                       # the typical case: black bands along the frame border
        t, b, l, r = rng.integers(0, H // 6), rng.integers(0, H // 6), rng.integers(0, W // 6), rng.integers(0, W // 6)
        m[:t] = 3; m[H - b:] = 1; m[:, :l] = 2; m[:, W - r:] = 5
    elif kind == "speckle":
        m[rng.random((H, W)) < 0.002] = 1
    elif kind == "ties":                       # symmetric holes -> several rectangles of equal area
        m[H // 2, :] = 1; m[:, W // 2] = 1
    elif kind == "full":
        m[:] = 1
    return m


@pytest.mark.parametrize("H,W", [(288, 512), (97, 131)])
@pytest.mark.parametrize("kind", ["border", "speckle", "ties", "empty", "full"])
def test_crop_search_matches_reference_loop(cuda, H, W, kind):
    from stabnet_amd import warp
    rng = np.random.default_rng(H + len(kind))
    m = _mask(rng, H, W, kind)
    want, area = O.max_inscribed_rect(m)
    got, garea = warp.max_inscribed_rect(torch.from_numpy(m).to(cuda))
    assert (got, garea) == (want, area)


def test_black_accumulate(cuda):
    from stabnet_amd import warp
    rng = np.random.default_rng(0)
    acc = torch.zeros((4, 9), dtype=torch.int32, device=cuda)
    tot = np.zeros((4, 9), np.int64)
    for _ in range(5):
        b = (rng.random((4, 9)) < 0.3).astype(np.float32)
        warp.black_accumulate(torch.from_numpy(b).to(cuda), acc)
        tot += np.round(b).astype(np.int64)
    assert np.array_equal(acc.cpu().numpy(), tot)


@pytest.mark.parametrize("H,W", [(97, 131), (288, 512), (720, 1280), (64, 65)])
def test_integral_image_exact(cuda, H, W):
    """The two scan kernels behind the search (one wave per row; 16 columns x 16 row groups per workgroup): the workspace starts
    with S[(H+1) x (W+1)] int64, S[y+1][x+1] = sum of all_black[:y+1, :x+1] -- integers, so exact."""
    from stabnet_amd import _lib
    from stabnet_amd._tensor import ptr, stream_ptr
    rng = np.random.default_rng(W)
    m = rng.integers(0, 1000, (H, W)).astype(np.int32)
    m[rng.random((H, W)) < 0.5] = 0
    a = torch.from_numpy(m).to(cuda)
    nbytes = _lib.lib().stabnet_crop_search_workspace_bytes(H, W, 10)
    ws = torch.full(((nbytes + 7) // 8,), -1, dtype=torch.int64, device=cuda)
    ans = torch.empty(5, dtype=torch.int32, device=cuda)
    _lib.call("stabnet_crop_search", ptr(a), H, W, 10, ptr(ans), ptr(ws), ws.numel() * 8, stream_ptr(a.device), device=a.device)
    S = ws[: (H + 1) * (W + 1)].cpu().numpy().reshape(H + 1, W + 1)
    want = np.zeros((H + 1, W + 1), np.int64)
    want[1:, 1:] = m.astype(np.int64).cumsum(0).cumsum(1)
    assert np.array_equal(S, want)
