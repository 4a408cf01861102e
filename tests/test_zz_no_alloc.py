"""pdekan_layer_forward_noalloc (the entry point that needs no workspace and allocates nothing) against
pdekan_layer_forward on the same inputs, through the C ABI.  Collected last on purpose: it is the newest
entry point and a failure here must not hide the rest of the suite."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer(I, O, G, seed):
    from pde_amd.kan import KANLinear
    torch.manual_seed(seed)
    lay = KANLinear(I, O, grid_size=G).cuda()
    with torch.no_grad():
        for p in lay.parameters():
            p.add_(0.1 * torch.randn_like(p))
    return lay


def _both(lay, xj, n_dir, n_second):
    from pde_amd import _lib, functional as K
    lib = _lib.load()
    grid, bw, sw, sc = lay._kernel_params()
    st = K.layer_struct(grid, bw, sw, sc, K.ACT_SILU, lay.grid_size, lay.spline_order)
    n = xj.size(0)
    y_ws = torch.empty(n, lay.out_features, xj.size(2), device="cuda")
    y_na = torch.full_like(y_ws, float("nan"))
    ws = torch.empty(lib.pdekan_layer_workspace_bytes(ctypes.byref(st), n, n_dir, n_second), dtype=torch.uint8,
                     device="cuda")
    stream = K._stream(xj.device)
    _lib.check(lib.pdekan_layer_forward(ctypes.byref(st), K._ptr(xj), K._ptr(y_ws), n, n_dir, n_second, K._ptr(ws),
                                        ws.numel(), stream))
    _lib.check(lib.pdekan_layer_forward_noalloc(ctypes.byref(st), K._ptr(xj), K._ptr(y_na), n, n_dir, n_second, stream))
    torch.cuda.synchronize()
    return y_ws, y_na


def test_noalloc_forward_equals_workspace_forward_twice_in_a_row():
    """Two calls in a row on the shared built-in scratch, the second on a smaller layer and fewer points: what the
    first call left in the scratch must not leak into the second.  Same kernels, same inputs: the same bits."""
    gen = torch.Generator().manual_seed(0)
    cases = [(64, 64, 10, 4099, 2, 1), (8, 5, 5, 301, 2, 1), (64, 128, 10, 1000, 0, 0), (2, 64, 10, 70, 2, 1)]
    for I, O, G, n, nd, ns in cases:
        lay = _layer(I, O, G, I + O)
        J = 1 + nd + ns
        xj = torch.randn(n, I, J, generator=gen)
        xj[:, :, 0] = torch.rand(n, I, generator=gen) * 2.8 - 1.4
        y_ws, y_na = _both(lay, xj.cuda(), nd, ns)
        assert torch.isfinite(y_na).all()
        assert torch.equal(y_ws, y_na), (I, O, G, n)


def test_noalloc_forward_allocates_nothing_and_refuses_oversized_layers():
    from pde_amd import _lib, functional as K
    lib = _lib.load()
    lay = _layer(64, 64, 10, 1)
    xj = torch.randn(512, 64, 4, device="cuda")
    _both(lay, xj, 2, 1)                                   # first call on the device: address lookup
    grid, bw, sw, sc = lay._kernel_params()
    st = K.layer_struct(grid, bw, sw, sc, K.ACT_SILU, lay.grid_size, lay.spline_order)
    y = torch.empty(512, 64, 4, device="cuda")
    for _ in range(3):
        free_before, _ = torch.cuda.mem_get_info()
        _lib.check(lib.pdekan_layer_forward_noalloc(ctypes.byref(st), K._ptr(xj), K._ptr(y), 512, 2, 1,
                                                    K._stream(xj.device)))
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after == free_before                   # the driver handed out no device memory
    big = _layer(256, 256, 20, 2)
    grid, bw, sw, sc = big._kernel_params()
    st = K.layer_struct(grid, bw, sw, sc, K.ACT_SILU, big.grid_size, big.spline_order)
    xb = torch.randn(16, 256, 4, device="cuda")
    yb = torch.empty(16, 256, 4, device="cuda")
    rc = lib.pdekan_layer_forward_noalloc(ctypes.byref(st), K._ptr(xb), K._ptr(yb), 16, 2, 1, K._stream(xb.device))
    assert rc == -2 and b"scratch" in lib.pdekan_last_error()


def test_noalloc_forward_under_graph_capture():
    """What the entry point is for: captured into a HIP graph (where hipMalloc is illegal) and replayed."""
    from pde_amd import _lib, functional as K
    lib = _lib.load()
    lay = _layer(64, 64, 10, 3)
    grid, bw, sw, sc = lay._kernel_params()
    st = K.layer_struct(grid, bw, sw, sc, K.ACT_SILU, lay.grid_size, lay.spline_order)
    xs = torch.randn(2048, 64, 4, device="cuda")
    y = torch.empty(2048, 64, 4, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                          # first call (address lookup) outside capture
        _lib.check(lib.pdekan_layer_forward_noalloc(ctypes.byref(st), K._ptr(xs), K._ptr(y), 2048, 2, 1,
                                                    K._stream(xs.device)))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _lib.check(lib.pdekan_layer_forward_noalloc(ctypes.byref(st), K._ptr(xs), K._ptr(y), 2048, 2, 1,
                                                    K._stream(xs.device)))
    for seed in (1, 2):
        xs.copy_(torch.randn(2048, 64, 4, generator=torch.Generator().manual_seed(seed)))
        g.replay()
        torch.cuda.synchronize()
        y_ws, y_na = _both(lay, xs, 2, 1)
        assert torch.equal(y, y_ws) and torch.equal(y_na, y_ws)
