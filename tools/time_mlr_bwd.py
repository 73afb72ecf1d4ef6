"""Timing aid: the fused HyperMLR backward alone (halo_hypermlr_backward: prep + pixels + weights + final), HIP events around
REP calls, at the training shape (2 x 64 x 160 x 320) and the v2 head's (1 x 64 x 640 x 1280)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)
from halo_amd import _lib
from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR

dev = torch.device("cuda:0")
L = _lib.lib()
for (B, C, O, h, w) in ((2, 64, 19, 160, 320), (1, 64, 19, 640, 1280)):
    x = HyperMapper(1.0).expmap(torch.randn((B, C, h, w), device=dev) * 0.1, dim=1).double()
    mlr = HyperMLR(C, O).to(dev)
    P, A = mlr.P_MLR.detach(), mlr.A_MLR.detach()
    gout = torch.randn((B, O, h, w), device=dev, dtype=torch.float64)
    gx = torch.empty_like(x); gP = torch.empty_like(P); gA = torch.empty_like(A)
    n = L.halo_hypermlr_backward_workspace_bytes(B, C, O, h * w)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)

    def call():
        _lib.check(L.halo_hypermlr_backward(_lib.ptr(x), _lib.ptr(P), _lib.ptr(A), _lib.ptr(gout), _lib.dtype_code(gout), B, C, O, h * w, 1.0, _lib.ptr(gx), _lib.ptr(gP),
                                            _lib.ptr(gA), _lib.ptr(ws), n, _lib.stream_ptr(dev)), "halo_hypermlr_backward")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    REP = 20
    e0.record()
    for _ in range(REP):
        call()
    e1.record(); torch.cuda.synchronize()
    by = (2 * x.numel() + gout.numel()) * 8
    ms = e0.elapsed_time(e1) / REP
    print(f"halo_hypermlr_backward {B}x{C}x{h}x{w}, {O} classes: {ms * 1e3:7.1f} us   ({by / ms / 1e6:.0f} GB/s of x + gout read, gx written)", flush=True)
