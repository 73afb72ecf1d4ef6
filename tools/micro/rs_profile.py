"""Where the main thread's time goes in a pipelined RegionSelection round without a backbone (cProfile sees the calling thread only)."""
import cProfile, os, pstats, shutil, sys, tempfile, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import halo_amd; halo_amd.configure(hw_queues=2)
from halo_amd.core.active.build import RegionSelection
from halo_amd.core.utils.hyperbolic import HyperMapper

dev = torch.device("cuda:0")
N, H, W, C, O = 192, 1024, 2048, 64, 19
cfg = types.SimpleNamespace(
    MODEL=types.SimpleNamespace(NUM_CLASSES=O, HYPER=True, CURVATURE=1.0),
    ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                 BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))
g = torch.Generator(device=dev).manual_seed(0)
emb = HyperMapper(1.0).expmap(torch.randn((1, C, 160, 320), generator=g, device=dev) * 0.1, dim=1)
logit = torch.nn.functional.interpolate(torch.randn((1, O, 160, 320), generator=g, device=dev), size=(640, 1280), mode="bilinear", align_corners=True)


class Ident(torch.nn.Module):
    def forward(self, x):
        return x


class Head(torch.nn.Module):
    def forward(self, x, size=None):
        return logit, emb


tmp = tempfile.mkdtemp(prefix="halo_rs_p_")
gt = torch.randint(0, O, (1, H, W)).pin_memory()
items = [{"img": torch.zeros(1, 3, 8, 8), "path_to_mask": [os.path.join(tmp, f"m{i}.png")],
          "origin_mask": torch.full((1, H, W), 255, dtype=torch.int64).pin_memory(), "origin_label": gt, "size": torch.tensor([[H, W]]),
          "active": torch.zeros(1, H, W, dtype=torch.bool).pin_memory(), "selected": torch.zeros(1, H, W, dtype=torch.bool).pin_memory(),
          "path_to_indicator": [os.path.join(tmp, f"i{i}.pth")], "name": [f"img{i}"]} for i in range(N)]
RegionSelection(cfg, Ident(), Head(), items[:16], 1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
RegionSelection(cfg, Ident(), Head(), items, 1)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
shutil.rmtree(tmp, ignore_errors=True)
