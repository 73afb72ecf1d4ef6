"""Multi-process (gloo, world_size 2, CPU) tests of the sharding + all-gather step.

The scorer is injected: here it is the CPU oracle standing in for the HIP path (tests may call
the oracle), so what is under test is the host logic of halo_amd/pool.py.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_shard_ranges_cover_the_pool_in_order():
    from halo_amd.pool import regions_per_image, shard_range
    for n, world in ((2975, 8), (500, 8), (7, 2), (3, 4), (0, 2)):
        blocks = [shard_range(n, r, world) for r in range(world)]
        flat = [i for lo, hi in blocks for i in range(lo, hi)]
        assert flat == list(range(n))
    assert shard_range(2975, 0, 8) == (0, 372) and shard_range(2975, 7, 8) == (2604, 2975)
    assert regions_per_image(1024, 2048, 0.05, 5, 1) == 2331          # build.py:148-150
    assert regions_per_image(256, 512, 0.05, 5, 1) == 146


def _make_images(n, H=40, W=64, C=6, O=19):
    from oracle import halo_oracle as ho
    out = []
    for i in range(n):
        rng = np.random.default_rng(100 + i)
        z = (rng.standard_normal((1, C, H // 4, W // 4)) * 0.1).astype(np.float32)
        emb = ho.bilinear(ho.expmap(z, 1.0, dim=1), (H, W))
        logit = ho.bilinear(rng.standard_normal((1, O, H // 4, W // 4)).astype(np.float32), (H, W))
        gt = rng.integers(0, O, (H, W)).astype(np.int64)
        out.append((logit, emb, gt))
    return out


def _oracle_acquire(n_regions):
    from oracle import halo_oracle as ho

    def fn(batch):
        picks = torch.zeros((len(batch), n_regions, 3), dtype=torch.float64)
        npk = torch.zeros((len(batch),), dtype=torch.int32)
        for j, (logit, emb, gt) in enumerate(batch):
            s, _, _ = ho.floating_region_score(logit, emb, "entropy", "radius", True, gt, size=3, purity_type="radius")
            H, W = gt.shape
            act = np.zeros((H, W), bool); sel = np.zeros((H, W), bool); am = np.full((H, W), 255, np.int64)
            _, _, _, _, p = ho.select_pixels_to_label(s, n_regions, 1, 5, act, sel, am, gt, True)
            picks[j, :len(p)] = torch.from_numpy(p)
            npk[j] = len(p)
        return picks, npk
    return fn


def _worker(rank, world, port, n_images, n_regions, outdir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from halo_amd.pool import acquire_pool
    images = _make_images(n_images)
    tables, counts, owner, keep = acquire_pool(images, _oracle_acquire(n_regions), n_regions, global_budget=9)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), tables=tables.numpy(), counts=counts.numpy(),
             owner=owner.numpy(), keep=keep.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_images", [5, 4])
def test_two_ranks_equal_one_process(tmp_path, n_images):
    """Sharded over 2 ranks (uneven shards for 5 images) the gathered tables are bit-identical to
    the single-process tables, on both ranks -- the result must not depend on the world size."""
    from halo_amd.pool import acquire_pool, global_budget_select
    n_regions = 6
    mp.spawn(_worker, args=(2, _free_port(), n_images, n_regions, str(tmp_path)), nprocs=2, join=True)
    images = _make_images(n_images)
    ref_t, ref_c, ref_o, _ = acquire_pool(images, _oracle_acquire(n_regions), n_regions)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for r in (r0, r1):
        assert np.array_equal(r["tables"], ref_t.numpy())
        assert np.array_equal(r["counts"], ref_c.numpy())
    assert np.array_equal(r0["owner"], r1["owner"])
    assert list(r0["owner"]) == [0] * ((n_images + 1) // 2) + [1] * (n_images - (n_images + 1) // 2)
    # optional global-budget mode: same mask on both ranks, exactly 9 picks, all from the top scores
    assert np.array_equal(r0["keep"], r1["keep"]) and int(r0["keep"].sum()) == 9
    keep = global_budget_select(ref_t, ref_c, 9).numpy()
    assert np.array_equal(keep, r0["keep"])
    kept = ref_t.numpy()[:, :, 2][keep]
    assert kept.min() >= np.sort(ref_t.numpy()[:, :, 2].reshape(-1))[-9]
