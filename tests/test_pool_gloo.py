"""Multi-process (gloo, world_size 2 / 4 / 8, CPU) tests of the sharding + all-gather step.

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


def _synthetic_rows(lo, hi, n):
    """Deterministic pick table of pool images lo..hi-1 (a function of the GLOBAL image index only): what any rank would have
    produced for them.  Counts vary per image (0..n), scores include -0.0 / inf / NaN bit patterns."""
    g = torch.arange(lo, hi, dtype=torch.float64)
    k = torch.arange(n, dtype=torch.float64)
    picks = torch.zeros((hi - lo, n, 3), dtype=torch.float64)
    picks[:, :, 0] = (g[:, None] * 7 + k[None, :] * 13) % 65536
    picks[:, :, 1] = (g[:, None] * 31 + k[None, :] * 3) % 65536
    picks[:, :, 2] = torch.sin(g[:, None] * 0.37 + k[None, :]) * 1e3
    if hi > lo:
        picks[0, 0, 2] = -0.0
        picks[-1, -1, 2] = float("nan")
        picks[(hi - lo) // 2, 0, 2] = float("inf")
    npk = ((torch.arange(lo, hi) * 5) % (n + 1)).to(torch.int32)
    return picks, npk


def _gather_worker(rank, world, port, n_images, n, outdir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from halo_amd.pool import gather_tables, shard_range
    lo, hi = shard_range(n_images, rank, world)
    picks, npk = _synthetic_rows(lo, hi, n)
    tables, counts, owner = gather_tables(picks, npk, n_images)
    want_t, want_c = _synthetic_rows(0, n_images, n)
    # the block-local special values sit at each block's own first / middle / last row: rebuild the expectation block by block
    want_t = torch.cat([_synthetic_rows(*shard_range(n_images, r, world), n)[0] for r in range(world)])
    ok = (np.array_equal(tables.numpy().view(np.int64), want_t.numpy().view(np.int64))
          and torch.equal(counts, want_c) and tables.shape == (n_images, n, 3)
          and list(owner) == [r for r in range(world) for _ in range(*shard_range(n_images, r, world))])
    open(os.path.join(outdir, "rank%d.%s" % (rank, "ok" if ok else "bad")), "w").write("%d %d" % (lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_images", [(2, 37), (4, 37), (8, 37), (8, 2975), (4, 2975), (8, 5), (8, 8), (4, 1)])
def test_gather_tables_at_world_2_4_8_with_uneven_and_empty_blocks(tmp_path, world, n_images):
    """SURVEY 8e at the world sizes the 8-GPU node runs (configs[3]: 2975 images -> seven blocks of 372 and one of 371): every
    rank packs its block (shorter than ceil(N/world) on the last non-empty rank, EMPTY on the ranks behind it when N < world),
    pads it to the fixed wire block, and ONE all_gather_into_tensor returns the pool's tables in pool order on every rank, bit for
    bit (NaN / -0.0 / inf score patterns included), with the owner map shard_range implies."""
    mp.spawn(_gather_worker, args=(world, _free_port(), n_images, 5, str(tmp_path)), nprocs=world, join=True)
    done = sorted(os.listdir(tmp_path))
    assert done == sorted("rank%d.ok" % r for r in range(world)), done
    from halo_amd.pool import shard_range
    if (world, n_images) == (8, 2975):
        sizes = [int(open(tmp_path / ("rank%d.ok" % r)).read().split()[1]) - int(open(tmp_path / ("rank%d.ok" % r)).read().split()[0])
                 for r in range(8)]
        assert sizes == [372] * 7 + [371]
    if (world, n_images) == (8, 5):
        assert [shard_range(5, r, 8) for r in (4, 5, 7)] == [(4, 5), (5, 5), (5, 5)]


# ------------------------------------------------------------------ wire format and the sharded driver
def test_wire_format_round_trips_bit_exactly():
    from halo_amd.pool import pack_tables, unpack_tables
    rng = np.random.default_rng(3)
    b, n = 3, 7
    picks = torch.zeros((b, n, 3), dtype=torch.float64)
    picks[:, :, 0] = torch.from_numpy(rng.integers(0, 65536, (b, n)).astype(np.float64))      # h up to the selector's limit
    picks[:, :, 1] = torch.from_numpy(rng.integers(0, 65536, (b, n)).astype(np.float64))
    picks[:, :, 2] = torch.from_numpy(rng.standard_normal((b, n)))
    picks[0, 0, 2] = float("nan"); picks[1, 1, 2] = -0.0; picks[2, 2, 2] = float("inf")
    picks[0, 1, :2] = 65535.0
    npk = torch.tensor([7, 0, 3], dtype=torch.int32)
    wire = pack_tables(picks, npk, rows=5)
    assert wire.dtype == torch.int32 and wire.shape == (5, 3 * n + 1)                           # 12 bytes per pick
    t, c = unpack_tables(wire, n)
    assert np.array_equal(t[:b].numpy().view(np.int64), picks.numpy().view(np.int64))           # bit patterns incl. NaN, -0
    assert list(c) == [7, 0, 3, 0, 0] and float(t[3:].abs().sum()) == 0.0


class _Pool(torch.utils.data.Dataset):
    """Loader items shaped like core/datasets/cityscapes.py:274-286 (what RegionSelection consumes)."""

    def __init__(self, root, n):
        from oracle import halo_oracle as ho
        self.items = []
        for i in range(n):
            rng = np.random.default_rng(500 + i)
            H, W = (40, 64) if i % 2 else (48, 56)
            emb = ho.expmap((rng.standard_normal((1, 6, 10, 16)) * 0.2).astype(np.float32), 1.0, dim=1)
            logit = rng.standard_normal((1, 19, 12, 20)).astype(np.float32)
            self.items.append({"img": torch.zeros(3, 8, 8), "path_to_mask": os.path.join(root, f"m{i}.png"),
                               "origin_mask": torch.full((H, W), 255, dtype=torch.int64),
                               "origin_label": torch.from_numpy(rng.integers(0, 19, (H, W)).astype(np.int64)),
                               "size": torch.tensor([H, W]), "active": torch.from_numpy(rng.random((H, W)) < 0.02),
                               "selected": torch.zeros(H, W, dtype=torch.bool),
                               "path_to_indicator": os.path.join(root, f"i{i}.pth"), "name": f"img{i}",
                               "logit_lr": torch.from_numpy(logit[0]), "embed_lr": torch.from_numpy(emb[0])})

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _cfg():
    import types
    return types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_CLASSES=19, HYPER=True, CURVATURE=1.0),
        ACTIVE=types.SimpleNamespace(UNCERTAINTY="entropy", PURITY="radius", NORMALIZE=True, RADIUS_K=1, MASK_RADIUS_K=5,
                                     BUDGET=0.05, SELECT_ITER=[0, 1, 2, 3, 4], K=100, VIZ_MASK=False))


def _oracle_driver(cfg, feature_extractor, classifier, loader, round_number, write_files=True):
    """Stand-in for the HIP RegionSelection with the same contract (files written through the product's own
    _persist unless write_files=False; per-image pick tables returned): the CPU oracle does the arithmetic."""
    from halo_amd.core.active.build import _persist
    from oracle import halo_oracle as ho
    tables = []
    for batch in loader:
        im = dict(logit_lr=batch["logit_lr"].numpy(), embed_lr=batch["embed_lr"].numpy(),
                  origin_label=batch["origin_label"][0].numpy(), origin_mask=batch["origin_mask"][0].numpy(),
                  active=batch["active"][0].numpy(), selected=batch["selected"][0].numpy())
        (mask, act, sel, picks), = ho.region_selection(cfg, [im])
        if write_files:
            _persist(mask, torch.from_numpy(act), torch.from_numpy(sel), batch["path_to_mask"][0], batch["path_to_indicator"][0])
        tables.append((torch.from_numpy(np.ascontiguousarray(picks)).reshape(-1, 3), len(picks)))
    return tables


def _sharded_worker(rank, world, port, n_images, root, n_regions):
    import sys
    torch.set_num_threads(1)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from halo_amd.pool import region_selection_sharded
    res = region_selection_sharded(_cfg(), None, None, _Pool(root, n_images), 1, driver=_oracle_driver,
                                   loader_kwargs=dict(pin_memory=False), n_regions=n_regions)
    np.savez(os.path.join(root, f"rank{rank}.npz"), tables=res["tables"].numpy(), counts=res["counts"].numpy(),
             owner=res["owner"].numpy(), rng=np.array(res["range"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_images,n_regions", [(2, 5, None), (2, 5, 40), (4, 11, None), (8, 11, 40), (8, 5, None)])
def test_region_selection_sharded_ranks_write_the_single_process_files(tmp_path, world, n_images, n_regions):
    """every rank drives its block of the pool and writes its own mask / indicator files;
    the union of the files and the all-gathered pick tables equal the single-process (reference order) run --
    at world 2, 4 and 8, with uneven blocks, and with ranks whose block is empty (8 ranks, 5 images)."""
    from PIL import Image
    from halo_amd.pool import region_selection_sharded, shard_range
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    ref = region_selection_sharded(_cfg(), None, None, _Pool(str(one), n_images), 1, driver=_oracle_driver,
                                   loader_kwargs=dict(pin_memory=False), n_regions=n_regions)
    assert ref["range"] == (0, n_images) and ref["keep"] is None
    mp.spawn(_sharded_worker, args=(world, _free_port(), n_images, str(two), n_regions), nprocs=world, join=True)
    for i in range(n_images):
        a = np.array(Image.open(one / f"m{i}.png")); b = np.array(Image.open(two / f"m{i}.png"))
        assert Image.open(two / f"m{i}.png").mode == "L" and np.array_equal(a, b), i
        ia, ib = torch.load(one / f"i{i}.pth"), torch.load(two / f"i{i}.pth")
        assert torch.equal(ia["active"], ib["active"]) and torch.equal(ia["selected"], ib["selected"]), i
        assert ib["active"].dtype == torch.bool and int(ib["selected"].sum()) > 0
    ranks = [np.load(two / ("rank%d.npz" % r)) for r in range(world)]
    blocks = [shard_range(n_images, r, world) for r in range(world)]
    assert [tuple(r["rng"]) for r in ranks] == blocks            # world 8, 5 images: ranks 5..7 hold EMPTY blocks and still take part
    if world == 2:
        assert blocks == [(0, 3), (3, 5)]
    for r in ranks:
        assert np.array_equal(r["tables"].view(np.int64), ref["tables"].numpy().view(np.int64))
        assert np.array_equal(r["counts"], ref["counts"].numpy())
        assert list(r["owner"]) == [k for k, (lo, hi) in enumerate(blocks) for _ in range(lo, hi)]
    assert int(ref["counts"].min()) > 0


def _budget_worker(rank, world, port, n_images, root, budget):
    import sys
    torch.set_num_threads(1)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from halo_amd.pool import region_selection_sharded
    res = region_selection_sharded(_cfg(), None, None, _Pool(root, n_images), 1, driver=_oracle_driver,
                                   loader_kwargs=dict(pin_memory=False), global_budget=budget, writer_threads=2)
    np.savez(os.path.join(root, f"rank{rank}.npz"), kept=res["kept"].numpy(), keep=res["keep"].numpy(), tables=res["tables"].numpy())
    dist.destroy_process_group()


def test_global_budget_round_writes_the_kept_picks_files_at_any_world_size(tmp_path):
    """The opt-in pool-wide budget (north_star's "global budget selection"; NOT reference behaviour, which budgets per image:
    build.py:148-150): every image proposes its n_regions greedy picks, the best G of the pool are kept, and each image's files
    are what the reference's loop leaves after as many iterations as picks of that image were kept -- checked against the ORACLE
    run with exactly that many regions -- identically at world 1, 2 and 8 (uneven and empty blocks)."""
    import copy
    from PIL import Image
    from halo_amd.pool import region_selection_sharded
    from oracle import halo_oracle as ho
    n_images, budget = 5, 8
    dirs = {w: tmp_path / ("w%d" % w) for w in (1, 2, 8)}
    for d in dirs.values():
        d.mkdir()
    one = region_selection_sharded(_cfg(), None, None, _Pool(str(dirs[1]), n_images), 1, driver=_oracle_driver,
                                   loader_kwargs=dict(pin_memory=False), global_budget=budget, writer_threads=2)
    kept = one["kept"].numpy()
    assert int(kept.sum()) == budget == int(one["keep"].sum()) and kept.max() > kept.min()       # the images do NOT get equal shares
    # the kept picks are the pool's top scores
    sc = one["tables"].numpy()[:, :, 2]
    valid = np.arange(sc.shape[1])[None, :] < one["counts"].numpy()[:, None]
    assert sc[one["keep"].numpy()].min() >= np.sort(sc[valid])[-budget]
    pool = _Pool(str(tmp_path), n_images)
    for i in range(n_images):
        it = pool[i]
        H, W = it["origin_label"].shape
        cfg_i = copy.deepcopy(_cfg())
        cfg_i.ACTIVE.SELECT_ITER = [0]
        cfg_i.ACTIVE.BUDGET = max(0.0, (int(kept[i]) - 0.5) * 9.0 / (H * W))                      # ceil(H W budget / 9) == kept[i]
        im = dict(logit_lr=it["logit_lr"][None].numpy(), embed_lr=it["embed_lr"][None].numpy(), origin_label=it["origin_label"].numpy(),
                  origin_mask=it["origin_mask"].numpy(), active=it["active"].numpy(), selected=it["selected"].numpy())
        (mask, act, sel, picks), = ho.region_selection(cfg_i, [im])
        assert len(picks) == kept[i], i
        assert np.array_equal(np.array(Image.open(dirs[1] / f"m{i}.png")), mask), i
        ind = torch.load(dirs[1] / f"i{i}.pth")
        assert np.array_equal(ind["active"].numpy(), act) and np.array_equal(ind["selected"].numpy(), sel), i
        assert ind["active"].dtype == torch.bool and Image.open(dirs[1] / f"m{i}.png").mode == "L"
    for world in (2, 8):
        mp.spawn(_budget_worker, args=(world, _free_port(), n_images, str(dirs[world]), budget), nprocs=world, join=True)
        for r in range(world):
            got = np.load(dirs[world] / ("rank%d.npz" % r))
            assert np.array_equal(got["kept"], kept) and np.array_equal(got["keep"], one["keep"].numpy())
            assert np.array_equal(got["tables"].view(np.int64), one["tables"].numpy().view(np.int64))
        for i in range(n_images):
            assert (dirs[world] / f"m{i}.png").read_bytes() == (dirs[1] / f"m{i}.png").read_bytes(), (world, i)
            a, b = torch.load(dirs[world] / f"i{i}.pth"), torch.load(dirs[1] / f"i{i}.pth")
            assert torch.equal(a["active"], b["active"]) and torch.equal(a["selected"], b["selected"]), (world, i)


def test_host_composition_of_the_indicator_maps_native_and_numpy():
    """halo_compose_indicators (libhalo_host.so) == its numpy statement == the oracle's select loop on the windows: clipped at the
    borders, prior maps kept, k = 0 leaves them unchanged."""
    from halo_amd import _hostlib
    from halo_amd.core.active.build import compose_indicators
    rng = np.random.default_rng(8)
    H, W = 37, 53
    pa, ps = rng.random((H, W)) < 0.1, rng.random((H, W)) < 0.05
    picks = np.stack([rng.integers(0, H, 40), rng.integers(0, W, 40), rng.standard_normal(40)], 1).astype(np.float64)
    picks[0, :2] = (0, 0); picks[1, :2] = (H - 1, W - 1); picks[2, :2] = (0, W - 1)
    for k in (0, 1, 40):
        for r, mr in ((1, 5), (2, 3), (0, 0)):
            a_np, s_np = compose_indicators(pa, ps, picks[:k], r, mr)
            a_c, s_c = _hostlib.compose_indicators(pa, ps, picks, k, r, mr)
            assert np.array_equal(a_np, a_c) and np.array_equal(s_np, s_c), (k, r, mr)
            ea, es = pa.copy(), ps.copy()
            for h, w in picks[:k, :2].astype(int):
                ea[max(h - mr, 0):h + mr + 1, max(w - mr, 0):w + mr + 1] = True
                es[max(h - r, 0):h + r + 1, max(w - r, 0):w + r + 1] = True
            assert np.array_equal(a_np, ea) and np.array_equal(s_np, es)


class _FakeLearner:
    """The attributes SourceFreeLearner.on_train_batch_start touches (core/train_learners.py:307-326)."""

    def __init__(self, root, rank, n_images):
        import types
        self.cfg = _cfg()
        self.cfg.SAVE_DIR = root
        self.local_rank, self.debug, self.active_iters, self.active_round = rank, False, [0, 7], 0
        self.feature_extractor = self.classifier = None
        self.active_loader = torch.utils.data.DataLoader(_Pool(root, n_images), batch_size=1)
        self.saved, self.logged = [], []
        self.trainer = types.SimpleNamespace(save_checkpoint=self.saved.append)
        self.acquisition_driver = _oracle_driver

    def log(self, *a, **k):
        self.logged.append(a)


def test_sharded_hook_replaces_the_rank0_gate(tmp_path):
    from halo_amd.hooks import sharded_on_train_batch_start, use_sharded_rounds
    cls = use_sharded_rounds(type("L", (_FakeLearner,), {"on_train_batch_start": lambda self, b, i: (b, i)}))
    assert cls.on_train_batch_start is sharded_on_train_batch_start and cls._reference_on_train_batch_start is not None
    ln = cls(str(tmp_path), 0, 3)
    assert ln.on_train_batch_start("batch", 3) == ("batch", 3) and ln.active_round == 0 and not ln.saved     # not an active iteration
    assert ln.on_train_batch_start("batch", 7) == ("batch", 7)
    assert ln.active_round == 1 and ln.saved == [os.path.join(str(tmp_path), "model_before_round_0.ckpt")]
    assert ln.logged and ln.last_round_tables["tables"].shape[0] == 3
    assert all(os.path.exists(tmp_path / f"m{i}.png") and os.path.exists(tmp_path / f"i{i}.pth") for i in range(3))
    assert ln.last_round_tables["keep"] is None and ln.last_round_tables["kept"] is None          # the default: the reference's per-image budget
    ln.debug = True
    ln.on_train_batch_start("batch", 0)
    assert ln.active_round == 1
    # the opt-in switch of the hook: a pool-wide budget for the round
    ln.debug = False
    ln.acquisition_global_budget = 7
    ln.active_iters = [9]
    ln.on_train_batch_start("batch", 9)
    assert int(ln.last_round_tables["kept"].sum()) == 7 and ln.active_round == 2


def test_global_budget_refuses_a_driver_without_write_files_and_a_reordering_loader(tmp_path):
    """The driver contract of the global-budget round: a custom driver that cannot take `write_files=False` would write
    the UNBUDGETED files -- it is refused before any work; and the second, file-writing pass must see the images in the order the
    tables were computed in -- a loader that yields another order is refused before a mismatched file is written."""
    from halo_amd.core.active.build import persist_from_tables
    from halo_amd.pool import region_selection_sharded

    def old_contract_driver(cfg, feature_extractor, classifier, loader, round_number):
        raise AssertionError("must not be called")

    with pytest.raises(TypeError, match="write_files"):
        region_selection_sharded(_cfg(), None, None, _Pool(str(tmp_path), 3), 1, driver=old_contract_driver,
                                 loader_kwargs=dict(pin_memory=False), global_budget=4)
    # without a global budget the five-argument contract stays valid
    res = region_selection_sharded(_cfg(), None, None, _Pool(str(tmp_path), 3), 1,
                                   driver=lambda c, f, k, loader, r: _oracle_driver(c, f, k, loader, r), loader_kwargs=dict(pin_memory=False))
    assert res["keep"] is None and res["tables"].shape[0] == 3
    # the order check of the second pass
    from torch.utils.data import DataLoader
    pool = _Pool(str(tmp_path), 3)
    loader = DataLoader(pool, batch_size=1, shuffle=False)
    paths = [b["path_to_mask"][0] for b in loader]
    with pytest.raises(RuntimeError, match="same order"):
        persist_from_tables(_cfg(), loader, res["tables"], res["counts"], writer_threads=1, expect_paths=paths[::-1])
    persist_from_tables(_cfg(), loader, res["tables"], res["counts"], writer_threads=1, expect_paths=paths)
