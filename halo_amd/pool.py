"""Image-wise sharding of the unlabeled pool and the one exchange step of the path.

The reference runs the whole acquisition round on rank 0, one image at a time, while the other
ranks wait in DDP (core/train_learners.py:308); there is no collective on its hot path.  Images
are fully independent (per-image normalisation floating_region.py:206-208, per-image budget
build.py:148-150, per-image suppression), so here every rank scores a contiguous block of the
image list and the ranks exchange only the fixed-size per-image pick tables with ONE all-gather
(RCCL over xGMI on the GPUs; gloo in the CPU tests).

Wire format (one int32 row per image, `3*n_regions + 1` words): per pick `(h << 16) | w` and the
float64 score as two words, then the pick count -- 12 bytes per pick, 28 KB per 1024x2048 image.
Shard sizes follow from (n_images, world) alone (`shard_range`), so no size exchange and no host
synchronisation is needed: every rank pads its block to ceil(N / world) rows.

With `global_budget=None` (default, = reference behaviour) the result does not depend on the
world size.
"""
import math

import torch
import torch.distributed as dist


def shard_range(n_images, rank, world):
    """Contiguous block of ceil(N/world) images per rank, in list order (2975 -> 372/371 at 8 ranks)."""
    per = math.ceil(n_images / world) if world > 0 else n_images
    lo = min(rank * per, n_images)
    return lo, min(lo + per, n_images)


def regions_per_image(height, width, budget, n_rounds, radius_k):
    """build.py:148-150: ceil(H*W * (BUDGET / len(SELECT_ITER)) / (2*RADIUS_K+1)^2)."""
    return math.ceil(height * width * (budget / n_rounds) / (2 * radius_k + 1) ** 2)


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def pack_tables(picks, n_picked, rows):
    """(b, n, 3) float64 (h, w, score) + (b,) counts -> (rows, 3n+1) int32 wire block, zero padded."""
    b, n = picks.shape[0], picks.shape[1]
    wire = torch.zeros((rows, 3 * n + 1), dtype=torch.int32, device=picks.device)
    if b:
        pos = ((picks[:, :, 0].to(torch.int64) << 16) | picks[:, :, 1].to(torch.int64)).to(torch.int32)   # sides <= 65535
        body = wire[:b, :3 * n].view(b, n, 3)
        body[:, :, 0] = pos
        body[:, :, 1:] = picks[:, :, 2].contiguous().view(torch.int32).view(b, n, 2)     # the score's bits, unchanged
        wire[:b, 3 * n] = n_picked.to(torch.int32)
    return wire


def unpack_tables(wire, n):
    """Inverse of pack_tables: -> (tables (rows, n, 3) float64, counts (rows,) int32)."""
    rows = wire.shape[0]
    body = wire[:, :3 * n].reshape(rows, n, 3)
    tables = torch.empty((rows, n, 3), dtype=torch.float64, device=wire.device)
    tables[:, :, 0] = (body[:, :, 0] >> 16) & 0xffff
    tables[:, :, 1] = body[:, :, 0] & 0xffff
    tables[:, :, 2] = body[:, :, 1:].contiguous().view(torch.float64).view(rows, n)
    return tables, wire[:, 3 * n].clone()


def gather_tables(picks, n_picked, n_images=None, group=None):
    """All-gather per-image pick tables with ONE fixed-size collective and no host synchronisation.

    picks (b, n, 3) float64, n_picked (b,) int32: this rank's block `shard_range(n_images, rank, world)`
    of the pool (n_images defaults to world * b, i.e. equal blocks).  Returns (tables (N, n, 3),
    counts (N,), owner (N,)) in pool order on every rank."""
    world, rank = _world(group)
    b, n = picks.shape[0], picks.shape[1]
    if n_images is None:
        n_images = world * b
    per = math.ceil(n_images / world)
    lo, hi = shard_range(n_images, rank, world)
    assert hi - lo == b, "this rank's block has %d images, shard_range says %d" % (b, hi - lo)
    owner = torch.arange(n_images, device=picks.device, dtype=torch.int64).div(max(per, 1), rounding_mode="floor").to(torch.int32)
    if not (dist.is_available() and dist.is_initialized()):
        return picks, n_picked.to(torch.int32), owner
    wire = pack_tables(picks, n_picked, per)
    out = torch.empty((world * per, 3 * n + 1), dtype=torch.int32, device=picks.device)
    dist.all_gather_into_tensor(out, wire, group=group)
    tables, counts = unpack_tables(out[:n_images], n)       # the padding rows all sit behind the last rank's block
    return tables, counts, owner


def global_budget_select(tables, counts, total_regions):
    """OPTIONAL, not reference behaviour: re-rank the gathered picks pool-wide and keep the best
    `total_regions` (ties: lower image index, then earlier pick).  Returns a bool mask (images, n).
    The reference spends exactly n_regions per image; keep this off to match it."""
    I, n, _ = tables.shape
    valid = torch.arange(n, device=tables.device)[None, :] < counts[:, None].to(torch.int64)
    score = torch.where(valid, tables[:, :, 2], torch.full_like(tables[:, :, 2], -float("inf")))
    order = torch.argsort(score.reshape(-1), descending=True, stable=True)
    keep = torch.zeros(I * n, dtype=torch.bool, device=tables.device)
    k = torch.clamp(valid.sum(), max=int(total_regions))
    keep[order] = torch.arange(I * n, device=tables.device) < k
    return keep.reshape(I, n) & valid


def acquire_pool(images, acquire_fn, n_regions, group=None, global_budget=None):
    """Score + select this rank's shard and exchange the pick tables.

    images: sequence (len N, same on every rank) of per-image inputs; acquire_fn(list_of_inputs) ->
    (picks (b,n,3) float64, n_picked (b,) int32) for a batch of this rank's images (the HIP path in
    production, any stand-in in tests).  Returns (tables, counts, owner) for the WHOLE pool on every
    rank, plus the optional global-budget keep mask."""
    world, rank = _world(group)
    lo, hi = shard_range(len(images), rank, world)
    picks, npk = acquire_fn([images[i] for i in range(lo, hi)])
    assert picks.shape[1] == n_regions and picks.shape[0] == hi - lo
    tables, counts, owner = gather_tables(picks, npk, len(images), group)
    keep = None
    if global_budget is not None:
        keep = global_budget_select(tables, counts, global_budget)
    return tables, counts, owner, keep


def region_selection_sharded(cfg, feature_extractor, classifier, dataset_or_loader, round_number, group=None,
                             loader_kwargs=None, global_budget=None, driver=None, n_regions=None):
    """RegionSelection (build.py:71-186) with the pool sharded over the ranks of `group`.

    Each rank runs the drop-in driver on its block of the dataset and writes its own mask /
    indicator files (the filesystem is the reference's hand-off, cityscapes.py:234-251); the
    per-image pick tables are then exchanged with one all-gather, which is also the point where
    every rank knows that all files are on disk (it replaces the reference's "ranks != 0 stall in
    the next DDP all-reduce", train_learners.py:308).  Returns a dict: `range` (this rank's block),
    `tables` (N, n, 3), `counts` (N,), `owner` (N,) for the whole pool on every rank, `keep` (the
    optional global-budget mask, None by default = reference behaviour).

    `driver(cfg, feature_extractor, classifier, loader, round_number) -> [(picks (n,3), count)]` per
    image defaults to the HIP RegionSelection; the CPU tests inject a stand-in.  `n_regions` = the
    table width (regions per image, build.py:148-150); when None the ranks agree on the widest table
    with one scalar all-reduce first (pools of mixed image sizes)."""
    from torch.utils.data import DataLoader, Subset
    world, rank = _world(group)
    dataset = getattr(dataset_or_loader, "dataset", dataset_or_loader)
    n_images = len(dataset)
    lo, hi = shard_range(n_images, rank, world)
    kw = dict(batch_size=1, shuffle=False, num_workers=0, pin_memory=True, drop_last=False)
    kw.update(loader_kwargs or {})
    loader = DataLoader(Subset(dataset, range(lo, hi)), **kw)
    if driver is None:
        from .core.active.build import RegionSelection

        def driver(*a):
            return RegionSelection(*a, return_tables=True)
    per_image = driver(cfg, feature_extractor, classifier, loader, round_number)      # pipelined, async writers
    assert len(per_image) == hi - lo
    n = max([p.shape[0] for p, _ in per_image], default=0) if n_regions is None else int(n_regions)
    if world > 1 and n_regions is None:     # images may differ in size: agree on the widest table
        nt = torch.tensor([n], dtype=torch.int64, device=_comm_device(per_image))
        dist.all_reduce(nt, op=dist.ReduceOp.MAX, group=group)
        n = int(nt.item())
    dev = _comm_device(per_image)
    picks = torch.zeros((hi - lo, max(n, 1), 3), dtype=torch.float64, device=dev)
    npk = torch.zeros((hi - lo,), dtype=torch.int32, device=dev)
    for j, (p, k) in enumerate(per_image):
        picks[j, :p.shape[0]] = p.to(dev)
        npk[j] = int(k)
    tables, counts, owner = gather_tables(picks, npk, n_images, group)
    keep = global_budget_select(tables, counts, global_budget) if global_budget is not None else None
    return {"range": (lo, hi), "tables": tables, "counts": counts, "owner": owner, "keep": keep}


def _comm_device(per_image):
    """Tensors handed to the collective live where the backend wants them: the GPU under RCCL, the host under gloo."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    for p, _ in per_image:
        return p.device
    return torch.device("cpu")
