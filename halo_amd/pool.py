"""Image-wise sharding of the unlabeled pool and the one exchange step of the path.

The reference runs the whole acquisition round on rank 0, one image at a time, while the other
ranks wait in DDP (core/train_learners.py:308); there is no collective on its hot path.  Images
are fully independent (per-image normalisation floating_region.py:206-208, per-image budget
build.py:148-150, per-image suppression), so here every rank scores a contiguous block of the
image list and the ranks exchange only the fixed-size per-image pick tables with ONE all-gather
(RCCL over xGMI on the GPUs; gloo in the CPU tests).

Table row = (h, w, score) float64, `n_regions` rows per image, unused rows zero; 56 KB per
1024x2048 image.  With `global_budget=False` (default, = reference behaviour) the result does not
depend on the world size.
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


def gather_tables(picks, n_picked, group=None):
    """All-gather per-image pick tables.  picks (b, n, 3) float64, n_picked (b,) int32 on this rank
    (b may differ between ranks by at most the sharding remainder).  Returns (tables, counts, owner):
    (sum_b, n, 3), (sum_b,), (sum_b,) in pool order.  One fixed-size all-gather: every rank pads to
    the largest shard."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return picks, n_picked, torch.zeros(picks.shape[0], dtype=torch.int32, device=picks.device)
    world = dist.get_world_size(group)
    b = torch.tensor([picks.shape[0]], dtype=torch.int64, device=picks.device)
    sizes = [torch.zeros_like(b) for _ in range(world)]
    dist.all_gather(sizes, b, group=group)
    sizes = [int(s.item()) for s in sizes]
    bmax = max(sizes)
    n = picks.shape[1]
    # pack counts into the same buffer so the exchange stays a single collective
    pack = torch.zeros((bmax, n * 3 + 1), dtype=torch.float64, device=picks.device)
    pack[:picks.shape[0], :n * 3] = picks.reshape(picks.shape[0], n * 3)
    pack[:picks.shape[0], n * 3] = n_picked.to(torch.float64)
    out = torch.empty((world * bmax, n * 3 + 1), dtype=torch.float64, device=picks.device)
    dist.all_gather_into_tensor(out, pack, group=group)
    rows, owner = [], []
    for r, s in enumerate(sizes):
        rows.append(out[r * bmax: r * bmax + s])
        owner.append(torch.full((s,), r, dtype=torch.int32, device=picks.device))
    allr = torch.cat(rows)
    return (allr[:, :n * 3].reshape(-1, n, 3).contiguous(), allr[:, n * 3].to(torch.int32),
            torch.cat(owner))


def global_budget_select(tables, counts, total_regions):
    """OPTIONAL, not reference behaviour: re-rank the gathered picks pool-wide and keep the best
    `total_regions` (ties: lower image index, then earlier pick).  Returns a bool mask (images, n).
    The reference spends exactly n_regions per image; keep this off to match it."""
    I, n, _ = tables.shape
    valid = torch.arange(n, device=tables.device)[None, :] < counts[:, None].to(torch.int64)
    score = torch.where(valid, tables[:, :, 2], torch.full_like(tables[:, :, 2], -float("inf")))
    order = torch.argsort(score.reshape(-1), descending=True, stable=True)
    keep = torch.zeros(I * n, dtype=torch.bool, device=tables.device)
    k = min(int(total_regions), int(valid.sum().item()))
    keep[order[:k]] = True
    return keep.reshape(I, n) & valid


def acquire_pool(images, acquire_fn, n_regions, group=None, global_budget=None):
    """Score + select this rank's shard and exchange the pick tables.

    images: sequence (len N, same on every rank) of per-image inputs; acquire_fn(list_of_inputs) ->
    (picks (b,n,3) float64, n_picked (b,) int32) for a batch of this rank's images (the HIP path in
    production, any stand-in in tests).  Returns (tables, counts, owner) for the WHOLE pool on every
    rank, plus the optional global-budget keep mask."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = shard_range(len(images), rank, world)
    picks, npk = acquire_fn([images[i] for i in range(lo, hi)])
    assert picks.shape[1] == n_regions and picks.shape[0] == hi - lo
    tables, counts, owner = gather_tables(picks, npk, group)
    keep = None
    if global_budget is not None:
        keep = global_budget_select(tables, counts, global_budget)
    return tables, counts, owner, keep


def region_selection_sharded(cfg, feature_extractor, classifier, dataset_or_loader, round_number, group=None,
                             loader_kwargs=None):
    """RegionSelection (build.py:71-186) with the pool sharded over the ranks of `group`: each rank
    runs the drop-in driver on its block of the dataset and writes its own mask / indicator files
    (the filesystem is the reference's hand-off, cityscapes.py:234-251); a barrier replaces the
    reference's "ranks != 0 stall in the next DDP all-reduce"."""
    from torch.utils.data import DataLoader, Subset
    from .core.active.build import RegionSelection
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    dataset = getattr(dataset_or_loader, "dataset", dataset_or_loader)
    lo, hi = shard_range(len(dataset), rank, world)
    kw = dict(batch_size=1, shuffle=False, num_workers=0, pin_memory=True, drop_last=False)
    kw.update(loader_kwargs or {})
    loader = DataLoader(Subset(dataset, range(lo, hi)), **kw)
    RegionSelection(cfg, feature_extractor, classifier, loader, round_number)      # pipelined, async writers
    if world > 1:
        dist.barrier(group=group)
    return lo, hi
