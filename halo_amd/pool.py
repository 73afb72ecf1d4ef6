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

With `global_budget=None` (default, = reference behaviour: build.py:148-150 budgets per image) the
result does not depend on the world size.  `global_budget=G` (north_star's "global budget
selection"; NOT reference behaviour, opt-in) spends G regions over the whole pool: every image
still proposes its n_regions greedy picks, the gathered tables are re-ranked pool-wide, and each
image's files are written from the KEPT prefix of its table -- also independent of the world size.
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
        pack_tables_into(wire[:b], picks, n_picked)
    return wire


def pack_tables_into(wire_rows, picks, n_picked):
    """Write the wire rows of `picks` (b, n, 3) / `n_picked` (b,) into `wire_rows` (b, 3n+1) int32 in place.  Device tensors
    go through ONE kernel launch on the current stream (halo_pack_pick_tables); host tensors (the gloo tests' stand-in
    drivers) through the equivalent index arithmetic."""
    b, n = picks.shape[0], picks.shape[1]
    assert wire_rows.shape == (b, 3 * n + 1) and wire_rows.dtype == torch.int32 and wire_rows.stride(1) == 1
    if b == 0:
        return wire_rows
    if picks.is_cuda:
        from . import _lib
        dev = _lib.require_device(picks, n_picked, wire_rows)
        pk = picks if (picks.dtype == torch.float64 and picks.is_contiguous()) else picks.to(torch.float64).contiguous()
        npk = n_picked if (n_picked.dtype == torch.int32 and n_picked.is_contiguous()) else n_picked.to(torch.int32).contiguous()
        rc = _lib.lib().halo_pack_pick_tables(_lib.ptr(pk), _lib.ptr(npk), b, n, _lib.ptr(wire_rows), wire_rows.stride(0),
                                              _lib.stream_ptr(dev))
        _lib.check(rc, "halo_pack_pick_tables")
        return wire_rows
    pos = ((picks[:, :, 0].to(torch.int64) << 16) | picks[:, :, 1].to(torch.int64)).to(torch.int32)   # sides <= 65535
    body = wire_rows[:, :3 * n].view(b, n, 3) if wire_rows.is_contiguous() else None
    if body is None:
        tmp = torch.zeros((b, 3 * n + 1), dtype=torch.int32)
        pack_tables_into(tmp, picks, n_picked)
        wire_rows.copy_(tmp)
        return wire_rows
    body[:, :, 0] = pos
    sc = torch.empty((b, n), dtype=torch.float64).copy_(picks[:, :, 2])         # fresh storage: unit strides whatever the slice's
    body[:, :, 1:] = sc.view(torch.int32).view(b, n, 2)                          # the score's bits, unchanged
    wire_rows[:, 3 * n] = n_picked.to(torch.int32)
    return wire_rows


def unpack_tables(wire, n):
    """Inverse of pack_tables: -> (tables (rows, n, 3) float64, counts (rows,) int32)."""
    rows = wire.shape[0]
    body = wire[:, :3 * n].reshape(rows, n, 3)
    tables = torch.empty((rows, n, 3), dtype=torch.float64, device=wire.device)
    tables[:, :, 0] = (body[:, :, 0] >> 16) & 0xffff
    tables[:, :, 1] = body[:, :, 0] & 0xffff
    sc = torch.empty((rows, n, 2), dtype=torch.int32, device=wire.device).copy_(body[:, :, 1:])
    tables[:, :, 2] = sc.view(torch.float64).view(rows, n)
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
    tables, counts = gather_wire(pack_tables(picks, n_picked, per), n_images, n, group)
    return tables, counts, owner


def gather_wire(wire, n_images, n, group=None):
    """The round's ONE collective on an already packed block: `wire` (ceil(N/world), 3n+1) int32 holds this rank's rows
    (pack_tables_into), zero padded.  Returns (tables (N, n, 3) float64, counts (N,) int32) in pool order on every rank.
    Device tensors go to RCCL as they are; under a host backend (gloo: the CPU tests, and the two-ranks-on-one-GPU
    hardware test) a device block is staged through the host and the result is returned on the device."""
    world, _ = _world(group)
    per = math.ceil(n_images / world)
    assert wire.shape == (per, 3 * n + 1) and wire.dtype == torch.int32 and wire.is_contiguous()
    if not (dist.is_available() and dist.is_initialized()):
        return unpack_tables(wire[:n_images], n)
    host_backend = wire.is_cuda and dist.get_backend(group) != "nccl"
    src = wire.cpu() if host_backend else wire
    out = torch.empty((world * per, 3 * n + 1), dtype=torch.int32, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    if host_backend:
        out = out.to(wire.device)
    return unpack_tables(out[:n_images], n)                 # the padding rows all sit behind the last rank's block


def reset_round_state(active, selected, active_mask):
    """The loader's round-1 state (core/datasets/cityscapes.py:245-251) written on the device in one pass:
    active = selected = False, active_mask = 255.  Contiguous device tensors of one shape, in place."""
    from . import _lib
    dev = _lib.require_device(active, selected, active_mask)
    assert active.dtype == torch.bool and selected.dtype == torch.bool and active_mask.dtype == torch.int64
    assert active.shape == selected.shape == active_mask.shape
    assert active.is_contiguous() and selected.is_contiguous() and active_mask.is_contiguous()
    rc = _lib.lib().halo_reset_round_state(_lib.ptr(active), _lib.ptr(selected), _lib.ptr(active_mask), active.numel(),
                                           _lib.stream_ptr(dev))
    _lib.check(rc, "halo_reset_round_state")


def undo_picks(picks, n_picked, active_radius, mask_radius, active, selected, active_mask):
    """Restore the round-1 state of (B,H,W) `active` / `selected` / `active_mask` after a selection that STARTED from it,
    from that selection's pick table: only the windows select_pixels_to_label wrote (build.py:52-62) are rewritten."""
    from . import _lib
    dev = _lib.require_device(picks, n_picked, active, selected, active_mask)
    B, H, W = active.shape
    assert picks.dtype == torch.float64 and picks.is_contiguous() and picks.shape[0] == B and picks.shape[2] == 3
    assert n_picked.dtype == torch.int32 and n_picked.is_contiguous()
    assert active.dtype == torch.bool and selected.dtype == torch.bool and active_mask.dtype == torch.int64
    assert active.is_contiguous() and selected.is_contiguous() and active_mask.is_contiguous()
    rc = _lib.lib().halo_undo_picks(_lib.ptr(picks), _lib.ptr(n_picked), B, H, W, picks.shape[1], int(active_radius),
                                    int(mask_radius), _lib.ptr(active), _lib.ptr(selected), _lib.ptr(active_mask),
                                    _lib.stream_ptr(dev))
    _lib.check(rc, "halo_undo_picks")


def device_identity(index):
    """'pci=<bus id> uuid=<hex>' of HIP device `index` (asked of the HIP runtime the kernels run on)."""
    import ctypes
    from . import _lib
    buf = ctypes.create_string_buffer(128)
    _lib.check(_lib.lib().halo_device_identity(int(index), buf, 128), "halo_device_identity")
    return buf.value.decode("ascii", "replace")


def assert_distinct_devices(device_index, group=None):
    """One process per GPU: every rank of `group` (one node) must hold a different physical device.  Exchanges the
    identity strings with one all_gather_object; raises on a duplicate.  Returns the list (rank order)."""
    world, rank = _world(group)
    me = device_identity(device_index)
    if world == 1:
        return [me]
    ids = [None] * world
    dist.all_gather_object(ids, me, group=group)
    if len(set(ids)) != world:
        raise RuntimeError("ranks share a GPU: %s" % ", ".join("rank %d -> %s" % (r, i) for r, i in enumerate(ids)))
    return ids


def global_budget_select(tables, counts, total_regions):
    """OPTIONAL, not reference behaviour: re-rank the gathered picks pool-wide and keep the best
    `total_regions` (ties: lower image index, then earlier pick; NaN scores rank first, as in the
    selector).  Returns a bool mask (images, n).  An image's greedy picks are non-increasing in
    score, so what is kept of an image is a PREFIX of its table: its files are the state of the
    reference's loop after that many iterations (kept_counts).  The reference spends exactly
    n_regions per image; keep this off to match it."""
    I, n, _ = tables.shape
    valid = torch.arange(n, device=tables.device)[None, :] < counts[:, None].to(torch.int64)
    score = torch.where(valid, tables[:, :, 2], torch.full_like(tables[:, :, 2], -float("inf")))
    score = torch.where(torch.isnan(score), torch.full_like(score, float("inf")), score)      # the selector's order: NaN above everything
    order = torch.argsort(score.reshape(-1), descending=True, stable=True)
    keep = torch.zeros(I * n, dtype=torch.bool, device=tables.device)
    k = torch.clamp(valid.sum(), max=int(total_regions))
    keep[order] = torch.arange(I * n, device=tables.device) < k
    return keep.reshape(I, n) & valid


def kept_counts(keep):
    """picks kept per image under a global budget; the kept picks of an image are the first k of its table"""
    k = keep.sum(dim=1).to(torch.int32)
    n = keep.shape[1]
    prefix = torch.arange(n, device=keep.device)[None, :] < k[:, None].to(torch.int64)
    assert bool((prefix == keep).all()), "global-budget keep-mask is not a prefix of every image's table"
    return k


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
                             loader_kwargs=None, global_budget=None, driver=None, n_regions=None, writer_threads=None):
    """RegionSelection (build.py:71-186) with the pool sharded over the ranks of `group`.

    Each rank runs the drop-in driver on its block of the dataset and writes its own mask /
    indicator files (the filesystem is the reference's hand-off, cityscapes.py:234-251); the
    per-image pick tables are then exchanged with one all-gather, which is also the point where
    every rank knows that all files are on disk (it replaces the reference's "ranks != 0 stall in
    the next DDP all-reduce", train_learners.py:308).  Returns a dict: `range` (this rank's block),
    `tables` (N, n, 3), `counts` (N,), `owner` (N,) for the whole pool on every rank, `keep` (the
    optional global-budget mask, None by default = reference behaviour), `kept` (picks kept per
    image, None by default).

    `global_budget=G` (opt-in; NOT reference behaviour, which budgets per image: build.py:148-150):
    the ranks score and select WITHOUT writing (`driver(..., write_files=False)`), exchange the
    tables, rank all proposed picks pool-wide (global_budget_select) and then every rank writes its
    block's files from the kept prefix of each table in a second, model-free pass over its loader
    (core.active.build.persist_from_tables).  Files and tables are independent of the world size.

    `driver(cfg, feature_extractor, classifier, loader, round_number) -> [(picks (n,3), count)]` per
    image defaults to the HIP RegionSelection; the CPU tests inject a stand-in.  Under `global_budget` the
    driver is called with one more KEYWORD, `write_files=False`, and must then score and select WITHOUT
    writing any file (a driver that cannot take the keyword is refused with a TypeError before any work
    is done); the second pass checks that the loader yields the same `path_to_mask` sequence as the
    first, so that every table meets the image it was computed for.  `n_regions` = the
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

        def driver(*a, **kw):
            return RegionSelection(*a, return_tables=True, **kw)
    defer = global_budget is not None
    seen_paths = None
    if defer:
        import inspect
        try:
            params = inspect.signature(driver).parameters
            takes = "write_files" in params or any(p_.kind is inspect.Parameter.VAR_KEYWORD for p_ in params.values())
        except (TypeError, ValueError):
            takes = True                                      # builtins without a signature: let the call decide
        if not takes:
            raise TypeError("region_selection_sharded(global_budget=...) calls the acquisition driver with write_files=False "
                            "(score and select only; the files follow from the pool-wide keep-mask): %r does not accept that "
                            "keyword" % (driver,))
        seen_paths = []
        loader = _RecordingLoader(loader, seen_paths)
    # pipelined, async writers; under a global budget the files wait for the pool-wide keep-mask
    per_image = driver(cfg, feature_extractor, classifier, loader, round_number, **({"write_files": False} if defer else {}))
    assert len(per_image) == hi - lo
    n = max([p.shape[0] for p, _ in per_image], default=0) if n_regions is None else int(n_regions)
    if world > 1 and n_regions is None:     # images may differ in size: agree on the widest table
        nt = torch.tensor([n], dtype=torch.int64, device=_comm_device(per_image))
        dist.all_reduce(nt, op=dist.ReduceOp.MAX, group=group)
        n = int(nt.item())
    dev = _comm_device(per_image)
    picks = torch.zeros((hi - lo, max(n, 1), 3), dtype=torch.float64, device=dev)
    if per_image and all(p.shape[0] == n for p, _ in per_image):
        picks = torch.stack([p for p, _ in per_image]).to(dev)        # one copy for the block (round 4: one per image)
    else:
        for j, (p, k) in enumerate(per_image):
            picks[j, :p.shape[0]] = p.to(dev)
    # the counts are host integers already (the writer threads read them from pinned memory): no per-image device sync
    npk = torch.tensor([int(k) for _, k in per_image], dtype=torch.int32).to(dev)
    tables, counts, owner = gather_tables(picks, npk, n_images, group)
    keep = kept = None
    if defer:
        keep = global_budget_select(tables, counts, global_budget)
        kept = kept_counts(keep)
        from .core.active.build import persist_from_tables
        persist_from_tables(cfg, loader.loader, tables[lo:hi], kept[lo:hi], writer_threads=writer_threads, expect_paths=seen_paths)
        if world > 1:
            dist.barrier(group=group)                                    # every rank's files are on disk when anyone returns
    return {"range": (lo, hi), "tables": tables, "counts": counts, "owner": owner, "keep": keep, "kept": kept}


class _RecordingLoader:
    """The rank's loader, noting the `path_to_mask` of every image it hands to the first (scoring) pass of a global-budget round."""

    def __init__(self, loader, seen):
        self.loader, self.seen = loader, seen

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):
        return getattr(self.loader, name)

    def __iter__(self):
        for batch in self.loader:
            try:
                self.seen.extend(str(p_) for p_ in batch["path_to_mask"])
            except (KeyError, TypeError):
                pass
            yield batch


def _comm_device(per_image):
    """Tensors handed to the collective live where the backend wants them: the GPU under RCCL, the host under gloo."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    for p, _ in per_image:
        return p.device
    return torch.device("cpu")
