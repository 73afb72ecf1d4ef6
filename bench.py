#!/usr/bin/env python3
"""bench.py -- acquisition-scored images/s on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W      # N > 1 without a launcher: starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one batch of B synthetic Cityscapes-shaped images (1024x2048, C=256 float64 embedding,
19 classes) through the whole unit of work of SURVEY.md 8(d): FloatingRegionScore.forward (HALO
branch: entropy x radius, normalised, 3x3) + `score[active] = -inf` + select_pixels_to_label
(2331 regions, radius 1, mask radius 5), inputs resident in HBM, all three output maps written.
Scoring of batch s+1 (HBM-bound) overlaps the selection of batch s on the slot's own HIP stream.

Workload at N=1: BASELINE.json configs[1] -- a pool of `steps*B` image evaluations (default 32 x 16 =
512) drawn from a ring of R = 32 distinct resident images, so consecutive steps read different
images (a 500-image pool does not fit in HBM at 4.3 GB/image); `--pool-images 2975` = configs[2].
N>1: the pool is sharded image-wise (contiguous blocks, halo_amd.pool.shard_range), every rank runs the same per-rank
workload (weak scaling; `--pool-images N` gives every rank its own block of the N images) and the per-image pick tables
are exchanged with ONE RCCL all-gather per ROUND: each step packs its tables into the rank's wire block, the
collective runs once behind the last step on its own normal-priority stream, inside the timed region.
`--branch ripu|hyper`, `--feat-dtype f32`, `--channels 512`, `--source lowres` are variants for
DESIGN.md / profiles/, not the BASELINE unit.

Prints ONE JSON line (rank 0).  roofline: the feature-reduction kernel's algorithmic bytes / its
average duration measured live with HIP events on its own stream.  cpu_baseline: the CPU oracle
(OpenMP over the host cores this process may use) on 1 warm-up + 8 of the same images (kind "port").
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# hardware queues: importing the package changes nothing; the bench opts in to the acquisition's measured optimum (2, read by the
# HIP runtime at initialisation; HALO_BENCH_HW_QUEUES overrides, an exported GPU_MAX_HW_QUEUES wins) -- halo_amd.configure()
import halo_amd  # noqa: E402
if "GPU_MAX_HW_QUEUES" not in os.environ:
    halo_amd.configure(hw_queues=int(os.environ.get("HALO_BENCH_HW_QUEUES", "2")))

import numpy as np
import torch
import torch.distributed as dist
from halo_amd import _lib as _halo_lib  # noqa: E402  (constants only; the library loads on first use)

LOGIT_LR_VALU_PER_PX = 1104.0      # SQ_INSTS_VALU per output pixel at 19 classes, profiles/r05_pmc_lowres.json (578.8 M per 16 images) (1162 before the integer forms, 1411 in round 3)
HBM_PEAK_GBPS = 8000.0       # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
H, W, O = 1024, 2048, 19


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="default 32 x 16 = 512 image evaluations (configs[1]: 500)")
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--pool-images", type=int, default=0,
                    help="score exactly this many image evaluations over all ranks (overrides --steps; 2975 = configs[2], "
                         "the Cityscapes train list), the last step of a rank being a partial batch")
    ap.add_argument("--batch", type=int, default=16, help="images per step and rank")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight (selection slots)")
    ap.add_argument("--ring", type=int, default=32, help="distinct resident images per rank (two batches: consecutive "
                                                         "steps read different images)")
    ap.add_argument("--branch", choices=["halo", "ripu", "hyper"], default="halo",
                    help="halo = entropy x radius, normalised, mask radius 5 (configs/gtav/source_target.yaml, the BASELINE "
                         "unit); ripu = entropy x ripu, not normalised, mask radius 3 (configs/gtav/ripu.yaml); hyper = "
                         "entropy x hyper (K=100 radius bins), the defaults.py:69 purity")
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--feat-dtype", choices=["f64", "f32"], default="f64",
                    help="f64 = what the reference's hyperbolic head hands over (hyperbolic.py:37)")
    ap.add_argument("--source", choices=["fullres", "lowres"], default="fullres",
                    help="fullres = SURVEY 8(d) unit of work (default, the BASELINE metric); lowres = the "
                         "RegionSelection boundary: x4 low-res head outputs, upsampling fused into the scorer (N1)")
    ap.add_argument("--lr-mode", choices=["gram", "exact"], default="exact",
                    help="--source lowres only: 'exact' (the product's default) interpolates every channel, bit-identical to "
                         "upsample-then-score; 'gram' (opt-in) evaluates a float64 embedding's radius through per-cell Gram terms "
                         "(SURVEY 8f N1), guarded against cancellation")
    ap.add_argument("--cpu-images", type=int, default=8, help="timed images in the CPU-baseline sample, after one "
                                                              "untimed warm-up image (0 = skip)")
    ap.add_argument("--resets", choices=["undo", "kernel", "fills", "side"], default="undo",
                    help="how a slot's round-1 state (active = selected = False, active_mask = 255: what the loader hands over in "
                         "the reference, cityscapes.py:245-251 -- harness work, not part of the unit) is restored before the slot is "
                         "scored again: undo = halo_undo_picks behind the slot's selection rewrites exactly the windows that "
                         "selection wrote (default; the state is checked after the run), kernel = halo_reset_round_state rewrites "
                         "all 17 bytes per pixel on the scoring stream, fills = three torch fills there (round 2), side = the "
                         "fills on a housekeeping stream")
    ap.add_argument("--dump-tables", default="", help="rank 0 writes the round's gathered pick tables / counts / owners (pool "
                                                      "order) to this .npz after the timed region (tests compare world sizes)")
    ap.add_argument("--tail", choices=["auto", "split", "inline"], default="auto",
                    help="split = the scorer's tail kernels on their own stream, forked behind the feature pass (halo_score_maps_split), "
                         "so that they overlap the next step's feature kernel; inline = on the scoring stream, between two feature "
                         "kernels; auto = split for the 'hyper' purity (0.95 ms of tail: +6 %%), inline otherwise (0.26 ms of tail: "
                         "beside the next feature kernel it costs that kernel 0.5-0.9 ms, profiles/archive/r03_tail_split.txt)")
    ap.add_argument("--settle", type=float, default=5.0,
                    help="seconds to wait before the GPU is touched when the resident pool is large (> 8 GiB): a run that starts "
                         "within a few seconds of the end of another large GPU process measures ~3 %% low -- the driver is still "
                         "busy with the memory that process gave back (profiles/archive/r03_process_alternation.txt); 0 = do not wait")
    ap.add_argument("--sel-priority", type=int, default=-1, help="stream priority of the selection streams (-1 = high)")
    ap.add_argument("--data", default="gaussian",
                    help="synthetic value distribution, '+'-joined modifiers of the SURVEY 8(d) default ('gaussian'): late_round = half of the "
                         "pixels already active (-inf) in 11x11 blocks, as after several acquisition rounds; saturated = latents x 40 over "
                         "the right half of the image (embeddings projected onto the ball's boundary: one exact radius there); peaked = "
                         "logits x 30 (saturated softmax: exactly equal entropies over large regions).  The selector's hand-over counters "
                         "(`selection`) say what each does to the value-binned sweep; plateau = the contrived worst case, a quarter of the image "
                         "with all-equal logits AND boundary embeddings: one exact score at the TOP of the map, every pick a tie-break")
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--width", type=int, default=W)
    return ap.parse_args()


DATA_MODS = ("gaussian", "late_round", "saturated", "peaked", "plateau")


def make_ring(dev, R, C, Hh, Ww, fdtype, seeds, lowres=False, data=("gaussian",)):
    """Synthetic pool per SURVEY.md 8(d): low-res latent z ~ N(0, 0.1^2), seed 1234 + image id (`seeds[r]` for ring slot r);
    embed = expmap0_project(z); logit = HyperMLR(embed), P/A ~ kaiming_uniform(a=sqrt 5) seed 7;
    both upsampled x4 (align_corners) -- all by this package's own kernels, untimed."""
    from halo_amd.core.utils.hyperbolic import HyperMapper, HyperMLR, bilinear_align_corners
    h, w = Hh // 4, Ww // 4
    mapper = HyperMapper(c=1.0)
    torch.manual_seed(7)
    mlr = HyperMLR(C, O, c=1.0).to(dev)
    fs, ls = ((h, w), (h, w)) if lowres else ((Hh, Ww), (Hh, Ww))
    feat = torch.empty((R, C) + fs, dtype=fdtype, device=dev)
    logit = torch.empty((R, O) + ls, dtype=torch.float32, device=dev)
    gt = torch.empty((R, Hh, Ww), dtype=torch.int64, device=dev)
    with torch.no_grad():
        for r in range(R):
            g = torch.Generator(device=dev).manual_seed(1234 + int(seeds[r]))
            z = torch.randn((1, C, h, w), generator=g, device=dev, dtype=torch.float32) * 0.1
            if "saturated" in data:
                z[..., w // 2:] *= 40.0          # tanh saturates: project() puts these vectors ON the ball's boundary
            emb = mapper.expmap(z, dim=1)
            lg = mlr._hyper_logits(emb, out_dtype=torch.float32)
            if "peaked" in data:
                lg *= 30.0                       # softmax saturates: p = 1 / 0 exactly over large regions
            if "plateau" in data:
                # all-equal logits (uniform softmax: the maximal entropy, the same bits in every pixel) on embeddings of norm 1.001
                # (outside the ball: dist0 clamps its argument to 1 - 1e-7, one exact radius): the TOP score of the map, exactly tied
                # over a quarter of the image -- every pick is a tie-break by position
                lg[..., : h // 2, : w // 2] = 0.0
                emb[..., : h // 2, : w // 2] = 1.001 / math.sqrt(C)
            if lowres:
                logit[r:r + 1] = lg
                feat[r:r + 1] = emb if fdtype == torch.float64 else emb.float()
            else:
                logit[r:r + 1] = bilinear_align_corners(lg, (Hh, Ww))
                up = bilinear_align_corners(emb, (Hh, Ww))
                feat[r:r + 1] = up if fdtype == torch.float64 else up.float()
                del up
            lab = torch.randint(0, O, (Hh, Ww), generator=g, device=dev, dtype=torch.int64)
            lab[torch.rand((Hh, Ww), generator=g, device=dev) < 0.05] = 255
            gt[r] = lab
        prior = None
        if "late_round" in data:                 # what earlier rounds left behind: `active` windows (build.py:56-57) over half the image
            prior = torch.empty((R, Hh, Ww), dtype=torch.bool, device=dev)
            for r in range(R):
                g = torch.Generator(device=dev).manual_seed(99991 + int(seeds[r]))
                blocks = torch.rand(((Hh + 10) // 11, (Ww + 10) // 11), generator=g, device=dev) < 0.5
                prior[r] = blocks.repeat_interleave(11, 0).repeat_interleave(11, 1)[:Hh, :Ww]
    torch.cuda.synchronize(dev)
    return feat, logit, gt, prior


BRANCHES = {   # name -> (unc_type, pur_type, normalize, mask radius, K)
    "halo": ("entropy", "radius", True, 5, 100),      # configs/gtav/source_target.yaml (HYPER head)
    "ripu": ("entropy", "ripu", False, 3, 100),       # configs/gtav/ripu.yaml:23-27
    "hyper": ("entropy", "hyper", True, 5, 100),      # core/configs/defaults.py:66-79
}


class Pipeline:
    """Two-stream pipeline: score(batch s+1) on `s_score` overlaps select(batch s) on `s_sel`; the round's tables
    collect in one wire block that is exchanged ONCE, behind the last step."""

    def __init__(self, dev, feat, logit, gt, B, n_regions, rows, depth, lowres=False, branch="halo",
                 resets="undo", sel_priority=-1, lr_mode="exact", tail="auto", prior=None):
        from halo_amd import _lib
        self.lib = _lib.lib()
        self.dev, self.feat, self.logit, self.gt, self.B, self.n = dev, feat, logit, gt, B, n_regions
        self.R = feat.shape[0]
        self.lowres = lowres
        self.lr_mode = lr_mode
        self.unc, self.pur, self.norm, self.mrad, self.K = BRANCHES[branch]
        Hh, Ww = gt.shape[-2:]
        self.size = (Hh, Ww)
        # Scoring is bandwidth-bound, the greedy selector is latency-bound (dependent steps, one workgroup
        # per image): keep `depth` batches in flight, each slot selecting on its own stream, so that
        # selection runs beside the scoring of later batches.
        D = self.D = depth
        self.s_score = torch.cuda.Stream(dev)
        self.s_sel = [torch.cuda.Stream(dev, priority=sel_priority) for _ in range(D)]   # -1: dispatch ahead of scoring
        self.s_comm = torch.cuda.Stream(dev)                # the round's one exchange: normal priority, its own stream
        from halo_amd.core.active.floating_region import score_dtype
        sdt = score_dtype(self.pur, feat)
        self.score = [torch.empty((B, Hh, Ww), dtype=sdt, device=dev) for _ in range(D)]
        self.active = [torch.zeros((B, Hh, Ww), dtype=torch.bool, device=dev) for _ in range(D)]
        self.selected = [torch.zeros((B, Hh, Ww), dtype=torch.bool, device=dev) for _ in range(D)]
        self.amask = [torch.full((B, Hh, Ww), 255, dtype=torch.int64, device=dev) for _ in range(D)]
        from halo_amd.core.active.floating_region import new_score_range
        # normalised maps: the scorer bounds their value range for free and the selector skips its range pass
        self.rng = [new_score_range(B, dev) if self.norm else None for _ in range(D)]
        self.prior = prior                 # (R,H,W) bool or None: the `active` map each image enters the round with
        self.resets = "prior" if prior is not None else resets
        self.s_house = torch.cuda.Stream(dev)
        self.reset_done = [torch.cuda.Event() for _ in range(D)]
        self.scored = [torch.cuda.Event() for _ in range(D)]
        self.selected_done = [torch.cuda.Event() for _ in range(D)]
        # the scorer's tail (min / max, normalise, product, mask: 0.26 ms of small kernels) on its own high-priority stream,
        # forked behind the feature pass: it overlaps the NEXT step's feature kernel instead of standing between two of them
        self.split = (tail == "split" or (tail == "auto" and self.pur == "hyper")) and not lowres and self.pur in ("radius", "euc_norm", "hyper")
        if self.split:
            from halo_amd.core.active.floating_region import score_workspace
            self.s_tail = torch.cuda.Stream(dev, priority=-1)
            self.ws = [score_workspace(B, Hh, Ww, dev) for _ in range(D)]                 # one per call in flight
            self.maps = [(torch.empty((B, Hh, Ww), dtype=sdt, device=dev), torch.empty((B, Hh, Ww), dtype=torch.float32, device=dev))
                         for _ in range(D)]
            self.fork_ev = [(self.lib.halo_event_create(), self.lib.halo_event_create()) for _ in range(D)]   # untimed steps
        self.ev = []                       # (start, stop) HIP events around k_feat_reduce
        self.ev_lr = []                    # low-res source: (logit start, logit stop, embedding start, embedding stop, images)
        self.tables = [torch.zeros((B, n_regions, 3), dtype=torch.float64, device=dev) for _ in range(D)]      # warm-up steps
        self.counts = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(D)]
        # the round's tables of this rank: the selector writes each timed step's rows in place; packed into the wire block
        # (halo_amd/pool.py, padded to ceil(N / world) rows) ONCE, right before the round's one collective
        self.round_tables = torch.zeros((rows, n_regions, 3), dtype=torch.float64, device=dev)
        self.round_counts = torch.zeros((rows,), dtype=torch.int32, device=dev)
        # what the value-binned sweep did with each image: {reason, picks before the hand-over} (halo_greedy_select_ex)
        self.handover = [torch.zeros((B, 2), dtype=torch.int32, device=dev) for _ in range(D)]
        self.round_handover = torch.zeros((rows, 2), dtype=torch.int32, device=dev)
        self.wire = torch.zeros((rows, 3 * n_regions + 1), dtype=torch.int32, device=dev)
        self.rows_done = 0
        self.slot_out = [None] * D
        self.step_no = 0
        self.slot_lo = [None] * D          # which ring slice each slot last processed
        self.ref_tables = {}               # ring offset -> pick table seen first (the ring repeats: tables must too)
        self.tables_consistent = True
        self.min_picked = n_regions
        self.exchange_ms = None

    def _restore_state(self, k):
        from halo_amd.pool import reset_round_state
        if self.resets == "kernel":
            reset_round_state(self.active[k], self.selected[k], self.amask[k])
        else:
            self.active[k].zero_()
            self.selected[k].zero_()
            self.amask[k].fill_(255)

    def step(self, timed, b=None, lo=None, row=None):
        """One batch of `b` (default B) images starting at ring slot `lo`: score on s_score, then mask + select on the
        slot's stream; `row` (timed steps): where the batch's tables go in the round's wire block."""
        from halo_amd.core.active.build import greedy_select
        from halo_amd.core.active.floating_region import score_maps, score_maps_lowres
        from halo_amd.pool import undo_picks
        B, R = self.B, self.R
        b = B if b is None else b
        k = self.step_no % self.D
        if lo is None:
            lo = (self.step_no * B) % R
        if lo + b > R:
            lo = 0
        fb, lb, gb = self.feat[lo:lo + b], self.logit[lo:lo + b], self.gt[lo:lo + b]
        evs = None
        timed_kernel = timed and not self.lowres and self.pur in ("radius", "euc_norm", "hyper")
        if timed_kernel:
            evs = (self.lib.halo_event_create(), self.lib.halo_event_create())
            self.ev.append(evs + (b,))
        with torch.cuda.stream(self.s_score):
            # round-1 state for this batch (the loader's job in the reference, cityscapes.py:245-251).
            # Kept on the scoring stream by default: moving the fills to the slot's select stream measured
            # 5-8 % SLOWER end to end (they then run at high priority beside the feature stream).
            if self.resets == "prior":                               # three torch ops per step: harness work on the scoring stream
                self.s_score.wait_event(self.selected_done[k])
                self.active[k][:b].copy_(self.prior[lo:lo + b])
                self.selected[k].zero_()
                self.amask[k].fill_(255)
            elif self.resets in ("kernel", "fills"):
                self.s_score.wait_event(self.selected_done[k])      # buffers k free again
                self._restore_state(k)
            elif self.resets == "undo" or self.step_no < self.D:
                self.s_score.wait_event(self.selected_done[k])      # undo: the restore ran behind the slot's selection
            else:
                self.s_score.wait_event(self.reset_done[k])
            # all three output maps of FloatingRegionScore.forward are written (floating_region.py:217)
            if self.lowres:
                lev = None
                if timed:
                    lev = tuple(self.lib.halo_event_create() for _ in range(6))
                    self.ev_lr.append(lev + (b,))
                sc, self.imp, self.unc_map = score_maps_lowres(lb, fb, self.size, self.unc, self.pur, self.norm, gb, ksize=3,
                                                               K=self.K, c=1.0, active=self.active[k][:b], want_maps=True,
                                                               mode=self.lr_mode, events=lev,
                                                               score_range=None if self.rng[k] is None else self.rng[k][:b])
                self.score[k][:b].copy_(sc)
            else:
                kw = {}
                if self.split:
                    kw = dict(tail_stream=self.s_tail, workspace=self.ws[k], maps=(self.maps[k][0][:b], self.maps[k][1][:b]))
                _, self.imp, self.unc_map = score_maps(lb, fb, self.unc, self.pur, self.norm, gb, size=3, K=self.K, c=1.0,
                                                       active=self.active[k][:b], want_maps=True, out=self.score[k][:b],
                                                       events=evs if evs is not None or not self.split else self.fork_ev[k],
                                                       score_range=None if self.rng[k] is None else self.rng[k][:b], **kw)
            self.scored[k].record(self.s_tail if self.split else self.s_score)
        with torch.cuda.stream(self.s_sel[k]):
            self.s_sel[k].wait_event(self.scored[k])
            dst = (self.tables[k][:b], self.counts[k][:b]) if row is None else (self.round_tables[row:row + b], self.round_counts[row:row + b])
            picks, npk = greedy_select(self.score[k][:b], self.n, 1, self.mrad, self.active[k][:b], self.selected[k][:b],
                                       self.amask[k][:b], gb, out=dst, score_range=None if self.rng[k] is None else self.rng[k][:b],
                                       handover=self.handover[k][:b] if row is None else self.round_handover[row:row + b])
            self.slot_out[k] = dst
            if row is not None:
                self.rows_done = max(self.rows_done, row + b)
            if self.resets == "undo":
                undo_picks(picks, npk, 1, self.mrad, self.active[k][:b], self.selected[k][:b], self.amask[k][:b])
            self.selected_done[k].record(self.s_sel[k])
        if self.resets == "side":
            with torch.cuda.stream(self.s_house):
                self.s_house.wait_event(self.selected_done[k])
                self._restore_state(k)
                self.reset_done[k].record(self.s_house)
        self.slot_lo[k] = (lo, b)
        self.step_no += 1

    def finish_round(self, n_images, host_backend=False, warm=False):
        """The path's one exchange step: behind the last selection of the round, ONE all-gather of the
        rank's wire block on the communication stream.  Returns (tables, counts) of the whole pool in pool order."""
        from halo_amd.pool import gather_wire, pack_tables_into
        with torch.cuda.stream(self.s_comm):
            for e in self.selected_done:
                self.s_comm.wait_event(e)
            # the round's exchange format: one launch for the whole block (warm-up: any rows, the tables are still zero)
            n = min(self.B, self.wire.shape[0]) if warm else self.rows_done
            pack_tables_into(self.wire[:n], self.round_tables[:n], self.round_counts[:n])
            if host_backend:
                self.s_comm.synchronize()
                t0 = time.perf_counter()
                out = gather_wire(self.wire, n_images, self.n)
                self.s_comm.synchronize()
                self.exchange_ms = (time.perf_counter() - t0) * 1e3
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.s_comm)
                out = gather_wire(self.wire, n_images, self.n)
                e1.record(self.s_comm)
                self._exchange_events = (e0, e1)
        return out

    def drain(self, check=True):
        """Wait for every stream of the pipeline; `check=False` (the timed region) leaves the host-synchronising self-checks --
        full-map reductions, table clones and compares -- to a later self_check() call."""
        self.s_score.synchronize()
        for st in self.s_sel:
            st.synchronize()
        self.s_house.synchronize()
        if self.split:
            self.s_tail.synchronize()
        self.s_comm.synchronize()
        ee = getattr(self, "_exchange_events", None)
        if ee is not None:
            self.exchange_ms = ee[0].elapsed_time(ee[1])
            self._exchange_events = None
        if check:
            self.self_check()

    def self_check(self):
        """Outside the timed region: the round-1 state the undo kernel must have left, and the same ring images giving the
        same pick tables whichever slot / step / overlap pattern processed them."""
        if self.resets == "undo" and self.step_no >= self.D:      # the restore must have left exactly the loader's round-1 state
            for k in range(self.D):
                assert not bool(self.active[k].any()) and not bool(self.selected[k].any()) and bool((self.amask[k] == 255).all()), \
                    "halo_undo_picks did not restore the round-1 state of slot %d" % k
        for k, ent in enumerate(self.slot_lo):
            if ent is None:
                continue
            lo, b = ent
            tab, cnt = self.slot_out[k]
            self.min_picked = min(self.min_picked, int(cnt.min()))
            if lo not in self.ref_tables:
                if b == self.B:
                    self.ref_tables[lo] = tab.clone()
            elif not torch.equal(self.ref_tables[lo][:b], tab):
                self.tables_consistent = False

    def lowres_pass_ms(self):
        """per full-batch timed step of the low-res source: {logit, feat (embedding pass), gram, radius (the two kernels of the gram
        route; None otherwise), tail} -> list of ms"""
        import ctypes
        out = {"logit": [], "feat": [], "gram": [], "radius": [], "tail": []}

        def ms(a, b):
            v = ctypes.c_float(0)
            return v.value if self.lib.halo_event_elapsed_ms(a, b, ctypes.byref(v)) == 0 else None
        for e0, e1, e2, e3, e4, e5, nimg in self.ev_lr:
            vals = {"logit": ms(e0, e1), "feat": ms(e2, e3), "tail": ms(e3, e5)}
            if self.lr_mode == "gram" and self.feat.dtype == torch.float64:
                vals["gram"], vals["radius"] = ms(e2, e4), ms(e4, e3)
            if nimg == self.B:
                for k, v in vals.items():
                    if v is not None:
                        out[k].append(v)
            for e in (e0, e1, e2, e3, e4, e5):
                self.lib.halo_event_destroy(e)
        self.ev_lr = []
        return out

    def feat_kernel_ms(self):
        import ctypes
        out = []
        for a, b, nimg in self.ev:
            ms = ctypes.c_float(0)
            if self.lib.halo_event_elapsed_ms(a, b, ctypes.byref(ms)) == 0 and nimg == self.B:      # full batches only
                out.append(ms.value)
            self.lib.halo_event_destroy(a)
            self.lib.halo_event_destroy(b)
        self.ev = []
        return out


def effective_cpus():
    """Host cores this process may actually use: min(visible CPUs, scheduler affinity, cgroup CPU quota).  The GPU
    boxes of this pool show 256 CPUs under a cgroup quota of 16 (cpu.max "1600000 100000"): an OpenMP team wider
    than the quota only gets throttled."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def parity_slots(R, B, n_rows, want):
    """Ring slots the oracle scores: spread over EVERY resident batch (offsets 0, B/3, 2B/3, B-1 of each), so that images
    whose feature tensor starts past 2^32 elements of the batch's allocation are covered, not only image 0."""
    offs = sorted({0, B // 3, (2 * B) // 3, B - 1})
    slots = [lo + o for lo in range(0, R, B) for o in offs if lo + o < min(R, n_rows)]
    if len(slots) > want:                      # keep the spread: every (len/want)-th slot, first and last included
        idx = sorted({round(i * (len(slots) - 1) / max(1, want - 1)) for i in range(want)})
        slots = [slots[i] for i in idx]
    return slots


def cpu_baseline(feat, logit, gt, prior, slots, n_regions, branch, lowres=False, lr_mode="exact"):
    """The CPU oracle (kind 'port': C restatement, OpenMP over all host cores) on ring images `slots` (after one untimed
    warm-up pass over the first of them): score + mask + select, median s/image -> images/s.  Returns every
    image's pick table so the caller can compare them with the rows the TIMED pipeline wrote for the same images."""
    import ctypes
    from oracle import halo_oracle as ho
    ho.lib()
    unc, pur, norm, mrad, K = BRANCHES[branch]
    cores = effective_cpus()
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)        # the oracle's OpenMP team = the usable cores
    except OSError:
        cores = os.cpu_count() or 1
    times, picks = [], {}
    for j, i in enumerate([slots[0]] + list(slots)):
        lg = logit[i].cpu().numpy()
        ft = feat[i].cpu().numpy()
        g = gt[i].cpu().numpy()
        Hh, Ww = g.shape
        act0 = prior[i].cpu().numpy() if prior is not None else np.zeros((Hh, Ww), bool)
        t0 = time.perf_counter()
        raw = None
        if lowres:      # the RegionSelection boundary: the reference resizes both head outputs first (build.py:122-135) -- part of the unit
            lg = ho.bilinear(lg[None], (Hh, Ww))
            if lr_mode == "gram" and ft.dtype == np.float64:
                raw = ho.gram_radius(ft[None], (Hh, Ww), "euc_norm" if pur == "euc_norm" else "radius", 1.0)
                ft = None
            else:
                ft = ho.bilinear(ft[None], (Hh, Ww))
        s, _, _ = ho.floating_region_score(lg, ft, unc, pur, norm, g, size=3, purity_type=pur, K=K, **({"impurity_raw": raw} if raw is not None else {}))
        act = act0.copy(); sel = np.zeros((Hh, Ww), bool); am = np.full((Hh, Ww), 255, np.int64)
        s[act] = -np.inf
        _, _, _, _, pk = ho.select_pixels_to_label(s, n_regions, 1, mrad, act, sel, am, g, True)
        if j > 0:
            times.append(time.perf_counter() - t0)
        picks[i] = pk
    return float(np.median(times)), cores, picks


def visible_devices():
    """Number of ROCm devices a child process of this one would see -- asked of a short-lived CHILD, so that the
    launcher parent never brings up the HIP runtime (on ROCm torch.cuda.device_count() falls back to hipGetDeviceCount
    when amdsmi is absent, which initialises HIP/HSA in the caller)."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes through
    torch.distributed.run and relay their output.  This process never touches the GPU (the device count comes from a
    child too) and only waits for the children -- no exec."""
    import socket
    import subprocess
    share = bool(os.environ.get("HALO_BENCH_SHARE_GPU"))
    have = visible_devices()
    if have < (1 if share else n):
        sys.exit("bench.py: --gpus %d but only %d ROCm device(s) visible" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if "RANK" not in os.environ and (a.gpus > 1 or os.environ.get("HALO_BENCH_SPAWN")):    # the env forces the launcher at N = 1 (test hook)
        sys.exit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    # host pools no wider than this rank's share of the cgroup quota (8 ranks x 16 threads on a 16-core quota only throttle)
    host_threads = halo_amd.host_threads_per_rank()                       # usable cores // LOCAL_WORLD_SIZE
    torch.set_num_threads(max(1, min(torch.get_num_threads(), host_threads)))
    pool_bytes = max(a.batch, (a.ring // max(1, a.batch)) * a.batch) * a.channels * a.height * a.width * (8 if a.feat_dtype == "f64" else 4)
    settled = a.settle if (a.settle > 0 and a.source == "fullres" and pool_bytes > (8 << 30)) else 0.0
    if settled:
        time.sleep(settled)                  # before the first HIP call of this process
    assert torch.cuda.is_available(), "bench.py needs ROCm devices"
    # HALO_BENCH_BACKEND=gloo + HALO_BENCH_SHARE_GPU=1: the hardware test of the N > 1 code on a ONE-GPU box -- every rank
    # on device 0, the wire block staged through the host.  Never set for a measurement.
    backend = os.environ.get("HALO_BENCH_BACKEND", "nccl")
    share_gpu = bool(os.environ.get("HALO_BENCH_SHARE_GPU"))
    dev = torch.device("cuda", local % torch.cuda.device_count() if share_gpu else local)
    torch.cuda.set_device(dev)
    # launched by torch.distributed.run (RANK set): one process per GPU over RCCL, also at world 1
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    device_ids = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world
    # one process per GPU: two ranks on one device would halve every number silently -- every rank's device identity (PCI bus
    # id + UUID) is exchanged and must be distinct, unless the test switch shares the one GPU of a test box on purpose
    from halo_amd.pool import device_identity
    device_ids = [device_identity(dev.index)]
    if use_dist:
        ids = [None] * world
        dist.all_gather_object(ids, device_ids[0])
        device_ids = ids
        if not share_gpu and len(set(device_ids)) != world:
            raise RuntimeError("ranks share a GPU: %s" % ", ".join("rank %d -> %s" % (r, i) for r, i in enumerate(device_ids)))
    host_backend = use_dist and backend != "nccl"
    fdtype = torch.float64 if a.feat_dtype == "f64" else torch.float32
    a.depth = max(1, a.depth)
    Hh, Ww, C, B = a.height, a.width, a.channels, a.batch
    R = max(B, (a.ring // B) * B)
    n_regions = math.ceil(Hh * Ww * (0.05 / 5) / 9)                    # build.py:148-150 -> 2331
    unc, pur, norm, mrad, K = BRANCHES[a.branch]

    lowres = a.source == "lowres"
    if lowres:
        a.cpu_images = min(a.cpu_images, 2)          # the oracle also resizes (4.3 GB per image at C = 256): a smaller sample
    data = tuple(a.data.split("+"))
    for m_ in data:
        if m_ not in DATA_MODS:
            sys.exit("bench.py: --data modifier %r (known: %s)" % (m_, ", ".join(DATA_MODS)))

    # ---- the pool and this rank's block of it (halo_amd.pool.shard_range: contiguous ceil(N/world) images per rank)
    from halo_amd.pool import shard_range
    pool_mode = a.pool_images > 0
    n_pool = a.pool_images if pool_mode else world * a.steps * B
    lo_r, hi_r = shard_range(n_pool, rank, world)
    rows = math.ceil(n_pool / world)
    n_local = hi_r - lo_r
    # full batches, plus a partial last one when the block does not divide evenly
    sched = [B] * (n_local // B) + ([n_local % B] if n_local % B else [])
    if pool_mode:
        # pool image g has content id g % R on EVERY rank (so the pool's results do not depend on the world size and
        # any rank can check any other rank's rows); rank r's ring is that base ring rotated by its block's offset
        seeds = [(lo_r + s_) % R for s_ in range(R)]
        a.steps = math.ceil(rows / B)                                  # steps of the largest block
    else:
        seeds = [rank * R + s_ for s_ in range(R)]                      # every rank its own R images
    feat, logit, gt, prior = make_ring(dev, R, C, Hh, Ww, fdtype, seeds, lowres, data)
    pipe = Pipeline(dev, feat, logit, gt, B, n_regions, rows, a.depth, lowres, a.branch, a.resets, a.sel_priority, a.lr_mode, a.tail,
                    prior=prior)

    for _ in range(a.warmup):
        pipe.step(False)
    if a.warmup > 0:
        # one untimed pass of the round's exchange as well: first use of a kernel / of the communicator costs a code-object
        # load or a lazy connection set-up (100+ ms on a fresh box), which is what warm-up steps are for
        pipe.finish_round(n_pool, host_backend, warm=True)
    pipe.drain()
    pipe.exchange_ms = None
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    row = 0
    for b in sched:
        pipe.step(True, b, lo=row % R, row=row)
        row += b
    tables, counts = pipe.finish_round(n_pool, host_backend)           # ONE collective per round
    pipe.drain(check=False)
    torch.cuda.synchronize(dev)
    dt_rank = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0                                       # the clock stops here, on both sides of the barrier
    pipe.self_check()                                                   # host-synchronising checks: after it
    rank_dts = [dt_rank]
    if use_dist:
        cdev = torch.device("cpu") if host_backend else dev
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        allt = torch.zeros((world,), dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allt, torch.tensor([dt_rank], dtype=torch.float64, device=cdev))
        rank_dts = [float(x) for x in allt.cpu()]

    # ---- what the exchange delivered (outside the timed region): this rank's rows are its local tables, and -- pool
    # mode -- every other rank's rows equal this rank's own results for the same content
    assert tables.shape[0] == n_pool and counts.shape[0] == n_pool
    if sched:      # what came back for this rank's block is what its selector wrote
        assert torch.equal(tables[lo_r:hi_r], pipe.round_tables[:n_local]), "gathered tables differ from the local ones"
        assert torch.equal(counts[lo_r:hi_r], pipe.round_counts[:n_local])
    if prior is None:       # (half of an image already active can leave fewer than n_regions pickable windows)
        assert int(counts.min()) == n_regions, "a gathered image has fewer picks than regions"
    exchange_checked = 0
    if pool_mode and n_local > 0:
        g_ = torch.arange(n_pool, device=dev)
        j_ = (g_ - lo_r) % R                                            # local image with the same content id
        known = j_ < n_local
        exchange_checked = int(known.sum())
        ref = tables[(lo_r + j_)[known]]
        assert torch.equal(tables[known], ref), "rows gathered from other ranks differ from this rank's results for the same images"

    feat_ms = pipe.feat_kernel_ms()
    lr_ms = pipe.lowres_pass_ms()
    lr_logit_ms, lr_feat_ms = lr_ms["logit"], lr_ms["feat"]
    batch_alone = None
    if rank == 0 and not lowres and feat_ms and os.environ.get("HALO_BENCH_BATCH_ALONE"):
        # diagnostic: the scoring call on each resident batch with nothing beside it, after the timed region
        from halo_amd.core.active.floating_region import score_maps
        batch_alone = []
        for lo_ in range(0, R, B):
            score_maps(logit[lo_:lo_ + B], feat[lo_:lo_ + B], "entropy", "radius", True, None, size=3)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                score_maps(logit[lo_:lo_ + B], feat[lo_:lo_ + B], "entropy", "radius", True, None, size=3)
            e1.record(); torch.cuda.synchronize(dev)
            batch_alone.append(round(e0.elapsed_time(e1) / 4, 3))
    flat = None
    if rank == 0 and not lowres and feat_ms:
        # this box's own ceiling for the same bytes, after the timed region: a flat non-temporal read (no arithmetic, no
        # plane structure, nothing written) of the B feature tensors one k_feat_reduce launch streams, nothing beside it
        # (tools/libhalo_probe.so: a measurement aid outside the product ABI; absent -> the field is null)
        try:
            from tools import halo_probe
            fb = feat[0:B]
            assert fb.is_contiguous()
            flat = halo_probe.flat_read_gbps(fb)
            flat["what"] = "flat non-temporal read of the same feature tensors, alone, after the timed region (tools/halo_probe.hip)"
        except Exception as exc:                  # never part of the measurement
            flat = None
            print("bench.py: flat-read probe unavailable (%s)" % exc, file=sys.stderr)
    assert prior is not None or pipe.min_picked == n_regions, "selection stopped early"
    # every rank's own k_feat_reduce average (HIP events on its scoring stream): the N > 1 line carries the per-rank roofline
    # fractions beside the images/s (north_star: "HBM-bandwidth fraction reported in the same run")
    my_feat_ms = float(np.mean(feat_ms)) if feat_ms else float("nan")
    rank_feat_ms = [my_feat_ms]
    if use_dist:
        cdev = torch.device("cpu") if host_backend else dev
        allf = torch.zeros((world,), dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allf, torch.tensor([my_feat_ms], dtype=torch.float64, device=cdev))
        rank_feat_ms = [float(x) for x in allf.cpu()]
    # what the value-binned sweep did with this rank's images (a cost counter; rank 0's block is reported)
    ho_ = pipe.round_handover[:n_local].cpu().numpy() if n_local else np.zeros((0, 2), np.int32)
    assert pipe.tables_consistent, "pick tables of the same images differ between steps (race in the pipeline)"

    if rank == 0:
        esz = 8 if fdtype == torch.float64 else 4
        ssz = pipe.score[0].element_size()
        images = n_pool
        value = images / dt
        # k_feat_reduce per launch.  kernel_bytes: what the kernel itself moves -- features read + radius map written + (fused)
        # logits read + entropy map written.  launch_bytes: the ALGORITHMIC bytes of SURVEY.md 8(d) for the images of one launch --
        # H W (C s_feat + O 4 + s_score) each -- which is what `achieved` / `frac` are computed from (0.2 % less).
        kernel_bytes = B * Hh * Ww * (C * esz + esz + O * 4 + 4)
        uses_feat = pur in ("radius", "euc_norm", "hyper")
        path_bytes_per_image = Hh * Ww * ((C * esz if uses_feat else 0) + O * 4 + ssz)
        launch_bytes = B * path_bytes_per_image
        avg_ms = float(np.mean(feat_ms)) if feat_ms else float("nan")
        achieved = launch_bytes / (avg_ms * 1e-3) / 1e9 if feat_ms else float("nan")
        out = {
            "metric": "acquisition-scored images/sec (1024x2048, C=256, 19 cls)",
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if pool_mode else "weak",       # --pool-images fixes the total work, --steps the per-GPU work
            "vs_baseline": None, "dtype": a.feat_dtype, "data": "synthetic",
            "config": {"workload": "configs[%d]: synthetic pool of %d image evaluations (%s), %dx%d, C=%d %s embedding, %d classes, %s branch "
                                   "(%s x %s, %s, 3x3), %d regions/image, radius 1, mask radius %d"
                                   % (2 if a.pool_images == 2975 else 1, images,
                                      "the Cityscapes train list" if a.pool_images == 2975 else "configs[1] names 500", Hh, Ww, C, a.feat_dtype,
                                      O, a.branch.upper(), unc, pur, "normalised" if norm else "not normalised", n_regions, mrad),
                       "data": a.data,
                       "images_per_step_per_gpu": B, "batches_in_flight": a.depth, "resident_ring": R, "image_evaluations": images,
                       "outputs_written": "score, region_impurity, prediction_uncertainty (+ masks, pick tables)",
                       "sharding": "image-wise, %d rank(s), contiguous blocks of %d image(s)%s"
                                   % (world, rows, (", one %s all-gather of pick tables per round" % ("RCCL" if backend == "nccl" else backend))
                                      if use_dist else "")},
            "roofline": {"bound": "hbm", "kernel": "k_feat_reduce", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "bytes_per_launch": launch_bytes, "bytes_per_image_survey_8d": path_bytes_per_image, "images_per_launch": B,
                         "kernel_bytes_per_launch": kernel_bytes,
                         "frac_of_kernel_bytes": round(kernel_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if feat_ms else None,
                         "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(feat_ms),
                         "avg_launch_ms_even_odd_steps": [round(float(np.mean(feat_ms[0::2])), 4), round(float(np.mean(feat_ms[1::2])), 4)] if len(feat_ms) > 1 else None,
                         "scoring_call_alone_ms_per_batch": batch_alone,
                         "launch_ms": [round(v, 3) for v in feat_ms] if os.environ.get("HALO_BENCH_LAUNCH_MS") else None,
                         "flat_read": flat, "frac_of_flat_read": None if not flat else round(achieved / flat["GB/s"], 4)},
            "path_algorithmic_GBps": round(path_bytes_per_image * value / world / 1e9, 1),
            "pipeline_tables_consistent": bool(pipe.tables_consistent),
            "exchange": {"collectives_per_round": 1 if use_dist else 0, "ms": None if pipe.exchange_ms is None else round(pipe.exchange_ms, 3),
                         "bytes_per_rank": int(pipe.wire.numel() * 4), "rows_checked_against_local_results": exchange_checked,
                         "ranks": dist.get_world_size() if use_dist else 1,
                         "backend": (dist.get_backend() if use_dist else None)},
            # halo_greedy_select_ex's counters for rank 0's images: how many the value-binned sweep handed to the serial kernel
            # (8-14 ms per image instead of 0.06 ms amortised), why, and after how many of its own picks
            "selection": {"images": int(ho_.shape[0]), "handed_over": int((ho_[:, 0] != 0).sum()),
                          "reasons": {_halo_lib.SWEEP_REASONS[r_]: int((ho_[:, 0] == r_).sum()) for r_ in range(1, 6) if (ho_[:, 0] == r_).any()},
                          "sweep_picks_before_handover": None if not (ho_[:, 0] != 0).any() else
                          {"min": int(ho_[ho_[:, 0] != 0, 1].min()), "mean": round(float(ho_[ho_[:, 0] != 0, 1].mean()), 1)},
                          "picks_per_image": {"min": int(pipe.round_counts[:n_local].min()), "max": int(pipe.round_counts[:n_local].max())} if n_local else None},
            "state_resets": a.resets, "host_threads_per_rank": host_threads, "settle_s_before_first_gpu_call": settled,
            "tail": "on its own stream beside the next feature kernel (halo_score_maps_split)" if pipe.split else "inline on the scoring stream",
        }
        per_rank = [(shard_range(n_pool, r_, world)[1] - shard_range(n_pool, r_, world)[0]) / rank_dts[r_] for r_ in range(world)]
        out["per_rank_images_per_s"] = {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3)}
        if feat_ms and not lowres:
            fr = [launch_bytes / (m_ * 1e-3) / 1e9 / HBM_PEAK_GBPS for m_ in rank_feat_ms if m_ == m_]
            if fr:
                out["roofline"]["per_rank_frac"] = {"min": round(min(fr), 4), "max": round(max(fr), 4)}
                out["roofline"]["per_rank_avg_launch_ms"] = [round(m_, 4) for m_ in rank_feat_ms]
        out["devices"] = device_ids                       # rank order; distinct unless HALO_BENCH_SHARE_GPU (test switch)
        out["distinct_devices"] = len(set(device_ids))
        assert share_gpu or out["distinct_devices"] == out["n_gpus"], "n_gpus must equal the number of distinct devices"
        if lowres:      # not the BASELINE unit of work: a different (smaller) input boundary, reported for DESIGN.md
            out["config"]["workload"] = "RegionSelection boundary (N1): x4 low-res head outputs (%dx%d), upsample fused into the scorer, " \
                                        "then the same mask + select; NOT the BASELINE unit of work" % (Hh // 4, Ww // 4)
            out["config"]["lowres_mode"] = a.lr_mode
        if lowres or not feat_ms:
            out["roofline"] = None
            if lowres:
                out["path_algorithmic_GBps"] = None
        if lowres and lr_feat_ms and uses_feat:
            # the embedding pass of the low-res boundary: what bounds it depends on the mode
            h4, w4 = Hh // 4, Ww // 4
            t_feat, t_logit = float(np.mean(lr_feat_ms)), float(np.mean(lr_logit_ms))
            if a.lr_mode == "gram" and fdtype == torch.float64:
                # k_gram_lr + k_radius_gram: the low-res tensor read once, the radius map written (the five Gram maps are
                # intermediates, 5/C of the tensor, not counted)
                nbytes = B * (C * h4 * w4 * 8 + Hh * Ww * 8)
                ach = nbytes / (t_feat * 1e-3) / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": "k_gram_lr + k_radius_gram", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS,
                                   "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "bytes_per_launch": nbytes,
                                   "avg_launch_ms": round(t_feat, 4), "launches_timed": len(lr_feat_ms)}
            else:
                # k_feat_reduce_lr*: per output pixel and channel ATen's bilinear (3 mul + 3 fma: columns first, rows second) and one
                # fma into the sum of squares = 11 flops in 7 VALU slots as upsample-then-reduce executes them.  The kernel shares
                # a source row's column interpolation between the 8 (4) vertically adjacent pixels of a lane: 30-32 slots per channel
                # and 8 pixels, ~3.9 per pixel at this geometry -- `valu_slot_frac` prices the slots
                # it actually issues against the FP64 (FP32) vector issue rate, `frac` the reference-formula flops against the spec.
                flops = 11.0 * B * Hh * Ww * C
                slots = 3.9 * B * Hh * Ww * C          # 8 pixels per lane at this x4 geometry (SQ_INSTS_VALU, profiles/archive/r04_pmc_lowres.json); 4.4 with 4
                peak = 78.6 if fdtype == torch.float64 else 157.3
                ach = flops / (t_feat * 1e-3) / 1e12
                out["roofline"] = {"bound": "valu", "kernel": "k_feat_reduce_lr_dmaf" if fdtype == torch.float64 else "k_feat_reduce_lr", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(ach / peak, 4), "traffic": None, "flops_per_launch": flops,
                                   "avg_launch_ms": round(t_feat, 4), "launches_timed": len(lr_feat_ms),
                                   "valu_slot_frac": round(slots / (t_feat * 1e-3) / 1e12 / (peak / 2.0), 4),
                                   "flops_convention": "the reference's formula (ATen bilinear + square-accumulate), 11 per pixel and channel; the kernel issues ~3.9 of its 7 instructions"}
            out["lowres_passes_ms"] = {"logit_pass(k_logit_maps_lr, f32 VALU-bound)": round(t_logit, 4), "embedding_pass": round(t_feat, 4)}
            # one roofline entry per kernel of the low-res step; `roofline` above stays the embedding pass
            def entry(kernel, bound, work, t_ms, peak, unit, what):
                ach = work / (t_ms * 1e-3) / (1e9 if unit == "GB/s" else 1e12)
                return {"kernel": kernel, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                        "avg_launch_ms": round(t_ms, 4), "work_per_launch": work, "what": what}
            ks = []
            if lr_ms["logit"]:
                # VALU wave-instructions per output pixel at 19 classes (SQ_INSTS_VALU, profiles/archive/r04_pmc_lowres.json; 1162 before the
                # integer forms of the exp / log cores): interpolation 6 + lean softmax / entropy ~52 per class; peak = 256 CUs x 4
                # SIMDs x 32 lanes x 2.4 GHz (fma / mul / add issue in 2 cycles per wave, everything else in 4: tools/micro/op_rate.hip)
                ks.append(entry("k_logit_maps_lr<%d>" % O, "valu", LOGIT_LR_VALU_PER_PX / 19 * O * B * Hh * Ww, float(np.mean(lr_ms["logit"])), 78.6, "T lane-ops/s",
                                "f32 VALU instruction issue (no flops convention: compares, selects and conversions count)"))
            if lr_ms["gram"]:
                ks.append(entry("k_gram_lr2", "hbm", B * (C * h4 * w4 * 8 + 5 * h4 * w4 * 8), float(np.mean(lr_ms["gram"])), HBM_PEAK_GBPS, "GB/s",
                                "low-res float64 embedding read once + five Gram maps written"))
                ks.append(entry("k_radius_gram", "valu", 155.0 * B * Hh * Ww, float(np.mean(lr_ms["radius"])), 19.65, "T lane-ops/s",
                                "float64 VALU issue (16 lanes per clock and SIMD): ~155 instructions per output pixel (taps, 10-term form + "
                                "guard, sqrt, two logs with a division each); also writes the radius map"))
            elif lr_ms["feat"] and out["roofline"]:
                ks.append(dict(out["roofline"]))
            if lr_ms["tail"]:
                ssz_ = pipe.score[0].element_size()
                ks.append(entry("k_box3_minmax + k_minmax_finalize2 + k_combine_box3", "hbm", B * Hh * Ww * (ssz_ + 4 + 4 + 1 + 2 * ssz_ + 4),
                                float(np.mean(lr_ms["tail"])), HBM_PEAK_GBPS, "GB/s",
                                "radius + entropy read (entropy twice: min/max pass and combine), active read, three maps written"))
            out["roofline_kernels"] = ks
        # HBM traffic of the kernel per launch: PMC counters need their own rocprofv3 passes (--pmc FETCH_SIZE / WRITE_SIZE, gfx950
        # correction applied by tools/distill_profiles.py).  HALO_BENCH_PMC=<file>: the summary of passes taken on THIS box in THIS
        # call, of this exact launch shape (tools/collect_profiles.sh) -> `traffic`.  Otherwise the tracked collection's figure is
        # quoted under its own name and `traffic` stays null.
        same_run = os.environ.get("HALO_BENCH_PMC")
        for name in ([same_run] if same_run else []) + ["r06_pmc_summary.json", "r05_pmc_summary.json"]:
            pmc = name if os.path.isabs(name) or name == same_run else os.path.join(ROOT, "profiles", name)
            if out["roofline"] is None or lowres or not os.path.exists(pmc):
                continue
            try:
                rec = json.load(open(pmc))
                if rec.get("batch") == B and rec.get("dtype") == a.feat_dtype and rec.get("shape_HWCO") == [Hh, Ww, C, O]:
                    if name == same_run:
                        out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                        out["roofline"]["traffic_over_bytes_per_launch"] = round(rec["hbm_bytes_per_launch"] / launch_bytes, 4)
                        out["roofline"]["traffic_source"] = "same gpurun call: " + rec.get("command", "rocprofv3 --pmc passes")
                    else:
                        out["roofline"]["traffic_from_profile"] = rec["hbm_bytes_per_launch"]
                        out["roofline"]["traffic_source"] = "profiles/" + name
                    break
            except Exception:
                pass
        if a.cpu_images > 0 and world == 1:
            slots = parity_slots(R, B, n_local, a.cpu_images)
            s_img, cores, picks_cpu = cpu_baseline(feat, logit, gt, prior, slots, n_regions, a.branch, lowres, a.lr_mode)
            out["cpu_baseline"] = {"value": round(1.0 / s_img, 4), "unit": "images/s", "cores": cores, "kind": "port",
                                   "sample": "1 warm-up + %d timed ring images (slots %s), oracle/halo_oracle.c (OpenMP, %d threads = usable "
                                             "host cores of %d visible), median s/image = %.2f"
                                             % (len(slots), ",".join(str(i_) for i_ in slots), cores, os.cpu_count() or 1, s_img),
                                   # the honest anchor for "the reference's PyTorch path on host cores": its own code, measured by the
                                   # survey in the build container (the reference cannot travel to the GPU box)
                                   "reference_pytorch_probe": "SURVEY.md section 6: the reference's own code (torch CPU, 8 threads, geoopt/yacs "
                                                              "stand-ins) scored + selected 0.11-0.17 images/s at this shape"}
            # the TIMED pipeline's own pick tables (B images per launch: ring slot i was written to rows i, i + R, ... of the
            # round's table) against the oracle's picks for the same images -- every occurrence, bit for bit
            rt = pipe.round_tables[:n_local].cpu().numpy()
            rc_ = pipe.round_counts[:n_local].cpu().numpy()
            ok, rows_checked = True, 0
            for i_ in slots:
                want = np.ascontiguousarray(picks_cpu[i_])
                for row_ in range(i_, n_local, R):
                    k_ = int(rc_[row_])
                    same = k_ == len(want) and np.array_equal(np.ascontiguousarray(rt[row_, :k_]).view(np.int64), want.view(np.int64))
                    ok = ok and same
                    rows_checked += 1
            out["parity_vs_cpu"] = bool(ok)
            out["parity_images_checked"] = len(slots)
            out["parity_rows_checked"] = rows_checked
            out["parity_what"] = "pick tables written by the timed %d-image launches vs the oracle, ring slots %s" % (B, ",".join(str(i_) for i_ in slots))
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
        if out.get("parity_vs_cpu") is False:
            sys.exit("bench.py: the timed pipeline's pick tables differ from the CPU oracle's")
    if rank == 0 and a.dump_tables:
        np.savez(a.dump_tables, tables=tables.cpu().numpy(), counts=counts.cpu().numpy(), n_pool=n_pool, world=world)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
