"""Pool side of the round on a real MI355X: the wire-block kernel, the round-1 state kernels, device identity,
and -- the N > 1 code path on the ONE GPU a test box has -- bench.py with TWO ranks sharing the device over gloo, whose
gathered pool tables must equal the one-rank run's for the same pool (reference: core/train_learners.py:307-326 runs the
round on rank 0 only; results must not depend on the world size)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from halo_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def test_wire_block_kernel_equals_host_packing_and_round_trips(dev):
    from halo_amd.pool import pack_tables, pack_tables_into, unpack_tables
    rng = np.random.default_rng(3)
    for (b, n, rows) in ((5, 2331, 7), (1, 1, 1), (3, 300, 3), (16, 583, 19)):
        picks = np.zeros((b, n, 3))
        picks[:, :, 0] = rng.integers(0, 65536, (b, n))
        picks[:, :, 1] = rng.integers(0, 65536, (b, n))
        picks[:, :, 2] = rng.standard_normal((b, n)) * 1e3
        picks[0, 0, 2] = -np.inf
        picks[-1, -1, 2] = np.nan
        picks[0, -1, :2] = (65535, 65535)
        npk = rng.integers(0, n + 1, (b,)).astype(np.int32)
        host = pack_tables(torch.from_numpy(picks), torch.from_numpy(npk), rows)
        devw = pack_tables(torch.from_numpy(picks).to(dev), torch.from_numpy(npk).to(dev), rows)
        assert devw.is_cuda and torch.equal(devw.cpu(), host)
        # into a strided slice of a larger round block (what bench.py / the pool driver do per step)
        block = torch.full((rows + 2, 3 * n + 1), -7, dtype=torch.int32, device=dev)
        pack_tables_into(block[1:1 + b], torch.from_numpy(picks).to(dev), torch.from_numpy(npk).to(dev))
        assert torch.equal(block[1:1 + b].cpu(), host[:b]) and int((block[0] != -7).sum()) == 0 and int((block[1 + b:] != -7).sum()) == 0
        tb, cn = unpack_tables(devw, n)
        assert np.array_equal(cn[:b].cpu().numpy(), npk)
        got = tb[:b].cpu().numpy()
        assert np.array_equal(got[:, :, :2], picks[:, :, :2])
        assert np.array_equal(got[:, :, 2].view(np.int64), picks[:, :, 2].copy().view(np.int64))      # score bits untouched


def test_round_state_kernels(dev):
    """halo_reset_round_state == the loader's three fills; halo_undo_picks restores exactly that state after a selection
    that started from it (any radius, windows clipped at the borders)."""
    from halo_amd.core.active.build import greedy_select
    from halo_amd.pool import reset_round_state, undo_picks
    g = torch.Generator(device=dev).manual_seed(5)
    for (B, H, W) in ((2, 64, 128), (1, 33, 47), (3, 8, 1024), (1, 1, 1)):
        active = torch.rand((B, H, W), generator=g, device=dev) < 0.5
        selected = torch.rand((B, H, W), generator=g, device=dev) < 0.5
        amask = torch.randint(0, 255, (B, H, W), generator=g, device=dev, dtype=torch.int64)
        reset_round_state(active, selected, amask)
        assert not bool(active.any()) and not bool(selected.any()) and bool((amask == 255).all())
        # unaligned views take the byte kernel
        big_a = torch.ones((B * H * W + 3,), dtype=torch.bool, device=dev)
        big_s = torch.ones_like(big_a)
        big_m = torch.zeros((B * H * W + 1,), dtype=torch.int64, device=dev)
        va, vs, vm = big_a[3:].view(B, H, W), big_s[1:1 + B * H * W].view(B, H, W), big_m[1:].view(B, H, W)
        reset_round_state(va, vs, vm)
        assert not bool(va.any()) and not bool(vs.any()) and bool((vm == 255).all())
        assert bool(big_a[:3].all()) and bool(big_s[0]) and int(big_m[0]) == 0
        for arad, mrad, n in ((1, 5, 40), (0, 0, 9), (2, 3, 25), (1, 14, 6)):
            score = torch.rand((B, H, W), generator=g, device=dev, dtype=torch.float64)
            gt = torch.randint(0, 19, (B, H, W), generator=g, device=dev, dtype=torch.int64)
            picks, npk = greedy_select(score, n, arad, mrad, active, selected, amask, gt)
            assert int(npk.max()) > 0 and (bool(active.any()) or H * W == 0)
            undo_picks(picks, npk, arad, mrad, active, selected, amask)
            assert not bool(active.any()) and not bool(selected.any()) and bool((amask == 255).all()), (B, H, W, arad, mrad)


def test_device_identity(dev):
    from halo_amd.pool import assert_distinct_devices, device_identity
    ident = device_identity(0)
    assert ident.startswith("pci=") and " uuid=" in ident and len(ident.split("uuid=")[1]) == 32
    assert assert_distinct_devices(0) == [ident]                      # no process group: one rank
    with pytest.raises(Exception):
        device_identity(torch.cuda.device_count() + 7)
    # the refused query must not leave a sticky HIP error behind for the next launch of anybody
    assert float(torch.ones(4, device=dev).sum()) == 4.0
    torch.cuda.synchronize(dev)


def _bench(args, env=None, timeout=1500):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_contiguous_allocation_and_bandwidth_probes(dev):
    """Measurement aids (tools/halo_probe, outside the product ABI): a tensor in its own physically contiguous range behaves like
    any tensor and frees its memory with its last view; the two probes read what they are given (rates are positive, bad geometry
    is refused)."""
    from tools import halo_probe
    from tools.halo_probe import alloc_contiguous, contiguous_memory_stats, probe_streaming
    before = contiguous_memory_stats()
    t = alloc_contiguous((3, 8, 64, 128), torch.float64, dev)                  # planes of 64 KiB
    assert t.shape == (3, 8, 64, 128) and t.dtype == torch.float64 and t.is_contiguous() and t.device == dev
    t.copy_(torch.arange(t.numel(), device=dev, dtype=torch.float64).view_as(t))
    assert float(t.sum()) == float(t.numel() * (t.numel() - 1) // 2)
    mid = contiguous_memory_stats()
    assert mid["live"] == before["live"] + 1 and mid["failed"] == before["failed"]
    assert mid["contiguous_bytes"] + mid["fallback_bytes"] == before["contiguous_bytes"] + before["fallback_bytes"] + t.numel() * 8
    rows = probe_streaming(t, planes=8, plane_bytes=64 * 128 * 8, window_bytes=2 * 8 * 64 * 128 * 8, reps=2)
    assert [r[0] for r in rows] == [0, 2 * 8 * 65536] and [r[1] for r in rows] == [2 * 8 * 65536, 8 * 65536]
    assert all(r[2] > 0 and r[3] > 0 for r in rows)
    # the walk probe computes sum of squares per pixel over the planes of a group: check it (16 bytes per lane -> 2 doubles)
    out = torch.zeros((3, 64, 128), dtype=torch.float64, device=dev)
    halo_probe.walk_probe(t, t.numel() * 8, 65536, 8, out)
    assert torch.equal(out, (t * t).sum(dim=1))
    with pytest.raises(halo_probe.ProbeError):
        halo_probe.walk_probe(t, t.numel() * 8, 1000, 8, out)
    with pytest.raises(halo_probe.ProbeError):
        halo_probe.read_probe(t, 1024, None, offset=8)
    assert halo_probe.flat_read_gbps(t, reps=2)["GB/s"] > 0
    view = t[1:2]
    del t
    assert contiguous_memory_stats()["live"] == mid["live"]                    # the view keeps the block alive
    assert float(view[0, 0, 0, 0]) == 8 * 64 * 128
    del view
    assert contiguous_memory_stats()["live"] == before["live"]


@pytest.mark.parametrize("resets", ["kernel", "undo"])
def test_bench_two_ranks_on_one_gpu_equal_the_one_rank_pool(dev, tmp_path, resets):
    """bench.py's N > 1 code with rank != 0 on hardware: two ranks share the one GPU, process group over gloo, the wire
    block staged through the host.  Uneven blocks (37 images -> 19 + 18), rank 1's rotated ring, pack / unpack on device
    tensors, the one collective per round, the cross-rank row check inside bench.py -- and the gathered pool tables
    must be bit-identical to the one-rank run of the same pool."""
    common = ["--height", "256", "--width", "512", "--channels", "32", "--batch", "8", "--ring", "16", "--warmup", "2",
              "--pool-images", "37", "--cpu-images", "0", "--resets", resets]
    one = _bench(common + ["--dump-tables", str(tmp_path / "one.npz")])
    two = _bench(common + ["--gpus", "2", "--dump-tables", str(tmp_path / "two.npz")],
                 env={"HALO_BENCH_BACKEND": "gloo", "HALO_BENCH_SHARE_GPU": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["config"]["image_evaluations"] == 37 and two["config"]["image_evaluations"] == 37
    assert two["scaling"] == "strong" and two["steps"] == 3 and one["steps"] == 5          # ceil(19 / 8), ceil(37 / 8)
    assert two["exchange"]["collectives_per_round"] == 1 and two["exchange"]["ms"] is not None
    assert "gloo all-gather of pick tables per round" in two["config"]["sharding"]
    # rank 0 holds 19 images = contents 0..15 and 0..2 again: it can check every one of the 37 gathered rows
    assert two["exchange"]["rows_checked_against_local_results"] == 37
    assert two["per_rank_images_per_s"]["min"] > 0 and two["pipeline_tables_consistent"] is True
    a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / "two.npz")
    assert int(a["n_pool"]) == int(b["n_pool"]) == 37 and int(b["world"]) == 2
    assert np.array_equal(a["counts"], b["counts"]) and a["tables"].shape == b["tables"].shape
    assert np.array_equal(a["tables"].view(np.int64), b["tables"].view(np.int64)), "pool tables depend on the world size"


def test_bench_eight_ranks_on_one_gpu_score_the_2975_image_pool_like_one_rank(dev, tmp_path):
    """configs[3] / configs[4] code at world 8 before an 8-GPU node sees it: bench.py --gpus 8 --pool-images 2975
    at a tiny shape, all eight ranks on the one GPU over gloo.  Seven blocks of 372 images and one of 371, seven ROTATED rings
    (rank r's ring = the base ring rotated by 372 r mod 16), 47 steps per rank the last a partial batch (372 = 46 x 8 + 4; rank 7:
    371 = 46 x 8 + 3), the padded wire block (rank 7 pads one row), ONE collective, every gathered row checked against rank 0's own
    results for the same content -- and the pool's tables bit-identical to the one-rank run of the same pool."""
    common = ["--height", "64", "--width", "128", "--channels", "8", "--batch", "8", "--ring", "16", "--warmup", "2",
              "--pool-images", "2975", "--cpu-images", "0"]
    one = _bench(common + ["--dump-tables", str(tmp_path / "one.npz")])
    eight = _bench(common + ["--gpus", "8", "--dump-tables", str(tmp_path / "eight.npz")],
                   env={"HALO_BENCH_BACKEND": "gloo", "HALO_BENCH_SHARE_GPU": "1"}, timeout=2400)
    assert one["n_gpus"] == 1 and eight["n_gpus"] == 8 and eight["scaling"] == "strong"
    assert one["config"]["image_evaluations"] == eight["config"]["image_evaluations"] == 2975
    assert eight["steps"] == 47 and one["steps"] == 372                 # ceil(372 / 8), ceil(2975 / 8)
    assert "contiguous blocks of 372 image(s)" in eight["config"]["sharding"]
    assert eight["exchange"]["collectives_per_round"] == 1 and eight["exchange"]["ms"] is not None
    assert eight["exchange"]["rows_checked_against_local_results"] == 2975
    assert eight["exchange"]["bytes_per_rank"] == 372 * (3 * 10 + 1) * 4   # 10 regions per 64x128 image, padded to 372 rows on every rank
    assert len(eight["devices"]) == 8 and eight["distinct_devices"] == 1    # the test switch: one shared GPU, reported as such
    assert len(one["devices"]) == 1 and one["distinct_devices"] == 1
    assert eight["per_rank_images_per_s"]["min"] > 0 and eight["pipeline_tables_consistent"] is True
    assert eight["host_threads_per_rank"] == max(1, one["host_threads_per_rank"] // 8)
    # per-rank evidence in the N > 1 line: every rank's own feature-kernel average and roofline fraction
    rf = eight["roofline"]
    assert len(rf["per_rank_avg_launch_ms"]) == 8 and all(v > 0 for v in rf["per_rank_avg_launch_ms"])
    assert 0 < rf["per_rank_frac"]["min"] <= rf["per_rank_frac"]["max"]
    assert eight["exchange"]["ranks"] == 8 and eight["exchange"]["backend"] == "gloo" and one["exchange"]["ranks"] == 1
    assert eight["selection"]["images"] == 372 and one["selection"]["images"] == 2975
    a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / "eight.npz")
    assert int(a["n_pool"]) == int(b["n_pool"]) == 2975 and int(b["world"]) == 8
    assert np.array_equal(a["counts"], b["counts"]) and a["tables"].shape == b["tables"].shape == (2975, 10, 3)
    assert np.array_equal(a["tables"].view(np.int64), b["tables"].view(np.int64)), "pool tables depend on the world size"


def test_ranks_that_share_a_gpu_are_refused(dev):
    """Without bench.py's test switch two ranks on one device are an error (every number would silently halve):
    two gloo ranks, both on device 0, ask halo_amd.pool.assert_distinct_devices."""
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = str(sk.getsockname()[1]); sk.close()
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "import torch, torch.distributed as dist\n"
            "from halo_amd.pool import assert_distinct_devices\n"
            "dist.init_process_group('gloo')\n"
            "try:\n"
            "    assert_distinct_devices(0)\n"
            "except RuntimeError as ex:\n"
            "    print('REFUSED', ex); sys.exit(0)\n"
            "sys.exit(3)\n") % ROOT
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port))
             for rk in (0, 1)]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0 and "REFUSED" in out and "share a GPU" in out, out + err[-2000:]
