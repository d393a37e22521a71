"""N>1 path on CPU: two gloo ranks shard the screen into 8x8 tiles (tile t -> rank t % world), each
contributes its tile buffer, all_gather + assemble must reproduce the full frame exactly."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, w, h, ref_path, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aten_amd import tiling
    full = np.load(ref_path)
    mine = tiling.extract_tiles(full, rank, world)                # what atn_tile_device holds on this rank
    t = torch.from_numpy(mine)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    img = tiling.assemble_tiles(torch.stack(gathered).numpy(), w, h, world)
    if rank == 0:
        np.save(out_path, img)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_gather(tmp_path):
    w, h, world = 100, 52, 2          # not multiples of 8 on purpose: ragged edge tiles
    rng = np.random.default_rng(0)
    full = rng.random((h, w, 4), dtype=np.float32)
    ref = str(tmp_path / "ref.npy"); out = str(tmp_path / "out.npy")
    np.save(ref, full)
    mp.spawn(_worker, args=(world, 29611, w, h, ref, out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), full)


def test_tiling_maps_are_a_partition():
    from aten_amd import tiling
    for (w, h, world) in [(1920, 1080, 8), (64, 64, 1), (100, 52, 3), (8, 8, 4)]:
        seen = np.zeros((h, w), np.int32)
        for r in range(world):
            xs, ys, valid = tiling.slot_pixels(w, h, r, world)
            assert len(xs) == tiling.slots_per_rank(w, h, world)
            seen[ys[valid], xs[valid]] += 1
        assert np.all(seen == 1)


def test_bench_self_launch_command():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1 (the command line the driver uses); under a launcher, with --mgpu or with one GPU it does not."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.needs_self_launch(8, False, {})
    assert not b.needs_self_launch(8, False, {"WORLD_SIZE": "8"})
    assert not b.needs_self_launch(8, True, {})
    assert not b.needs_self_launch(1, False, {})
    cmd = b.launcher_cmd(4, ["--gpus", "4", "--steps", "7"], port=29511)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    free = b.launcher_cmd(2, [])
    assert int(free[free.index("--master-port") + 1]) > 0


def test_bench_refuses_counters_taken_on_other_kernel_sources(tmp_path, monkeypatch):
    """bench.py's roofline uses committed PMC counters only when they were collected on the kernel sources it is running
    (content hash of aten_amd/csrc + include/); a record from other sources is named and refused, never used silently."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    from aten_amd.build import kernel_sources_sha16
    sha = kernel_sources_sha16()
    assert len(sha) == 16 and sha == kernel_sources_sha16()
    (tmp_path / "profiles").mkdir()
    tag = "sponza_lod 1920x1080 1spp 5-bounce"
    rec = {"workload": tag, "kernel_sources_sha16": "0" * 16, "kernels": {"k_trace_fused<true, false>": {"launches_sampled": 5}}}
    (tmp_path / "profiles" / "r00_counters_x.json").write_text(json.dumps(rec))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    best, stale, got = b.profile_counters("sponza_lod", 1920, 1080, 1, 5, False)
    assert best is None and stale == os.path.join("profiles", "r00_counters_x.json") and got == sha
    rec["kernel_sources_sha16"] = sha
    (tmp_path / "profiles" / "r01_counters_x.json").write_text(json.dumps(rec))
    best, stale, _ = b.profile_counters("sponza_lod", 1920, 1080, 1, 5, False)
    assert best[0] == os.path.join("profiles", "r01_counters_x.json") and stale is not None
    assert b.kernel_entry((best,), "k_trace_fused")["launches_sampled"] == 5
    assert b.profile_counters("cornell", 1920, 1080, 1, 5, False)[0] is None
    # the hash of the source TEXT is not enough: the record must also name the build id of the loaded binary --
    # sources hash + extra compile flags, compiled into the library (atn_build_id) -- and that binary must be what the tree builds
    from aten_amd.build import build_id, loaded_build_id
    assert loaded_build_id() == build_id() == sha + "|"
    rec["build_id"] = build_id(["-DATN_INNER_BURST=4"])          # counters of a variant build (tools/build_variants.sh)
    (tmp_path / "profiles" / "r02_counters_x.json").write_text(json.dumps(rec))
    best, stale, _ = b.profile_counters("sponza_lod", 1920, 1080, 1, 5, False)
    assert best[0] == os.path.join("profiles", "r01_counters_x.json") and stale == os.path.join("profiles", "r02_counters_x.json")
    rec["build_id"] = build_id()
    (tmp_path / "profiles" / "r03_counters_x.json").write_text(json.dumps(rec))
    assert b.profile_counters("sponza_lod", 1920, 1080, 1, 5, False)[0][0] == os.path.join("profiles", "r03_counters_x.json")
    monkeypatch.setattr("aten_amd.build.loaded_build_id", lambda: build_id(["-DATN_INNER_BURST=4"]))     # ATEN_AMD_LIB = a variant
    # ... then only the record taken on that variant counts
    assert b.profile_counters("sponza_lod", 1920, 1080, 1, 5, False)[0][0] == os.path.join("profiles", "r02_counters_x.json")
    monkeypatch.setattr("aten_amd.build.loaded_build_id", lambda: "0123456789abcdef|")       # a binary built from other sources
    assert b.profile_counters("sponza_lod", 1920, 1080, 1, 5, False)[0] is None
