"""CPU, gloo, world_size 2 and 3: the sharding + ordered-gather logic of the multi-GPU path
(pipeline.shard_range / gather_in_order).  The per-rank compute is replaced by a deterministic
stand-in render (the HIP modules have no CPU fallback); what is tested is that rank 0 receives every
vehicle's crop exactly once, in original vehicle order, for even and ragged shard sizes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from future_urban_scene_generation_amd.pipeline import broadcast_state_dicts, gather_in_order, load_schema, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_render(vehicle_ids):
    """Stand-in for VehiclePipeline.run: a uint8 'crop' that encodes the global vehicle index."""
    out = torch.zeros((len(vehicle_ids), 4, 4, 3), dtype=torch.uint8)
    for i, v in enumerate(vehicle_ids):
        out[i] = (v * 3 + torch.arange(48).reshape(4, 4, 3)) % 251
    return out


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_items, rank, world)
        local = _fake_render(list(range(lo, hi)))
        gather_in_order(_fake_render(list(range(hi, lo, -1))), n_items)     # an earlier pass: the staging buffers are reused
        full = gather_in_order(local, n_items)
        kp = gather_in_order(torch.arange(lo, hi, dtype=torch.int32).view(-1, 1).repeat(1, 12), n_items)
        if rank == 0:
            q.put((full.clone(), kp.clone()))
        else:
            assert full is None and kp is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_items", [(2, 8), (2, 7), (3, 10), (8, 64)])
def test_gather_in_vehicle_order(world, n_items):
    """(8, 64) = BASELINE configs[3]'s shape: 64 vehicles of one frame, 8 shards of 8 (gloo rehearsal of --gpus 8
    --vehicles 64; RCCL itself has never run in the development loop)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, kp = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(full, _fake_render(list(range(n_items))))          # == the single-process result
    assert torch.equal(kp[:, 0], torch.arange(n_items, dtype=torch.int32))


def test_single_process_is_identity():
    x = _fake_render([0, 1, 2])
    assert gather_in_order(x, 3) is x


def _force_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FUSG_DIST_FORCE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        x = _fake_render([0, 1, 2])
        got = gather_in_order(x, 3)                                       # through the collective, not the shortcut
        from future_urban_scene_generation_amd.synth import synth_state_dict
        want = {"icn": synth_state_dict("icn", load_schema("icn"), 5)}
        sd = broadcast_state_dicts(want, nets=("icn",), src=0)
        ok = got is not x and torch.equal(got, x) and all(torch.equal(sd["icn"][k], want["icn"][k]) for k in want["icn"])
        os.environ.pop("FUSG_DIST_FORCE")
        ok = ok and gather_in_order(x, 3) is x                            # the shortcut again
        q.put(ok)
    finally:
        dist.destroy_process_group()


def test_one_rank_group_takes_the_collectives_when_forced():
    """FUSG_DIST_FORCE=1 (the hook behind tests/test_bench_gpu.py::test_rccl_collectives_with_a_one_rank_communicator): a process group
    of ONE rank goes through gather / broadcast instead of the single-process shortcuts - same results."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_force_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(timeout=60)
    assert p.exitcode == 0


def test_vehicle_noise_streams_are_shard_invariant():
    """Host half of the 8(e) noise caveat: with one generator per vehicle the draws of vehicle v are the
    same whatever batch it sits in; without (reference mode) they are `torch.randn(*shape)` on the global RNG."""
    from future_urban_scene_generation_amd.vunet.models import Vunet_fix_res

    def gens(seeds):
        out = []
        for sd in seeds:
            g = torch.Generator(device="cpu")
            g.manual_seed(sd)
            out.append(g)
        return out

    def draw(seeds, B):
        shapes = [(B, 128, 2, 2)] * 2 + [(B, 128, 4, 4)]
        buf = torch.empty(sum(int(torch.Size(s).numel()) for s in shapes))
        views = Vunet_fix_res._fill_noise(buf, shapes, gens(seeds) if seeds else None)
        return [buf[o:o + n].view(*s).clone() for o, n, s in views]

    full = draw([10, 11, 12, 13], 4)
    for lo, hi in (shard_range(4, 0, 2), shard_range(4, 1, 2)):
        part = draw([10 + i for i in range(lo, hi)], hi - lo)
        for f, p in zip(full, part):
            assert torch.equal(f[lo:hi], p)
    torch.manual_seed(5)
    ref = [torch.randn(2, 128, 2, 2), torch.randn(2, 128, 2, 2), torch.randn(2, 128, 4, 4)]
    torch.manual_seed(5)
    got = draw(None, 2)
    assert all(torch.equal(a, b) for a, b in zip(ref, got))
    with pytest.raises(ValueError):
        draw([1, 2, 3], 2)


def _bcast_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from future_urban_scene_generation_amd.synth import synth_state_dict
        nets = ("hg", "icn")
        mine = {n: synth_state_dict(n, load_schema(n), 3) for n in nets} if rank == 0 else None
        got = broadcast_state_dicts(mine, nets=nets)
        want = {n: synth_state_dict(n, load_schema(n), 3) for n in nets}      # what rank 0 holds
        ok = all(list(got[n].keys()) == list(want[n].keys()) and
                 all(got[n][k].dtype == want[n][k].dtype and torch.equal(got[n][k], want[n][k]) for k in want[n]) for n in nets)
        q.put((rank, ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_from_rank0():
    """Start-up collective of SURVEY 8(e): only rank 0 holds the checkpoints; after the broadcast every rank has
    bit-identical state_dicts in the reference's key order and dtypes (incl. the hourglass's int64 BatchNorm counters)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_slice_scene_partitions_the_vehicles():
    """Host half of the sharded frame driver: the per-vehicle entries of a scene are cut at shard_range's bounds, the
    frame-level ones are shared, and the shards put together are the scene again."""
    import numpy as np
    from future_urban_scene_generation_amd.pipeline import PER_VEHICLE_KEYS, slice_scene
    V = 7
    scene = {"frame": torch.zeros(4, 6, 3, dtype=torch.uint8), "focals": np.ones(2), "centers": np.ones(2),
             "bboxes": np.arange(V * 4).reshape(V, 4), "masks": torch.arange(V).view(V, 1, 1).expand(V, 4, 6),
             "src_sketch": torch.zeros(V, 4, 6, 3), "dst_sketch": torch.zeros(V, 4, 6, 3), "src_planes": torch.zeros(V, 5, 4, 6, 3),
             "src_kp": [[v] for v in range(V)], "dst_kp": [[v] for v in range(V)], "src_vis": np.zeros((V, 5)), "dst_vis": np.zeros((V, 5)),
             "kp3d": np.zeros((V, 12, 3)), "vehicle_seeds": list(range(V)),
             "inpaint": {"boxes": np.arange(V * 4).reshape(V, 4), "img": torch.zeros(V, 3, 2, 2)}}
    seen = []
    for r in range(3):
        lo, hi = shard_range(V, r, 3)
        sub = slice_scene(scene, lo, hi)
        assert sub["frame"] is scene["frame"] and sub["focals"] is scene["focals"]
        for k in PER_VEHICLE_KEYS:
            assert len(sub[k]) == hi - lo, k
        assert len(sub["inpaint"]["boxes"]) == hi - lo and sub["inpaint"]["img"].shape[0] == hi - lo
        seen += list(sub["vehicle_seeds"])
        assert sub["src_kp"] == [[v] for v in range(lo, hi)] and sub["masks"][:, 0, 0].tolist() == list(range(lo, hi))
    assert seen == list(range(V))
    assert len(scene["bboxes"]) == V                                           # the caller's dict is untouched


def test_later_frame_scenes_shard_like_first_frames():
    """Round 4 (a clip sharded over ranks): a later frame's scene has no detector boxes or CAD keypoints - `slice_scene` cuts what
    is there at the same bounds - and `synth_later_frame` keeps the vehicles (masks, first-frame planes) while giving every
    (vehicle, trajectory step) its own noise seed, so a vehicle's images do not depend on the rank that renders its clip."""
    import numpy as np
    from future_urban_scene_generation_amd.pipeline import slice_scene, synth_later_frame
    V = 5
    first = {"frame": torch.zeros(4, 6, 3, dtype=torch.uint8), "masks": torch.arange(V).view(V, 1, 1).expand(V, 4, 6),
             "src_sketch": torch.ones(V, 4, 6, 3), "dst_sketch": torch.zeros(V, 4, 6, 3), "src_planes": torch.zeros(V, 5, 4, 6, 3),
             "src_kp": [[np.full((4, 2), v, np.int32)] * 5 for v in range(V)], "dst_kp": [[np.zeros((4, 2), np.int32)] * 5 for v in range(V)],
             "src_vis": np.ones((V, 5)), "dst_vis": np.ones((V, 5)), "vehicle_seeds": [10 + v for v in range(V)],
             "bboxes": np.zeros((V, 4)), "kp3d": np.zeros((V, 12, 3))}
    later = synth_later_frame(first, 3)
    assert later["masks"] is first["masks"] and later["src_planes"] is first["src_planes"] and later["src_kp"] is first["src_kp"]
    assert later["dst_sketch"] is first["src_sketch"]
    assert later["vehicle_seeds"] == [(10 + v) * 64 + 3 for v in range(V)] and first["vehicle_seeds"] == [10 + v for v in range(V)]
    assert all(d.shape == s.shape and d.dtype == np.int32 for dv, sv in zip(later["dst_kp"], first["src_kp"]) for d, s in zip(dv, sv))
    assert synth_later_frame(first, 3)["dst_kp"][2][1].tolist() == later["dst_kp"][2][1].tolist()       # seeded by the step
    later = {k: v for k, v in later.items() if k not in ("bboxes", "kp3d")}      # what a real later frame carries
    seen = []
    for r in range(2):
        lo, hi = shard_range(V, r, 2)
        sub = slice_scene(later, lo, hi)
        assert "bboxes" not in sub and sub["masks"][:, 0, 0].tolist() == list(range(lo, hi)) and len(sub["dst_kp"]) == hi - lo
        seen += sub["vehicle_seeds"]
    assert seen == later["vehicle_seeds"]
