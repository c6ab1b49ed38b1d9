"""Worker of tests/test_bench_gpu.py::test_run_frame_sharded_over_two_ranks (not a test module): one rank of a gloo group on the
box's card.  Every rank builds the same pipeline (synthetic weights of one seed) and the same scenes.

  * `run_frame` shards the frame's vehicles over the ranks; rank 0 compares the result with its own unsharded run_frame of the same
    scene (integers exact, images to the last place); every rank gets a RANK-LOCAL state (its own vehicles' appearance codes);
  * `run_later_frame(scene, state)` renders a future frame of the clip from that state, sharded the same way (a vehicle's frames
    stay on one rank): rank 0 compares with the unsharded later frame;
  * `run_frames` over three sharded scenes (one frame in flight) gives what `run_frame` gives, bit for bit.

Rank 0 prints one `OBS {json}` line with the differences it observed (the test records them: profiles/r04_parity.json) and
SHARD_OK / SHARD_FAILED.  The bars below are ~10x the observations of round 4 (a shard is a smaller batch: other split-K factors
and tile shapes; Lab -> BGR amplifies one Lab step)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# (max abs difference in uint8 levels, largest fraction of differing pixels) - sharded vs unsharded.  Observed in round 4
# (profiles/r04_parity.json, first / later frame): vunet_u8 1 level on 1.8e-5 / 1.6e-5 of the pixels, icn_u8 4 levels on 8.2e-5 /
# 7.5e-5 (Lab -> BGR amplifies one Lab step), frame_icn 2 / 1 levels on 1.2e-5 / 7e-6, frame_vunet 1 level on 2.9e-6 / 1.4e-6
BARS = {"vunet_u8": (1, 2e-4), "icn_u8": (8, 1e-3), "frame_icn": (6, 2e-4), "frame_vunet": (2, 5e-5), "inpaint_u8": (1, 2e-4)}


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.set_num_threads(8)                                          # two ranks share the box's host
    backend = os.environ.get("FUSG_TEST_BACKEND", "gloo")
    if backend == "nccl":
        # RCCL: one card per rank.  On a box with ONE card the test starts a single rank with FUSG_DIST_FORCE=1 - the sharded code
        # paths then run with a one-rank RCCL communicator (every shard is the whole frame: the results equal the unsharded ones bit for bit)
        torch.cuda.set_device(rank % torch.cuda.device_count())
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank % torch.cuda.device_count()))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from future_urban_scene_generation_amd import ops
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, shard_range, synth_frame, synth_later_frame as later_scene
    dev = torch.device("cuda", rank % torch.cuda.device_count()) if backend == "nccl" else torch.device("cuda:0")
    ops.set_precision("f16x3")
    inpaint = len(sys.argv) > 1 and sys.argv[1] == "inpaint"
    if backend == "nccl":
        # the start-up collective as well: rank 0's weights reach every rank as one flat device blob per network (north_star)
        from future_urban_scene_generation_amd.pipeline import load_schema
        from future_urban_scene_generation_amd.synth import synth_state_dict
        nets = ("hg", "icn", "vunet") + (("edge", "inpaint") if inpaint else ())
        mine = {n: synth_state_dict(n, load_schema(n), 3) for n in nets} if rank == 0 else None
        pipe = VehiclePipeline(dev, inpaint=inpaint, state_dicts=mine, broadcast_src=0)
    else:
        pipe = VehiclePipeline(dev, inpaint=inpaint, seed=3)
    ok = True
    obs = {}

    def compare(tag, got, single, keys):
        nonlocal ok
        for k in keys:
            lim, flim = BARS[k]
            d = (got[k].to(torch.int32) - single[k].to(torch.int32)).abs()
            frac = float((d > 0).float().mean())
            obs[f"{tag}_{k}_max_diff"] = max(obs.get(f"{tag}_{k}_max_diff", 0), int(d.max()))
            obs[f"{tag}_{k}_frac_differing"] = max(obs.get(f"{tag}_{k}_frac_differing", 0.0), frac)
            if int(d.max()) > lim or frac > flim:
                ok = False
                print("MISMATCH", tag, k, int(d.max()), frac, flush=True)

    scenes = {}
    for V in (5, 1):                                                  # ragged shards (3 + 2), and an empty shard on rank 1
        sc = synth_frame(V, (360, 640), dev, seed=20 + V, inpaint=inpaint)
        sc["vehicle_seeds"] = [90 + v for v in range(V)]
        scenes[V] = sc
        got = pipe.run_frame(sc)
        lo, hi = shard_range(V, rank, world)
        st = got["state"]
        assert st["sharded"] and st["shard"] == (lo, hi, V) and st["central"].shape[0] == hi - lo and st["appearance"][0].shape[0] == hi - lo
        assert (set(got) == {"state"}) == (rank != 0)
        if world == 1:
            assert dist.get_backend() == "nccl" and os.environ.get("FUSG_DIST_FORCE") == "1"
        later = later_scene(sc, 7)
        got_l = pipe.run_later_frame(later, st)                       # the clip's next frame: same shards, codes never moved
        assert (got_l is None) == (rank != 0)
        if rank == 0:
            single = pipe.run_frame({**sc, "shard": False})
            assert not single["state"]["sharded"]
            # the integer results are the same bits; the rendered crops may differ in the last place, because a network's
            # launches (split-K factors, tile shapes) depend on the batch it is given and a shard is a smaller batch
            for k in ("kp_idx", "kp_xy", "geom"):
                if not torch.equal(got[k], single[k]):
                    ok = False
                    print("MISMATCH", V, k, flush=True)
            compare("first", got, single, ("vunet_u8", "icn_u8", "frame_icn", "frame_vunet") + (("inpaint_u8",) if inpaint else ()))
            for a, b in zip(got["pose"], single["pose"]):
                for x, y in zip(a, b):
                    if not np.allclose(np.asarray(x), np.asarray(y), rtol=1e-4, atol=1e-5, equal_nan=True):
                        ok = False
                        print("MISMATCH pose", V, flush=True)
            single_l = pipe.run_later_frame({**later, "shard": False}, single["state"])
            if not torch.equal(got_l["geom"], single_l["geom"]):
                ok = False
                print("MISMATCH later geom", V, flush=True)
            compare("later", got_l, single_l, ("vunet_u8", "icn_u8", "frame_icn", "frame_vunet"))
            # (the state of the sharded first frame equals rank 0's slice of the unsharded one to the last place)
            d = float((st["appearance"][1] - single["state"]["appearance"][1][lo:hi]).abs().max())
            obs["state_appearance_abs_diff"] = max(obs.get("state_appearance_abs_diff", 0.0), d)
    # one frame in flight, sharded: frame i+1's shard is issued before frame i's crops are gathered - same bits as run_frame
    seq_scenes = [scenes[5], scenes[1], scenes[5]]
    want = [pipe.run_frame(sc) for sc in seq_scenes]
    frames = list(pipe.run_frames(seq_scenes))
    assert len(frames) == 3
    for f, w in zip(frames, want):
        assert set(f) == set(w), (sorted(f), sorted(w))
        if rank == 0:
            for k in ("kp_idx", "kp_xy", "geom", "icn_u8", "vunet_u8", "frame_icn", "frame_vunet"):
                if not torch.equal(f[k], w[k]):
                    ok = False
                    print("MISMATCH pipelined", k, flush=True)
            for a, b in zip(f["pose"], w["pose"]):
                if not all(np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True) for x, y in zip(a, b)):
                    ok = False
                    print("MISMATCH pipelined pose", flush=True)
        assert torch.equal(f["state"]["appearance"][0], w["state"]["appearance"][0])
    # a whole clip through the generator form: first frame + two later frames
    clip = list(pipe.run_clip_frames(scenes[5], [later_scene(scenes[5], 7), later_scene(scenes[5], 8)]))
    assert len(clip) == 3 and all((c is None) == (rank != 0) for c in clip)
    dist.barrier()
    if rank == 0:
        print("OBS " + json.dumps(obs, sort_keys=True), flush=True)
        print("SHARD_OK" if ok else "SHARD_FAILED", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
