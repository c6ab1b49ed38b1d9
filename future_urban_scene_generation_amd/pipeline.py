"""Batched, shardable driver of the per-vehicle hot path (replaces the reference's vehicle-serial,
batch-1 loop of trajectory_inference.py:55-250 for the network part).

One *crop* = one first-frame pass for one vehicle:
    hourglass heat-maps + argmax  ->  ICN completion  ->  VUnet enc_up/enc_down/dec_up/dec_down
    [-> EdgeGenerator -> InpaintGenerator with ``inpaint=True``]
Vehicles are independent, so a frame's vehicles are sharded contiguously over the
ranks of one node with NO collective in the data path; the only exchange is the final gather of
the rendered uint8 crops (and int32 keypoint indices) to rank 0, which pastes them in original
vehicle order (later vehicles overwrite earlier ones, trajectory_inference.py:197-198).
"""
from __future__ import annotations

import json
import os
from argparse import Namespace
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, size-balanced shard [lo, hi) of `n_items` vehicles for `rank` of `world`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_GATHER_BUFS = {}


def _one_rank(group=None) -> bool:
    """No process group, or a group of one rank: the multi-rank code paths are skipped.  FUSG_DIST_FORCE=1 (test hook) keeps them for
    a group of ONE rank, so that a box with a single card executes the RCCL collectives of every sharded path (broadcast of the
    weights, gather of the crops / states, the comm-stream pipelining) with a one-rank communicator."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and os.environ.get("FUSG_DIST_FORCE") != "1"


def gather_in_order(local: torch.Tensor, n_items: int, group=None, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather per-rank shards ([n_local, ...], same trailing shape) to `dst` in vehicle order.
    Works for any backend (RCCL on GPU tensors, gloo on CPU tensors).  Returns the full tensor on
    `dst`, None elsewhere.  Shards may be ragged (n_items not a multiple of the world size).
    The padded send buffer and rank `dst`'s receive buffers are allocated ONCE per (shape, dtype, device, world) and
    reused by every later call (a frame loop calls this once per pass): the result is a fresh tensor, the staging
    buffers never escape."""
    import torch.distributed as dist
    if _one_rank(group):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_range(n_items, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    if local.is_cuda and dist.get_backend(group) == "gloo":       # rehearsal of the multi-rank path without RCCL
        local = local.cpu()
    stream_id = torch.cuda.current_stream(local.device).cuda_stream if local.is_cuda else 0     # staging buffers live on one stream
    key = (max_n, tuple(local.shape[1:]), local.dtype, str(local.device), world, rank == dst, id(group), stream_id)
    bufs = _GATHER_BUFS.get(key)
    if bufs is None:
        pad = torch.zeros((max_n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        recv = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        bufs = _GATHER_BUFS[key] = (pad, recv)
    pad, recv = bufs
    pad[: local.shape[0]].copy_(local)                             # (rows past a short shard keep their zeros: never read)
    dist.gather(pad, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([recv[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def broadcast_state_dicts(state_dicts: Optional[Dict[str, dict]], nets: Sequence[str] = ("hg", "icn", "vunet"),
                          src: int = 0, group=None, device=None) -> Dict[str, "OrderedDict[str, torch.Tensor]"]:
    """The start-up collective of the multi-GPU path (north_star: "RCCL ... for the broadcast of shared
    weights"): rank `src` holds the checkpoints of `nets` (state_dicts with the reference's keys), every rank returns
    identical CPU state_dicts.  The keys / shapes / dtypes are known everywhere (the schemas shipped with the package),
    so each network travels as ONE flat buffer per dtype class - a float32 blob (27 / 35 / 181 / 43 / 43 MB for hg / icn
    / vunet / edge / inpaint) and, for the hourglass, an int64 blob of the BatchNorm counters - i.e. few, large messages
    for xGMI's point-to-point links rather than one message per tensor (1174 tensors in total).  Works on any backend:
    with RCCL ("nccl") the blobs are staged on `device` (default: the current HIP device), with gloo on the host.
    Without an initialised process group (or world size 1) it returns rank src's dicts unchanged."""
    import torch.distributed as dist
    if _one_rank(group):
        assert state_dicts is not None, "broadcast_state_dicts: no process group and no state_dicts"
        return {n: state_dicts[n] for n in nets}
    rank = dist.get_rank(group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device(device) if device is not None else (torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu"))
    out = {}
    for net in nets:
        schema = load_schema(net)
        classes = {"f": [k for k, (_, dt) in schema.items() if dt.startswith("float")],
                   "i": [k for k, (_, dt) in schema.items() if not dt.startswith("float")]}
        sd = OrderedDict()
        for cls, keys in classes.items():
            if not keys:
                continue
            dtype = torch.float32 if cls == "f" else torch.int64
            numel = [int(torch.Size(schema[k][0]).numel()) for k in keys]
            if rank == src:
                assert state_dicts is not None and net in state_dicts, f"rank {src} has no state_dict for '{net}'"
                flat = torch.cat([state_dicts[net][k].detach().to("cpu", dtype).reshape(-1) for k in keys]).to(dev)
            else:
                flat = torch.empty(sum(numel), dtype=dtype, device=dev)
            dist.broadcast(flat, src=src, group=group)
            flat = flat.cpu()
            off = 0
            for k, n in zip(keys, numel):
                want = getattr(torch, schema[k][1])
                sd[k] = flat[off:off + n].view(schema[k][0]).to(want).clone()
                off += n
        out[net] = OrderedDict((k, sd[k]) for k in schema)              # the reference's key order
    return out


def load_schema(net: str):
    """state_dict schema (key -> (shape, dtype)) of one of the five reference networks, as dumped from the
    reference's own modules (tools/gen_golden.py); shipped inside the package (schemas/)."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "schemas", f"schema_{net}.json")) as f:
        raw = json.load(f, object_pairs_hook=OrderedDict)
    return OrderedDict((k, (tuple(v[0]), v[1])) for k, v in raw.items())


# FUSG_PLAN_MT=1: a recorded pass is replayed by one host thread per recorded stream (fusg_plan_run_mt) instead of one thread for all
PLAN_THREADS = os.environ.get("FUSG_PLAN_MT", "0") == "1"
FRAME_PLANS = int(os.environ.get("FUSG_FRAME_PLANS", "6"))     # recorded passes kept by run_frame(replay=True), one per vehicle count

PER_VEHICLE_KEYS = ("bboxes", "masks", "src_sketch", "dst_sketch", "src_planes", "src_kp", "dst_kp", "src_vis", "dst_vis", "kp3d",
                    "vehicle_seeds")


def slice_scene(scene: Dict, lo: int, hi: int) -> Dict:
    """The scene of `run_frame` restricted to vehicles [lo, hi): per-vehicle entries sliced (views, nothing copied), frame-level
    entries (frame, background, focals, centers) shared."""
    out = dict(scene)
    for k in PER_VEHICLE_KEYS:
        if scene.get(k) is not None:
            out[k] = scene[k][lo:hi]
    if scene.get("inpaint") is not None:
        out["inpaint"] = {k: v[lo:hi] for k, v in scene["inpaint"].items()}
    return out


class VehiclePipeline:
    """Holds the five networks on one device and runs batches of crops through them."""

    def __init__(self, device, inpaint: bool = False, state_dicts: Optional[Dict[str, dict]] = None, seed: int = 0,
                 broadcast_src: Optional[int] = None, group=None, cad: bool = False):
        """state_dicts: checkpoints (the reference's keys) per network; a missing network gets the synthetic weights of
        `seed`.  broadcast_src: with an initialised process group, only that rank needs to hold `state_dicts` (a real
        checkpoint read from disk on rank 0): they are distributed with `broadcast_state_dicts` first (north_star: RCCL
        broadcast of the shared weights), so every rank renders with identical parameters.
        cad: also hold the reference's CAD-model classifier (VGG-19, 10 classes; state_dicts['vgg'] = `cads/model.pth`,
        run_test.py:47-58) and run it on the hourglass's crop in every pass (trajectory_inference.py:66-69): 'cad_logits' /
        `run_frame`'s 'cad_idx'.  Not part of BASELINE's crop pass: `bench.py` leaves it off."""
        from .edgeconnect.models import EdgeModel, InpaintingModel
        from .stacked_hourglass.models import HourglassNet
        from .synth import synth_state_dict
        from .vunet.models import Vunet_fix_res
        from .warp_learn.models import G_Resnet
        self.device = torch.device(device)
        self.inpaint = inpaint
        self.group = group
        self.cad = None
        if broadcast_src is not None:
            import torch.distributed as dist
            if not _one_rank(group):
                nets_ = ("hg", "icn", "vunet") + (("edge", "inpaint") if inpaint else ())
                coll_dev = self.device if dist.get_backend(group) == "nccl" else "cpu"
                state_dicts = broadcast_state_dicts(state_dicts if dist.get_rank(group) == broadcast_src else None,
                                                    nets=nets_, src=broadcast_src, group=group, device=coll_dev)

        def sd(net):
            if state_dicts is not None and net in state_dicts:
                return state_dicts[net]
            return synth_state_dict(net, load_schema(net), seed)

        self.hg = HourglassNet(num_stacks=2, num_blocks=1, num_classes=12)             # run_test.py:62
        self.icn = G_Resnet(21)                                                         # run_test.py:74
        self.vunet = Vunet_fix_res(Namespace(up_mode="subpixel", w_norm=True, drop_prob=0.2, vunet_256=True))
        self.hg.load_state_dict(sd("hg"))
        self.icn.load_state_dict(sd("icn"))
        self.vunet.load_state_dict(sd("vunet"))
        nets = [self.hg, self.icn, self.vunet]
        if inpaint:
            self.edge, self.inp = EdgeModel(None), InpaintingModel(None)
            self.edge.generator.load_state_dict(sd("edge"))
            self.inp.generator.load_state_dict(sd("inpaint"))
            nets += [self.edge, self.inp]
        if cad:
            from .cad_classifier import VGG19Classifier, vgg19_schema
            self.cad = VGG19Classifier(10)
            self.cad.load_state_dict(state_dicts["vgg"] if state_dicts is not None and "vgg" in state_dicts
                                     else synth_state_dict("vgg", vgg19_schema(10), seed))
            nets.append(self.cad)
        for n in nets:
            n.to(self.device).eval()
        # the FusedNets whose packed-weight caches a recorded pass points into (EdgeModel / InpaintingModel wrap theirs)
        self._nets = [getattr(n, "generator", n) for n in nets]
        self._status = None                       # this pipeline's own range-status word (ops.status_scope), made on first use

    # The networks of one crop pass do not depend on each other (hourglass / ICN / VUnet; edge -> inpaint is one
    # chain), so each branch runs on its own HIP stream: the many small, latency-bound launches of the hourglass
    # and of the VUnet's low-resolution levels fill the CUs the big ICN layers leave idle between their waves of
    # workgroups.  The VUnet - the longest chain of dependent launches - gets a high-priority stream, so that its
    # small kernels never queue behind the ICN's big ones (measured 1180 -> 1260 crops/s).
    # FUSG_STREAMS=0 serialises everything on the caller's stream.
    HIGH_PRIORITY = ("vunet",)

    def _branches(self, jobs):
        """jobs: list of (name, zero-argument callable returning a dict of output tensors); the first one runs on
        the caller's stream."""
        out = {}
        if os.environ.get("FUSG_STREAMS", "1") == "0" or self.device.type != "cuda" or len(jobs) == 1:
            for _, j in jobs:
                out.update(j())
            return out
        from . import ops
        rec = ops.RECORDER                                  # recording a fusg_plan: dependencies go through it
        main = torch.cuda.current_stream(self.device)
        pool = self.__dict__.setdefault("_streams", {})
        streams = []
        for name, _ in jobs[1:]:
            st = pool.get(name)
            if st is None:
                st = pool[name] = torch.cuda.Stream(device=self.device, priority=-1 if name in self.HIGH_PRIORITY else 0)
            streams.append(st)
        # fork: the side branches may start as soon as what is on the caller's stream NOW is done (not after branch 0)
        if rec is None:
            ready = torch.cuda.Event()
            ready.record(main)
            for st in streams:
                st.wait_event(ready)
        else:
            for st in streams:
                rec.dependency(st.cuda_stream, main.cuda_stream)
        used = []
        for i, (name, j) in enumerate(jobs):
            if i == 0:
                out.update(j())
                continue
            st = streams[i - 1]
            with torch.cuda.stream(st):
                res = j()
            for t in res.values():
                t.record_stream(main)
            out.update(res)
            used.append(st)
        for st in used:
            if rec is None:
                main.wait_stream(st)
            else:
                rec.dependency(main.cuda_stream, st.cuda_stream)
        return out

    def _side(self, name: str, fn):
        """Run fn() on the side stream `name`, forked from the CURRENT stream (what is queued there now is finished
        before fn's launches start); returns (fn's result, join): join() makes the current stream wait for the side
        stream.  Used for the VUnet's two independent encoders: the shape encoder (forward_dec_up) does not depend
        on the appearance half (forward_enc_up -> forward_enc_down), so the VUnet's chain of dependent launches - the
        longest of the pass - is a third shorter.  FUSG_VUNET_SPLIT=0 (or FUSG_STREAMS=0) keeps one stream."""
        from . import ops
        if not ops.side_streams_enabled() or self.device.type != "cuda":
            return fn(), (lambda: None)
        st = ops.side_stream(name, self.device)
        ops.fork_to(st)
        with torch.cuda.stream(st):
            res = fn()

        def tensors(o):
            if torch.is_tensor(o):
                yield o
            elif isinstance(o, (list, tuple)):
                for q in o:
                    yield from tensors(q)

        return res, (lambda: ops.join_from(st, list(tensors(res))))

    def compile(self, batch: Dict[str, torch.Tensor], vehicle_seeds: Optional[Sequence[int]] = None, fn=None) -> "CompiledPass":
        """Record one crop pass for inputs of `batch`'s shapes into a fusg_plan and return the object that replays it:
        `compiled.run(batch, vehicle_seeds)` gives what `self.run` gives, with one library call instead of ~370 (the
        reference's batch-1 call pattern is bound by the interpreter, not by the GPU).  `fn(batch, vehicle_seeds) -> dict of
        tensors`: record that pass instead of `self._run` (e.g. `self._vunet_forward`: BASELINE configs[0])."""
        return CompiledPass(self, batch, vehicle_seeds, fn)

    @torch.no_grad()
    def _vunet_forward(self, batch, vehicle_seeds=None):
        """BASELINE configs[0]: `Vunet_fix_res.forward(y_tilde, x)` alone (vunet/models.py:461-481, mean_appearance: the
        decoder is conditioned on the SAMPLED appearance code z_app, :476).  batch: 'vu_y' [B,3,R,R], 'vu_x' [B,6,R,R]."""
        from . import ops
        vu = self.vunet
        vu.set_vehicle_seeds(vehicle_seeds)
        # forward()'s four calls (:472-476) with the shape encoder - which draws no noise and depends on nothing else - on
        # the side stream, as in `_run`: same launches, same noise order, same bits as `self.vunet.forward(y_tilde, x)`
        (do, ds), join = self._side("vunet_shape", lambda: vu.forward_dec_up(batch["vu_y"]))
        eo, es = vu.forward_enc_up(batch["vu_x"])
        mu_app, z_app = vu.forward_enc_down(eo, es)
        join()
        xt, mu_shape, _ = vu.forward_dec_down(do, ds, z_app)
        return {"x_tilde": xt, "vunet_u8": ops.to_image_u8(xt), "mu_app_0": mu_app[0], "mu_app_1": mu_app[1],
                "mu_shape_0": mu_shape[0], "mu_shape_1": mu_shape[1]}

    def vunet_forward(self, batch, vehicle_seeds=None, check: Optional[str] = "sync"):
        """`_vunet_forward` under the pipeline's range guard (see `run`)."""
        rng = torch.get_rng_state() if (vehicle_seeds is None and check == "sync") else None
        return self._guarded(self._vunet_forward, (batch, vehicle_seeds), check, rng)

    # Range guard of the split-fp16 contraction (ops.py): the networks' own per-call checks are deferred while a pass
    # is being issued (they would synchronise the host once per network and undo the stream overlap); the pass is
    # checked as a whole instead.  check="sync" (default): read the status word after the pass and, if an operand
    # left the split's range, redo the pass in exact fp32 - the returned tensors are always valid.  check="async":
    # return without synchronising (the caller keeps issuing passes); the status word is sticky, and `finish()`
    # says whether any pass since the last call was affected - call it before consuming outputs.
    def status_word(self) -> torch.Tensor:
        """The range-status word this pipeline's passes report to - its own, not the device-wide one the module entry
        points read and clear: an entry-point call between async passes and `finish()` can neither clear nor inherit
        a flag raised by those passes."""
        from . import ops
        if self._status is None:
            self._status = ops.new_status_word(self.device)
        return self._status

    def _guarded(self, fn, args, check: str, rng_state):
        from . import ops
        if not ops.range_guarded() or check is None:
            return fn(*args)
        word = self.status_word()
        with ops.defer_range_check(), ops.status_scope(word):
            out = fn(*args)
        if check == "async":
            return out
        with torch.cuda.device(self.device):
            hit = ops.range_exceeded(self.device, word=word)
        if not hit:
            return out
        if rng_state is not None:
            torch.set_rng_state(rng_state)
        with ops.defer_range_check(), ops.precision("f32"):
            return fn(*args)

    def finish(self) -> bool:
        """Synchronise the device and report (and clear) the range status of the passes issued with check="async":
        True = some pass staged an operand outside the split-fp16 range; its outputs must be recomputed
        (`ops.precision("f32")`)."""
        from . import ops
        torch.cuda.synchronize(self.device)
        if not ops.range_guarded():
            return False
        with torch.cuda.device(self.device):
            return ops.range_exceeded(self.device, word=self.status_word())

    def run(self, batch: Dict[str, torch.Tensor], vehicle_seeds: Optional[Sequence[int]] = None,
            check: Optional[str] = "sync") -> Dict[str, torch.Tensor]:
        """batch: 'hg_x' [B,3,R,R], 'icn_x' [B,21,R,R], 'vu_x' [B,6,R,R], 'vu_y' [B,3,R,R]
        (+ 'ec_img','ec_gray','ec_edge','ec_mask' with inpaint).  All device-resident.
        Returns 'kp_idx' int32 [B,12], 'icn_u8' / 'vunet_u8' uint8 [B,R,R,3] (+ 'inpaint_u8').
        vehicle_seeds: one VUnet noise seed per sample (e.g. base + global vehicle index) - makes the result of
        a vehicle independent of how the vehicles are sharded over ranks; None = the reference's global RNG.
        check: range guard of the split-fp16 path, see `_guarded`."""
        rng = torch.get_rng_state() if (vehicle_seeds is None and check == "sync") else None
        return self._guarded(self._run, (batch, vehicle_seeds), check, rng)

    @torch.no_grad()
    def _run(self, batch, vehicle_seeds):
        from . import ops
        B, R = batch["hg_x"].shape[0], batch["hg_x"].shape[-1]
        if B == 0:                                                # a rank whose shard is empty (fewer vehicles than ranks)
            dev = self.device
            out = {"kp_idx": torch.empty((0, 12), dtype=torch.int32, device=dev),
                   "icn_u8": torch.empty((0, R, R, 3), dtype=torch.uint8, device=dev),
                   "vunet_u8": torch.empty((0, R, R, 3), dtype=torch.uint8, device=dev)}
            if self.inpaint:
                out["inpaint_u8"] = torch.empty((0, R, R, 3), dtype=torch.uint8, device=dev)
            if self.cad is not None:
                out["cad_logits"] = torch.empty((0, 10), dtype=torch.float32, device=dev)
            return out
        self.vunet.set_vehicle_seeds(vehicle_seeds)

        def hg():
            out = {"kp_idx": ops.argmax_hw(self.hg(batch["hg_x"])["heatmaps"][-1])}
            if self.cad is not None:                              # same crop, same branch (:66-69): a stream of its own costs more
                out["cad_logits"] = self.cad(batch["hg_x"])
            return out

        def icn():
            return {"icn_u8": ops.to_image_u8(self.icn(batch["icn_x"]))}

        def vunet():
            vu = self.vunet
            # trajectory_inference.py:230-233; the shape encoder runs beside the appearance half (it draws no noise,
            # so the host-side draw order is the reference's either way)
            (do, ds), join = self._side("vunet_shape", lambda: vu.forward_dec_up(batch["vu_y"]))
            eo, es = vu.forward_enc_up(batch["vu_x"])
            mu_app, _ = vu.forward_enc_down(eo, es)
            join()
            xt, _, _ = vu.forward_dec_down(do, ds, mu_app)
            # (the appearance code is what a vehicle's FUTURE frames are rendered with, :424-426: run_frame hands it out)
            return {"vunet_u8": ops.to_image_u8(xt), "mu_app_0": mu_app[0], "mu_app_1": mu_app[1]}

        def inpaint():
            e = self.edge(batch["ec_gray"], batch["ec_edge"], batch["ec_mask"])      # :124-129
            p = self.inp(batch["ec_img"], e, batch["ec_mask"])
            return {"inpaint_u8": ops.merge_u8(p, batch["ec_img"], batch["ec_mask"])}

        # the ICN's big launches are issued first, on the caller's stream
        return self._branches([("icn", icn), ("vunet", vunet), ("hg", hg)] + ([("inpaint", inpaint)] if self.inpaint else []))

    # ------------------------------------------------------------------------------------------ per-frame chain
    def run_frame(self, scene: Dict, check: Optional[str] = "sync", replay: bool = False) -> Dict:
        """One frame's vehicles from detector boxes to the two composited frames, device-resident: the reference's
        per-vehicle order (trajectory_inference.py:55-250, first frame) batched over the V vehicles of the frame:

            box crop -> hourglass -> argmax -> keypoints in frame pixels -> pose fit (4 LM starts per vehicle)
            plane warp (homographies fitted on the host) -> ICN inputs -> ICN -> to_image(from_LAB=True)
            VUnet inputs -> VUnet first-frame -> to_image
            resize-back + masked paste of every vehicle, in vehicle order (later vehicles overwrite earlier ones)

        scene (see `synth_frame`): 'frame' uint8 [H, W, 3] (CUDA); 'bboxes' int [V, 4] (host: the detector's boxes);
        'masks' uint8 [V, H, W] (CUDA, non-zero = vehicle: the rendered sketch mask, the reference's ~sketch_mask);
        'src_sketch' / 'dst_sketch' uint8 [V, H, W, 3]; 'src_planes' uint8 [V, 5, H, W, 3]; 'src_kp' / 'dst_kp' host lists
        [V][5] of int32 [n, 2] plane corner points; 'src_vis' / 'dst_vis' host [V, 5]; 'kp3d' float32 [V, 12, 3] (host: the
        CAD model's keypoints); 'focals' / 'centers' [2] (host); optional 'background' uint8 [H, W, 3] (default: the frame).
        What the reference computes BETWEEN the pose fit and the plane warp - rendering the posed CAD model into sketches,
        masks and plane corner points with Open3D (warp_learn/vehicle_utils.py) - is out of scope: those
        are inputs here, and the pose is an output for that renderer.  A pipeline built with inpaint=True also takes
        'inpaint' = {'boxes' host int [V, 4] (bbox_new_img: the 1.3x detector box clipped to the frame), 'img' float32
        [V, 3, R, R], 'gray' / 'edge' / 'mask' float32 [V, 1, R, R] in [0, 1]} - what create_inpaint_inputs_shape hands
        EdgeConnect (utils/inpaint_utils.py:35-58: Mask R-CNN mask, dilation, Canny - host steps, out of scope) - runs
        EdgeModel -> InpaintingModel -> merge as a fourth branch (:124-129), composites every vehicle's inpainted box under
        its pasted crop in the reference's per-vehicle order (:130-145) and returns 'inpaint_u8' [V, R, R, 3] as well.

        replay=True issues the three networks as ONE recorded-plan replay (`CompiledPass`, recorded on the first frame with
        this many vehicles and kept per vehicle count) instead of ~370 launches from Python: at 8 vehicles per frame the
        interpreter, not the GPU, bounds the eager form.

        A pipeline built with cad=True classifies every vehicle's box crop (VGG-19, :66-69) and returns 'cad_idx' int64 [V]; with
        'kp3d_bank' float32 [n_cad, 12, 3] in the scene the pose fit uses the chosen model's keypoints (:82-88) instead of 'kp3d'.
        With an initialised process group of more than one rank the frame's vehicles are sharded over the ranks (every rank
        passes the same scene; rank 0 returns the result, the other ranks a dict holding only 'state'; scene['shard'] = False
        keeps a rank on its own).  'state' is RANK-LOCAL then: the appearance codes and central crops of the rank's own
        vehicles [lo, hi) - all frames of a vehicle stay on one rank, `run_later_frame` picks the shard up from it.

        Returns 'kp_idx' int32 [V, 12], 'kp_xy' float32 [V, 12, 2], 'pose' = list of (error, rvec [3, 1], tvec [3, 1]),
        'icn_u8' / 'vunet_u8' uint8 [V, R, R, 3] (BGR), 'frame_icn' / 'frame_vunet' uint8 [H, W, 3], 'geom' int32 [V, 8],
        'state' = what `run_later_frame` needs to render the same vehicles' future frames (VUnet appearance code, central crop)."""
        rng = torch.get_rng_state() if check == "sync" else None
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        if not _one_rank(self.group) and scene.get("shard", True):
            # One frame's vehicles over the ranks (BASELINE configs[3]: 64 vehicles of a frame, 8 shards of 8):
            # every rank holds the scene and renders vehicles shard_range(V, rank, world); the uint8 crops, keypoint indices
            # and crop rows travel to rank 0 (gather_in_order: five small messages per frame, no other exchange), which fits
            # the poses and pastes in vehicle order.  The range guard stays local to a rank's own vehicles - it runs
            # before the first collective, so a rank that redoes its shard in fp32 cannot desynchronise the gathers.
            rank = dist.get_rank(self.group)
            V = len(scene["bboxes"])
            lo, hi = shard_range(V, rank, world)
            local = self._guarded(self._frame_local, (slice_scene(scene, lo, hi), replay), check, rng)
            out = self._frame_gather_finish(scene, local, (lo, hi, V))
            if rank != 0:
                return out                                        # {'state': ...}: this rank's vehicles' appearance codes stay here
        else:
            out = self._guarded(self._run_frame, (scene, replay), check, rng)
        # the reference's host epilogue of the pose fit (argmin over the four starts, sign flip): 4 x 7 numbers per vehicle
        from .utils.pnp_utils import select_and_flip
        rv, tv, er = (t.cpu().numpy() for t in out.pop("_pose_raw"))
        out["pose"] = [select_and_flip(rv[i], tv[i], er[i]) for i in range(rv.shape[0])]
        return out

    def run_frames(self, scenes, replay: bool = True):
        """`run_frame` over a sequence of frames (the reference's outer loop, trajectory_inference.py:283-300), software-
        pipelined one frame deep: frame i+1's host work (40 homography fits, ~40 launches of glue, the networks' replay)
        is issued while the GPU still runs frame i, and frame i's results are collected afterwards - its range status
        and raw pose come back through pinned buffers filled by stream-ordered copies and one event, so reading them
        waits for frame i only.  A generator: yields one `run_frame`-shaped dict per scene, in order; every tensor in it
        is the caller's (nothing aliases a later frame's buffers).  A frame whose split-fp16 range status is raised is
        redone in exact fp32 before it is yielded, with the RNG state it was issued under."""
        import torch.distributed as dist
        if not _one_rank(self.group):
            # sharded frames, one frame deep as well: every rank issues its shard of frame i+1 before frame i's crops are
            # gathered; the gather and rank 0's frame-level part run on a communication stream that waits for frame i's
            # launches only, so they overlap frame i+1's networks.  Yields `run_frame`'s sharded results (rank 0: the frame,
            # other ranks: {'state': ...}); a scene with shard = False is not pipelined.
            pending = None
            for scene in scenes:
                if not scene.get("shard", True):
                    if pending is not None:
                        yield self._collect_sharded(pending)
                        pending = None
                    yield self.run_frame(scene, replay=replay)
                    continue
                ticket = self._issue_sharded(scene, replay)
                if pending is not None:
                    yield self._collect_sharded(pending)
                pending = ticket
            if pending is not None:
                yield self._collect_sharded(pending)
            return
        pending = None
        for scene in scenes:
            ticket = self._issue_frame(scene, replay)
            if pending is not None:
                yield self._collect_frame(pending)
            pending = ticket
        if pending is not None:
            yield self._collect_frame(pending)

    def _issue_frame(self, scene, replay):
        from . import ops
        guarded = ops.range_guarded()
        rng = torch.get_rng_state() if (guarded and scene.get("vehicle_seeds") is None) else None
        word = self.status_word() if guarded else None
        with torch.cuda.device(self.device):
            if guarded:
                with ops.defer_range_check(), ops.status_scope(word):
                    out = self._run_frame(scene, replay)
            else:
                out = self._run_frame(scene, replay)
            # stream-ordered read-back: the three raw pose arrays and the status word into pinned memory, the word cleared
            # for the next frame (whose launches queue behind these copies), one event to wait on
            ring = self.__dict__.setdefault("_frame_pins", [])
            raw = out.pop("_pose_raw")
            hit = [i for i, pr in enumerate(ring) if all(a.shape == b.shape for a, b in zip(pr[0], raw))]
            pins = ring.pop(hit[0]) if hit else \
                ([torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in raw], torch.zeros(1, dtype=torch.int32, pin_memory=True))
            for h, t in zip(pins[0], raw):
                h.copy_(t, non_blocking=True)
            if guarded:
                pins[1].copy_(word[:1], non_blocking=True)
                word.zero_()
            ev = torch.cuda.Event()
            ev.record()
        return {"out": out, "pins": pins, "event": ev, "scene": scene, "replay": replay, "rng": rng, "guarded": guarded}

    def _collect_frame(self, t):
        from . import ops
        from .utils.pnp_utils import select_and_flip
        t["event"].synchronize()
        (rv, tv, er), status = t["pins"]
        if t["guarded"] and int(status[0]) != 0:                  # rare: this frame again, in exact fp32
            cur = torch.get_rng_state()
            if t["rng"] is not None:
                torch.set_rng_state(t["rng"])
            with ops.defer_range_check(), ops.precision("f32"):
                out = self.run_frame(t["scene"], check=None, replay=False)
            if t["rng"] is not None:
                torch.set_rng_state(cur)
            self.__dict__["_frame_pins"].append(t["pins"])
            return out
        out = t["out"]
        rv, tv, er = rv.numpy().copy(), tv.numpy().copy(), er.numpy().copy()
        self.__dict__["_frame_pins"].append(t["pins"])
        out["pose"] = [select_and_flip(rv[i], tv[i], er[i]) for i in range(rv.shape[0])]
        return out

    def _issue_sharded(self, scene, replay):
        """This rank's shard of a frame, issued without waiting for it (the sharded counterpart of `_issue_frame`)."""
        import torch.distributed as dist
        from . import ops
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        V = len(scene["bboxes"])
        lo, hi = shard_range(V, rank, world)
        sub = slice_scene(scene, lo, hi)
        guarded = ops.range_guarded()
        rng = torch.get_rng_state() if (guarded and scene.get("vehicle_seeds") is None) else None
        word = self.status_word() if guarded else None
        with torch.cuda.device(self.device):
            if guarded:
                with ops.defer_range_check(), ops.status_scope(word):
                    local = self._frame_local(sub, replay)
            else:
                local = self._frame_local(sub, replay)
            pin = None
            if guarded:
                ring = self.__dict__.setdefault("_shard_pins", [])
                pin = ring.pop() if ring else torch.zeros(1, dtype=torch.int32, pin_memory=True)
                pin.copy_(word[:1], non_blocking=True)
                word.zero_()
            ev = torch.cuda.Event()
            ev.record()
        return {"local": local, "pin": pin, "event": ev, "scene": scene, "sub": sub, "shard": (lo, hi, V), "rng": rng, "guarded": guarded}

    def _collect_sharded(self, t):
        from . import ops
        from .utils.pnp_utils import select_and_flip
        t["event"].synchronize()                                  # this frame's launches only; the next frame's are already queued
        local = t["local"]
        if t["guarded"]:
            hit = int(t["pin"][0]) != 0
            self.__dict__["_shard_pins"].append(t["pin"])
            if hit:                                               # rare: this rank's shard again, in exact fp32 (before any collective)
                cur = torch.get_rng_state()
                if t["rng"] is not None:
                    torch.set_rng_state(t["rng"])
                with ops.defer_range_check(), ops.precision("f32"):
                    local = self._frame_local(t["sub"], False)
                if t["rng"] is not None:
                    torch.set_rng_state(cur)
                torch.cuda.synchronize(self.device)
        main = torch.cuda.current_stream(self.device)
        comm = self.__dict__.get("_comm_stream")
        if comm is None:
            comm = self.__dict__["_comm_stream"] = torch.cuda.Stream(device=self.device)
        for v in local.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(comm)
        with torch.cuda.device(self.device), torch.cuda.stream(comm):
            out = self._frame_gather_finish(t["scene"], local, t["shard"])
            if "_pose_raw" in out:
                rv, tv, er = (x.cpu().numpy() for x in out.pop("_pose_raw"))          # waits for the communication stream only
                out["pose"] = [select_and_flip(rv[i], tv[i], er[i]) for i in range(rv.shape[0])]
        main.wait_stream(comm)                                    # the caller consumes the results on its own stream
        for v in out.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(main)
        return out

    def _run_frame(self, scene, replay=False):
        """One rank, every vehicle: the per-vehicle part, then the frame-level part."""
        return self._frame_finish(scene, self._frame_local(scene, replay))

    GATHERED = ("kp_idx", "icn_u8", "vunet_u8", "geom")

    def _gather_keys(self, first_frame: bool):
        if not first_frame:
            return ("icn_u8", "vunet_u8", "geom")
        return self.GATHERED + (("inpaint_u8",) if self.inpaint else ()) + (("cad_idx",) if self.cad is not None else ())

    def _gather_local(self, local, V, first_frame=True):
        """The frame's only exchange: this rank's crops / keypoint indices / crop rows -> rank 0, in vehicle order (a few small
        messages; RCCL gather on device tensors, gloo through the host).  Returns the full dict on rank 0, None elsewhere."""
        import torch.distributed as dist
        full = {}
        for k in self._gather_keys(first_frame):
            g = gather_in_order(local[k].contiguous(), V, self.group)
            full[k] = None if g is None else g.to(self.device)
        return full if dist.get_rank(self.group) == 0 else None

    @staticmethod
    def _local_state(local, shard):
        """What `run_later_frame` needs for THIS rank's vehicles (they never move: SURVEY.md 8e)."""
        if "mu_app_0" not in local:
            return None
        return {"appearance": [local["mu_app_0"], local["mu_app_1"]], "central": local["central"], "shard": tuple(shard), "sharded": True}

    def _frame_gather_finish(self, scene, local, shard):
        """Sharded first frame after the rank's local part: gather -> (rank 0) frame-level part.  Every rank gets 'state'."""
        state = self._local_state(local, shard)
        full = self._gather_local(local, shard[2], True)
        if full is None:
            return {"state": state}
        out = self._frame_finish(scene, full)                      # (no appearance codes in `full`: they stay on their ranks)
        out["state"] = state
        return out

    @torch.no_grad()
    def _frame_local(self, scene, replay=False):
        """The per-vehicle part of a frame for the vehicles `scene` lists (all of them, or one rank's shard): glue, the
        networks, Lab -> BGR.  Returns 'kp_idx', 'icn_u8' (BGR), 'vunet_u8', 'geom' (+ 'inpaint_u8'), device tensors."""
        import numpy as np

        from . import frame_ops as fo
        from . import ops
        from .warp_learn import planes_utils as pu
        dev = self.device
        frame = scene["frame"]
        H, W, _ = frame.shape
        bboxes = np.asarray(scene["bboxes"]).reshape(-1, 4)
        V, R = bboxes.shape[0], 256
        seeds = scene.get("vehicle_seeds")
        with torch.cuda.device(dev):
            inp = scene.get("inpaint") if self.inpaint else None
            if self.inpaint and inp is None:
                raise ValueError("run_frame: this pipeline was built with inpaint=True; the scene needs 'inpaint' = "
                                 "{'boxes' [V, 4], 'img' [V, 3, R, R], 'gray' / 'edge' / 'mask' [V, 1, R, R]}")
            if V == 0:                                            # no vehicle in the frame (or an empty shard)
                e8 = lambda: torch.empty((0, R, R, 3), dtype=torch.uint8, device=dev)   # noqa: E731
                out = {"kp_idx": torch.empty((0, 12), dtype=torch.int32, device=dev), "icn_u8": e8(), "vunet_u8": e8(),
                       "geom": torch.empty((0, 8), dtype=torch.int32, device=dev)}
                if inp is not None:
                    out["inpaint_u8"] = e8()
                if self.cad is not None:
                    out["cad_idx"] = torch.empty((0,), dtype=torch.int64, device=dev)
                # an empty shard still hands out a (zero-vehicle) state, so that `run_later_frame` takes part in the gathers
                out.update(mu_app_0=torch.empty((0, 128, R // 64, R // 64), device=dev), mu_app_1=torch.empty((0, 128, R // 32, R // 32), device=dev),
                           central=e8())
                return out
            replay = replay and ops.RECORDER is None
            cps = self.__dict__.setdefault("_frame_plans", {})
            cp = cps.get((V, ops.PRECISION)) if replay else None
            if cp is not None and [n.generation for n in self._nets] != cp.generations:
                cp = None
            tgt = cp.inputs if cp is not None else {}            # a recorded pass's inputs are written in place
            # ---- host: the homography fits of every plane of every vehicle (1.2 ms for 8 vehicles), before any launch
            jobs = pu.warp_jobs_frame(scene["src_kp"], scene["dst_kp"], scene["src_vis"], scene["dst_vis"])
            # ---- uint8 glue on the caller's stream
            geom_box = fo.box_geometry((H, W), bboxes, dev)
            img_bbox = fo.crop_resize(frame, geom_box, (R, R), 0)                              # :58-60
            hg_x = fo.crop_resize(frame, geom_box, (R, R), 1, fo.IMAGENET_MEAN, fo.IMAGENET_STD, out=tgt.get("hg_x"))   # :61-65
            central = fo.central_crop(img_bbox)                                                # vehicle_utils.py:49-52
            warped = pu.warp_planes_batch(scene["src_planes"], jobs)                           # :171-175
            _, geom = fo.mask_bbox_geom(scene["masks"])
            icn_x = pu.icn_inputs_device(warped, scene["dst_sketch"], central, geom, R, R, out=tgt.get("icn_x"))   # :179-180
            vu_x, vu_y = fo.vunet_inputs(frame, scene["masks"], scene["src_sketch"], scene["dst_sketch"], geom, R,
                                         out=(tgt["vu_x"], tgt["vu_y"]) if tgt else None)       # :203-228
            # ---- the three networks: the crop pass of `run` (three stream branches), eagerly or as one plan replay
            nets_in = {"hg_x": hg_x, "icn_x": icn_x, "vu_x": vu_x, "vu_y": vu_y}
            if inp is not None:                                   # :121: create_inpaint_inputs_shape's four tensors, given
                nets_in.update(ec_img=inp["img"], ec_gray=inp["gray"], ec_edge=inp["edge"], ec_mask=inp["mask"])
            if replay:
                if cp is None:
                    # one recorded pass (with its private pool of intermediates) per vehicle count; a video whose count varies
                    # keeps the FRAME_PLANS most recently used ones
                    cps.pop((V, ops.PRECISION), None)
                    while len(cps) >= FRAME_PLANS:
                        cps.pop(next(iter(cps)))
                    cp = cps[(V, ops.PRECISION)] = CompiledPass(self, nets_in, seeds)
                else:
                    cps[(V, ops.PRECISION)] = cps.pop((V, ops.PRECISION))     # most recently used last
                out = dict(cp._issue(nets_in, seeds))
                for k in ("vunet_u8", "kp_idx", "inpaint_u8", "cad_logits", "mu_app_0", "mu_app_1"):    # the plan's buffers belong to its next replay
                    if k in out:
                        out[k] = out[k].clone()
            else:
                out = self._run(nets_in, seeds)                                                # :75-79, :182, :230-234
            out["icn_u8"] = pu.lab2bgr(out["icn_u8"])                                          # to_image(from_LAB=True), :182
            out["geom"] = geom
            out["central"] = central
            if "cad_logits" in out:
                out["cad_idx"] = out.pop("cad_logits").argmax(1)                               # :69
        return out

    @torch.no_grad()
    def _frame_finish(self, scene, out):
        """The frame-level part, on the rank that holds every vehicle's crops (`out`: _frame_local's dict for ALL the
        scene's vehicles, in vehicle order): keypoints -> frame pixels -> pose fit, ordered paste into the two frames."""
        import numpy as np

        from . import frame_ops as fo
        from . import ops
        from .utils.pnp_utils import cpc_fit_device
        from .warp_learn import planes_utils as pu
        dev = self.device
        frame = scene["frame"]
        H, W, _ = frame.shape
        bboxes = np.asarray(scene["bboxes"]).reshape(-1, 4)
        V, R = bboxes.shape[0], 256
        inp = scene.get("inpaint") if self.inpaint else None
        out = dict(out)
        with torch.cuda.device(dev):
            back = frame if inp is not None else scene.get("background", frame)   # :133-143: with --inpaint the composite starts from the frame
            if V == 0:
                out["kp_xy"] = torch.empty((0, 12, 2), dtype=torch.float32, device=dev)
                out["_pose_raw"] = tuple(torch.empty(sh, dtype=torch.float32, device=dev) for sh in ((0, 4, 3), (0, 4, 3), (0, 4)))
                out["frame_icn"], out["frame_vunet"] = back.clone(), back.clone()
                return out
            f32 = lambda a: ops.h2d(np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float32).reshape(-1, 2), (V, 2))), dev)   # noqa: E731
            if self.cad is not None and scene.get("kp3d_bank") is not None:                    # :82-88: the chosen CAD model's keypoints
                kp3d = ops.h2d(np.asarray(scene["kp3d_bank"], np.float32), dev)[out["cad_idx"]]
            else:
                kp3d = ops.h2d(np.asarray(scene["kp3d"], np.float32), dev)
            geom_box = fo.box_geometry((H, W), bboxes, dev)
            out["kp_xy"] = fo.keypoints_to_frame(out["kp_idx"], geom_box, (R // 4, R // 4))   # :95-97 (64 x 64 heat-maps)
            out["_pose_raw"] = cpc_fit_device(f32(scene["focals"]), f32(scene["centers"]), out["kp_xy"], kp3d)   # :104-105
            box = {}
            if inp is not None:
                rows = [[int(b[0]), int(b[1]), int(b[2]), int(b[3]), 0, 0, 0, 0] for b in np.asarray(inp["boxes"]).reshape(-1, 4)]
                box = dict(box_images=out["inpaint_u8"], box_geom=ops.h2d(rows, dev, torch.int32))
            out["frame_icn"] = pu.paste_back_device(back, out["icn_u8"], out["geom"], scene["masks"], **box)       # :184-198
            out["frame_vunet"] = pu.paste_back_device(back, out["vunet_u8"], out["geom"], scene["masks"], **box)   # :236-250
            if "mu_app_0" in out:                                 # what `run_later_frame` renders these vehicles' future frames with
                out["state"] = {"appearance": [out.pop("mu_app_0"), out.pop("mu_app_1")], "central": out.pop("central"),
                                "shard": (0, V, V), "sharded": False}
        return out

    # ------------------------------------------------------------------------------------------ future frames of a clip
    def run_later_frame(self, scene: Dict, state: Dict, check: Optional[str] = "sync", replay: bool = False) -> Dict:
        """A FUTURE frame of vehicles whose first frame `run_frame` rendered (trajectory_inference.py:283-450, per vehicle and
        trajectory step): no hourglass, no pose fit, no appearance encoder - the source planes warped to the new pose -> ICN
        inputs (with the first frame's central crop) -> ICN -> Lab image (:376-391); the new sketch -> VUnet shape encoder ->
        decoder conditioned on the FIRST frame's appearance code (:415-426); ordered paste of both (:393-410, :428-445).

        scene: as for `run_frame`, but rendered for the new pose - 'frame' (what is pasted onto; 'background' overrides),
        'masks', 'dst_sketch', 'dst_kp', 'dst_vis' of the new pose, 'src_planes' / 'src_kp' / 'src_vis' of the first frame,
        optional 'vehicle_seeds' (the shape decoder draws its sampler noise: a seed per vehicle and frame keeps a vehicle's
        images independent of batching); state: `run_frame(...)["state"]` of the same vehicles, same order.
        replay=True issues the two networks (ICN, VUnet shape half) as ONE recorded-plan replay per vehicle count, like
        `run_frame(replay=True)`: five of a clip's six frames are later frames, and at 8 vehicles the interpreter bounds the eager form.
        Returns 'icn_u8' / 'vunet_u8' uint8 [V, R, R, 3] (BGR), 'frame_icn' / 'frame_vunet' uint8 [H, W, 3], 'geom'."""
        rng = torch.get_rng_state() if (check == "sync" and scene.get("vehicle_seeds") is None) else None
        import torch.distributed as dist
        if not _one_rank(self.group) and state.get("sharded"):
            # a clip sharded over ranks (north_star: "vehicles-in-a-frame and frames-in-a-clip shard"): every rank passes the
            # same scene and ITS OWN state (run_frame's, rank-local); it renders its vehicles [lo, hi) from the appearance codes
            # it kept, the uint8 crops and crop rows travel to rank 0, which pastes in vehicle order.  As for the first
            # frame the range guard runs before the collectives.  Rank 0 returns the result, the others None.
            lo, hi, V = state["shard"]
            if int(scene["masks"].shape[0]) != V:
                raise ValueError(f"run_later_frame: the state was made for a frame of {V} vehicles, the scene holds {int(scene['masks'].shape[0])}")
            local = self._guarded(self._later_local, (slice_scene(scene, lo, hi), state, replay), check, rng)
            full = self._gather_local(local, V, False)
            return None if full is None else self._later_finish(scene, full)
        return self._guarded(self._run_later_frame, (scene, state, replay), check, rng)

    def _run_later_frame(self, scene, state, replay=False):
        return self._later_finish(scene, self._later_local(scene, state, replay))

    @torch.no_grad()
    def _later_nets(self, batch, vehicle_seeds=None):
        """The two networks of a later frame (trajectory_inference.py:389, :424-426) as a pass function (`CompiledPass(fn=...)`):
        batch = 'icn_x' [V, 21, R, R], 'vu_y' [V, 3, R, R], 'app0' / 'app1' = the first frame's appearance code."""
        from . import ops
        self.vunet.set_vehicle_seeds(vehicle_seeds)

        def icn():
            return {"icn_u8": ops.to_image_u8(self.icn(batch["icn_x"]))}                   # :389

        def vunet():
            vu = self.vunet
            do, ds = vu.forward_dec_up(batch["vu_y"])                                      # :424
            xt, _, _ = vu.forward_dec_down(do, ds, [batch["app0"], batch["app1"]])         # :425
            return {"vunet_u8": ops.to_image_u8(xt)}                                       # :426

        return self._branches([("icn", icn), ("vunet", vunet)])

    @torch.no_grad()
    def _later_local(self, scene, state, replay=False):
        """The per-vehicle part of a later frame for the vehicles `scene` lists (all, or one rank's shard - `state` holds
        exactly these vehicles): warp, ICN, VUnet shape half.  Returns 'icn_u8' (BGR), 'vunet_u8', 'geom'."""
        from . import frame_ops as fo
        from . import ops
        from .warp_learn import planes_utils as pu
        dev = self.device
        frame = scene["frame"]
        R = 256
        V = int(scene["masks"].shape[0])
        if state["central"].shape[0] != V:
            raise ValueError(f"run_later_frame: the state holds {state['central'].shape[0]} vehicles, the scene {V}")
        with torch.cuda.device(dev):
            if V == 0:
                e8 = torch.empty((0, R, R, 3), dtype=torch.uint8, device=dev)
                return {"icn_u8": e8, "vunet_u8": e8.clone(), "geom": torch.empty((0, 8), dtype=torch.int32, device=dev)}
            jobs = pu.warp_jobs_frame(scene["src_kp"], scene["dst_kp"], scene["src_vis"], scene["dst_vis"])
            warped = pu.warp_planes_batch(scene["src_planes"], jobs)                           # :376-381
            _, geom = fo.mask_bbox_geom(scene["masks"])
            icn_x = pu.icn_inputs_device(warped, scene["dst_sketch"], state["central"], geom, R, R)   # :385-387
            _, vu_y = fo.vunet_inputs(frame, scene["masks"], scene["dst_sketch"], scene["dst_sketch"], geom, R)   # :415-420 (y_tilde only)
            seeds = scene.get("vehicle_seeds")
            nets_in = {"icn_x": icn_x, "vu_y": vu_y, "app0": state["appearance"][0], "app1": state["appearance"][1]}
            replay = replay and ops.RECORDER is None
            if replay:
                cps = self.__dict__.setdefault("_frame_plans", {})
                key = ("later", V, ops.PRECISION)
                cp = cps.get(key)
                if cp is not None and [n.generation for n in self._nets] != cp.generations:
                    cp = None
                if cp is None:
                    cps.pop(key, None)
                    while len(cps) >= FRAME_PLANS:
                        cps.pop(next(iter(cps)))
                    cp = cps[key] = CompiledPass(self, nets_in, seeds, fn=self._later_nets)
                else:
                    cps[key] = cps.pop(key)                   # most recently used last
                out = {k: v.clone() for k, v in cp._issue(nets_in, seeds).items()}     # the plan's buffers belong to its next replay
            else:
                out = self._later_nets(nets_in, seeds)
            out["icn_u8"] = pu.lab2bgr(out["icn_u8"])
            out["geom"] = geom
        return out

    @torch.no_grad()
    def _later_finish(self, scene, out):
        """The frame-level part of a later frame, on the rank that holds every vehicle's crops: the ordered paste."""
        from .warp_learn import planes_utils as pu
        frame = scene["frame"]
        out = dict(out)
        with torch.cuda.device(self.device):
            back = scene.get("background", frame)
            if int(scene["masks"].shape[0]) == 0:
                out["frame_icn"], out["frame_vunet"] = back.clone(), back.clone()
                return out
            out["frame_icn"] = pu.paste_back_device(back, out["icn_u8"], out["geom"], scene["masks"])       # :393-410
            out["frame_vunet"] = pu.paste_back_device(back, out["vunet_u8"], out["geom"], scene["masks"])   # :428-445
        return out

    def run_later_frames(self, scenes, state: Dict, replay: bool = False):
        """`run_later_frame` over the later frames of a clip with ONE frame in flight, like `run_frames` for first frames: frame i+1
        is issued - its homographies fitted on the host, its launches queued - before frame i's range status is read back (pinned,
        one event per frame), so the host part of a later frame (1 ms of homography fits) and the read-back overlap the previous
        frame's networks.  A generator: one `run_later_frame`-shaped dict per scene, in order; a frame whose status is raised is
        redone in exact fp32 before it is yielded.  A sharded state (process group) is not pipelined: frame by frame."""
        from . import ops
        if not _one_rank(self.group) and state.get("sharded"):
            for sc in scenes:
                yield self.run_later_frame(sc, state, replay=replay)
            return
        pending = None
        for scene in scenes:
            guarded = ops.range_guarded()
            rng = torch.get_rng_state() if (guarded and scene.get("vehicle_seeds") is None) else None
            word = self.status_word() if guarded else None
            with torch.cuda.device(self.device):
                if guarded:
                    with ops.defer_range_check(), ops.status_scope(word):
                        out = self._run_later_frame(scene, state, replay)
                else:
                    out = self._run_later_frame(scene, state, replay)
                ring = self.__dict__.setdefault("_later_pins", [])
                pin = ring.pop() if ring else torch.zeros(1, dtype=torch.int32, pin_memory=True)
                if guarded:
                    pin.copy_(word[:1], non_blocking=True)
                    word.zero_()
                ev = torch.cuda.Event()
                ev.record()
            ticket = {"out": out, "pin": pin, "event": ev, "scene": scene, "rng": rng, "guarded": guarded}
            if pending is not None:
                yield self._collect_later(pending, state)
            pending = ticket
        if pending is not None:
            yield self._collect_later(pending, state)

    def _collect_later(self, t, state):
        from . import ops
        t["event"].synchronize()
        hit = t["guarded"] and int(t["pin"][0]) != 0
        self.__dict__["_later_pins"].append(t["pin"])
        if not hit:
            return t["out"]
        cur = torch.get_rng_state()                                # rare: this frame again, in exact fp32
        if t["rng"] is not None:
            torch.set_rng_state(t["rng"])
        with ops.defer_range_check(), ops.precision("f32"):
            out = self.run_later_frame(t["scene"], state, check=None, replay=False)
        if t["rng"] is not None:
            torch.set_rng_state(cur)
        return out

    def run_clip_frames(self, first_scene: Dict, later_scenes, replay: bool = False):
        """A vehicle clip the reference's way (trajectory_inference.py:55-250 then :267-450): the first frame through
        `run_frame`, every future frame through `run_later_frame` with the first frame's state.  Generator of 1 + len(later_scenes)
        results.  Under a process group the whole clip is sharded by vehicle: a vehicle's six frames stay on one rank."""
        first = self.run_frame(first_scene, replay=replay)
        state = first["state"]
        yield first if len(first) > 1 else None
        yield from self.run_later_frames(later_scenes, state, replay=replay)       # one later frame in flight

    def run_clip(self, clip: Dict[str, torch.Tensor], vehicle_seeds: Optional[Sequence[int]] = None,
                 check: Optional[str] = "sync") -> Dict[str, torch.Tensor]:
        rng = torch.get_rng_state() if (vehicle_seeds is None and check == "sync") else None
        return self._guarded(self._run_clip, (clip, vehicle_seeds), check, rng)

    @torch.no_grad()
    def _run_clip(self, clip, vehicle_seeds):
        """Clip mode = what traj_test does per vehicle over a 6-frame clip: 1x hourglass,
        F x ICN, 1x VUnet appearance half, F x VUnet shape half with the frame-0 appearance code reused
        (trajectory_inference.py:75-79, 182, 387-391, 230-233, 424-426) - but batched over vehicles AND
        frames instead of the reference's two nested serial loops.

        clip: 'hg_x' [V,3,R,R], 'icn_x' [V,F,21,R,R], 'vu_x' [V,6,R,R], 'vu_y' [V,F,3,R,R].
        Returns 'kp_idx' int32 [V,12], 'icn_u8' / 'vunet_u8' uint8 [V,F,R,R,3]."""
        from . import ops
        V, F = clip["icn_x"].shape[:2]
        R = clip["hg_x"].shape[-1]

        def hg():
            return {"kp_idx": ops.argmax_hw(self.hg(clip["hg_x"])["heatmaps"][-1])}

        def icn():
            y = self.icn(clip["icn_x"].reshape(V * F, 21, R, R))
            return {"icn_u8": ops.to_image_u8(y).view(V, F, R, R, 3)}

        def vunet():
            vu = self.vunet
            vu.set_vehicle_seeds(vehicle_seeds)                   # appearance half: one stream per vehicle
            eo, es = vu.forward_enc_up(clip["vu_x"])
            mu_app, _ = vu.forward_enc_down(eo, es)
            if vehicle_seeds is not None:                         # shape half: one stream per (vehicle, frame)
                vu.set_vehicle_seeds([int(sd) * 64 + f + 1 for sd in vehicle_seeds for f in range(F)])
            # every frame of a vehicle conditions on that vehicle's appearance code: frame-major repeat
            mu_rep = [m.repeat_interleave(F, dim=0) for m in mu_app]
            do, ds = vu.forward_dec_up(clip["vu_y"].reshape(V * F, 3, R, R))
            xt, _, _ = vu.forward_dec_down(do, ds, mu_rep)
            return {"vunet_u8": ops.to_image_u8(xt).view(V, F, R, R, 3)}

        return self._branches([("icn", icn), ("vunet", vunet), ("hg", hg)])


def _like_layout(t: torch.Tensor) -> torch.Tensor:
    """A zeroed tensor with exactly `t`'s sizes and strides (an NHWC-physical view keeps its padded channel pitch, so
    the glue kernels of `run_frame` can write a recorded pass's inputs in place and the stems read them unchanged)."""
    extent = 1 + sum((n - 1) * st for n, st in zip(t.shape, t.stride())) if t.numel() else 0
    base = torch.zeros(extent + 4, dtype=t.dtype, device=t.device)
    v = base.as_strided(t.shape, t.stride())
    v._fusg_zero_pad = True             # only ever written through views of the logical channels (ops.as_nhwc)
    return v


class CompiledPass:
    """A recorded crop pass of a VehiclePipeline (include/fusg.h, fusg_plan): fixed input shapes, persistent input /
    intermediate / output buffers (a private torch memory pool that lives as long as this object), per-pass host data
    (the VUnet's CPU-drawn sampler noise: same generator, order and shapes as the reference) refreshed before every
    replay.  The returned tensors are the plan's output buffers: the next `run` overwrites them."""

    def __init__(self, pipe: "VehiclePipeline", batch: Dict[str, torch.Tensor], vehicle_seeds=None, fn=None):
        from . import _lib as L
        from . import ops
        self.pipe, self.device = pipe, pipe.device
        self.fn = fn if fn is not None else pipe._run                          # the pass being recorded
        self.precision = ops.PRECISION                                        # recorded into every conv descriptor
        self.keys = sorted(batch.keys())
        with torch.cuda.device(self.device):
            pipe._guarded(self.fn, (batch, vehicle_seeds), None, None)        # warm-up: weight upload, workspaces, streams
            torch.cuda.synchronize(self.device)
            self.stream = torch.cuda.current_stream(self.device)
            self.pool = torch.cuda.MemPool()
            self.rec = ops.PlanRecorder()
            with torch.cuda.use_mem_pool(self.pool, device=self.device):
                self.inputs = {k: _like_layout(batch[k]).copy_(batch[k]) for k in self.keys}
                L.check(L.lib().fusg_plan_begin(self.rec.handle), "plan_begin")
                ops.RECORDER = self.rec
                try:
                    with ops.defer_range_check(), ops.status_scope(pipe.status_word()):
                        self.outputs = self.fn(self.inputs, vehicle_seeds)
                finally:
                    ops.RECORDER = None
                    L.check(L.lib().fusg_plan_end(self.rec.handle), "plan_end")
            torch.cuda.synchronize(self.device)
        self.size = int(L.lib().fusg_plan_size(self.rec.handle))
        # the plan holds raw pointers into each network's packed-weight cache: remember which cache it was recorded on
        self.generations = [n.generation for n in pipe._nets]

    def __del__(self):
        try:
            from . import _lib as L
            torch.cuda.synchronize(self.device)
            L.lib().fusg_plan_destroy(self.rec.handle)
        except Exception:                                                     # interpreter shutdown
            pass

    def _issue(self, batch, vehicle_seeds):
        from . import _lib as L
        for k in self.keys:
            src = batch[k]
            if src.data_ptr() != self.inputs[k].data_ptr():
                if src.shape != self.inputs[k].shape:
                    raise ValueError(f"CompiledPass: input '{k}' has shape {tuple(src.shape)}, recorded {tuple(self.inputs[k].shape)}")
                self.inputs[k].copy_(src, non_blocking=True)
        lib = L.lib()
        slot = lib.fusg_plan_next_slot(self.rec.handle)                       # which pinned ring slot this run reads
        if slot < 0:
            L.check(-1, "plan_next_slot")
        vu = self.pipe.vunet
        vu.set_vehicle_seeds(vehicle_seeds)
        gens = vu.__dict__.get("_vehicle_gens")
        for ring, shapes in self.rec.noise_slots:                             # the reference's draw order
            vu._fill_noise(ring[slot], shapes, gens)
        if PLAN_THREADS:
            L.check(lib.fusg_plan_run_mt(self.rec.handle), "plan_run_mt")     # one issuing host thread per recorded stream
        else:
            L.check(lib.fusg_plan_run(self.rec.handle), "plan_run")
        return self.outputs

    # ---- the recording as one hipGraph: a measurement path (tools/graph_capture_probe.py), not used by the product
    def capture_graph(self) -> int:
        """Capture this recording into ONE hipGraph (fusg_plan_graph_capture: the library re-issues its launch closures
        under a thread-local stream capture; no Python runs inside the capture).  Returns the graph's node count."""
        from . import _lib as L
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            L.check(L.lib().fusg_plan_graph_capture(self.rec.handle, self.stream.cuda_stream, 0), "plan_graph_capture")
        return int(L.lib().fusg_plan_graph_nodes(self.rec.handle))

    def run_graph(self, batch, vehicle_seeds=None):
        """`_issue` through the captured graph: inputs refreshed in place, the noise drawn into the graph's pinned slot."""
        from . import _lib as L
        lib = L.lib()
        with torch.cuda.device(self.device):
            for k in self.keys:
                src = batch[k]
                if src.data_ptr() != self.inputs[k].data_ptr():
                    self.inputs[k].copy_(src, non_blocking=True)
            slot = lib.fusg_plan_graph_slot(self.rec.handle)
            if slot < 0:
                L.check(-1, "plan_graph_slot")
            vu = self.pipe.vunet
            vu.set_vehicle_seeds(vehicle_seeds)
            gens = vu.__dict__.get("_vehicle_gens")
            for ring, shapes in self.rec.noise_slots:
                vu._fill_noise(ring[slot], shapes, gens)
            L.check(lib.fusg_plan_graph_launch(self.rec.handle, self.stream.cuda_stream), "plan_graph_launch")
        return self.outputs

    def run(self, batch: Dict[str, torch.Tensor], vehicle_seeds: Optional[Sequence[int]] = None,
            check: Optional[str] = "sync") -> Dict[str, torch.Tensor]:
        """Same contract as VehiclePipeline.run (incl. the range guard: on a raised status the pass is redone eagerly in
        exact fp32).  Must be called with the stream that was current at compile time."""
        from . import ops
        with torch.cuda.device(self.device):
            if torch.cuda.current_stream(self.device) != self.stream:
                raise RuntimeError("CompiledPass.run: the current stream differs from the one the pass was recorded on")
            if ops.PRECISION != self.precision:
                raise RuntimeError(f"CompiledPass.run: recorded with precision {self.precision}, now {ops.PRECISION}")
            if [n.generation for n in self.pipe._nets] != self.generations:
                raise RuntimeError("CompiledPass.run: a network's parameters changed (load_state_dict / .to() / refresh()) "
                                   "after this pass was recorded - its launches point into the old packed weights; "
                                   "call pipe.compile() again")
            rng = torch.get_rng_state() if (vehicle_seeds is None and check == "sync" and ops.range_guarded()) else None
            out = self._issue(batch, vehicle_seeds)
            if not ops.range_guarded() or check != "sync":
                return out
            if not ops.range_exceeded(self.device, word=self.pipe.status_word()):
                return out
            if rng is not None:
                torch.set_rng_state(rng)
            with ops.defer_range_check(), ops.precision("f32"):
                return self.fn(batch, vehicle_seeds)


def synth_clip(vehicles: int, frames: int, res: int, device, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Synthetic clip-mode inputs (vehicle-major, then frame)."""
    from .synth import synth_inputs
    v = synth_inputs("vunet", vehicles, res, seed)
    return {"hg_x": synth_inputs("hg", vehicles, res, seed)["x"].to(device),
            "icn_x": synth_inputs("icn", vehicles * frames, res, seed)["x"].view(vehicles, frames, 21, res, res).to(device),
            "vu_x": v["x"].to(device),
            "vu_y": synth_inputs("vunet", vehicles * frames, res, seed + 1)["y_tilde"].view(vehicles, frames, 3, res, res).to(device)}


def synth_batch(batch: int, res: int, device, inpaint: bool = False, seed: int = 0, nhwc: bool = False) -> Dict[str, torch.Tensor]:
    """Synthetic, device-resident inputs of the shapes/ranges the reference feeds.
    nhwc=True (CUDA devices): the tensors are handed over in the layout the frame driver's glue kernels write and the
    stems read in place - NHWC-physical, channel pitch padded with zeros to the stem's K-channels (4 / 24 / 8 / 4; logical
    shape and values unchanged) - so a pass starts without the four NCHW -> NHWC conversion launches a caller with
    standard-contiguous tensors pays (`ops.as_nhwc` at every network's entry)."""
    from .synth import synth_inputs
    b = {"hg_x": synth_inputs("hg", batch, res, seed)["x"], "icn_x": synth_inputs("icn", batch, res, seed)["x"]}
    v = synth_inputs("vunet", batch, res, seed)
    b["vu_x"], b["vu_y"] = v["x"], v["y_tilde"]
    if inpaint:
        e = synth_inputs("edge", batch, res, seed)
        b.update(ec_img=e["img"], ec_gray=e["gray"], ec_edge=e["edge"], ec_mask=e["mask"])
    out = {k: t.to(device) for k, t in b.items()}
    if nhwc and torch.device(device).type == "cuda":
        from . import ops
        pitch = {"hg_x": 4, "icn_x": 24, "vu_x": 8, "vu_y": 4}
        with torch.cuda.device(device):
            for k, cp in pitch.items():
                out[k] = ops.as_nhwc(out[k].contiguous(), cpad=cp)
    return out


def synth_frame(vehicles: int, frame_hw=(720, 1280), device="cuda", seed: int = 0, inpaint: bool = False) -> Dict:
    """A synthetic frame for `VehiclePipeline.run_frame`: smooth random texture, `vehicles` detector boxes, and per
    vehicle what the reference's renderer would hand over - an elliptical sketch (normal-map colours) with its mask,
    five texture-plane quadrilaterals (corner points before and after a small pose change, visibilities) and the planes
    cut out of the frame with them (`fusg_fill_poly_planes_u8`), plus 12 CAD keypoints and camera intrinsics.  All
    pixel data lives on `device`; point lists and boxes are host arrays, as in the reference."""
    import numpy as np

    from .warp_learn import planes_utils as pu
    H, W = frame_hw
    g = np.random.default_rng(seed)
    dev = torch.device(device)
    tg = torch.Generator().manual_seed(seed)
    low = torch.rand((1, 3, H // 16 + 2, W // 16 + 2), generator=tg)
    tex = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)[0]
    tex = (tex + 0.08 * torch.rand((3, H, W), generator=tg)).clamp(0, 1)
    frame = (tex.permute(1, 2, 0) * 255).to(torch.uint8).contiguous().to(dev)
    yy, xx = np.mgrid[0:H, 0:W]
    bboxes, masks, sk_src, sk_dst, planes, src_kp, dst_kp, src_vis, dst_vis, kp3d = [], [], [], [], [], [], [], [], [], []
    for v in range(vehicles):
        bw, bh = int(g.integers(W // 8, W // 4)), int(g.integers(H // 6, H // 3))
        x0 = int(g.integers(-bw // 6, W - bw + bw // 6))                       # some boxes run off the frame
        y0 = int(g.integers(-bh // 6, H - bh + bh // 6))
        bboxes.append([x0, y0, x0 + bw, y0 + bh])
        cx, cy = x0 + bw / 2, y0 + bh / 2
        ell = (((xx - cx) / (0.45 * bw)) ** 2 + ((yy - cy) / (0.42 * bh)) ** 2) <= 1.0
        if not ell.any():
            ell[min(max(int(cy), 0), H - 1), min(max(int(cx), 0), W - 1)] = True
        m = ell.astype(np.uint8)
        nrm = np.stack([(xx - cx) / (0.45 * bw), (yy - cy) / (0.42 * bh), np.ones_like(xx, dtype=np.float64) * 0.6], -1)
        col = np.clip((nrm * 0.5 + 0.5) * 255, 1, 255).astype(np.uint8) * m[..., None]
        masks.append(m)
        sk_src.append(col)
        sk_dst.append(np.ascontiguousarray(col[..., ::-1]) if v % 2 else col)
        # five quadrilaterals / hexagons inside the box, and the same after a small perturbation
        def poly(n, ox, oy, sx, sy):
            ang = np.sort(g.uniform(0, 2 * np.pi, n))
            return np.stack([cx + ox * bw + sx * bw * np.cos(ang), cy + oy * bh + sy * bh * np.sin(ang)], 1)
        spec = [(6, -0.18, 0.0, 0.22, 0.3), (6, 0.18, 0.0, 0.22, 0.3), (4, 0.0, -0.2, 0.25, 0.15), (4, 0.0, 0.05, 0.2, 0.2),
                (4, 0.0, 0.25, 0.25, 0.12)]
        s_pts = [poly(*sp) for sp in spec]
        d_pts = [p + g.normal(0, 0.02 * bw, p.shape) for p in s_pts]
        src_kp.append([np.int32(p) for p in s_pts])
        dst_kp.append([np.int32(p) for p in d_pts])
        vis = g.integers(0, 2, 5).astype(np.uint8)
        vis[2] = 1
        src_vis.append(vis)
        dv = vis.copy()
        if v % 3 == 1:                                                        # exercise the left/right symmetry swap
            dv[0], dv[1] = 0, 1
        dst_vis.append(dv)
        planes.append(pu.fill_planes(frame, src_kp[-1]))
        kp3d.append((g.uniform(-1, 1, (12, 3)) * np.array([0.9, 0.5, 2.0]) * 5).astype(np.float32))
    t8 = lambda a: torch.from_numpy(np.ascontiguousarray(np.stack(a))).to(dev)   # noqa: E731
    extra = {}
    if inpaint:                                                               # EdgeConnect's inputs per vehicle (given, see run_frame)
        from .synth import synth_inputs
        e = synth_inputs("edge", vehicles, 256, seed)
        boxes = []
        for x0, y0, x1, y1 in bboxes:                                         # the 1.3x box, clipped to the frame
            cx, cy, bw, bh = (x0 + x1) / 2, (y0 + y1) / 2, 1.3 * (x1 - x0), 1.3 * (y1 - y0)
            bx0, by0 = max(0, int(cx - bw / 2)), max(0, int(cy - bh / 2))
            boxes.append([bx0, by0, max(bx0 + 2, min(W - 1, int(cx + bw / 2))), max(by0 + 2, min(H - 1, int(cy + bh / 2)))])
        extra["inpaint"] = {"boxes": np.asarray(boxes, dtype=np.int64), "img": e["img"].to(dev), "gray": e["gray"].to(dev),
                            "edge": e["edge"].to(dev), "mask": e["mask"].to(dev)}
    return {**extra, "frame": frame, "bboxes": np.asarray(bboxes, dtype=np.int64), "masks": t8(masks), "src_sketch": t8(sk_src),
            "dst_sketch": t8(sk_dst), "src_planes": torch.stack(planes), "src_kp": src_kp, "dst_kp": dst_kp,
            "src_vis": np.stack(src_vis), "dst_vis": np.stack(dst_vis), "kp3d": np.stack(kp3d),
            "focals": np.array([1.1 * W, 1.1 * W], np.float32), "centers": np.array([W / 2, H / 2], np.float32)}


def synth_later_frame(scene: Dict, step: int) -> Dict:
    """The scene of `synth_frame` one trajectory step later, for `run_later_frame`: the same vehicles (masks, first-frame planes
    and their corner points), new plane corner points for the new pose (seeded by `step`), the source sketch as the new pose's
    sketch, and one noise seed per (vehicle, step) when the first frame had per-vehicle seeds."""
    import numpy as np
    g = np.random.default_rng(1000 + step)
    out = dict(scene)
    out["dst_kp"] = [[np.int32(p + g.normal(0, 3.0, p.shape)) for p in veh] for veh in scene["src_kp"]]
    out["dst_sketch"] = scene["src_sketch"]
    if scene.get("vehicle_seeds") is not None:
        out["vehicle_seeds"] = [int(sd) * 64 + step for sd in scene["vehicle_seeds"]]
    return out
