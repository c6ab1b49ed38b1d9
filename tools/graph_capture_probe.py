#!/usr/bin/env python3
"""GPU box: root cause of round 3's failed hipGraph capture of the crop pass, and the hipGraph-vs-plan measurement.

Round 3 (`gpurun_out/r03graph/trace.log`): `torch.cuda.graph` around `VehiclePipeline._run` ended with
hipErrorStreamCaptureInvalidated, the experiment did not end the capture, and torch's allocator asserted at exit.

    python tools/graph_capture_probe.py                 # driver: one child process per probe, then the measurement
    python tools/graph_capture_probe.py probe NAME      # one probe (what the driver starts)
    python tools/graph_capture_probe.py measure [B]     # plan replay vs the same recording as ONE hipGraph

A probe captures ONE candidate under `torch.cuda.CUDAGraph` in GLOBAL error mode - where HIP fails the offending call itself,
so the traceback names the call site - with the capture ended in a `finally` and the process left through os._exit (a
failed capture cannot trip a teardown assert).  Candidates: the whole pass, each network branch alone, and the individual
suspects VERDICT r3 listed (first-use pageable upload of `ops.border_tiles`, `ops._workspace` growth, the status-word read,
the pinned-ring noise copy of `Vunet_fix_res._draw_noise`, `ops.h2d`).
The measurement does not capture Python at all: `fusg_plan_graph_capture` re-issues a recorded plan's launch closures under
a native stream capture (csrc/plan.hip), so eager issue / plan replay / hipGraph replay are the same launches on the same streams."""
import json
import os
import subprocess
import sys
import time
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

PROBES = ["one_launch", "border_tiles_first_use", "workspace_growth", "status_read", "h2d_pinned", "noise_ring", "hg", "icn",
          "vunet_enc", "vunet_dec", "whole_pass_one_stream", "whole_pass_streams"]


def _setup(B=1):
    import torch
    from future_urban_scene_generation_amd import ops
    from future_urban_scene_generation_amd.pipeline import VehiclePipeline, synth_batch
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    ops.set_precision("f16x3")
    pipe = VehiclePipeline(dev)
    batch = synth_batch(B, 256, dev, seed=1)
    return torch, ops, pipe, batch, dev


def probe(name):
    if name != "whole_pass_streams":
        os.environ["FUSG_STREAMS"] = "0"
    torch, ops, pipe, batch, dev = _setup()
    from future_urban_scene_generation_amd import _lib as L
    for _ in range(3):                                  # warm: weights packed and uploaded, workspaces, caches, streams
        pipe.run(batch, check=None)
    torch.cuda.synchronize()
    vu = pipe.vunet

    def whole():
        with ops.defer_range_check(), ops.status_scope(pipe.status_word()):
            pipe._run(batch, None)

    def vunet_enc():
        with ops.defer_range_check():
            eo, es = vu.forward_enc_up(batch["vu_x"])
            vu.forward_enc_down(eo, es)

    def vunet_dec():
        with ops.defer_range_check():
            do, ds = vu.forward_dec_up(batch["vu_y"])
            vu.forward_dec_down(do, ds)

    def noise_ring():
        vu._draw_noise([(1, 128, 4, 4), (1, 128, 8, 8)], dev)

    def border_first():
        ops._BORDER_TILES.clear()                       # a shape never seen: torch.tensor(list, device=...) = pageable upload
        ops.border_tiles(24, 48, dev)

    def ws_growth():
        ops._WS.clear()
        ops._workspace(dev, 1 << 24)

    t32 = ops.nhwc_empty(1, 32, 64, 64, dev, zero=True)
    cands = {"one_launch": lambda: ops.maxpool2(t32),
             "border_tiles_first_use": border_first, "workspace_growth": ws_growth,
             "status_read": lambda: ops.range_exceeded(dev, word=pipe.status_word()),
             "h2d_pinned": lambda: ops.h2d([1.0, 2.0, 3.0], dev),
             "noise_ring": noise_ring,
             "hg": lambda: _deferred(ops, lambda: pipe.hg(batch["hg_x"])),
             "icn": lambda: _deferred(ops, lambda: pipe.icn(batch["icn_x"])),
             "vunet_enc": vunet_enc, "vunet_dec": vunet_dec,
             "whole_pass_one_stream": whole, "whole_pass_streams": whole}
    fn = cands[name]
    res = {"probe": name, "captured": False, "error": None, "where": None, "replay_ok": None}
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    began = False
    with torch.cuda.stream(st):
        try:
            g.capture_begin(capture_error_mode="global")
            began = True
            fn()
        except BaseException as e:                      # the offending call raises at its own site in global mode
            res["error"] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:300])
            tb = traceback.extract_tb(e.__traceback__)
            res["where"] = ["%s:%d %s" % (os.path.relpath(f.filename, REPO) if f.filename.startswith(REPO) else os.path.basename(f.filename), f.lineno, f.name)
                            for f in tb if "graph_capture_probe" not in f.filename][-4:]
        finally:
            if began:
                try:
                    g.capture_end()                     # ALWAYS end the capture
                    res["captured"] = res["error"] is None
                except BaseException as e:
                    res["end_error"] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200])
                    try:                                # the allocator still thinks a capture is under way: tell it otherwise
                        torch._C._cuda_endAllocateToPool(dev.index or 0, g.pool())
                    except BaseException:
                        pass
    if res["captured"]:
        try:
            g.replay()
            torch.cuda.synchronize()
            res["replay_ok"] = True
        except BaseException as e:
            res["replay_ok"] = False
            res["replay_error"] = str(e).splitlines()[0][:200]
    print("PROBE " + json.dumps(res), flush=True)
    sys.stdout.flush()
    os._exit(0)                                         # no interpreter teardown: nothing to assert on


def _deferred(ops, fn):
    with ops.defer_range_check():
        return fn()


def measure(B=1):
    import faulthandler
    faulthandler.enable()
    torch, ops, pipe, batch, dev = _setup(B)
    from future_urban_scene_generation_amd import _lib as L
    out = {"batch": B}
    # (everything on a stream of our own: the legacy null stream cannot be captured, and a capture elsewhere forbids launches on it)
    own = torch.cuda.Stream()
    own.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(own)
    for tag, streams in (("one_stream", "0"), ("streams", "1")):
        os.environ["FUSG_STREAMS"] = streams
        for _ in range(3):
            pipe.run(batch, check=None)
        torch.cuda.synchronize()

        def lat(f, n=40):
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                f()
            torch.cuda.synchronize()
            thr = (time.perf_counter() - t0) / n
            ts = []
            for _ in range(n):
                t1 = time.perf_counter()
                f()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            ts.sort()
            return round(thr * 1e3, 3), round(ts[len(ts) // 2] * 1e3, 3)

        r = {"eager_ms": lat(lambda: pipe.run(batch, check=None))}
        cp = pipe.compile(batch)
        r["plan_ops"] = cp.size
        r["plan_replay_ms"] = lat(lambda: cp._issue(batch, None))
        try:
            r["graph_nodes"] = cp.capture_graph()
            torch.manual_seed(3)
            a = {k: v.clone() for k, v in cp._issue(batch, None).items()}
            torch.manual_seed(3)
            b = cp.run_graph(batch, None)
            torch.cuda.synchronize()
            r["graph_equals_plan_bits"] = all(torch.equal(a[k], b[k]) for k in a)
            r["graph_replay_ms"] = lat(lambda: cp.run_graph(batch, None))
        except BaseException as e:
            r["graph_error"] = "%s: %s" % (type(e).__name__, str(e)[:600])
        out[tag] = r
        print("PARTIAL " + json.dumps({tag: r}), flush=True)
        del cp
    out["columns"] = "[back-to-back ms per pass, median ms of one pass waited for]"
    print("MEASURE " + json.dumps(out), flush=True)
    os._exit(0)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "probe":
        return probe(sys.argv[2])
    if len(sys.argv) > 1 and sys.argv[1] == "measure":
        return measure(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for name in PROBES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "probe", name], capture_output=True, text=True, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("PROBE ")]
        print(lines[-1] if lines else "PROBE " + json.dumps({"probe": name, "rc": r.returncode, "stderr": r.stderr[-600:]}), flush=True)
    for B in (1, 8):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "measure", str(B)], capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("MEASURE ")]
        print(lines[-1] if lines else "MEASURE " + json.dumps({"batch": B, "rc": r.returncode, "stdout": r.stdout[-1500:], "stderr": r.stderr[-3000:]}), flush=True)


if __name__ == "__main__":
    main()
