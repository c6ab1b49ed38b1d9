"""The PYTHONPATH route of INTEGRATION.md §2: with dropin/ in front of a reference checkout, the reference's own import
statements (run_test.py:13-21, trajectory_inference.py:26-29, edgeconnect/models.py:5) resolve to the MI355X modules,
while the reference's other sub-modules of the merged packages keep resolving to the checkout.  CPU only: imports,
constructors and state_dict schemas (no forward)."""
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_lines_resolve_to_the_dropins(tmp_path):
    # a stand-in for the reference checkout: only the sibling modules the merged packages must keep finding
    ref = tmp_path / "reference"
    for pkg, mod in (("warp_learn", "planes_utils"), ("warp_learn", "online_visibility"), ("edgeconnect", "config"),
                     ("edgeconnect", "utils"), ("utils", "crop_utils"), ("utils", "geometry"), ("utils", "pnp_utils")):
        d = ref / pkg
        d.mkdir(parents=True, exist_ok=True)
        (d / "__init__.py").write_text("")
        (d / f"{mod}.py").write_text(f"MARKER = 'reference {pkg}.{mod}'\n")
    (ref / "vunet").mkdir()                                    # the reference's vunet/ has no __init__.py
    (ref / "vunet" / "data_utils.py").write_text("MARKER = 'reference vunet.data_utils'\n")
    code = textwrap.dedent("""
        import json
        from argparse import Namespace
        from stacked_hourglass.models import HourglassNet                      # run_test.py:15
        from warp_learn.models import G_Resnet, get_icn_inputs                 # run_test.py:21, trajectory_inference.py:26
        from vunet.models import Vunet_fix_res                                 # run_test.py:20
        from edgeconnect.models import EdgeModel, InpaintingModel              # run_test.py:13-14
        from edgeconnect.networks import InpaintGenerator, EdgeGenerator, Discriminator   # edgeconnect/models.py:5
        import warp_learn.planes_utils, warp_learn.online_visibility, edgeconnect.config, edgeconnect.utils
        import vunet.data_utils
        import utils.pnp_utils                                                 # trajectory_inference.py:25 (opt-in shim)
        import utils.crop_utils, utils.geometry
        from future_urban_scene_generation_amd.pipeline import load_schema
        from future_urban_scene_generation_amd.synth import schema_of
        nets = {"hg": HourglassNet(num_stacks=2, num_blocks=1, num_classes=12), "icn": G_Resnet(21),
                "vunet": Vunet_fix_res(Namespace(up_mode="subpixel", w_norm=True, drop_prob=0.2, vunet_256=True)),
                "edge": EdgeModel(None).generator, "inpaint": InpaintingModel(None).generator}
        out = {"modules": {k: type(v).__module__ for k, v in nets.items()},
               "schema_ok": {k: list(schema_of(v.state_dict()).items()) == list(load_schema(k).items()) for k, v in nets.items()},
               "siblings": [warp_learn.planes_utils.MARKER, warp_learn.online_visibility.MARKER, edgeconnect.config.MARKER,
                            edgeconnect.utils.MARKER, vunet.data_utils.MARKER],
               "pnp": [utils.pnp_utils.MARKER, utils.pnp_utils.FUSG_DROPIN, utils.pnp_utils.__name__, utils.pnp_utils.__package__],
               "planes": [warp_learn.planes_utils.FUSG_DROPIN, warp_learn.planes_utils.__name__, warp_learn.planes_utils.__package__,
                          warp_learn.planes_utils.__file__],
               "utils_siblings": [utils.crop_utils.MARKER, utils.geometry.MARKER],
               "to_str": str(next(nets["hg"].to("cpu").parameters()).device)}
        print("RESULT " + json.dumps(out))
    """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(REPO, "dropin"), REPO, str(ref)]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert all(m.startswith("future_urban_scene_generation_amd.") for m in out["modules"].values()), out["modules"]
    assert all(out["schema_ok"].values()), out["schema_ok"]
    # both OpenCV-touching shims are opt-in: by default they ARE the reference's modules, loaded as regular modules
    assert out["pnp"] == ["reference utils.pnp_utils", False, "utils.pnp_utils", "utils"]
    assert out["planes"][:3] == [False, "warp_learn.planes_utils", "warp_learn"] and out["planes"][3].startswith(str(ref))
    assert out["utils_siblings"] == ["reference utils.crop_utils", "reference utils.geometry"]
    assert out["siblings"] == ["reference warp_learn.planes_utils", "reference warp_learn.online_visibility",
                               "reference edgeconnect.config", "reference edgeconnect.utils", "reference vunet.data_utils"]


def test_planes_utils_shim_is_opt_in(tmp_path):
    """`from warp_learn.planes_utils import to_image, warp_unwarp_planes` (trajectory_inference.py:28-29): by default the
    shim executes the reference checkout's own file; FUSG_DROPIN_PLANES_UTILS=1 (or install(planes_utils=True)) selects
    the device versions."""
    ref = tmp_path / "reference" / "warp_learn"
    ref.mkdir(parents=True)
    (ref / "__init__.py").write_text("")
    (ref / "planes_utils.py").write_text("def to_image(x, from_LAB):\n    return 'reference'\n\ndef warp_unwarp_planes(*a):\n    return 'reference'\n")
    code = "from warp_learn.planes_utils import to_image, warp_unwarp_planes; import warp_learn.planes_utils as m; print('RESULT', m.FUSG_DROPIN, to_image.__module__, to_image(0, False) if not m.FUSG_DROPIN else '-')"
    base = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(REPO, "dropin"), REPO, str(tmp_path / "reference")]))
    base.pop("FUSG_DROPIN_PLANES_UTILS", None)
    r = subprocess.run([sys.executable, "-c", code], env=base, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESULT False warp_learn.planes_utils reference" in r.stdout
    r = subprocess.run([sys.executable, "-c", code], env=dict(base, FUSG_DROPIN_PLANES_UTILS="1"), capture_output=True, text=True,
                       timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESULT True future_urban_scene_generation_amd.warp_learn.planes_utils" in r.stdout
    code2 = ("import sys, future_urban_scene_generation_amd as f; f.install(); print('A', 'warp_learn.planes_utils' in sys.modules); "
             "f.install(planes_utils=True); import warp_learn.planes_utils as m; print('B', m.__name__)")
    r = subprocess.run([sys.executable, "-c", code2], env=dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, str(tmp_path / "reference")])),
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "A False" in r.stdout and "B future_urban_scene_generation_amd.warp_learn.planes_utils" in r.stdout


def test_pnp_utils_shim_is_opt_in(tmp_path):
    """`from utils.pnp_utils import cpc_rodr_4_angles` (trajectory_inference.py:25): the same policy as planes_utils -
    the reference's own file by default (north_star: PnP stays host-side; the Rodrigues epilogue is unpinned),
    FUSG_DROPIN_PNP=1 / install(pnp=True) selects the device pose fit."""
    ref = tmp_path / "reference" / "utils"
    ref.mkdir(parents=True)
    (ref / "__init__.py").write_text("")
    (ref / "geometry.py").write_text("X = 3\n")
    # (a relative import inside the reference file must keep working when the shim loads it)
    (ref / "pnp_utils.py").write_text("from .geometry import X\n\ndef cpc_rodr_4_angles(*a):\n    return 'reference', X\n")
    code = "from utils.pnp_utils import cpc_rodr_4_angles as f; import utils.pnp_utils as m; print('RESULT', m.FUSG_DROPIN, f.__module__, f() if not m.FUSG_DROPIN else '-')"
    base = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(REPO, "dropin"), REPO, str(tmp_path / "reference")]))
    base.pop("FUSG_DROPIN_PNP", None)
    r = subprocess.run([sys.executable, "-c", code], env=base, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESULT False utils.pnp_utils ('reference', 3)" in r.stdout
    r = subprocess.run([sys.executable, "-c", code], env=dict(base, FUSG_DROPIN_PNP="1"), capture_output=True, text=True,
                       timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESULT True future_urban_scene_generation_amd.utils.pnp_utils" in r.stdout
    code2 = ("import sys, future_urban_scene_generation_amd as f; f.install(); print('A', 'utils.pnp_utils' in sys.modules); "
             "f.install(pnp=True); import utils.pnp_utils as m; print('B', m.__name__)")
    r = subprocess.run([sys.executable, "-c", code2], env=dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, str(tmp_path / "reference")])),
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "A False" in r.stdout and "B future_urban_scene_generation_amd.utils.pnp_utils" in r.stdout
