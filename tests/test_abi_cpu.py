"""CPU: the C-ABI library loads and exports every symbol include/fusg.h declares; host-side
validation rejects bad descriptors without touching a GPU; drop-in schemas match the reference."""
import ctypes as C
import os
import re
from argparse import Namespace

import pytest
import torch

from conftest import REPO, load_schema
from future_urban_scene_generation_amd import _lib as L
from future_urban_scene_generation_amd.synth import schema_of


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.lib()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(REPO, "include", "fusg.h")).read()
    declared = set(re.findall(r"\b(fusg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found"
    assert declared == set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fusg_version() == int(re.search(r"#define FUSG_VERSION (\d+)", hdr).group(1))
    assert lib.fusg_arch() == b"gfx950"


def test_struct_sizes_match_compiled_header(lib):
    assert C.sizeof(L.Tensor) == lib.fusg_sizeof_tensor() == 80
    assert C.sizeof(L.ConvDesc) == lib.fusg_sizeof_conv_desc()


def test_validation_without_gpu(lib):
    d = L.ConvDesc()
    assert lib.fusg_conv2d(C.byref(d), None) == -1          # FUSG_ERR_INVALID before any launch
    assert b"src0" in lib.fusg_last_error()
    t = L.Tensor()
    assert lib.fusg_maxpool2(C.byref(t), C.byref(t), None) == -1
    assert lib.fusg_argmax_hw(C.byref(t), None, None) == -1


def test_plan_heuristic(lib):
    d = L.ConvDesc()
    d.src0.n, d.qh, d.qw = 32, 64, 64
    d.cout, d.cout_pad, d.k_pad, d.nphase = 256, 256, 2304, 1
    assert lib.fusg_conv2d_plan(C.byref(d)) == 0 and d.tile == L.TILE_128x128 and d.ksplit == 1
    d = L.ConvDesc()
    d.src0.n, d.qh, d.qw = 32, 2, 2
    d.cout, d.cout_pad, d.k_pad, d.nphase = 512, 512, 9216, 1
    nbytes = lib.fusg_conv2d_plan(C.byref(d))
    assert d.ksplit > 1 and nbytes == d.ksplit * 128 * 512 * 4


def test_dropin_schemas_match_reference():
    from future_urban_scene_generation_amd.edgeconnect.networks import EdgeGenerator, InpaintGenerator
    from future_urban_scene_generation_amd.stacked_hourglass.models import HourglassNet
    from future_urban_scene_generation_amd.vunet.models import Vunet_fix_res
    from future_urban_scene_generation_amd.warp_learn.models import G_Resnet
    nets = {"hg": HourglassNet(2, 1, 12), "icn": G_Resnet(21),
            "vunet": Vunet_fix_res(Namespace(up_mode="subpixel", w_norm=True, drop_prob=0.2, vunet_256=True)),
            "edge": EdgeGenerator(), "inpaint": InpaintGenerator()}
    for name, m in nets.items():
        assert list(schema_of(m.state_dict()).items()) == list(load_schema(name).items()), name


def test_no_cpu_fallback():
    from future_urban_scene_generation_amd.warp_learn.models import G_Resnet
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G_Resnet(21).eval()(torch.zeros(1, 21, 64, 64))


def test_install_aliases():
    import sys
    import future_urban_scene_generation_amd as pkg
    saved = {k: sys.modules.get(k) for k in pkg._ALIASES}
    try:
        pkg.install()
        from stacked_hourglass.models import HourglassNet          # noqa: F401
        from vunet.models import Vunet_fix_res                      # noqa: F401
        from edgeconnect.models import EdgeModel, InpaintingModel   # noqa: F401
        from warp_learn.models import G_Resnet                      # noqa: F401
        assert HourglassNet.__module__.startswith("future_urban_scene_generation_amd")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
