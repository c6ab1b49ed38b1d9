"""CPU oracle for the per-vehicle novel-view synthesis hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and there only
as the checker / CPU baseline - never as the thing measured or shipped.  The product modules
(``future_urban_scene_generation_amd``) never import this package and raise if the HIP library
is missing.

What it is: a from-scratch functional restatement (plain ``torch.nn.functional`` calls on CPU
tensors, driven by a flat ``state_dict``) of the arithmetic the reference's four network packages
perform at inference time.  Each function cites the reference ``file:line`` it follows (paths are
relative to the reference checkout).  The reference hot path is Python-on-PyTorch (there is no
native code to compile), so the restatement is Python-on-PyTorch too; PyTorch itself is the
reference's un-vendored third-party dependency (README.md:52 "version 1.3 or above"; here
2.10.0 CPU kernels).

Parity pin: tools/gen_golden.py imports the *reference* modules in the build container, loads the
synthetic weights of ``future_urban_scene_generation_amd.synth`` into them and writes their
outputs to ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks every oracle function
against those vectors (bit-exact on the generating machine, <=1e-5 elsewhere because oneDNN picks
different blockings on different CPUs).

PARITY UNPINNED for ``oracle/cv_host.py``: the OpenCV-defined host steps of the reference (homography warp,
fillPoly masks, resize, Lab conversion, findHomography) are restated there from OpenCV's published 8-bit
algorithms, because opencv-python is un-vendored, unpinned (requirements.txt:5) and absent from the build
container, and the reference holds no fixtures for them.  The scikit-image Canny and Open3D
rendering steps are not restated at all.
"""
from .hourglass import hourglass_forward, heatmap_argmax, get_maxima          # noqa: F401
from .icn import icn_forward                                                  # noqa: F401
from .vunet import (vunet_enc_up, vunet_enc_down, vunet_dec_up, vunet_dec_down,    # noqa: F401
                    vunet_forward, depth_to_space, space_to_depth)
from .edgeconnect import (edge_generator_forward, inpaint_generator_forward,  # noqa: F401
                          edge_model_forward, inpaint_model_forward)
from .host import to_image_u8, to_tensor_pm1, ssim                            # noqa: F401
from .pipeline import crop_pass                                               # noqa: F401
from .frame import frame_pass, later_frame_pass                               # noqa: F401
from . import cv_host                                                         # noqa: F401
from . import pnp                                                             # noqa: F401
from .vgg import vgg19_forward                                                # noqa: F401
