# Opt-in shim for `from warp_learn.planes_utils import to_image, warp_unwarp_planes` (trajectory_inference.py:28-29).
# FUSG_DROPIN_PLANES_UTILS=1 selects the MI355X versions (csrc/cvops.hip); their parity with the OpenCV build the
# reference runs on is UNPINNED (oracle/cv_host.py), so by default this module IS the reference's own planes_utils.py,
# loaded as a regular module from the checkout further down the merged package's path (`FUSG_DROPIN` False).
import os as _os

if _os.environ.get("FUSG_DROPIN_PLANES_UTILS") == "1":
    from future_urban_scene_generation_amd.warp_learn.planes_utils import *  # noqa: F401,F403
    from future_urban_scene_generation_amd.warp_learn.planes_utils import (get_planes, pascal_texture_planes,  # noqa: F401
                                                                            planes_to_torch, to_image, warp_unwarp_planes)
    FUSG_DROPIN = True
else:
    from future_urban_scene_generation_amd._shim import become_reference_module as _become
    _become(__name__, __file__)
