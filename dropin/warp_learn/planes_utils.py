# Opt-in shim for `from warp_learn.planes_utils import to_image, warp_unwarp_planes` (trajectory_inference.py:28-29).
# FUSG_DROPIN_PLANES_UTILS=1 selects the MI355X versions (csrc/cvops.hip); their parity with the OpenCV build the
# reference runs on is UNPINNED (oracle/cv_host.py), so by default this module IS the reference's own planes_utils.py,
# executed from the checkout further down the merged package's path.
import os as _os

if _os.environ.get("FUSG_DROPIN_PLANES_UTILS") == "1":
    from future_urban_scene_generation_amd.warp_learn.planes_utils import *  # noqa: F401,F403
    from future_urban_scene_generation_amd.warp_learn.planes_utils import (get_planes, pascal_texture_planes,  # noqa: F401
                                                                            planes_to_torch, to_image, warp_unwarp_planes)
    FUSG_DROPIN = True
else:
    import warp_learn as _pkg
    _here = _os.path.dirname(_os.path.abspath(__file__))
    for _d in _pkg.__path__:
        _f = _os.path.join(_d, "planes_utils.py")
        if _os.path.abspath(_d) != _here and _os.path.exists(_f):
            with open(_f) as _fh:
                exec(compile(_fh.read(), _f, "exec"), globals())
            break
    else:
        raise ImportError("warp_learn.planes_utils: no reference checkout behind dropin/ on sys.path "
                          "(set FUSG_DROPIN_PLANES_UTILS=1 to use the MI355X versions)")
    FUSG_DROPIN = False
