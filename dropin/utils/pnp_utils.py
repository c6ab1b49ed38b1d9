# Opt-in shim for `from utils.pnp_utils import cpc_rodr_4_angles` (trajectory_inference.py:25).
# north_star: "3D keypoint projection / PnP (utils.geometry, utils.pnp_utils) stays host-side" - so by default this module IS
# the reference's own pnp_utils.py (loaded from the checkout further down the merged package's path, `FUSG_DROPIN` False).
# FUSG_DROPIN_PNP=1 selects the device pose fit (csrc/pnp.hip: the four Levenberg-Marquardt runs pinned to the reference's
# CPC_R within 1e-5; its epilogue's cv2.Rodrigues round trip is OpenCV's formula in numpy and UNPINNED, like planes_utils).
import os as _os

if _os.environ.get("FUSG_DROPIN_PNP") == "1":
    from future_urban_scene_generation_amd.utils.pnp_utils import *  # noqa: F401,F403
    from future_urban_scene_generation_amd.utils.pnp_utils import cpc_rodr_4_angles, cpc_rodr_4_angles_batch  # noqa: F401
    FUSG_DROPIN = True
else:
    from future_urban_scene_generation_amd._shim import become_reference_module as _become
    _become(__name__, __file__)
