"""MI355X-native per-vehicle novel-view synthesis path (stacked-hourglass keypoints ->
Warp&Learn ICN -> VUnet appearance transfer -> EdgeConnect inpainting).

Drop-in module packages (same class names, constructor arguments, state_dict schema and forward
signatures as the reference's) live in the sub-packages ``stacked_hourglass``, ``warp_learn``,
``vunet`` and ``edgeconnect``; ``install()`` aliases them under the reference's top-level import
names so that ``run_test.py`` / ``trajectory_inference.py`` pick them up unchanged.
"""
import importlib
import sys

__version__ = "0.1.0"

_ALIASES = {
    "stacked_hourglass": "future_urban_scene_generation_amd.stacked_hourglass",
    "stacked_hourglass.models": "future_urban_scene_generation_amd.stacked_hourglass.models",
    "warp_learn.models": "future_urban_scene_generation_amd.warp_learn.models",
    "vunet": "future_urban_scene_generation_amd.vunet",
    "vunet.models": "future_urban_scene_generation_amd.vunet.models",
    "vunet.layers": "future_urban_scene_generation_amd.vunet.layers",
    "edgeconnect.networks": "future_urban_scene_generation_amd.edgeconnect.networks",
    "edgeconnect.models": "future_urban_scene_generation_amd.edgeconnect.models",
}


# Opt-in only - ONE policy for every module that restates third-party (OpenCV) arithmetic whose parity with the build the
# reference runs on is unpinned (oracle/cv_host.py, oracle/pnp.py):
#   planes_utils: the uint8 image steps (to_image, warp_unwarp_planes, get_planes, planes_to_torch);
#   pnp:          the pose fit (cpc_rodr_4_angles).  north_star keeps PnP host-side; the device fit itself is pinned to the
#                 reference's CPC_R, its cv2.Rodrigues epilogue is not.
_OPT_IN_ALIASES = {"planes_utils": {"warp_learn.planes_utils": "future_urban_scene_generation_amd.warp_learn.planes_utils"},
                   "pnp": {"utils.pnp_utils": "future_urban_scene_generation_amd.utils.pnp_utils"}}


def install(names=None, planes_utils=None, pnp=None) -> None:
    """Register the drop-in modules in ``sys.modules`` under the reference's import names.

    By default only the four network packages are replaced; ``warp_learn``, ``edgeconnect`` and ``utils`` keep resolving
    their other sub-modules (config, online_visibility, pnp_utils, ...) from the reference checkout on ``sys.path``.
    ``planes_utils=True`` (or FUSG_DROPIN_PLANES_UTILS=1) additionally routes ``from warp_learn.planes_utils import
    to_image, warp_unwarp_planes`` (trajectory_inference.py:28-29) to the device versions; ``pnp=True`` (or
    FUSG_DROPIN_PNP=1) routes ``from utils.pnp_utils import cpc_rodr_4_angles`` (:25) to the device pose fit - both
    explicit, because their parity with OpenCV is unpinned."""
    import os
    if planes_utils is None:
        planes_utils = os.environ.get("FUSG_DROPIN_PLANES_UTILS") == "1"
    if pnp is None:
        pnp = os.environ.get("FUSG_DROPIN_PNP") == "1"
    aliases = dict(_ALIASES)
    if planes_utils:
        aliases.update(_OPT_IN_ALIASES["planes_utils"])
    if pnp:
        aliases.update(_OPT_IN_ALIASES["pnp"])
    for alias, target in aliases.items():
        if names is not None and alias.split(".")[0] not in names:
            continue
        mod = importlib.import_module(target)
        sys.modules[alias] = mod
        if "." in alias:
            parent, leaf = alias.rsplit(".", 1)
            if parent in sys.modules:
                setattr(sys.modules[parent], leaf, mod)
