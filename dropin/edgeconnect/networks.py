from future_urban_scene_generation_amd.edgeconnect.networks import *  # noqa: F401,F403
from future_urban_scene_generation_amd.edgeconnect.networks import EdgeGenerator, InpaintGenerator  # noqa: F401
