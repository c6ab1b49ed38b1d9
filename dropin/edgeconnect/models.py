from future_urban_scene_generation_amd.edgeconnect.models import *  # noqa: F401,F403
from future_urban_scene_generation_amd.edgeconnect.models import EdgeModel, InpaintingModel  # noqa: F401
