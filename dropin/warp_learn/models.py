from future_urban_scene_generation_amd.warp_learn.models import *  # noqa: F401,F403
from future_urban_scene_generation_amd.warp_learn.models import G_Resnet, get_icn_inputs  # noqa: F401
