from future_urban_scene_generation_amd.vunet.models import *  # noqa: F401,F403
from future_urban_scene_generation_amd.vunet.models import Vunet_fix_res  # noqa: F401
