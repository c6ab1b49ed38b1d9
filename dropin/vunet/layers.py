from future_urban_scene_generation_amd.vunet.layers import *  # noqa: F401,F403
