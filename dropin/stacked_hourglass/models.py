from future_urban_scene_generation_amd.stacked_hourglass.models import *  # noqa: F401,F403
from future_urban_scene_generation_amd.stacked_hourglass.models import HourglassNet, Bottleneck, Hourglass  # noqa: F401
