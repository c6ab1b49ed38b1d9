from future_urban_scene_generation_amd.utils.pnp_utils import *  # noqa: F401,F403
from future_urban_scene_generation_amd.utils.pnp_utils import cpc_rodr_4_angles, cpc_rodr_4_angles_batch  # noqa: F401
