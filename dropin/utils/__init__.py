# merged package: modules missing here (crop_utils, geometry, keypoint_utils, ...) resolve to the reference checkout
# further down sys.path; only pnp_utils is replaced (the reference's own utils/__init__.py is empty)
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
