# merged package: modules missing here resolve to the reference checkout further down sys.path
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
