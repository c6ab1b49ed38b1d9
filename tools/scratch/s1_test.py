import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
os.environ['FUSG_H3_SCALE'] = '1'
from future_urban_scene_generation_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), 'libfusg_s1.so')
import pytest
sys.exit(pytest.main(['tests/test_gpu_nets.py', '-m', 'gpu', '-x', '-q', '-k', 'f16x3']))
