import sys, os
sys.path.insert(0, os.getcwd())
from future_urban_scene_generation_amd import _lib as L
v = sys.argv[1]
if v != '0':
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f'libfusg_abl{v}.so')
print('variant', v, L.LIB_PATH)
exec(open('tools/scratch/abl.py').read())
