import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
def maps(tag):
    libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'hip' in l or 'hsa' in l))
    print(tag, libs)
import torch
print(torch.__version__, torch.cuda.is_available(), torch.version.hip)
maps('after torch')
from gpitch_amd import _lib
lib = _lib.load_library()
maps('after lib')
hip = C.CDLL('libamdhip64.so.7') if False else None
h = C.c_void_p()
print('gp_create', lib.gp_create(0, None, C.byref(h)))
x = torch.zeros(4, device='cuda')
print('gp_create after torch init', lib.gp_create(0, None, C.byref(h)))
