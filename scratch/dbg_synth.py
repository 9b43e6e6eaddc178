import sys, ctypes, numpy as np
sys.path.insert(0,'.')
from eigenkernel_amd import solver
from oracle import ek_oracle as ok
lib=solver.load_library(); lib.ek_hip_init(0)
n=384; nn=n*n*8
p=ctypes.c_void_p(); lib.ek_hip_malloc(ctypes.byref(p), nn)
lib.ek_hip_synth_matrix_device(n,1,p,n)
h=np.zeros((n,n),order='F'); lib.ek_hip_memcpy_d2h(h.ctypes.data,p,nn)
o=ok.synth_matrix(n,1)
d=np.argwhere(h!=o)
print(len(d), d[:5])
for i,j in d[:5]:
    print(i,j,repr(h[i,j]),repr(o[i,j]), h[i,j]-o[i,j])
