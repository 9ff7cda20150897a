import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, oracle_np as O
from spectral_cube_amd import ops
from spectral_cube_amd.device import DeviceArray
shape=(700,4,64)
k=np.hanning(81)+0.01
rng=np.random.default_rng(21)
d=rng.standard_normal(shape).astype(np.float32)
d[3:5,1,1]=np.nan; d[:,2,3]=np.nan
out=ops.spectral_conv(DeviceArray.from_numpy(d),k).get()
exp=O.spectral_smooth(d,None,k)
diff=np.argwhere(np.isnan(out)!=np.isnan(exp))
print(len(diff), diff[:5], diff[-5:])
print(out[[0,1,39,40,41,659,660,661,699],2,3])
print(exp[[0,1,39,40,41,659,660,661,699],2,3])
from spectral_cube_amd import _lib
import ctypes as C
o2=DeviceArray(shape,np.float32)
_lib.call("spc_memset",0,C.c_void_p(o2.ptr),255,o2.nbytes,None)
out=ops.spectral_conv(DeviceArray.from_numpy(d),k,out=o2).get()
diff=np.argwhere(np.isnan(out)!=np.isnan(exp))
print("poisoned:",len(diff), diff[:3], diff[-3:])
ok=np.isfinite(exp)
print("maxabs", np.abs(out[ok&np.isfinite(out)]-exp[ok&np.isfinite(out)]).max())
