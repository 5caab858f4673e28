import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
from normalizing_flows_amd import _lib, flow, densities
lib=_lib.load(); _lib.require_device()
for lg in (20, 22, 24, 26):
    rows=1<<lg; out=flow.DeviceArray((rows,2))
    for dens in ("banana","gaussian_blobs","energy3"):
        call=lambda i: _lib.check(lib.nf_density_sample(densities.DENSITIES[dens], C.c_uint64(7+i), rows, out.ptr, _lib.MEM_DEVICE))
        for i in range(5): call(i)
        _lib.check(lib.nf_device_synchronize())
        t0=time.perf_counter()
        for i in range(20): call(i)
        _lib.check(lib.nf_device_synchronize())
        dt=(time.perf_counter()-t0)/20
        print(f"2^{lg} {dens}: {dt*1e6:.1f} us  {rows/dt:.3e} rows/s  {rows*8/dt/1e9:.0f} GB/s")
    out.free()
