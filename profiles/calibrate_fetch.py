#!/usr/bin/env python3
"""Calibration workload for rocprofv3 --pmc FETCH_SIZE: stream a [rows, 2] float32 matrix
once with one float2 per lane (the hot path's read shape).  512 MiB > the 256 MiB Infinity
Cache, so every byte comes from HBM.  Compare FETCH_SIZE * 1024 with the byte count printed."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from normalizing_flows_amd import _lib, flow  # noqa: E402

rows = 1 << 26  # 512 MiB
lib = _lib.load()
x = flow.DeviceArray((rows, 2))
out = flow.DeviceArray((4,))
for _ in range(3):
    _lib.check(lib.nf_debug_read_f2(x.ptr, rows, out.ptr))
print(f"debug_read_f2_kernel read {rows * 8} bytes per launch")
