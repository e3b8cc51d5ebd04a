#!/usr/bin/env python
"""Known byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: "calibrate on a known byte
count in your own access pattern") and the box's copy / read / gather rates with this library's own kernels.
usage (GPU box): python tools/calib_fetch.py            -- prints one JSON line: mode -> GB/s, algorithmic bytes per launch
                 tools/calib_fetch.sh <tag>            -- the same under two --pmc passes; factors in gpurun_out/calib_<tag>.txt"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ranklib_amd import _native as N  # noqa: E402

GIB = 1 << 30
out = {}
# buffers far beyond the 256 MiB Infinity Cache; strides: a node holding 1/2, 1/4, 1/16 of the rows of a 4 GiB row matrix
for name, mode, nbytes, stride in (("copy_16B_per_lane", 0, 2 * GIB, 1), ("read_16B_per_lane", 1, 4 * GIB, 1), ("write_16B_per_lane", 2, 4 * GIB, 1),
                                   ("gather32_every_row", 3, 4 * GIB, 1), ("gather32_1_in_2", 3, 4 * GIB, 2), ("gather32_1_in_4", 3, 4 * GIB, 4),
                                   ("gather32_1_in_16", 3, 4 * GIB, 16)):
    ms, b = N.membench(mode, nbytes, stride, iters=5)
    out[name] = {"avg_ms": round(ms, 4), "alg_bytes": b, "GBps": round(b / ms / 1e6, 1), "launches": 6}
print(json.dumps(out))
