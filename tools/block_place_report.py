#!/usr/bin/env python
"""Load balance of the child-histogram passes from a tools/block_place.py dump: blocks per CU ("place" = XCD, SE, CU from HW_ID), when the
places finish, and a block's duration against the number of blocks that shared its CU.   usage: python tools/block_place_report.py raw.npy [steps..]"""
import sys

import numpy as np

tr = np.load(sys.argv[1])
steps = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 3, 5]
for step in steps:
    b = tr[step, 1]
    m = b[:, 7] > 0
    if not m.any():
        continue
    t0 = b[m, 0] * 0.01
    t1 = b[m, 7] * 0.01
    base = t0.min()
    t0 -= base
    t1 -= base
    hw = b[m, 6]
    xcc = (hw >> 32) & 0xf
    hwid = hw & 0xffffffff
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 0x7
    place = xcc * 1000 + se * 100 + sh * 16 + cu
    print("step %d: %d blocks, span %.1f us, block duration median %.1f max %.1f" % (step, m.sum(), t1.max(), np.median(t1 - t0), (t1 - t0).max()))
    print("  blocks per XCD", np.bincount(xcc.astype(int), minlength=8))
    ups = np.unique(place)
    cnts = np.array([(place == p).sum() for p in ups])
    ends = np.array([t1[place == p].max() for p in ups])
    print("  %d places (CUs); blocks per place: min %d median %d max %d; a place's last block ends at: min %.1f median %.1f max %.1f us"
          % (len(ups), cnts.min(), np.median(cnts), cnts.max(), ends.min(), np.median(ends), ends.max()))
    print("  block starts (percentiles 0 10 50 90 100):", np.percentile(t0, [0, 10, 50, 90, 100]).round(1), " ends:", np.percentile(t1, [0, 10, 50, 90, 100]).round(1))
    d = t1 - t0
    conc = np.array([((place == place[i]) & (t0 < t1[i]) & (t1 > t0[i])).sum() for i in range(len(d))])
    for cv in np.unique(conc):
        print("    blocks that overlapped %d block(s) on their CU (themselves included): %4d, median duration %.1f us" % (cv, (conc == cv).sum(), np.median(d[conc == cv])))
