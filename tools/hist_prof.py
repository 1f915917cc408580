#!/usr/bin/env python3
"""Reader of MP_HIST_PROF=<file> (unique.hip): per-workgroup phase stamps of the histogram kernel of one mp_window_unique call.
hist_kernel: clock stamps after init / row loop / patch rows / final merge; hist_mask_kernel: cycles of compaction, hashing, patch rows,
final merge and the number of marked rows."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] != 0]
print("workgroups", len(a))
if len(sys.argv) > 2 and sys.argv[2] == "mask":
    d = a[:, 1:5].astype(np.int64)
    print("mean cycles per phase: compaction %.0f  hashing %.0f  patch rows %.0f  final merge %.0f   sum %.0f" % (*d.mean(axis=0), d.sum(axis=1).mean()))
    print("marked rows per workgroup: mean %.0f of %.0f rows (%.1f %%)" % (a[:, 5].mean(), a[:, 6].mean(), 100.0 * a[:, 5].sum() / a[:, 6].sum()))
else:
    t = a[:, :5].astype(np.int64)
    d = np.diff(t, axis=1)
    print("mean cycles per phase: init %.0f  row loop %.0f  patch rows %.0f  final merge %.0f   total %.0f" % (*d.mean(axis=0), (t[:, 4] - t[:, 0]).mean()))
    print("entries in the LDS table at the end of the row loop (mean, max):", a[:, 6].mean(), a[:, 6].max())
