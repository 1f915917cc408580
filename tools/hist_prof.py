import numpy as np, sys
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] != 0]
t = a[:, :5].astype(np.int64)
d = np.diff(t, axis=1)
print("workgroups", len(a))
print("mean cycles per phase: init %.0f  main %.0f  patch %.0f  final flush %.0f   total %.0f" % (*d.mean(axis=0), (t[:, 4] - t[:, 0]).mean()))
print("span (first start .. last end) cycles:", int(t[:, 4].max() - t[:, 0].min()))
print("entries in LDS at end of main (mean, max):", a[:, 6].mean(), a[:, 6].max(), " before final flush:", a[:, 7].mean())
