import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
a = a.reshape(a.shape[0], -1, 4) if a.ndim == 2 else a
b = b.reshape(a.shape)
d = (a != b).any(-1)
print("shape", a.shape, "differing pixels", int(d.sum()), "of", d.size)
if d.any():
    ys, xs = np.nonzero(d)
    print("rows with differences:", len(set(ys.tolist())), "first rows", sorted(set(ys.tolist()))[:20])
    cb = np.bincount(xs // 256, minlength=(a.shape[1] + 255) // 256)
    print("per 256-column block:", cb.tolist())
    print("zero (unshaded) among differing in b:", int((b[d] == 0).all(-1).sum()), " in a:", int((a[d] == 0).all(-1).sum()))
    rb = np.bincount(ys, minlength=a.shape[0])
    print("per-row counts (first 40 rows with any):", [(int(y), int(rb[y])) for y in np.nonzero(rb)[0][:40]])
