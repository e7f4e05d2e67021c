import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    print(f"{k:44s} rel {np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30):.2e}")
