import numpy as np, sys
b=np.load("/tmp/b.npz"); lb=b["log_q"]
for name in sys.argv[1:]:
    a=np.load(name); la=a["log_q"]
    both=np.isfinite(la)&np.isfinite(lb)
    d=np.abs(la-lb)[both]
    dx=np.abs(a["x"]-b["x"]); rows=np.nonzero(dx.max(axis=1)>1e-3)[0]
    print(name, "finite", np.isfinite(la).sum(), "vs", np.isfinite(lb).sum(), "max dlogq", d.max(), "rows x diff", len(rows))
