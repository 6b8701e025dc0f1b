import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np, orc
from knee import run, setp, G, dts
s,_ = run(setp(), dts[:2]); rate=(s[1]-s[0])/dts[1]; d2=dts.copy(); d2[1]=(G[1]-s[0])/rate
H = {
 "ce100": setp(i8=lambda v: v*3.7771), "s2.607": setp(i16=lambda v: v*2.607), "gamwo": setp(i12=lambda v: v/25600.0),
 "taua": setp(i9=lambda v: v+4.53), "p3.54": setp(i10=lambda v: v*3.5367), "base": setp(),
}
name = sys.argv[1]; nrows = int(sys.argv[2]); mod = H[name]
cur = list(d2)            # dts of rows 1..11
dt_class = dts[10]        # the planned dt of row 11 (un-calibrated chain)
kprev = None
for row in range(12, nrows+1):
    cands = {}
    for k in range(1, 26):
        for j in range(3):
            d = max(dt_class*25*0.333333/k*(0.333333**j), 0.05)
            cands.setdefault(round(d, 12), (k, j))
    # the stress increment is monotone in dt: bracket by bisection over the sorted candidates
    ds = sorted(cands)
    lo, hi = 0, len(ds)-1
    cache = {}
    def val(i):
        if i not in cache:
            s,f = run(mod, np.array(cur+[ds[i]])); cache[i] = s[-1]
        return cache[i]
    target = G[row-1]
    while hi - lo > 1:
        mid = (lo+hi)//2
        if val(mid) > target: lo = mid      # stress is negative and decreasing with dt
        else: hi = mid
    best = min((lo, hi), key=lambda i: abs(val(i)-target))
    k, j = cands[ds[best]]
    print(name, "row", row, "dt", ds[best], "k", k, "cuts", j, "residual %.4f" % (val(best)-target), "neighbour residuals", ["%.3f"%(val(i)-target) for i in (lo,hi)], flush=True)
    cur.append(ds[best]); dt_class = ds[best]
