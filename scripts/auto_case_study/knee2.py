import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np, orc
from knee import run, setp, G, dts
s,_ = run(setp(), dts[:2]); rate=(s[1]-s[0])/dts[1]; d2=dts.copy(); d2[1]=(G[1]-s[0])/rate
def resid(mod):
    s,f = run(mod, d2); return (s-G[:11])[8:11]
fams = {
 "s*a": lambda a: setp(i16=lambda v: v*a),
 "go*a": lambda a: setp(i15=lambda v: v*a),
 "c1*a": lambda a: setp(i8=lambda v: v*a),
 "gam_wo/a": lambda a: setp(i12=lambda v: v/a),
 "tau_a+a": lambda a: setp(i9=lambda v: v+a),
 "q*a": lambda a: setp(i11=lambda v: v*a),
 "p*a": lambda a: setp(i10=lambda v: v*a),
}
start = {"s*a": (2.5, 3.2), "go*a": (20., 30.), "c1*a": (4., 7.), "gam_wo/a": (1e4, 1e5), "tau_a+a": (0.5, 1.5), "q*a": (0.5, 0.7), "p*a": (1.5, 2.5)}
for name in sys.argv[1:] or fams:
    f = fams[name]; a0, a1 = start[name]
    logp = name in ("gam_wo/a",)
    x0, x1 = (np.log(a0), np.log(a1)) if logp else (a0, a1)
    ev = lambda x: resid(f(np.exp(x) if logp else x))
    r0 = ev(x0); r1 = ev(x1)
    for it in range(4):
        if abs(r1[2]) < 0.002 or r1[2] == r0[2]: break
        x2 = x1 - r1[2]*(x1-x0)/(r1[2]-r0[2]); x0, r0 = x1, r1; x1 = x2; r1 = ev(x1)
    print(name, "a =", np.exp(x1) if logp else x1, "residual rows 9,10,11:", np.round(r1,4), flush=True)
