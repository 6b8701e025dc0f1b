import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np, orc
from knee import run, setp, G, dts
s,_ = run(setp(), dts[:2]); rate=(s[1]-s[0])/dts[1]; d2=dts.copy(); d2[1]=(G[1]-s[0])/rate
def resid(mod):
    s,f = run(mod, d2); return (s-G[:11])[8:11]
H = {
 "swap p,q": setp(i10=1.4, i11=0.8),
 "p->1/p": setp(i10=1.25), "q->1/q": setp(i11=1/1.4), "both inv": setp(i10=1.25, i11=1/1.4),
 "mu=c44": setp(i6=117720.0), "mu=(c11-c12)/2": setp(i6=43355.0), "c1x4": setp(i8=0.4),
 "mu=c44,both": setp(i6=117720.0, i10=1.25), "c1x2pi": setp(i8=0.1*2*np.pi/ (2*np.pi) * 3.82),
}
for name in sys.argv[1:]:
    print(name, np.round(resid(H[name]),4), flush=True)
