import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np, orc
from knee import run, setp, G, dts
s,_ = run(setp(), dts[:2]); rate=(s[1]-s[0])/dts[1]; d2=dts.copy(); d2[1]=(G[1]-s[0])/rate
rows = np.array(list(d2) + [0.064129115905, 0.059378751644, 0.054980270616])
a = float(sys.argv[1])
for b in [float(x) for x in sys.argv[2:]]:
    def mod(p): p[8]*=a; p[16]*=b
    s,f = run(mod, rows)
    print("c1 x %g  s x %g  rows 9-14:" % (a,b), np.round((s-G[:14])[8:],4), flush=True)
