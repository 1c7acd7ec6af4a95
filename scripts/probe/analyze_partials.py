"""Replays the fmaf chain of every mismatching per-lane partial a CHECK build of scripts/probe/rowdot_variants.hip logged (w, x operands\nincluded) and says which chain reproduces the wrong value bit for bit: python scripts/probe/analyze_partials.py <probe logs>"""
import re, sys, itertools, numpy as np
from fractions import Fraction
def f32(x): return np.float32(x)
def fma(a,b,c):
    r = Fraction(float(a))*Fraction(float(b))+Fraction(float(c))
    return np.float32(float(r)) if True else None   # float(Fraction) rounds to double first: double rounding possible but rare
rows=[]
for fn in sys.argv[1:]:
    for line in open(fn):
        if 'PARTIAL' not in line: continue
        m=re.search(r'launch (\d+) wave (\d+) value (\d+) .*lane (\d+): want (\S+) got (\S+) \| w: (.*) \| x: (.*)$', line)
        L,W,V,lane,want,got,w,x=m.groups()
        rows.append((int(L),int(W),int(V),int(lane),f32(want),f32(got),[f32(t) for t in w.split()],[f32(t) for t in x.split()]))
for (L,W,V,lane,want,got,w,x) in rows:
    acc=f32(0)
    for k in range(8): acc=fma(w[k],x[k],acc)
    ok = (acc==want)
    hit=None
    for mask in range(1,256):
        a=f32(0)
        for k in range(8):
            if not (mask>>k)&1: a=fma(w[k],x[k],a)
        if a==got: hit=[k for k in range(8) if (mask>>k)&1]; break
    # doubled products?
    hit2=None
    if hit is None:
        for k2 in range(8):
            a=f32(0)
            for k in range(8):
                a=fma(w[k],x[k],a)
                if k==k2: a=fma(w[k],x[k],a)
            if a==got: hit2=k2
    print(L,W,V,lane,"replay==want",ok,"got = chain skipping",hit,"doubled",hit2)
