# -*- coding: utf-8 -*-
"""NumPy float64 restatement of csrc/common.h: digamma_pos_for_f32 (round 6: branch-free six-step recurrence as D'(x) / D(x), one
Newton-refined reciprocal each, the short atanh-series logarithm) and of the loop it replaced, against scipy.special.digamma on
float32 arguments: absolute error and how many results differ after rounding to float32.   python tools/digamma_check.py   (CPU)"""
import numpy as np, scipy.special as sp
rng=np.random.default_rng(1)
def rcp_nr(x):
    r=(1.0/x).astype(np.float32).astype(np.float64)  # crude seed like a low-precision rcp
    for _ in range(3):
        e=1.0 - x*r; r = r + r*e
    return r
LN2=0.6931471805599453
def log_c(x):
    m,e=np.frexp(x)  # m in [0.5,1)
    lo = m < 0.7071067811865476
    m=np.where(lo, 2*m, m); e=np.where(lo, e-1, e)
    t=(m-1.0)*rcp_nr(m+1.0)
    z=t*t
    p=1/15.
    for c in (1/13.,1/11.,1/9.,1/7.,1/5.,1/3.,1.0):
        p=p*z+c
    return e*LN2 + 2*t*p
def dg(x):
    small = x<6.0
    D = x*(x+1)*(x+2)*(x+3)*(x+4)*(x+5)
    Dp = ((((6*x+75)*x+340)*x+675)*x+548)*x+120
    q = Dp*rcp_nr(D)
    xs = np.where(small, x+6.0, x)
    ix = rcp_nr(xs); z=ix*ix
    y = z*(1/12. - z*(1/120. - z*(1/252. - z*(1/240. - z*(1/132. - z*(691/32760. - z*(1/12.)))))))
    return np.where(small,-q,0.0) + log_c(xs) - 0.5*ix - y
for name,x in [('loguniform 1e-15..1e8', np.exp(rng.uniform(np.log(1e-15),np.log(1e8),2000000))), ('0.01..20', rng.uniform(0.01,20,2000000)), ('f32 grid near root', np.float32(1.4616321)+np.arange(-2000,2000,dtype=np.float32)*np.float32(1.2e-7))]:
    x=x.astype(np.float32).astype(np.float64)
    a=dg(x); b=sp.digamma(x)
    err=np.abs(a-b); rel=err/np.maximum(np.abs(b),1e-300)
    neq=(a.astype(np.float32)!=b.astype(np.float32))
    print(name,'max abs',err.max(),'max rel',rel.max(),'f32 mismatches',neq.sum(),'of',x.size, 'max ulp diff', np.abs(a.astype(np.float32).view(np.int32).astype(np.int64)-b.astype(np.float32).view(np.int32).astype(np.int64)).max())
print('--- current implementation (recurrence two terms per division to x >= 6, library log)')
def dg_old(x):
    x=x.copy(); r=np.zeros_like(x)
    for _ in range(3):
        s = x<6.0
        r=np.where(s, r-(2*x+1)/(x*(x+1)), r); x=np.where(s, x+2, x)
    z=1/(x*x)
    y = z*(1/12. - z*(1/120. - z*(1/252. - z*(1/240. - z*(1/132. - z*(691/32760. - z*(1/12.)))))))
    return r+np.log(x)-0.5/x-y
for name,x in [('0.01..20', rng.uniform(0.01,20,2000000)), ('f32 grid near root', np.float32(1.4616321)+np.arange(-2000,2000,dtype=np.float32)*np.float32(1.2e-7))]:
    x=x.astype(np.float32).astype(np.float64)
    a=dg_old(x); b=sp.digamma(x); c=dg(x)
    for nm,a in (('old',a),('new',c)):
        neq=(a.astype(np.float32)!=b.astype(np.float32))
        print(name,nm,'max abs',np.abs(a-b).max(),'f32 mismatches',neq.sum(), 'max ulp', np.abs(a.astype(np.float32).view(np.int32).astype(np.int64)-b.astype(np.float32).view(np.int32).astype(np.int64)).max())
