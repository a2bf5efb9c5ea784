"""Fits the branch-free fp32 erf of the fp32x sampling kernel (amuse_amd/csrc/amuse_dev.hpp erf_bf): |z| < 0.875: z + z P(z^2),
P of degree 6; otherwise 1 - exp2(Q(min(|z|, 4))), Q of degree 9 fitted to log2(erfc) - Lawson-weighted least squares, then the
fp32 evaluation is checked against float64: max |erf error| 7.9e-8 (1.3 ulp at 1), GELU error 4.5e-7 = torch's own fp32 GELU.
Build-container script; prints the coefficients that are pasted into the header."""
import numpy as np
from scipy.special import erf, erfc
np.set_printoptions(precision=17)
def cheb_nodes(a,b,n): k=np.arange(n); return 0.5*(a+b)+0.5*(b-a)*np.cos((2*k+1)*np.pi/(2*n))
def fit_minimax(f, a, b, deg, w=None, iters=40):
    # iteratively reweighted least squares approximating minimax (Lawson)
    xs = np.linspace(a,b,4001); y=f(xs); wt=np.ones_like(xs)
    for _ in range(iters):
        V=np.vander(xs,deg+1,increasing=True)
        W=np.sqrt(wt)[:,None]
        c,*_=np.linalg.lstsq(V*W, y*np.sqrt(wt), rcond=None)
        e=np.abs(V@c-y)
        if w is not None: e=e*w(xs)
        wt=wt*(e/e.mean()+1e-12); wt/=wt.sum()
    return c
B=0.875
# region 1: erf(z) = z + z*P(t), t = z^2, P degree 5 -> fit g(t) = (erf(z)/z - 1) as function of t on [0, B^2]
def g1(t):
    z=np.sqrt(np.maximum(t,1e-300)); return np.where(t<1e-12, 2/np.sqrt(np.pi)-1 - 2/np.sqrt(np.pi)*t/3, erf(z)/z-1)
c1=fit_minimax(g1,0.0,B*B,6)
# region 2: log2(erfc(z)) on [B, 4.0] as poly in z of degree 8
def g2(z): return np.log2(erfc(z))
c2=fit_minimax(g2,B,4.0,9)
print("c1", [float(np.float32(v)) for v in c1]); print("c2",[float(np.float32(v)) for v in c2])
# fp32 evaluation check
def erf32(z):
    z=np.asarray(z,np.float32); az=np.abs(z); t=az*az
    p=np.float32(c1[-1])
    for c in c1[-2::-1]: p=np.float32(p*t+np.float32(c))
    r1=np.float32(az*p+az)
    zc=np.minimum(az,np.float32(4.0))
    q=np.float32(c2[-1])
    for c in c2[-2::-1]: q=np.float32(q*zc+np.float32(c))
    r2=np.float32(1.0)-np.exp2(q.astype(np.float32)).astype(np.float32)
    r=np.where(az<np.float32(B),r1,r2)
    return np.copysign(r,z)
zz=np.concatenate([np.linspace(-6,6,2000001), np.linspace(-1e-3,1e-3,20001)]).astype(np.float32)
err=np.abs(erf32(zz).astype(np.float64)-erf(zz.astype(np.float64)))
print("max abs err", err.max(), "at", zz[err.argmax()])
rel=err/np.maximum(np.abs(erf(zz.astype(np.float64))),1e-30); print("max rel err (|z|>1e-6)", rel[np.abs(zz)>1e-6].max())
# gelu error
x=np.linspace(-8,8,2000001).astype(np.float32)
z=(x*np.float32(0.70710678118654752)).astype(np.float32)
gel=(np.float32(0.5)*x*(np.float32(1.0)+erf32(z))).astype(np.float32)
ref=0.5*x.astype(np.float64)*(1+erf(x.astype(np.float64)/np.sqrt(2)))
print("gelu max abs err", np.abs(gel-ref).max())
from math import erf as merf
import torch
t=torch.from_numpy(x); gt=(0.5*t*(1+torch.erf(t/ (2**0.5)))).numpy()
print("torch fp32 gelu-erf max abs err vs f64", np.abs(gt-ref).max())
