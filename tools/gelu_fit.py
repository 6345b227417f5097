"""Fit of the erf-GELU polynomial used by gelu_fast() (lam_slide_amd/csrc/common.hip.h): q(a) ~ log2 Phi(-a) on [0, 9], degree 5,
iteratively re-weighted least squares towards the minimax error of a * Phi(-a); prints the fp32 coefficients and the error of an fp32 evaluation."""
import numpy as np
from scipy.special import ndtr, log_ndtr, erf
a = np.linspace(0.0, 9.0, 200001)
q = log_ndtr(-a) / np.log(2.0)
h = ndtr(-a)
deg=5
A = np.stack([a**k for k in range(deg + 1)], axis=1)
w = np.ones_like(a); best=None
for it in range(600):
    c, *_ = np.linalg.lstsq(A * w[:, None], q * w, rcond=None)
    hh = np.exp2(A @ c)
    err = np.abs(a * (hh - h))
    m = err.max()
    if best is None or m < best[0]: best = (m, c.copy())
    w = w * (1.0 + 1.5 * err / (m + 1e-30)); w /= w.max(); w = np.maximum(w, 1e-8)
m, c = best
print("float64 max err", m)
c32 = c.astype(np.float32)
print("coeffs f32:", [float(v) for v in c32])
print("hex:", [hex(np.float32(v).view(np.uint32)) for v in c32])
# float32 emulation
x = np.linspace(-12, 12, 2400001).astype(np.float32)
ax = np.abs(x)
p = np.float32(c32[5])
for k in (4,3,2,1,0):
    p = (p * ax + c32[k]).astype(np.float32)      # (fma emulated as mul+add in f32: slightly worse than real fma)
hh = np.exp2(p.astype(np.float64)).astype(np.float32)
relu = np.maximum(x, np.float32(0))
g = (relu - ax * hh).astype(np.float32)
ref = x.astype(np.float64) * 0.5 * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
e = np.abs(g.astype(np.float64) - ref)
print("f32 emulation: max abs err", e.max(), "at x", x[e.argmax()], " max rel err (|x|>0.05 pos side)", (e/np.maximum(np.abs(ref),1e-30))[(x>0.05)].max())
# old formula for comparison
def old(x):
    ax=np.abs(x); t=1/(1+0.3275911*0.70710678*ax)
    pp=0.5*1.061405429*t+0.5*-1.453152027; pp=pp*t+0.5*1.421413741; pp=pp*t+0.5*-0.284496736; pp=pp*t+0.5*0.254829592
    ee=np.exp2(-0.5*1.4426950408889634*x*x)
    return np.maximum(x,0)-ax*pp*t*ee
xo=x.astype(np.float64)
print("old formula max abs err (f64 arithmetic)", np.abs(old(xo)-ref).max())
for big in (15.0, 30.0, 100.0, 1e4, 3e38):
    ax=np.float32(big); p=np.float32(c32[5])
    for k in (4,3,2,1,0): p=np.float32(p*ax+c32[k])
    print("a=",big,"q=",p)
