"""Developer tool: the coefficients of the branch-free fp32 erf of csrc/common.h (erf_f32) - least-squares fits at Chebyshev nodes in
float64, rounded to float32 - and their error against scipy over [-6, 6] with the evaluation emulated in float32; also the fp32 GELU
built on it against the exact function and against torch's own fp32 GELU.
    python tools/fit_erf.py"""
import numpy as np
from scipy.special import erf, erfc
from numpy.polynomial import chebyshev as C, polynomial as P
f32 = np.float32
SPLIT = 0.95
# small range: erf(x) = x * Ps(s), s = x^2 in [0, SPLIT^2]
s = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) * 0.5 + 0.5
s = s * SPLIT ** 2
x = np.sqrt(np.maximum(s, 1e-300))
ys = np.where(s > 0, erf(x) / np.where(x > 0, x, 1), 2 / np.sqrt(np.pi))
best = None
for deg in (5, 6):
    cs = P.polyfit(s, ys, deg)
    err = np.abs(P.polyval(s, cs) - ys).max()
    print("small deg", deg, "double fit err", err)
    if deg == 6: small = cs
# large range: q(t) = -ln(erfc(t)) = t * G(t), t in [SPLIT, TMAX]
TMAX = 4.2
t = np.cos(np.pi * (np.arange(6000) + 0.5) / 6000) * 0.5 + 0.5
t = SPLIT * 0.98 + t * (TMAX - SPLIT * 0.98)
g = -np.log(erfc(t)) / t
for deg in (6, 7, 8, 9):
    cl = P.polyfit(t, g, deg)
    err = np.abs(P.polyval(t, cl) - g).max()
    print("large deg", deg, "double fit err in g", err)
    if deg == 8: large = cl
def horner32(c, v):
    r = np.full_like(v, f32(c[-1]))
    for k in c[-2::-1]:
        r = (r * v + f32(k)).astype(f32)      # (fma in hardware: this is slightly pessimistic)
    return r
def erf32(a):
    a = a.astype(f32)
    tt = np.minimum(np.abs(a), f32(TMAX)).astype(f32)
    ss = (a * a).astype(f32)
    sm = (horner32(small, ss) * a).astype(f32)
    q = (horner32(large, tt) * tt).astype(f32)
    e = np.exp2((q * f32(-1.4426950408889634)).astype(f32)).astype(f32)
    lg = np.copysign((f32(1) - e).astype(f32), a)
    return np.where(tt < f32(SPLIT), sm, lg)
a = np.linspace(-6, 6, 2000001)
ref = erf(a)
got = erf32(a).astype(np.float64)
ulp = np.abs(got - ref) / np.spacing(np.abs(ref).astype(f32)).astype(np.float64)
print("max abs err", np.abs(got - ref).max(), "max ulp", ulp[np.abs(a) > 1e-3].max())
z = a / np.sqrt(2)
gel_ref = 0.5 * a * (1 + erf(z))
gel = (f32(0.5) * a.astype(f32) * (f32(1) + erf32((a.astype(f32) * f32(0.70710678118654752)).astype(f32)))).astype(np.float64)
print("gelu max abs err", np.abs(gel - gel_ref).max())
import torch
tg = torch.nn.functional.gelu(torch.tensor(a, dtype=torch.float32)).double().numpy()
print("torch fp32 gelu max abs err vs exact", np.abs(tg - gel_ref).max(), " ours vs torch", np.abs(gel - tg).max())
np.set_printoptions(precision=10)
print("small", [float(f32(c)) for c in small])
print("large", [float(f32(c)) for c in large])
