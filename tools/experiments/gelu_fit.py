"""Exact minimax fits (linear programming) of the clamped odd polynomial behind csrc/mst_common.h::gelu_poly2: Phi(x) - 1/2 = xc R(xc^2), xc = clamp(x, -A, A),\nPhi(+-A) pinned to 1 / 0; prints the float32 coefficient tables (AVX_GELU_COEFFS) and the error of the float32 evaluation.  python tools/experiments/gelu_fit.py"""
import numpy as np
from scipy.special import erf
from scipy.optimize import linprog
def Phi(x): return 0.5*(1+erf(x/np.sqrt(2)))
# Minimax via LP: minimise E s.t. |w_i (V c - f)_i| <= E, with exact pin A*R(A^2) = 0.5.  Chebyshev-like scaled basis for conditioning.
def fit(n, A, relw=1.0, N=3000):
    x = (np.cos(np.pi*(np.arange(N)+0.5)/N)*0.5+0.5)*A       # (0, A)
    x = np.concatenate([x, [A]])
    t = x*x; s = t/(A*A)                                       # s in [0,1]
    V = np.stack([s**k for k in range(n+1)],1)                 # R(t) = sum a_k s^k
    f = (Phi(x)-0.5)/x
    g = x*Phi(x)
    # gelu error for +x: x*x*dR ; relative to gelu(x): x*x*dR/g ; for -x: abs error same magnitude x*x*dR, but gelu(-x) is tiny: use abs there
    w = np.maximum(x*x, relw*x*x/np.maximum(g,1e-3))           # weight: max(abs, rel*relw)
    # LP variables: a_0..a_n, E
    nv = n+2
    c = np.zeros(nv); c[-1]=1
    Aub = np.vstack([np.hstack([ V*w[:,None], -np.ones((len(x),1))]), np.hstack([-V*w[:,None], -np.ones((len(x),1))])])
    bub = np.concatenate([f*w, -f*w])
    Aeq = np.hstack([np.ones((1,n+1)), np.zeros((1,1))]) * 1.0   # R(A^2) = sum a_k = 0.5/A
    beq = np.array([0.5/A])
    res = linprog(c, A_ub=Aub, b_ub=bub, A_eq=Aeq, b_eq=beq, bounds=[(None,None)]*(n+1)+[(0,None)], method="highs")
    a = res.x[:n+1]
    return np.array([a[k]/(A*A)**k for k in range(n+1)]), res.x[-1]
def evalg(c, A, xs):
    xs32 = xs.astype(np.float32); xc = np.clip(xs32, -A, A).astype(np.float32); t = (xc*xc).astype(np.float32)
    c32 = [np.float32(v) for v in c]
    r = np.full_like(t, c32[-1])
    for k in range(len(c)-2, -1, -1): r = (r*t + c32[k]).astype(np.float32)
    return (xs32*(xc*r + np.float32(0.5)).astype(np.float32)).astype(np.float32)
if __name__ == "__main__":
    xs = np.linspace(-9, 9, 720001); ref = xs*Phi(xs)
    for n,A,rw in ((7,4.0,1.0),(6,4.0,1.0),(6,3.75,1.0),(5,3.75,1.0),(5,3.5,1.0),(5,3.75,0.3),(6,4.0,0.3)):
        c,E = fit(n, A, rw)
        g = evalg(c, A, xs).astype(np.float64); err = np.abs(g-ref); pos = xs>0.02
        print(f"n={n} A={A} relw={rw}: E={E:.2e} max abs {err.max():.2e} @ {xs[err.argmax()]:.2f}  max rel(x>.02) {np.max(err[pos]/ref[pos]):.2e}  coeffs:", ", ".join(float(np.float32(v)).hex().replace('0000000p','p') for v in c))
