"""Dataflow model (numpy, float64) of stft_wave1024x2_kernel: two frames side by side in the half-waves,
512 = 16 x 16 x 2.  Checks the lane / register maps, the twiddle tables and the split against numpy's rfft.
Diagnostic, not shipped; no oracle import."""
import numpy as np

rng = np.random.default_rng(1)
N = 1024
x = rng.standard_normal(N)
w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)
z = (x * w * 0.5).reshape(512, 2)
z = z[:, 0] + 1j * z[:, 1]                      # z[m] = 0.5 w x (2m, 2m+1)
W = lambda n, e: np.exp(-2j * np.pi * e / n)

lanes = np.arange(32)
# 1. v[j][l] = z[32 j + l]
v = np.array([[z[32 * j + l] for l in lanes] for j in range(16)])
# 2. radix-16 over j
A = np.array([[sum(v[j][l] * W(16, j * k1) for j in range(16)) for l in lanes] for k1 in range(16)])
# 3. twiddle W512^(l k1)
T = np.array([[A[k1][l] * W(512, l * k1) for l in lanes] for k1 in range(16)])
# 4. exchange: lane lam = kappa + 16 h gets u[i] = T[kappa][2 i + h]
u = np.zeros((16, 32), complex)
for lam in lanes:
    kap, h = lam & 15, lam >> 4
    for i in range(16):
        u[i][lam] = T[kap][2 * i + h]
# 5. radix-16 over i
B = np.array([[sum(u[i][lam] * W(16, i * q) for i in range(16)) for lam in lanes] for q in range(16)])
# 6. permlane16_swap(r[2p], r[2p+1]) + radix-2 with c[p] = W32^(2p+h)
R = np.zeros((16, 32), complex)
for lam in lanes:
    kap, h = lam & 15, lam >> 4
    partner = lam ^ 16
    for p in range(8):
        if h == 0:
            e, o = B[2 * p][lam], B[2 * p][partner]          # own r[2p], partner's r[2p]
        else:
            e, o = B[2 * p + 1][partner], B[2 * p + 1][lam]  # partner's r[2p+1], own r[2p+1]
        c = W(32, 2 * p + h)
        R[p][lam] = e + c * o
        R[p + 8][lam] = e - c * o
Zref = np.fft.fft(z)
for lam in lanes:
    for r in range(16):
        assert abs(R[r][lam] - Zref[lam + 32 * r]) < 1e-9, (lam, r)
# 7. split
strip = np.zeros(257, complex)
for lam in lanes:
    for r in range(8, 16):
        strip[lam + 32 * (r - 8)] = R[r][lam]
strip[256] = R[0][0]
X = np.zeros(513, complex)
for lam in lanes:
    for r in range(8):
        k = lam + 32 * r
        prt = strip[256 - k]
        th = 2 * np.pi * k / 1024
        tws = -np.sin(th) - 1j * np.cos(th)
        se, so = R[r][lam] + np.conj(prt), R[r][lam] - np.conj(prt)
        st = so * tws
        X[k] = se + st
        X[512 - k] = np.conj(se - st)
X[256] = 2 * np.conj(R[8][0])
ref = np.fft.rfft(x * w)
print("max err vs rfft:", np.abs(X - ref).max())
assert np.abs(X - ref).max() < 1e-9
print("ok")
