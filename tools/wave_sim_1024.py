"""Lane-level numpy simulation of the 1024-pt wave-per-frame kernel.

Design aid only (not product, not oracle): each "register" is a complex array of
64 lanes; cross-lane ops follow the documented DPP / v_permlane*_swap semantics.
Validates the index math, twiddle tables and the real-FFT split before the HIP
transcription runs on hardware.
"""
import numpy as np

L = np.arange(64)


def dft8(v):
    """8-point DFT over the register index (list of 8 lane-arrays), natural order."""
    w = np.exp(-2j * np.pi * np.outer(np.arange(8), np.arange(8)) / 8)
    return [sum(w[k, j] * v[j] for j in range(8)) for k in range(8)]


def transpose_bit(regs, p, q):
    """swap bit p of the register index with bit q of the lane index."""
    s = 1 << p
    out = list(regs)
    lanebit = (L >> q) & 1
    partner = L ^ (1 << q)
    for i in range(len(regs)):
        if i & s:
            continue
        ri, rj = regs[i], regs[i | s]
        out[i] = np.where(lanebit == 1, rj[partner], ri)
        out[i | s] = np.where(lanebit == 0, ri[partner], rj)
    return out


def frame_fft_1024(xw):
    """xw: 1024 windowed real samples (already scaled by 0.5). returns X[0..512]."""
    z = xw[0::2] + 1j * xw[1::2]              # 512 complex
    a = L & 7
    b = L >> 3
    n2 = 8 * a + b
    v = [z[64 * j + n2] for j in range(8)]    # LDS read
    v = dft8(v)                                # over j -> k1
    v = [v[k1] * np.exp(-2j * np.pi * n2 * k1 / 512) for k1 in range(8)]
    for p in range(3):                         # E1: reg bit p <-> lane bit p
        v = transpose_bit(v, p, p)
    # lane = (k1 = L&7, b = L>>3), regs a
    v = dft8(v)                                # over a -> k_a
    v = [v[ka] * np.exp(-2j * np.pi * b * ka / 64) for ka in range(8)]
    for p in range(3):                         # E2: reg bit p <-> lane bit 3+p
        v = transpose_bit(v, p, 3 + p)
    v = dft8(v)                                # over b -> k_b ; Z[L + 64 kb]
    Z = np.zeros(513, complex)
    for kb in range(8):
        Z[L + 64 * kb] = v[kb]
    Z[512] = Z[0]
    # split, lane handles k = L + 64 j (j<4) and k' = 512 - k
    X = np.zeros(513, complex)
    for j in range(4):
        k = L + 64 * j
        A = v[j]
        Bc = np.conj(Z[512 - k])               # partner via LDS
        E, O = A + Bc, A - Bc
        th = 2 * np.pi * k / 1024
        tw = -np.sin(th) - 1j * np.cos(th)
        T = tw * O
        X[k] = E + T
        X[512 - k] = np.conj(E - T)
    X[256] = 2 * np.conj(Z[256])
    return X


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    x = rng.standard_normal(1024)
    ref = np.fft.rfft(x)
    got = frame_fft_1024(0.5 * x)
    print("max err", np.abs(got - ref).max() / np.abs(ref).max())
