"""Lane-level numpy simulation of the 256-point (onset geometry) kernel: 16 lanes per frame,
8 complex points per lane, radix-8 in registers then a 16-point radix-2 DIF across lanes.
Design aid only."""
import numpy as np

Q = np.arange(16)


def bitrev4(q):
    return ((q & 1) << 3) | ((q & 2) << 1) | ((q & 4) >> 1) | ((q & 8) >> 3)


def frame_fft_256(xw):
    """xw: 256 windowed reals scaled by 0.5 -> X[0..128]"""
    z = xw[0::2] + 1j * xw[1::2]                  # 128 complex
    v = [z[16 * j + Q] for j in range(8)]          # lane q reads m = 16 j + q
    w8 = np.exp(-2j * np.pi * np.outer(np.arange(8), np.arange(8)) / 8)
    v = [sum(w8[k, j] * v[j] for j in range(8)) for k in range(8)]          # radix-8 over j -> k1
    v = [v[k1] * np.exp(-2j * np.pi * Q * k1 / 128) for k1 in range(8)]
    # 16-point DIF across lanes, partner distance d = 8, 4, 2, 1
    for d in (8, 4, 2, 1):
        upper = (Q & d) != 0
        sigma = np.where(upper, -1.0, 1.0)
        # stage twiddle for upper lanes: W_(2d)^(q mod d)
        tw = np.where(upper, np.exp(-2j * np.pi * (Q % d) / (2 * d)), 1.0)
        v = [(sigma * x + x[Q ^ d]) * tw for x in v]
    # lane q holds Z[k1 + 8 * bitrev4(q)]
    Z = np.zeros(129, complex)
    for k1 in range(8):
        Z[k1 + 8 * bitrev4(Q)] = v[k1]
    Z[128] = Z[0]
    X = np.zeros(129, complex)
    for k in range(65):
        a, p = Z[k], Z[128 - k]
        e, o = a + np.conj(p), a - np.conj(p)
        th = 2 * np.pi * k / 256
        t = (-np.sin(th) - 1j * np.cos(th)) * o
        X[k] = e + t
        X[128 - k] = np.conj(e - t)
    return X


if __name__ == "__main__":
    rng = np.random.default_rng(2)
    x = rng.standard_normal(256)
    ref = np.fft.rfft(x)
    got = frame_fft_256(0.5 * x)
    print("max err", np.abs(got - ref).max() / np.abs(ref).max())
