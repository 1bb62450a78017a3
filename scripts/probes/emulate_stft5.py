"""Lane-level emulation (NumPy, float64) of csrc/stft5.hip's index algebra: BOTH radix-16 passes of the 16 x 16 Cooley-Tukey
split as v_mfma_f32_16x16x32_f16 products, one frame per product chain, chained through registers:

  stage 1  D1[n2][k1]  = sum over (n1, re/im) A1[n2][(n1, ri)] B1[(n1, ri)][k1]       A1 = windowed PCM (gathered), B1 constant
  twiddle  Y'[n2][k1]  = D1 x W256^(n2 k1)                                            4 per-lane complex constants
  stage 2  D2[k2][k1]  = sum over (n2, re/im) A2[k2][(n2, ri)] B2[(n2, ri)][k1]       A2 constant, B2 = Y' AS IT LIES in the
                                                                                      registers of D1 (rows of D1 = K of stage 2)
  split    X[k], X[256-k] from Z[k] and conj Z[256-k]: the partner sits in the mirrored lane of the 16-lane row (column order
           P1) and two registers further (row order P2); only the k1 = 0 and k1 = 8 columns pick other sources.

Development aid: run it when a mapping changes - it compares the emulated magnitudes with numpy.fft.rfft (nothing here
touches a GPU or the product)."""
import numpy as np

P1 = [8, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 0]       # column c -> k1; mirror c <-> 15 - c pairs k1 with 16 - k1
P2 = [0] * 16                                                       # row 4 g + i -> k2; register i <-> i + 2 pairs k2 with 15 - k2
for g in range(4):
    P2[4 * g + 0], P2[4 * g + 1], P2[4 * g + 2], P2[4 * g + 3] = 2 * g, 2 * g + 1, 15 - 2 * g, 14 - 2 * g


def mfma_16x16x32(a, b):
    """a[64][8], b[64][8] per-lane fragments -> d[64][4]: A[l&15][8(l>>4)+j], B[8(l>>4)+j][l&15], D[4(l>>4)+i][l&15]."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        for j in range(8):
            A[l & 15, 8 * (l >> 4) + j] = a[l, j]
            B[8 * (l >> 4) + j, l & 15] = b[l, j]
    D = A @ B
    return np.array([[D[4 * (l >> 4) + i, l & 15] for i in range(4)] for l in range(64)])


def dft_operand(lane, j, ct, perm):
    """the constant operand of either stage for (lane, element j, output tile ct = 0 re / 1 im): index q = 4 kg + j / 2 of the
    contraction, part j % 2 (re / im of the data), output index perm[lane % 16]"""
    kg, o = lane >> 4, perm[lane & 15]
    q, ri = 4 * kg + (j >> 1), j & 1
    th = 2 * np.pi * q * o / 16.0
    if ct == 0:
        return np.cos(th) if ri == 0 else np.sin(th)
    return -np.sin(th) if ri == 0 else np.cos(th)


def emulate_frame(fr):
    """fr[512]: one windowed zero-padded frame -> (Zre, Zim)[64 lanes][4 regs] = Z[k1 + 16 k2] / 2 of lane (c, g), register i"""
    a1 = np.zeros((64, 8))
    for l in range(64):
        n2, kg = l & 15, l >> 4
        for j in range(8):
            n1, ri = 4 * kg + (j >> 1), j & 1
            a1[l, j] = fr[2 * (16 * n1 + n2) + ri]
    d1 = []
    for ct in range(2):
        b1 = np.array([[0.5 * dft_operand(l, j, ct, P1) for j in range(8)] for l in range(64)])
        d1.append(mfma_16x16x32(a1, b1))
    b2 = np.zeros((64, 8))
    for l in range(64):
        c, g = l & 15, l >> 4
        for i in range(4):
            y = (d1[0][l, i] + 1j * d1[1][l, i]) * np.exp(-2j * np.pi * (4 * g + i) * P1[c] / 256.0)
            b2[l, 2 * i], b2[l, 2 * i + 1] = y.real, y.imag
    d2 = []
    for ct in range(2):
        a2 = np.array([[dft_operand(l, j, ct, P2) for j in range(8)] for l in range(64)])
        d2.append(mfma_16x16x32(a2, b2))
    return d2[0], d2[1]


def split(zr, zi):
    """real-input split of one frame from the D2 layout -> magnitudes[257]"""
    Z = zr + 1j * zi
    mags = np.full(257, np.nan)
    for l in range(64):
        c, g = l & 15, l >> 4
        k1 = P1[c]
        mirror = (l & ~15) | (15 - c)
        for i in range(2):
            a = P2[4 * g + i]
            zk = Z[l, i]
            if c == 0:                       # k1 = 8: its own partner column
                zn0 = Z[l, i + 2]
            elif c == 15:                    # k1 = 0: pairs k2 with 16 - k2
                if i == 1:
                    zn0 = Z[l, 2]            # 16 - (2 g + 1) = 15 - 2 g: own register 2
                elif g == 0:
                    zn0 = Z[l, 0]            # bin 0 with itself (gives X[0] and X[256])
                else:
                    zn0 = Z[l - 16, 3]       # 16 - 2 g = 14 - 2 (g - 1): lane 15 of the row above, register 3 (row_bcast:15)
            else:
                zn0 = Z[mirror, i + 2]
            zn = np.conj(zn0)
            E = zk + zn
            O = (zk - zn) / 1j
            kk = k1 + 16 * a
            T = np.exp(-2j * np.pi * kk / 512.0) * O
            assert np.isnan(mags[kk]) and (kk == 0 or np.isnan(mags[256 - kk])), (l, i, kk)
            mags[kk] = abs(E + T)
            mags[256 - kk] = abs(E - T)
        if c == 15 and g == 3:
            mags[128] = 2 * abs(Z[l, 3])     # k2 = 8 of the k1 = 0 column: Z[128] pairs with itself
    assert not np.isnan(mags).any()
    return mags


def main():
    rng = np.random.RandomState(0)
    win = np.zeros(512)
    win[:480] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(480) / 480)
    x = rng.randn(480 + 3 * 160)
    worst = 0.0
    for u in range(4):
        fr = np.zeros(512)
        fr[:480] = x[160 * u: 160 * u + 480]
        fr *= win
        got = split(*emulate_frame(fr))
        ref = np.abs(np.fft.rfft(fr))
        worst = max(worst, np.abs(got - ref).max())
    print("max |emulated - rfft| =", worst)
    assert worst < 1e-9


if __name__ == "__main__":
    main()
