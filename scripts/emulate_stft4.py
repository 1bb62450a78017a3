"""Lane-level emulation (NumPy, float64) of csrc/stft4.hip's index algebra: the gather rows of the stage-1 MFMA
A operand, the constant B operand with its column permutation, the D layout handed to the in-register 16-point FFT,
the row-mirror partner of the real-input split.  Development aid: run it when the mapping changes - it compares the
emulated magnitudes with numpy.fft.rfft (nothing here touches a GPU or the product)."""
import numpy as np

KPERM = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 8]      # column c -> k1 ; mirror c <-> 15 - c pairs k1 with 16 - k1


def mfma_16x16x4(a, b, acc):
    """a[64], b[64] per-lane scalars, acc[64][4]: A[l&15][l>>4], B[l>>4][l&15], D[4(l>>4)+i][l&15]."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    for l in range(64):
        A[l & 15, l >> 4] = a[l]
        B[l >> 4, l & 15] = b[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        for i in range(4):
            out[l, i] += D[4 * (l >> 4) + i, l & 15]
    return out


def bmat(j, ct, lane):
    """constant B operand of k-chunk j (0..7), column tile ct (0 re, 1 im) for lane `lane`: 0.5 * DFT16 entry."""
    s, c = lane >> 4, lane & 15
    jp, p = j >> 1, j & 1
    n1 = 4 * jp + s
    th = 2 * np.pi * n1 * KPERM[c] / 16.0
    if ct == 0:
        v = np.cos(th) if p == 0 else np.sin(th)
    else:
        v = -np.sin(th) if p == 0 else np.cos(th)
    return 0.5 * v


def emulate_pass(frames_xw):
    """frames_xw [4][512] windowed zero-padded frames -> magnitudes [4][257]."""
    acc = np.zeros((4, 2, 64, 4))          # [t][ct][lane][i]
    for t in range(4):
        for jp in range(4):
            xv = np.zeros((64, 2))
            for l in range(64):
                r, s = l & 15, l >> 4
                u, i = r >> 2, r & 3
                m = 16 * (4 * jp + s) + 4 * t + i
                xv[l] = frames_xw[u][2 * m: 2 * m + 2]
            for p in range(2):
                j = 2 * jp + p
                for ct in range(2):
                    b = np.array([bmat(j, ct, l) for l in range(64)])
                    acc[t, ct] = mfma_16x16x4(xv[:, p], b, acc[t, ct])
    mags = np.zeros((4, 257))
    Z = np.zeros((64, 16), complex)
    for l in range(64):
        g, c = l >> 4, l & 15
        k1 = KPERM[c]
        y = np.array([acc[n2 >> 2, 0, l, n2 & 3] + 1j * acc[n2 >> 2, 1, l, n2 & 3] for n2 in range(16)])
        y = y * np.exp(-2j * np.pi * np.arange(16) * k1 / 256.0)
        Z[l] = np.fft.fft(y)              # Z[k1 + 16 k2] / 2 in register k2
    for l in range(64):
        g, c = l >> 4, l & 15
        k1 = KPERM[c]
        mirror = (l & ~15) | (15 - c)
        for k2 in range(8):
            zk = Z[l, k2]
            if c == 0:
                zn0 = Z[l, (16 - k2) & 15]
            elif c == 15:
                zn0 = Z[l, 15 - k2]
            else:
                zn0 = Z[mirror, 15 - k2]
            zn = np.conj(zn0)
            E = zk + zn                   # the 0.5 lives in the B operand
            O = (zk - zn) / 1j
            kk = k1 + 16 * k2
            T = np.exp(-2j * np.pi * kk / 512.0) * O
            mags[g, kk] = abs(E + T)
            mags[g, 256 - kk] = abs(E - T)
        if c == 0:
            mags[g, 128] = 2 * abs(Z[l, 8])
    return mags


def main():
    rng = np.random.RandomState(0)
    win = np.zeros(512)
    win[:480] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(480) / 480)
    x = rng.randn(480 + 3 * 160)
    frames = np.zeros((4, 512))
    for u in range(4):
        frames[u, :480] = x[160 * u: 160 * u + 480]
    frames *= win
    got = emulate_pass(frames)
    ref = np.abs(np.fft.rfft(frames, axis=1))
    print("max |emulated - rfft| =", np.abs(got - ref).max())
    assert np.abs(got - ref).max() < 1e-9


if __name__ == "__main__":
    main()
