"""NumPy study behind the split-GEMM arms (csrc/gemm_f16x2.hip, and round 2's bf16 x 3 arm, removed in round 3): how close to the float64
product are
  * f32 operands with f32 accumulation (what an f32 matrix pipe does),
  * three bf16 parts per operand, six products (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi),
  * a power-of-two scale from the operand's |x| maximum, two fp16 parts per operand, three products (h2.h1, h1.h2, h1.h1)
on activations of order one, gradients of order 1e-7, heavy tails and operands whose elements span e^(+-6 sigma)?
Partial products of bf16 / fp16 parts are exact in f32; only the accumulation (in blocks of 16 k, like the MFMA) rounds.
usage: python scripts/study_split_gemm.py   (prints one line per case; tests/test_split_gemm_study_cpu.py asserts on it)"""
import numpy as np
import torch


def split_bf16x3(x):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    hi = t.bfloat16().float()
    r1 = t - hi
    mid = r1.bfloat16().float()
    lo = (r1 - mid).bfloat16().float()
    return hi.numpy(), mid.numpy(), lo.numpy()


def pow2_scale(x):
    """2^(14 - floor(log2 max|x|)): the device's kws_absmax_scale (common.h), exponent clamped the same way"""
    m = np.float32(np.abs(x).max())
    e = int((m.view(np.uint32) >> 23) & 0xFF)
    e = max(e, 16)
    return float(np.uint32((268 - e) << 23).view(np.float32)), float(np.uint32((e - 14) << 23).view(np.float32))


def split_f16x2(x, scale):
    xs = (x.astype(np.float32) * np.float32(scale)).astype(np.float32)
    h1 = xs.astype(np.float16)
    h2 = (xs - h1.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h1.astype(np.float32), h2.astype(np.float32)


def accumulate(terms, K, block):
    """sum of exact partial products, rounded to f32 after every block of `block` k (one matrix instruction)"""
    acc = np.zeros((terms[0][0].shape[0], terms[0][1].shape[1]), np.float32)
    for k0 in range(0, K, block):
        sl = slice(k0, k0 + block)
        for a, b in terms:
            acc = (acc.astype(np.float64) + a[:, sl].astype(np.float64) @ b[sl].astype(np.float64)).astype(np.float32)
    return acc


def gemm_f32(A, B):
    return accumulate([(A.astype(np.float32), B.astype(np.float32))], A.shape[1], 2)      # v_mfma_f32_32x32x2_f32


def gemm_bf16x3(A, B):
    ah, am, al = split_bf16x3(A)
    bh, bm, bl = split_bf16x3(B)
    return accumulate([(al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)], A.shape[1], 16)


def gemm_f16x2(A, B):
    sa, ia = pow2_scale(A)
    sb, ib = pow2_scale(B)
    a1, a2 = split_f16x2(A, sa)
    b1, b2 = split_f16x2(B, sb)
    assert np.isfinite(a1).all() and np.isfinite(b1).all()
    return accumulate([(a2, b1), (a1, b2), (a1, b1)], A.shape[1], 16) * np.float32(ia) * np.float32(ib)


CASES = [
    ("activations, order one", lambda r, s: r.randn(*s)),
    ("ReLU6-like, non-negative", lambda r, s: np.clip(r.randn(*s) * 2 + 1, 0, 6)),
    ("gradients, order 1e-7", lambda r, s: r.randn(*s) * 1e-7),
    ("heavy tails (Student t, 2 dof)", lambda r, s: r.standard_t(2, size=s)),
    ("elements e^(6 sigma) apart", lambda r, s: r.randn(*s) * np.exp(r.randn(*s) * 6)),
    ("huge, order 1e12", lambda r, s: r.randn(*s) * 1e12),
]


def run(M=192, shapes=((128, 128), (512, 512)), seed=0):
    rng = np.random.RandomState(seed)
    rows = []
    for name, make in CASES:
        for K, N in shapes:
            A = make(rng, (M, K)).astype(np.float32)
            B = (rng.randn(K, N) * 0.1).astype(np.float32)
            ref = A.astype(np.float64) @ B.astype(np.float64)
            top = np.abs(ref).max()
            rows.append((name, K, N, np.abs(gemm_f32(A, B) - ref).max() / top, np.abs(gemm_bf16x3(A, B) - ref).max() / top,
                         np.abs(gemm_f16x2(A, B) - ref).max() / top))
    return rows


if __name__ == "__main__":
    for name, K, N, e32, e3, e2 in run():
        print("%-32s K=%3d N=%3d: f32 %.2e | bf16 x 3 %.2e | fp16 x 2 %.2e of the maximum" % (name, K, N, e32, e3, e2))
