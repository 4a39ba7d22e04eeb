"""Shared synthetic inputs for tests and bench."""
import numpy as np

GL_P = 2**64 - 2**32 + 1
BB_P = 2013265921
MODULUS = {0: GL_P, 1: BB_P}
EXT = {0: 2, 1: 4}
MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def field(self, p):
        return self.next() % p

    def nonzero(self, p):
        while True:
            v = self.next() % p
            if v:
                return v


def fibonacci_trace(field, N, secret_b=2, pad_seed=0x5EED):
    """tests/e2e_goldilocks.rs:20-63: w=3, rows (a,b,c), steps=N-1 filled rows,
    last row = padding (quirk Q5 stand-in: SplitMix64 mod p)."""
    p = MODULUS[field]
    t = np.zeros((N, 3), dtype=np.uint64)
    a, b = 1, secret_b % p
    c = (a + b) % p
    for i in range(N - 1):
        t[i] = (a, b, c)
        a, b = b, c
        c = (a + b) % p
    rng = SplitMix64(pad_seed)
    t[N - 1] = [rng.field(p) for _ in range(3)]
    return t


def fibonacci_trace_fast(field, N, secret_b=2, pad_seed=0x5EED):
    """Same values as fibonacci_trace, vectorised by doubling (for N >= 2^16)."""
    if N <= 1 << 12:
        return fibonacci_trace(field, N, secret_b, pad_seed)
    p = MODULUS[field]
    seq = [1, secret_b % p]
    # python big-int loop is ~1 us/step; fine up to 2^24 in ~20 s, so use object-free ints
    a, b = seq
    out = np.empty(N + 1, dtype=np.uint64)
    out[0], out[1] = a, b
    for i in range(2, N + 1):
        a, b = b, (a + b) % p
        out[i] = b
    t = np.empty((N, 3), dtype=np.uint64)
    t[:, 0] = out[0:N]
    t[:, 1] = out[1:N + 1]
    t[: N - 1, 2] = out[2:N + 1]
    rng = SplitMix64(pad_seed)
    t[N - 1] = [rng.field(p) for _ in range(3)]
    return t


def fibonacci_closures(field, N, omega):
    """Transition polys of tests/e2e_goldilocks.rs:48-59 as (scalars, idx) lincombs:
    w*P0 - P1 (twice, quirk Q2) and P2 - P0 - P1."""
    p = MODULUS[field]
    m1 = p - 1
    return [([omega, m1], [0, 1]), ([omega, m1], [0, 1]), ([1, m1, m1], [2, 0, 1])]
