"""Synthetic inputs of the benchmark workload: the Fibonacci
AIR trace of tests/e2e_goldilocks.rs:20-63 with SplitMix64 padding rows."""
import numpy as np

_MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)


def fibonacci_rows(p, length, steps, secret_b=2, pad_seed=0x5EED):
    """rows i < steps: (a, b, c) with a0 = 1, b0 = secret_b, c = a + b, (a,b,c) <- (b,c,b+c);
    rows >= steps: padding (stand-in for the reference's per-cell test_rng(), quirk Q5)."""
    seq = np.empty(steps + 2, dtype=np.uint64)
    a, b = 1, secret_b % p
    seq[0], seq[1] = a, b
    for i in range(2, steps + 2):
        a, b = b, (a + b) % p
        seq[i] = b
    t = np.empty((length, 3), dtype=np.uint64)
    t[:steps, 0] = seq[0:steps]
    t[:steps, 1] = seq[1:steps + 1]
    t[:steps, 2] = seq[2:steps + 2]
    rng = SplitMix64(pad_seed)
    for i in range(steps, length):
        t[i] = [rng.next() % p for _ in range(3)]
    return t
