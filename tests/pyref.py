"""Definition-level big-int restatement of the reference path (pure Python).

Second, independent implementation used only by tests to cross-check the C++
oracle at tiny sizes: O(n^2) DFTs straight from the definition, schoolbook
polynomial arithmetic, hashlib SHA-256.  Follows the same reference lines as
oracle/ministark_oracle.cpp (src/starks.rs:59-169, src/fri.rs:64-189,
src/merkle.rs:81-289, src/field.rs:36-109) but shares no code with it.
"""
import hashlib

GL_P = 2**64 - 2**32 + 1
BB_P = 2013265921

FIELDS = {
    0: dict(p=GL_P, gen=7, adicity=32, ext=2),
    1: dict(p=BB_P, gen=440564289, adicity=27, ext=4),
}


def root_of_unity(field, n):
    f = FIELDS[field]
    p = f["p"]
    w = pow(f["gen"], (p - 1) >> f["adicity"], p)
    k = n.bit_length() - 1
    assert 1 << k == n and k <= f["adicity"]
    for _ in range(k, f["adicity"]):
        w = w * w % p
    return w


# --- extension towers as tuples of base limbs -------------------------------
class Tower:
    def __init__(self, field, e):
        self.field, self.e, self.p = field, e, FIELDS[field]["p"]
        self.nr2 = 7 if field == 0 else 11
        self.nr4 = (2013265910, 1)  # field.rs:98 (BabyBear only)

    def zero(self):
        return (0,) * self.e

    def one(self):
        return (1,) + (0,) * (self.e - 1)

    def from_base(self, b):
        return (b % self.p,) + (0,) * (self.e - 1)

    def add(self, a, b):
        return tuple((x + y) % self.p for x, y in zip(a, b))

    def sub(self, a, b):
        return tuple((x - y) % self.p for x, y in zip(a, b))

    def neg(self, a):
        return tuple((-x) % self.p for x in a)

    def _mul2(self, a, b):
        p = self.p
        return ((a[0] * b[0] + self.nr2 * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def mul(self, a, b):
        p = self.p
        if self.e == 1:
            return (a[0] * b[0] % p,)
        if self.e == 2:
            return self._mul2(a, b)
        a0, a1, b0, b1 = a[:2], a[2:], b[:2], b[2:]
        t = self._mul2(self.nr4, self._mul2(a1, b1))
        m = self._mul2(a0, b0)
        r0 = ((m[0] + t[0]) % p, (m[1] + t[1]) % p)
        x, y = self._mul2(a0, b1), self._mul2(a1, b0)
        return r0 + ((x[0] + y[0]) % p, (x[1] + y[1]) % p)

    def pow(self, a, e):
        r = self.one()
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.mul(a, a)
            e >>= 1
        return r

    def inv(self, a):
        # Fermat in the extension: a^(p^e - 2)
        return self.pow(a, self.p**self.e - 2)

    def is_zero(self, a):
        return all(x == 0 for x in a)


def display(limbs, zero_as_empty=True):
    """[ark-mem] ark-ff 0.5 Display: Fp decimal with '0' trimmed (zero -> '');
    QuadExtField -> 'QuadExtField(c0 + c1 * u)' (nested for Fp4)."""
    if len(limbs) == 1:
        v = limbs[0]
        if v == 0:
            return "" if zero_as_empty else "0"
        return str(v)
    h = len(limbs) // 2
    return "QuadExtField(" + display(limbs[:h], zero_as_empty) + " + " + display(limbs[h:], zero_as_empty) + " * u)"


# --- src/merkle.rs:81-148 ----------------------------------------------------
def merkle_nodes(leafs, lpn, ic, zero_as_empty=True):
    """leafs: list of limb tuples.  Returns list of 32-byte nodes, level-major, root last."""
    n = len(leafs) // lpn
    assert len(leafs) % lpn == 0 and n > 0
    m = n
    while m > 1:
        assert m % ic == 0
        m //= ic
    nodes = []
    for g in range(n):
        s = "".join(display(x, zero_as_empty) for x in leafs[g * lpn:(g + 1) * lpn])
        nodes.append(hashlib.sha256(s.encode()).digest())
    idx = 0
    level = n
    total = 0
    m = n
    while True:
        total += m
        if m == 1:
            break
        m //= ic
    while len(nodes) < total:
        nodes.append(hashlib.sha256(b"".join(nodes[idx:idx + ic])).digest())
        idx += ic
    return nodes


# --- transforms from the definition -------------------------------------------
def dft(field, a, inverse=False):
    p = FIELDS[field]["p"]
    n = len(a)
    w = root_of_unity(field, n)
    if inverse:
        w = pow(w, p - 2, p)
    out = []
    for i in range(n):
        wi = pow(w, i, p)
        acc, x = 0, 1
        for k in range(n):
            acc = (acc + a[k] * x) % p
            x = x * wi % p
        out.append(acc)
    if inverse:
        ninv = pow(n, p - 2, p)
        out = [v * ninv % p for v in out]
    return out


def coset_eval(field, coeffs, shift, L):
    p = FIELDS[field]["p"]
    g = root_of_unity(field, L)
    out = []
    for i in range(L):
        x = shift * pow(g, i, p) % p
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * x + c) % p
        out.append(acc)
    return out


# --- polynomials over a Tower (lists of limb tuples, trimmed) ------------------
def trim(T, f):
    f = list(f)
    while f and T.is_zero(f[-1]):
        f.pop()
    return f


def peval(T, f, x):
    acc = T.zero()
    for c in reversed(f):
        acc = T.add(T.mul(acc, x), c)
    return acc


def psub(T, a, b):
    n = max(len(a), len(b))
    a = list(a) + [T.zero()] * (n - len(a))
    b = list(b) + [T.zero()] * (n - len(b))
    return trim(T, [T.sub(x, y) for x, y in zip(a, b)])


def pdiv(T, num, den):
    """schoolbook long division, returns quotient (den monic or not)."""
    num, den = trim(T, num), trim(T, den)
    if not num or len(num) < len(den):
        return []
    r = list(num)
    dd = len(den) - 1
    lead_inv = T.inv(den[-1])
    q = [T.zero()] * (len(num) - dd)
    for i in range(len(num) - 1, dd - 1, -1):
        c = T.mul(r[i], lead_inv)
        q[i - dd] = c
        for k in range(dd + 1):
            r[i - dd + k] = T.sub(r[i - dd + k], T.mul(c, den[k]))
    return trim(T, q)


def pmul(T, a, b):
    if not a or not b:
        return []
    out = [T.zero()] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = T.add(out[i + j], T.mul(x, y))
    return trim(T, out)


def ext_domain_eval(T, field, f, D):
    p = FIELDS[field]["p"]
    g = root_of_unity(field, D)
    return [peval(T, f, T.from_base(pow(g, i, p))) for i in range(D)]


def next_pow2(n):
    k = 1
    while k < n:
        k <<= 1
    return k


class PyProver:
    """Naive Stark::prove (src/starks.rs:59-169) with challenges as inputs."""

    def __init__(self, field, zero_as_empty=True, ext=None):
        self.field = field
        self.p = FIELDS[field]["p"]
        self.e = ext or FIELDS[field]["ext"]
        self.T = Tower(field, self.e)
        self.zae = zero_as_empty

    def trace_commit(self, trace, lpn):
        self.trace = [list(map(int, r)) for r in trace]
        self.N, self.w = len(trace), len(trace[0])
        flat = [(v,) for r in self.trace for v in r]
        return merkle_nodes(flat, lpn, 2, self.zae)[-1]

    def interpolate(self):
        self.polys = [dft(self.field, [self.trace[j][c] for j in range(self.N)], inverse=True) for c in range(self.w)]

    def lincomb(self, scalars, idx):
        r = [0] * self.N
        for s, i in zip(scalars, idx):
            r = [(x + s * y) % self.p for x, y in zip(r, self.polys[i])]
        self.polys.append(r)

    def lde_commit(self, blowup, shift, lpn):
        L = self.N * blowup
        cols = [coset_eval(self.field, f, shift, L) for f in self.polys]
        self.lde = [[cols[c][i] for c in range(len(cols))] for i in range(L)]
        flat = [(v,) for r in self.lde for v in r]
        return merkle_nodes(flat, lpn, 2, self.zae)[-1]

    def mix(self, r):
        v = [0] * self.N
        ri = 1
        for f in self.polys:
            v = [(x + ri * y) % self.p for x, y in zip(v, f)]
            ri = ri * r % self.p
        self.validity = v

    def eval_ext(self, zs):
        T = self.T
        out = []
        for z in zs:
            row = []
            for f in self.polys + [self.validity]:
                row.append(peval(T, trim(T, [T.from_base(c) for c in f]), tuple(z)))
            out.append(row)
        return out

    def _new_round(self, poly, dsize):
        T = self.T
        D = next_pow2(dsize)
        ev = ext_domain_eval(T, self.field, poly, D)
        nodes = merkle_nodes(ev, 2, 2, self.zae)
        split = [trim(T, poly[0::2]), trim(T, poly[1::2])]
        self.rounds.append(dict(poly=poly, D=D, ev=ev, nodes=nodes, split=split))
        return nodes[-1]

    def fri_begin(self, blowup, rounds):
        T = self.T
        self.rounds = []
        self.nrounds = rounds
        p = trim(T, [T.from_base(c) for c in self.validity])
        deg = len(p) - 1 if p else 0
        return self._new_round(p, (deg + 1) * blowup)

    def fri_deep(self, z):
        T = self.T
        pr = self.rounds[-1]
        self.z = tuple(z)
        self.B = [peval(T, pr["split"][0], self.z), peval(T, pr["split"][1], self.z)]
        return self.B

    def fri_fold_commit(self, alpha):
        T = self.T
        alpha = tuple(alpha)
        pr = self.rounds[-1]
        ev, od = pr["split"]
        n = max(len(ev), len(od))
        folded = []
        for i in range(n):
            a = ev[i] if i < len(ev) else T.zero()
            b = od[i] if i < len(od) else T.zero()
            folded.append(T.add(a, T.mul(alpha, b)))
        folded = trim(T, folded)
        deep_value = T.add(self.B[0], T.mul(self.B[1], alpha))
        num = psub(T, folded, trim(T, [deep_value]))
        rp = pdiv(T, num, [T.neg(self.z), T.one()])
        return self._new_round(rp, pr["D"] // 2)

    def _open(self, rnd, y):
        ev, nodes, D = rnd["ev"], rnd["nodes"], rnd["D"]
        idx = ev.index(tuple(y))
        start = idx - idx % 2
        out = [("idx", idx), ("neigh", ev[start:start + 2])]
        levels = (D // 2).bit_length()  # log2(D/2)+1
        path = []
        cur = D + idx // 2
        total = D + len(nodes)
        for _ in range(1, levels):
            sh = cur - D
            s = sh - sh % 2
            path.append(nodes[s:s + 2])
            cur = cur + (total - cur + 1) // 2
        out.append(("path", path))
        return out

    def fri_query(self, betas):
        T, p = self.T, self.p
        res = []
        for prev, cur in zip(self.rounds[:-1], self.rounds[1:]):
            gp, gc = root_of_unity(self.field, prev["D"]), root_of_unity(self.field, cur["D"])
            rr = []
            for beta in betas:
                if beta > prev["D"]:
                    beta %= prev["D"]
                x1 = T.from_base(pow(gp, beta, p))
                x2 = T.from_base(pow(gp, cur["D"] + beta, p))
                x3 = T.from_base(pow(gc, beta, p))
                y1, y2, y3 = peval(T, prev["poly"], x1), peval(T, prev["poly"], x2), peval(T, cur["poly"], x3)
                a = T.mul(T.sub(y2, y1), T.inv(T.sub(x2, x1)))
                b = T.sub(y1, T.mul(a, x1))
                g = trim(T, [b, a])
                num = psub(T, prev["poly"], g)
                van = pmul(T, [T.neg(x1), T.one()], [T.neg(x2), T.one()])
                q = pdiv(T, num, van)
                rr.append(dict(points=[x1, y1, x2, y2, x3, y3], q=q, p1=self._open(prev, y1), p2=self._open(prev, y2)))
            res.append(rr)
        return res

    def serialise_fri(self, res):
        """Same byte layout as the oracle / HIP library (include/ministark.h 'MSFP')."""
        import struct
        out = bytearray()

        def put(v):
            out.extend(struct.pack("<Q", v))

        def putE(x):
            for c in x:
                put(c)
        for rr in res:
            for it in rr:
                for x in it["points"]:
                    putE(x)
                put(len(it["q"]))
                for c in it["q"]:
                    putE(c)
                for pth in (it["p1"], it["p2"]):
                    d = dict(pth)
                    put(d["idx"])
                    for x in d["neigh"]:
                        putE(x)
                    put(len(d["path"]))
                    for lvl in d["path"]:
                        for h in lvl:
                            out.extend(h)
        return bytes(out)
