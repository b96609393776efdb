"""First-principles big-integer model of the arithmetic on the hot path (test infrastructure).

Nothing here follows the structure of the reference's Rust code: it is textbook modular arithmetic
(Python ints), textbook affine chord-and-tangent group law and the O(n^2) DFT definition.  It is
what the golden vectors under tests/golden/ are generated from (the reference
holds no MSM/FFT known-answer vectors on the 753-bit curves), and what the small-size parity
tests compare the C++ oracle and the HIP path against.
"""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CONST = json.load(open(os.path.join(_HERE, "golden", "constants.json")))

MASK64 = (1 << 64) - 1


class Field:
    def __init__(self, tag):
        f = CONST["fields"][tag]
        self.tag = tag
        self.p = int(f["p"], 16)
        self.R = int(f["R"], 16)            # 2^768 mod p  (ABI Montgomery radix)
        self.Rinv = int(f["Rinv"], 16)
        self.two_adicity = f["two_adicity"]
        self.generator = f["generator"]
        self.root_of_unity = int(f["root_of_unity"], 16)

    def to_mont(self, x):
        return (x * self.R) % self.p

    def from_mont(self, x):
        return (x * self.Rinv) % self.p


P4 = Field("p4")
P6 = Field("p6")
FIELDS = {"p4": P4, "p6": P6}


def int_to_limbs(x, n=12):
    return [(x >> (64 * i)) & MASK64 for i in range(n)]


def limbs_to_int(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


# ---------------------------------------------------------------- extension towers (tuples of ints)
class Ext:
    """Fp[X]/(X^k - nr), elements are k-tuples of ints (k = 1, 2, 3)."""

    def __init__(self, field, k, nr):
        self.F, self.k, self.nr, self.p = field, k, nr, field.p

    def zero(self):
        return (0,) * self.k

    def one(self):
        return (1,) + (0,) * (self.k - 1)

    def add(self, a, b):
        return tuple((x + y) % self.p for x, y in zip(a, b))

    def sub(self, a, b):
        return tuple((x - y) % self.p for x, y in zip(a, b))

    def neg(self, a):
        return tuple((-x) % self.p for x in a)

    def mul(self, a, b):
        k = self.k
        out = [0] * (2 * k - 1)
        for i in range(k):
            for j in range(k):
                out[i + j] += a[i] * b[j]
        for i in range(2 * k - 2, k - 1, -1):
            out[i - k] += self.nr * out[i]
        return tuple(v % self.p for v in out[:k])

    def smul(self, s, a):
        return tuple((s * x) % self.p for x in a)

    def pow(self, a, e):
        r = self.one()
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.mul(a, a)
            e >>= 1
        return r

    def inv(self, a):
        p, nr = self.p, self.nr
        if self.k == 1:
            return (pow(a[0], -1, p),)
        if self.k == 2:  # (a0 - a1 X) / (a0^2 - nr a1^2)
            n = pow((a[0] * a[0] - nr * a[1] * a[1]) % p, -1, p)
            return ((a[0] * n) % p, (-a[1] * n) % p)
        # k == 3: adjugate / norm
        a0, a1, a2 = a
        c0 = (a0 * a0 - nr * a1 * a2) % p
        c1 = (nr * a2 * a2 - a0 * a1) % p
        c2 = (a1 * a1 - a0 * a2) % p
        n = pow((a0 * c0 + nr * (a2 * c1 + a1 * c2)) % p, -1, p)
        return ((c0 * n) % p, (c1 * n) % p, (c2 * n) % p)

    def is_zero(self, a):
        return all(x == 0 for x in a)


class Curve:
    """y^2 = x^3 + a x + b over an Ext field; affine points are (x, y) tuples or None (infinity)."""

    def __init__(self, name):
        c = CONST["curves"][name]
        self.name = name
        self.F = FIELDS[c["field"]]
        self.E = Ext(self.F, c["ext"], c["nonresidue"])
        self.order = FIELDS[c["order"]].p
        self.a = tuple(int(v, 16) for v in c["a"])
        self.b = tuple(int(v, 16) for v in c["b"])
        self.G = (tuple(int(v, 16) for v in c["gx"]), tuple(int(v, 16) for v in c["gy"]))
        self.deg = c["ext"]

    def on_curve(self, P):
        if P is None:
            return True
        E = self.E
        x, y = P
        return E.mul(y, y) == E.add(E.add(E.mul(E.mul(x, x), x), E.mul(self.a, x)), self.b)

    def neg(self, P):
        return None if P is None else (P[0], self.E.neg(P[1]))

    def add(self, P, Q):
        E = self.E
        if P is None:
            return Q
        if Q is None:
            return P
        if P[0] == Q[0]:
            if E.is_zero(E.add(P[1], Q[1])):
                return None
            lam = E.mul(E.add(E.smul(3, E.mul(P[0], P[0])), self.a), E.inv(E.smul(2, P[1])))
        else:
            lam = E.mul(E.sub(Q[1], P[1]), E.inv(E.sub(Q[0], P[0])))
        x3 = E.sub(E.sub(E.mul(lam, lam), P[0]), Q[0])
        y3 = E.sub(E.mul(lam, E.sub(P[0], x3)), P[1])
        return (x3, y3)

    def mul(self, k, P):
        R = None
        while k:
            if k & 1:
                R = self.add(R, P)
            P = self.add(P, P)
            k >>= 1
        return R

    def msm(self, bases, scalars):
        """Naive sum s_i * P_i, zip-truncating like variable_base.rs:36."""
        acc = None
        for P, s in zip(bases, scalars):
            acc = self.add(acc, self.mul(s, P))
        return acc

    def proj_to_affine(self, X, Y, Z):
        E = self.E
        if E.is_zero(Z):
            return None
        zi = E.inv(Z)
        return (E.mul(X, zi), E.mul(Y, zi))


CURVES = {n: Curve(n) for n in ("mnt4753_g1", "mnt4753_g2", "mnt6753_g1", "mnt6753_g2")}


# ---------------------------------------------------------------- ABI (de)serialisation helpers
def fe_to_abi(F, x):
    """canonical int -> 12 u64 limbs of the Montgomery form x*2^768 (the reference's in-memory form)."""
    return int_to_limbs(F.to_mont(x))


def fe_from_abi(F, limbs):
    return F.from_mont(limbs_to_int(limbs))


def ext_to_abi(F, e):
    out = []
    for c in e:
        out += fe_to_abi(F, c)
    return out


def ext_from_abi(F, limbs, k):
    return tuple(fe_from_abi(F, limbs[12 * i:12 * i + 12]) for i in range(k))


# ---------------------------------------------------------------- DFT by definition
def domain_params(F, log_n):
    """group_gen = ROOT_OF_UNITY^(2^(s - log_n)) (algebra/src/fft/domain.rs:76-79)."""
    assert log_n < F.two_adicity
    w = F.root_of_unity
    for _ in range(log_n, F.two_adicity):
        w = (w * w) % F.p
    return w


def dft(F, a, log_n, inverse=False, coset=False):
    """O(n^2) definition of fft / ifft / coset_fft / coset_ifft on canonical ints
    (contract: domain.rs:113-179; natural order in and out; input padded / truncated to n)."""
    n = 1 << log_n
    p = F.p
    a = list(a[:n]) + [0] * max(0, n - len(a))
    w = domain_params(F, log_n)
    g = F.generator
    if inverse:
        w = pow(w, -1, p)
    if coset and not inverse:
        a = [(x * pow(g, i, p)) % p for i, x in enumerate(a)]
    pw = [pow(w, i, p) for i in range(n)]
    out = [sum(a[j] * pw[(j * k) % n] for j in range(n)) % p for k in range(n)]
    if inverse:
        ninv = pow(n, -1, p)
        out = [(x * ninv) % p for x in out]
        if coset:
            gi = pow(g, -1, p)
            out = [(x * pow(gi, i, p)) % p for i, x in enumerate(out)]
    return out


def ntt_fast(F, a, log_n, inverse=False, coset=False):
    """Same contract as dft(), O(n log n) recursive radix-2 (used to check larger sizes quickly)."""
    n = 1 << log_n
    p = F.p
    a = list(a[:n]) + [0] * max(0, n - len(a))
    w = domain_params(F, log_n)
    g = F.generator
    if inverse:
        w = pow(w, -1, p)
    if coset and not inverse:
        gp = 1
        for i in range(n):
            a[i] = (a[i] * gp) % p
            gp = (gp * g) % p

    def rec(v, w):
        m = len(v)
        if m == 1:
            return v
        e = rec(v[0::2], (w * w) % p)
        o = rec(v[1::2], (w * w) % p)
        out = [0] * m
        t = 1
        for k in range(m // 2):
            x = (t * o[k]) % p
            out[k] = (e[k] + x) % p
            out[k + m // 2] = (e[k] - x) % p
            t = (t * w) % p
        return out

    out = rec(a, w)
    if inverse:
        ninv = pow(n, -1, p)
        out = [(x * ninv) % p for x in out]
        if coset:
            gi = pow(g, -1, p)
            gp = 1
            for i in range(n):
                out[i] = (out[i] * gp) % p
                gp = (gp * gi) % p
    return out


# ---------------------------------------------------------------- deterministic PRNG (xoshiro256**, SplitMix64 seeded)
class Rng:
    def __init__(self, seed):
        s = seed & MASK64
        st = []
        for _ in range(4):
            s = (s + 0x9E3779B97F4A7C15) & MASK64
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
            st.append(z ^ (z >> 31))
        self.s = st

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & MASK64

    def next_u64(self):
        s = self.s
        r = (self._rotl((s[1] * 5) & MASK64, 7) * 9) & MASK64
        t = (s[1] << 17) & MASK64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return r

    def field_elem(self, p):
        """12 random u64, top limb masked to 49 bits, rejection-sampled < p
        (same distribution as algebra/src/fields/macros.rs:11-28)."""
        while True:
            l = [self.next_u64() for _ in range(12)]
            l[11] &= (1 << 49) - 1
            x = limbs_to_int(l)
            if x < p:
                return x
