"""Math-level reference in pure Python ints (tiny sizes only).

Independent of oracle/tf_oracle.c: works on *canonical values* with `% P` arithmetic and
converts to/from Montgomery words only where the reference algorithm itself looks at raw
words (Tip5's byte lookup).  Used to cross-check the C oracle beyond the literal KATs.
"""
P = (1 << 64) - (1 << 32) + 1
R = 1 << 64
R_INV = pow(R, P - 2, P)


def to_raw(v):  # BFieldElement::new, math/b_field_element.rs:235
    return (v % P) * R % P


def to_val(raw):  # BFieldElement::value, :248
    return raw * R_INV % P


def root_of_unity(n):  # omega_n = 7^((p-1)/n); SURVEY 7a (checked against the table in test_oracle_kat)
    return pow(7, (P - 1) // n, P)


def dft(vals, inverse=False):
    """Naive O(n^2) DFT on canonical values: out[k] = sum_j x[j] w^(jk)  (ntt.rs:67-82 convention)."""
    n = len(vals)
    if n == 0:
        return []
    w = root_of_unity(n)
    if inverse:
        w = pow(w, P - 2, P)
    out = [sum(x * pow(w, j * k, P) for j, x in enumerate(vals)) % P for k in range(n)]
    if inverse:
        ninv = pow(n, P - 2, P)
        out = [o * ninv % P for o in out]
    return out


LOOKUP = [((x + 1) ** 3 + 256) % 257 & 0xFF for x in range(256)]  # tip5/mod.rs:1022-1026
MDS_COL = [61402, 1108, 28750, 33823, 7454, 43244, 53865, 12034, 56951, 27521, 41351, 40901, 12021, 59689, 26798, 17845]


def tip5_permutation(state_vals, round_constants_vals):
    """Tip5 on canonical values; tip5/naive.rs:26-76."""
    s = list(state_vals)
    for rnd in range(5):
        for i in range(4):
            raw = to_raw(s[i])
            looked = int.from_bytes(bytes(LOOKUP[b] for b in raw.to_bytes(8, "little")), "little")
            s[i] = to_val(looked)  # from_raw_bytes then interpreted mod p
        for i in range(4, 16):
            s[i] = pow(s[i], 7, P)
        s = [sum(MDS_COL[(r - c) % 16] * s[c] for c in range(16)) % P for r in range(16)]
        s = [(s[i] + round_constants_vals[16 * rnd + i]) % P for i in range(16)]
    return s


# ---- the callers on either side of the path, pure Python: canonical values, schoolbook algorithms only.
# Nothing below shares code or structure with oracle/tf_oracle.c (which restates the reference's fast routes): these are the
# textbook definitions the reference's property tests pin its routes to (math/polynomial.rs:3440-3800, :4593-4638).
def splitmix_values(count, seed, first_index=0):
    """Canonical values of SURVEY.md 8(d)'s synthetic inputs: element i = SplitMix64(seed ^ i) mod p."""
    m = (1 << 64) - 1
    out = []
    for i in range(first_index, first_index + count):
        z = ((seed ^ i) + 0x9E3779B97F4A7C15) & m
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        out.append((z ^ (z >> 31)) % P)
    return out


def xfe(v):  # lift a base-field value (x_field_element.rs:133-137)
    return (v % P, 0, 0)


def xfe_add(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def xfe_sub(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def xfe_mul(a, b):
    """Product in F_p[x] / (x^3 - x + 1)  (x_field_element.rs:56-59: the Shah polynomial), schoolbook then x^3 = x - 1, x^4 = x^2 - x."""
    c = [0] * 5
    for i in range(3):
        for j in range(3):
            c[i + j] += a[i] * b[j]
    return ((c[0] - c[3]) % P, (c[1] + c[3] - c[4]) % P, (c[2] + c[4]) % P)


def xfe_pow(a, e):
    r = (1, 0, 0)
    while e:
        if e & 1:
            r = xfe_mul(r, a)
        a = xfe_mul(a, a)
        e >>= 1
    return r


def xfe_inv(a):  # Fermat in the field of p^3 elements
    assert any(a)
    return xfe_pow(a, P ** 3 - 2)


class Field:
    """Element arithmetic of width 1 (ints mod p) or width 3 (extension-field triples), so every routine below is written once."""

    def __init__(self, width):
        self.w = width
        self.zero = 0 if width == 1 else (0, 0, 0)
        self.one = 1 if width == 1 else (1, 0, 0)

    def add(self, a, b):
        return (a + b) % P if self.w == 1 else xfe_add(a, b)

    def sub(self, a, b):
        return (a - b) % P if self.w == 1 else xfe_sub(a, b)

    def mul(self, a, b):
        return a * b % P if self.w == 1 else xfe_mul(a, b)

    def inv(self, a):
        return pow(a, P - 2, P) if self.w == 1 else xfe_inv(a)

    def lift(self, v):
        return v % P if self.w == 1 else xfe(v)

    def group(self, flat):  # flat list of canonical words -> list of elements
        return list(flat) if self.w == 1 else [tuple(flat[3 * i:3 * i + 3]) for i in range(len(flat) // 3)]

    def flat(self, elems):
        return list(elems) if self.w == 1 else [c for e in elems for c in e]


def poly_mul(F, a, b):
    if not a or not b:
        return []
    out = [F.zero] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = F.add(out[i + j], F.mul(x, y))
    return out


def poly_eval(F, coeffs, x):
    acc = F.zero
    for c in reversed(coeffs):
        acc = F.add(F.mul(acc, x), c)
    return acc


def zerofier(F, roots):
    """prod (x - r), coefficients low to high (polynomial.rs:1462-1475 computes the same product through a tree)."""
    out = [F.one]
    for r in roots:
        out = poly_mul(F, out, [F.sub(F.zero, r), F.one])
    return out


def lagrange_interpolate(F, domain, values):
    """sum_i v_i prod_{j != i} (x - d_j) / (d_i - d_j): the definition (polynomial.rs:1565-1606 reaches it through zerofier trees)."""
    n = len(domain)
    out = [F.zero] * n
    for i in range(n):
        num, den = [F.one], F.one
        for j in range(n):
            if j != i:
                num = poly_mul(F, num, [F.sub(F.zero, domain[j]), F.one])
                den = F.mul(den, F.sub(domain[i], domain[j]))
        s = F.mul(values[i], F.inv(den))
        for k in range(n):
            out[k] = F.add(out[k], F.mul(num[k], s))
    return out


def long_divide(a, b):
    """Schoolbook division of base-field polynomials: (quotient, remainder); clean_divide (polynomial.rs:2358-2411) must return the
    quotient whenever the remainder is zero."""
    a = list(a)
    while b and b[-1] == 0:
        b = b[:-1]
    assert b
    q = [0] * max(0, len(a) - len(b) + 1)
    binv = pow(b[-1], P - 2, P)
    for k in range(len(a) - len(b), -1, -1):
        f = a[k + len(b) - 1] * binv % P
        q[k] = f
        for j, c in enumerate(b):
            a[k + j] = (a[k + j] - f * c) % P
    return q, a[:len(b) - 1]


def coset_evaluate(F, coeffs, offset, order):
    """Values of the polynomial at offset * w_order^i, i < order, by Horner (fast_coset_evaluate, polynomial.rs:1374-1399, computes
    them by scaling and one NTT).  `offset` is an element of F; w_order is lifted."""
    w = root_of_unity(order)
    return [poly_eval(F, coeffs, F.mul(offset, F.lift(pow(w, i, P)))) for i in range(order)]


def barycentric_evaluate(F, codeword, x):
    """The value at the extension-field point x of the polynomial of degree < n that takes codeword[i] at w_n^i: the interpolant
    through the naive inverse DFT, then Horner at x (barycentric_evaluate, polynomial.rs:2609-2637, uses the barycentric formula).
    Returns an extension-field triple."""
    n = len(codeword)
    if F.w == 1:
        limbs = [dft(list(codeword), inverse=True)]
    else:
        limbs = [dft([e[k] for e in codeword], inverse=True) for k in range(3)]
    acc = (0, 0, 0)
    for i in range(n - 1, -1, -1):
        c = xfe(limbs[0][i]) if F.w == 1 else (limbs[0][i], limbs[1][i], limbs[2][i])
        acc = xfe_add(xfe_mul(acc, x), c)
    return acc
