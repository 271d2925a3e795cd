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
