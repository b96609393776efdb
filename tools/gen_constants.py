#!/usr/bin/env python3
"""Derive every numeric constant the hot path needs and emit
   ginger-lib_amd/csrc/constants_gen.h  (C/HIP header, u64 limbs + 32-bit views)
   tests/golden/constants.json          (same numbers for the Python side)

Run in the authoring container only (it reads DATA constants -- moduli, curve
coefficients, generators -- from /root/reference and cross-checks every one of
them against first-principles Python big-int arithmetic).  The outputs are
committed; nothing at run time reads /root/reference.

Reference data locations (file:line are cited in the emitted header):
  algebra/src/fields/mnt4753/fq.rs:18-112   p4, R, R2, INV, GENERATOR, ROOT_OF_UNITY
  algebra/src/fields/mnt6753/fq.rs:17-111   p6, ...
  algebra/src/curves/mnt4753/g1.rs:20-112, g2.rs:21-200
  algebra/src/curves/mnt6753/g1.rs:20-112, g2.rs:20-273
"""
import json, os, re, sys

REF = "/root/reference/algebra/src"
OUT_H = os.path.join(os.path.dirname(__file__), "..", "ginger-lib_amd", "csrc", "constants_gen.h")
OUT_J = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "constants.json")

NL = 12
MASK64 = (1 << 64) - 1


def limbs_to_int(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def int_to_limbs(x, n=NL):
    return [(x >> (64 * i)) & MASK64 for i in range(n)]


def parse_const(path, name, occurrence=0):
    """Return the integer of the `occurrence`-th BigInteger768([...]) / BigInteger([...])
    that follows `name` in file `path`."""
    src = open(os.path.join(REF, path)).read()
    idx = [m.start() for m in re.finditer(r"\b" + re.escape(name) + r"\b\s*:", src)]
    if not idx:
        raise KeyError(name)
    pos = idx[0]
    arrs = list(re.finditer(r"BigInteger(?:768)?\(\[(.*?)\]\)", src[pos:], re.S))
    body = arrs[occurrence].group(1)
    toks = [t.strip() for t in body.replace("\n", " ").split(",") if t.strip()]
    vals = [int(t, 0) for t in toks]
    assert len(vals) == NL, (name, len(vals))
    return limbs_to_int(vals)


def parse_u64(path, name):
    src = open(os.path.join(REF, path)).read()
    m = re.search(r"const\s+" + name + r"\s*:\s*u(?:64|32)\s*=\s*([0-9xXa-fA-F_]+)", src)
    return int(m.group(1).replace("_", ""), 0)


def field_params(tag, path):
    p = parse_const(path, "MODULUS")
    R = (1 << 768) % p
    Rinv = pow(R, -1, p)
    d = dict(tag=tag, p=p, R=R, R2=(R * R) % p, R3=(R * R * R) % p,
             inv64=(-pow(p, -1, 1 << 64)) % (1 << 64),
             inv32=(-pow(p, -1, 1 << 32)) % (1 << 32))
    # cross-check against the reference's own constants
    assert parse_const(path, "R") == R, tag
    assert parse_const(path, "R2") == d["R2"], tag
    assert parse_u64(path, "INV") == d["inv64"], tag
    assert parse_u64(path, "MODULUS_BITS") == 753
    s = parse_u64(path, "TWO_ADICITY")
    d["two_adicity"] = s
    assert (p - 1) % (1 << s) == 0 and ((p - 1) >> s) & 1
    T = (p - 1) >> s
    gen_m = parse_const(path, "GENERATOR")
    assert gen_m == (17 * R) % p, tag
    d["generator"] = 17
    rou = pow(17, T, p)
    assert parse_const(path, "ROOT_OF_UNITY") == (rou * R) % p, tag
    assert pow(rou, 1 << s, p) == 1 and pow(rou, 1 << (s - 1), p) != 1
    d["root_of_unity"] = rou
    d["Rinv"] = Rinv
    return d


def main():
    F4 = field_params("p4", "fields/mnt4753/fq.rs")   # MNT4 Fq == MNT6 Fr
    F6 = field_params("p6", "fields/mnt6753/fq.rs")   # MNT6 Fq == MNT4 Fr
    p4, p6 = F4["p"], F6["p"]
    # the scalar-field files must be the swapped moduli
    assert parse_const("fields/mnt4753/fr.rs", "MODULUS") == p6 if os.path.exists(
        os.path.join(REF, "fields/mnt4753/fr.rs")) and "MODULUS" in open(
        os.path.join(REF, "fields/mnt4753/fr.rs")).read() else True

    def unm(x, F):  # Montgomery limbs -> canonical integer
        return (x * F["Rinv"]) % F["p"]

    curves = {}
    # ---- MNT4-753 G1: y^2 = x^3 + a x + b over p4, order p6
    a = unm(parse_const("curves/mnt4753/g1.rs", "COEFF_A"), F4)
    b = unm(parse_const("curves/mnt4753/g1.rs", "COEFF_B"), F4)
    gx = unm(parse_const("curves/mnt4753/g1.rs", "G1_GENERATOR_X"), F4)
    gy = unm(parse_const("curves/mnt4753/g1.rs", "G1_GENERATOR_Y"), F4)
    assert a == 2 and (gy * gy - (gx ** 3 + a * gx + b)) % p4 == 0
    curves["mnt4753_g1"] = dict(field="p4", ext=1, order="p6", a=[a], b=[b], gx=[gx], gy=[gy], nonresidue=0)
    # ---- MNT6-753 G1 over p6, order p4
    a6 = unm(parse_const("curves/mnt6753/g1.rs", "COEFF_A"), F6)
    b6 = unm(parse_const("curves/mnt6753/g1.rs", "COEFF_B"), F6)
    gx6 = unm(parse_const("curves/mnt6753/g1.rs", "G1_GENERATOR_X"), F6)
    gy6 = unm(parse_const("curves/mnt6753/g1.rs", "G1_GENERATOR_Y"), F6)
    assert a6 == 11 and (gy6 * gy6 - (gx6 ** 3 + a6 * gx6 + b6)) % p6 == 0
    curves["mnt6753_g1"] = dict(field="p6", ext=1, order="p4", a=[a6], b=[b6], gx=[gx6], gy=[gy6], nonresidue=0)
    # ---- MNT4-753 G2 over Fq2 = p4[X]/(X^2-13): a' = (13a, 0), b' = (0, 13 b)
    nr4 = 13
    g2 = {}
    for nm in ("X_C0", "X_C1", "Y_C0", "Y_C1"):
        g2[nm] = unm(parse_const("curves/mnt4753/g2.rs", "G2_GENERATOR_" + nm), F4)
    a2 = [(nr4 * a) % p4, 0]
    b2 = [0, (nr4 * b) % p4]
    assert unm(parse_const("curves/mnt4753/g2.rs", "MUL_BY_A_C0"), F4) == a2[0]
    assert unm(parse_const("curves/mnt4753/g2.rs", "COEFF_B", 1), F4) == b2[1]
    curves["mnt4753_g2"] = dict(field="p4", ext=2, order="p6", a=a2, b=b2,
                                gx=[g2["X_C0"], g2["X_C1"]], gy=[g2["Y_C0"], g2["Y_C1"]], nonresidue=nr4)
    # ---- MNT6-753 G2 over Fq3 = p6[X]/(X^3-11): a' = (0,0,a), b' = (11 b,0,0)
    nr6 = 11
    g3 = {}
    for nm in ("X_C0", "X_C1", "X_C2", "Y_C0", "Y_C1", "Y_C2"):
        g3[nm] = unm(parse_const("curves/mnt6753/g2.rs", "G2_GENERATOR_" + nm), F6)
    a3 = [0, 0, a6]
    b3 = [(nr6 * b6) % p6, 0, 0]
    assert unm(parse_const("curves/mnt6753/g2.rs", "COEFF_B", 0), F6) == b3[0]
    assert unm(parse_const("curves/mnt6753/g2.rs", "MUL_BY_A_C0"), F6) == (nr6 * a6) % p6
    assert unm(parse_const("curves/mnt6753/g2.rs", "MUL_BY_A_C2"), F6) == a6
    curves["mnt6753_g2"] = dict(field="p6", ext=3, order="p4", a=a3, b=b3,
                                gx=[g3["X_C0"], g3["X_C1"], g3["X_C2"]],
                                gy=[g3["Y_C0"], g3["Y_C1"], g3["Y_C2"]], nonresidue=nr6)

    # on-curve checks for the extension curves (first principles)
    def ext_mul(x, y, p, nr):
        k = len(x)
        out = [0] * (2 * k - 1)
        for i in range(k):
            for j in range(k):
                out[i + j] += x[i] * y[j]
        for i in range(2 * k - 2, k - 1, -1):
            out[i - k] += nr * out[i]
        return [v % p for v in out[:k]]

    for nm in ("mnt4753_g2", "mnt6753_g2"):
        c = curves[nm]
        p = F4["p"] if c["field"] == "p4" else F6["p"]
        nr = c["nonresidue"]
        x, y = c["gx"], c["gy"]
        lhs = ext_mul(y, y, p, nr)
        x3 = ext_mul(ext_mul(x, x, p, nr), x, p, nr)
        ax = ext_mul(c["a"], x, p, nr)
        rhs = [(x3[i] + ax[i] + c["b"][i]) % p for i in range(c["ext"])]
        assert lhs == rhs, nm

    # ---------------- JSON
    def hx(v):
        return hex(v)
    J = {"fields": {}, "curves": {}}
    for F in (F4, F6):
        J["fields"][F["tag"]] = {k: (hx(v) if isinstance(v, int) and k not in ("two_adicity", "generator") else v)
                                 for k, v in F.items() if k != "tag"}
    for nm, c in curves.items():
        J["curves"][nm] = dict(field=c["field"], ext=c["ext"], order=c["order"], nonresidue=c["nonresidue"],
                               a=[hx(v) for v in c["a"]], b=[hx(v) for v in c["b"]],
                               gx=[hx(v) for v in c["gx"]], gy=[hx(v) for v in c["gy"]])
    os.makedirs(os.path.dirname(OUT_J), exist_ok=True)
    json.dump(J, open(OUT_J, "w"), indent=1)

    # ---------------- C header
    def arr64(x):
        return "{" + ", ".join("0x%016xULL" % v for v in int_to_limbs(x)) + "}"

    def arr32(x, n=24):
        return "{" + ", ".join("0x%08xu" % ((x >> (32 * i)) & 0xFFFFFFFF) for i in range(n)) + "}"

    def mont(x, F):
        return (x * F["R"]) % F["p"]

    L = []
    L.append("// GENERATED by tools/gen_constants.py -- do not edit.")
    L.append("// Numeric parameters of the MNT4-753 / MNT6-753 cycle, cross-checked against")
    L.append("// reference data constants (algebra/src/fields/mnt4753/fq.rs:18-112,")
    L.append("// algebra/src/fields/mnt6753/fq.rs:17-111, algebra/src/curves/mnt{4,6}753/{g1,g2}.rs)")
    L.append("// and re-derived from first principles (R = 2^768 mod p, INV = -p^-1, 17^T, ...).")
    L.append("#pragma once")
    L.append("#include <stdint.h>")
    L.append("#define GH_NLIMB64 12")
    L.append("#define GH_NLIMB32 24")
    for F in (F4, F6):
        t = F["tag"].upper()
        p = F["p"]
        L.append("// ---- field %s (753-bit prime, 2-adicity %d)" % (F["tag"], F["two_adicity"]))
        L.append("#define GH_%s_TWO_ADICITY %d" % (t, F["two_adicity"]))
        L.append("#define GH_%s_INV64 0x%016xULL" % (t, F["inv64"]))
        L.append("#define GH_%s_INV32 0x%08xu" % (t, F["inv32"]))
        for nm, v in (("P", p), ("R", F["R"]), ("R2", F["R2"]), ("R3", F["R3"]),
                      ("GEN17_M", mont(17, F)), ("GEN17INV_M", mont(pow(17, -1, p), F)),
                      ("ROOT_M", mont(F["root_of_unity"], F)),
                      ("ROOTINV_M", mont(pow(F["root_of_unity"], -1, p), F)),
                      ("TWOINV_M", mont(pow(2, -1, p), F))):
            L.append("#define GH_%s_%s_64 %s" % (t, nm, arr64(v)))
            L.append("#define GH_%s_%s_32 %s" % (t, nm, arr32(v)))
    # ---- device-internal representation: 26 limbs of 29 bits, Montgomery radix 2^754
    def arr29(x):
        assert 0 <= x < (1 << 754)
        return "{" + ", ".join("0x%08xu" % ((x >> (29 * i)) & ((1 << 29) - 1)) for i in range(26)) + "}"
    for F in (F4, F6):
        t = F["tag"].upper()
        p = F["p"]
        L.append("// ---- field %s, device-internal radix-2^29 form (Montgomery radix 2^754)" % F["tag"])
        L.append("#define GH_%s_INV29 0x%08xu" % (t, (-pow(p, -1, 1 << 29)) % (1 << 29)))
        for nm, v in (("P29", p), ("ONE_I29", pow(2, 754, p)), ("CIN29", pow(2, 740, p)),
                      ("COUT29", pow(2, 768, p)), ("R2I29", pow(2, 1508, p))):
            L.append("#define GH_%s_%s %s" % (t, nm, arr29(v)))
    for nm, c in curves.items():
        F = F4 if c["field"] == "p4" else F6
        t = nm.upper()
        L.append("// ---- curve %s over %s^%d" % (nm, c["field"], c["ext"]))
        for i, v in enumerate(c["a"]):
            L.append("#define GH_%s_A%d_I29 %s" % (t, i, arr29((v * pow(2, 754, F["p"])) % F["p"])))
        L.append("#define GH_%s_EXT %d" % (t, c["ext"]))
        L.append("#define GH_%s_NONRESIDUE %d" % (t, c["nonresidue"]))
        for key in ("a", "b", "gx", "gy"):
            for i, v in enumerate(c[key]):
                L.append("#define GH_%s_%s%d_M_64 %s" % (t, key.upper(), i, arr64(mont(v, F))))
    L.append("")
    os.makedirs(os.path.dirname(OUT_H), exist_ok=True)
    open(OUT_H, "w").write("\n".join(L))
    # the oracle keeps its own copy so that it does not include anything from the product tree
    open(os.path.join(os.path.dirname(__file__), "..", "oracle", "constants_gen.h"), "w").write("\n".join(L))
    print("wrote", os.path.normpath(OUT_H), "and", os.path.normpath(OUT_J))


if __name__ == "__main__":
    main()
