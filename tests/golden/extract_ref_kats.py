#!/usr/bin/env python3
"""Extract the numeric known-answer vectors (inputs and expected outputs) that the reference's own
unit tests hold for the 753-bit fields and curves into tests/golden/ref_kats.json.

Only DATA is taken: for every listed test function, the ordered list of BigInteger768 literals
with the constructor they are wrapped in ("new" = raw Montgomery limbs, "from_repr" = canonical
integer).  How each list is to be read (which entries are operands, which the expected result)
is documented next to the test that consumes it (tests/test_ref_kats.py), from the reference
lines cited there.  Run in the authoring container only (/root/reference is not on the GPU box).
"""
import json, os, re

REF = "/root/reference/algebra/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kats.json")

WANT = {
    "fields/mnt4753/tests.rs": ["test_neg_one", "test_fq_add_assign", "test_fq_sub_assign", "test_fq_mul_assign", "test_fq_squaring",
                                "test_fq2_squaring", "test_fq2_mul", "test_fq2_inverse", "test_fq2_addition",
                                "test_fq2_subtraction", "test_fq2_negation", "test_fq2_doubling"],
    "fields/mnt6753/tests.rs": ["test_neg_one", "test_fq_add_assign", "test_fq_sub_assign", "test_fq_mul_assign", "test_fq_squaring",
                                "test_fq3_squaring", "test_fq3_mul", "test_fq3_inverse", "test_fq3_addition",
                                "test_fq3_subtraction", "test_fq3_negation", "test_fq3_doubling"],
    "curves/mnt4753/tests.rs": ["test_g1_addition_correctness", "test_g1_doubling_correctness", "test_g1_scalar_multiplication",
                                "test_g1_affine_projective_conversion", "test_g2_addition_correctness",
                                "test_g2_doubling_correctness", "test_g2_affine_projective_conversion"],
    "curves/mnt6753/tests.rs": ["test_g1_addition_correctness", "test_g1_doubling_correctness", "test_g1_scalar_multiplication",
                                "test_g1_affine_projective_conversion", "test_g2_addition_correctness",
                                "test_g2_doubling_correctness", "test_g2_affine_projective_conversion"],
}


def main():
    out = {}
    for path, names in WANT.items():
        src = open(os.path.join(REF, path)).read()
        starts = [(m.start(), m.group(1)) for m in re.finditer(r"\nfn (test_\w+)\s*\(", src)] + [(len(src), None)]
        bodies = {name: src[a:b] for (a, name), (b, _) in zip(starts, starts[1:])}
        out[path] = {}
        for name in names:
            body = bodies[name]
            line = src[:src.index("fn " + name)].count("\n") + 1
            vecs = []
            for m in re.finditer(r"(\w+)::(new|from_repr)\(\s*BigInteger768\(\[(.*?)\]\)", body, re.S):
                toks = [t.strip() for t in m.group(3).replace("\n", " ").split(",") if t.strip()]
                limbs = [int(t, 0) for t in toks]
                assert len(limbs) == 12, (path, name)
                vecs.append({"type": m.group(1), "ctor": m.group(2), "v": hex(sum(v << (64 * i) for i, v in enumerate(limbs)))})
            total = len(re.findall(r"BigInteger768\(\[", body))
            assert len(vecs) == total, (path, name, len(vecs), total)
            out[path][name] = {"line": line, "vectors": vecs}
    # the two 96-byte serialisation fixtures (fields/mnt{4,6}753/tests.rs test_fq_bytes) are data files
    for tag in ("mnt4753", "mnt6753"):
        p = os.path.join(REF, "fields", tag, "test_vec", "%s_tobyte" % tag)
        out["test_vec/%s_tobyte" % tag] = open(p, "rb").read().hex()
    json.dump(out, open(OUT, "w"), indent=0)
    print("wrote", OUT, {k: (len(v) if isinstance(v, dict) else "bytes") for k, v in out.items()})


if __name__ == "__main__":
    main()
