#!/usr/bin/env python3
"""Regenerate tests/golden/* from the read-only reference checkout.

This script is the ONLY thing that reads /root/reference; it runs in the build
container (the GPU box has no reference checkout) and its outputs are committed.
It copies DATA (numeric constants, golden vectors, SRS points) - never source code.

Outputs
  constants.json     field / curve constants parsed out of
                       curves/src/bls12_377/{fr,fq,fq2,g1,g2}.rs
  varuna_circuit0.json  the size-8 Varuna test-vector files
                       algorithms/src/snark/varuna/resources/circuit_0/**
  srs_g1_1024.bin    first 1024 points of
                       parameters/src/mainnet/resources/powers-of-beta-15.usrs
  srs_g1_32768.bin   all 32768 points of the same file (3 MB): SURVEY.md 8(d) config 1 runs on exactly these points followed by
                       their negations (algorithms/benches/msm/variable_base.rs:29-32, msm/tests.rs:39-67)
                       (96 B each, uncompressed, canonical LE integers, flag bits
                       cleared - format per curves/src/templates/macros.rs:86-95)
  beta_h_g2.bin      the one G2 point of beta-h.usrs (192 B)
"""
import json
import os
import re
import sys

REF = os.environ.get("SNARKVM_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def read(rel):
    with open(os.path.join(REF, rel), "r") as f:
        return f.read()


def limbs_to_int(limbs):
    v = 0
    for i, l in enumerate(limbs):
        v |= l << (64 * i)
    return v


def parse_limb_list(txt):
    out = []
    for tok in re.findall(r"0x[0-9a-fA-F_]+|\d[\d_]*", txt):
        tok = tok.replace("_", "")
        if tok.endswith("u64"):
            tok = tok[:-3]
        out.append(int(tok, 0))
    return out


def const_block(src, name):
    """Return the limb list of `const NAME: ... = BigInteger([ ... ]);`"""
    m = re.search(r"const\s+" + name + r"\s*:[^=]*=\s*BigInteger(?:\d+)?(?:::new)?\(\[(.*?)\]\)", src, re.S)
    if not m:
        raise KeyError(name)
    body = re.sub(r"u64", "", m.group(1))
    return parse_limb_list(body)


def field_constants(rel):
    src = read(rel)
    d = {}
    for name in ["MODULUS", "R", "R2", "GENERATOR", "TWO_ADIC_ROOT_OF_UNITY", "MODULUS_MINUS_ONE_DIV_TWO", "T"]:
        d[name] = const_block(src, name)
    d["INV"] = int(re.search(r"const\s+INV\s*:\s*u64\s*=\s*(\d+)u64", src).group(1))
    d["TWO_ADICITY"] = int(re.search(r"const\s+TWO_ADICITY\s*:\s*u32\s*=\s*(\d+)", src).group(1))
    d["MODULUS_BITS"] = int(re.search(r"const\s+MODULUS_BITS\s*:\s*u32\s*=\s*(\d+)", src).group(1))
    d["REPR_SHAVE_BITS"] = int(re.search(r"const\s+REPR_SHAVE_BITS\s*:\s*u32\s*=\s*(\d+)", src).group(1))
    m = re.search(r"POWERS_OF_ROOTS_OF_UNITY[^=]*=\s*&\[(.*?)\];", src, re.S)
    d["POWERS_OF_ROOTS_OF_UNITY"] = [parse_limb_list(x) for x in re.findall(r"BigInteger\(\[(.*?)\]\)", m.group(1), re.S)]
    return d


def pub_const_fq(src, name):
    m = re.search(r"pub const " + name + r"\s*:\s*Fq\s*=\s*field!\(\s*Fq,\s*BigInteger384::new\(\[(.*?)\]\)", src, re.S)
    return parse_limb_list(m.group(1))


def main():
    consts = {
        "source": "AleoNet/snarkVM v1.0.0 (reference @ 2024-10-16)",
        "fr": field_constants("curves/src/bls12_377/fr.rs"),
        "fq": field_constants("curves/src/bls12_377/fq.rs"),
    }
    g1 = read("curves/src/bls12_377/g1.rs")
    consts["g1"] = {
        "GENERATOR_X_MONT": pub_const_fq(g1, "G1_GENERATOR_X"),
        "GENERATOR_Y_MONT": pub_const_fq(g1, "G1_GENERATOR_Y"),
        "GENERATOR_X_DEC": re.search(r"G1_GENERATOR_X =\s*\n\s*///\s*(\d+)", g1).group(1),
        "GENERATOR_Y_DEC": re.search(r"G1_GENERATOR_Y =\s*\n\s*///\s*(\d+)", g1).group(1),
        "COFACTOR": parse_limb_list(re.search(r"const COFACTOR: &'static \[u64\] = &\[(.*?)\];", g1).group(1)),
    }
    m = re.search(r"const WEIERSTRASS_B: Fq = field!\(\s*Fq,\s*BigInteger384\(\[(.*?)\]\)", g1, re.S)
    consts["g1"]["WEIERSTRASS_B_MONT"] = parse_limb_list(m.group(1))
    g2 = read("curves/src/bls12_377/g2.rs")
    g2c = {}
    for name in ["G2_GENERATOR_X_C0", "G2_GENERATOR_X_C1", "G2_GENERATOR_Y_C0", "G2_GENERATOR_Y_C1"]:
        m = re.search(r"pub const " + name + r"\s*:\s*Fq\s*=\s*field!\(\s*Fq,\s*BigInteger384::new\(\[(.*?)\]\)", g2, re.S)
        g2c[name + "_MONT"] = parse_limb_list(m.group(1))
    m = re.search(r"const WEIERSTRASS_B: Fq2 = field!\(\s*Fq2,(.*?)\);\s*\n", g2, re.S)
    g2c["WEIERSTRASS_B_MONT"] = [parse_limb_list(x) for x in re.findall(r"BigInteger384(?:::new)?\(\[(.*?)\]\)", m.group(1), re.S)]
    consts["g2"] = g2c
    fq2 = read("curves/src/bls12_377/fq2.rs")
    m = re.search(r"const NONRESIDUE: Fq = field!\(\s*Fq,\s*BigInteger(?:384)?(?:::new)?\(\[(.*?)\]\)", fq2, re.S)
    consts["fq2"] = {"NONRESIDUE_MONT": parse_limb_list(m.group(1))}
    with open(os.path.join(OUT, "constants.json"), "w") as f:
        json.dump(consts, f, indent=1)

    base = "algorithms/src/snark/varuna/resources/circuit_0/"
    tv = {"domain": {}, "polynomials": {}}
    for n in "RCK":
        tv["domain"][n] = [str(x) for x in json.loads(read(base + f"domain/{n}.txt"))]
    for n in ["w_lde", "z_lde", "h_0", "g_1", "h_1", "g_a", "g_b", "g_c", "h_2"]:
        tv["polynomials"][n] = [str(x) for x in json.loads(read(base + f"polynomials/{n}.txt"))]
    wit = read(base + "witness.input").strip().splitlines()
    tv["witness"] = [json.loads(l) for l in wit]
    inst = read(base + "instance.input")
    mats = {}
    cur = None
    for line in inst.splitlines():
        line = line.strip()
        if line in ("A", "B", "C"):
            cur = line
            mats[cur] = []
        elif line and cur:
            mats[cur].append([int(x) for x in line.strip(",").split(",")])
    tv["instance"] = mats
    tv["challenges"] = read(base + "challenges.input")
    with open(os.path.join(OUT, "varuna_circuit0.json"), "w") as f:
        json.dump(tv, f, indent=1)

    with open(os.path.join(REF, "parameters/src/mainnet/resources/powers-of-beta-15.usrs"), "rb") as f:
        raw = f.read()
    count = int.from_bytes(raw[:8], "little")
    assert count == 32768 and len(raw) == 8 + 96 * count
    n = 1024
    pts = bytearray(raw[8 : 8 + 96 * n])
    for i in range(n):
        flags = pts[96 * i + 95] & 0xC0
        assert flags == 0, "unexpected SW flags on an SRS point"
    with open(os.path.join(OUT, "srs_g1_1024.bin"), "wb") as f:
        f.write(bytes(pts))
    allpts = raw[8 : 8 + 96 * count]
    assert all((allpts[96 * i + 95] & 0xC0) == 0 for i in range(count)), "unexpected SW flags on an SRS point"
    with open(os.path.join(OUT, "srs_g1_32768.bin"), "wb") as f:
        f.write(allpts)
    with open(os.path.join(REF, "parameters/src/mainnet/resources/beta-h.usrs"), "rb") as f:
        raw = f.read()
    with open(os.path.join(OUT, "beta_h_g2.bin"), "wb") as f:
        f.write(raw)
    print("wrote golden fixtures to", OUT)


if __name__ == "__main__":
    sys.exit(main())
