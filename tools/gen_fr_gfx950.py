#!/usr/bin/env python3
"""Generates stark_mlwe_amd/csrc/fr_gfx950.inc: the gfx950 device implementations of the Montgomery
product and of the fused multiply-accumulate ("wide") primitives, as inline-asm blocks of
    v_mad_u64_u32  acc.ml, vcc, x, y, acc.ml      (32x32+64 MAC; ~4 cycles per wave64 on CDNA4)
    v_addc_co_u32  acc.hi, vcc, 0, acc.hi, vcc    (carry count)
One asm statement per column / per operand row keeps the compiler from putting a hazard `s_nop` after
every MAC pair (it adds one after each asm statement) and stays under the 30-operand limit.

Run:  python tools/gen_fr_gfx950.py   (the output is committed; it only depends on the two moduli)
"""
import os

FIELDS = {
    "PallasFr": 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
    "Bls12381Fr": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
}
MAC = "v_mad_u64_u32 %[{ml}], vcc, %[{x}], %[{y}], %[{ml}]\\n\\tv_addc_co_u32 %[{hi}], vcc, 0, %[{hi}], vcc\\n\\t"


def limbs(p):
    return [(p >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def asm_block(macs, ml="ml", hi="hi", indent="    "):
    """macs: list of (xname, xexpr, xcons, yname, yexpr, ycons).  Returns one asm statement."""
    if not macs:
        return ""
    text = "".join(MAC.format(ml="ml", hi="hi", x=m[0], y=m[3]) for m in macs)
    ins = {}
    for m in macs:
        ins[m[0]] = (m[1], m[2]); ins[m[3]] = (m[4], m[5])
    in_ops = ", ".join(f'[{n}] "{c}"({e})' for n, (e, c) in ins.items())
    assert len(ins) + 2 <= 30, len(ins)
    return f'{indent}asm("{text}" : [ml] "+v"({ml}), [hi] "+v"({hi}) : {in_ops} : "vcc");\n'


def split_blocks(macs, ml="ml", hi="hi", indent="    "):
    out, cur, names = "", [], set()
    for m in macs:
        nn = names | {m[0], m[3]}
        if len(nn) + 2 > 30:
            out += asm_block(cur, ml, hi, indent); cur, names = [], set(); nn = {m[0], m[3]}
        cur.append(m); names = nn
    return out + asm_block(cur, ml, hi, indent)


def gen_mul(name, p):
    P = limbs(p)
    assert P[0] == 1 and (-pow(p, -1, 1 << 32)) % (1 << 32) == 0xFFFFFFFF
    s = f"// ---- {name}: Montgomery product, finely integrated product scanning ------------------------------\n"
    s += f"template <> __device__ __forceinline__ fr_t fr_mul_dev<{name}>(const fr_t& a, const fr_t& b) {{\n"
    s += "    uint64_t ml = 0; uint32_t hi = 0, lo; uint32_t m0, m1, m2, m3, m4, m5, m6, m7; uint32_t t[9];\n"
    for c in range(15):
        macs = []
        for i in range(max(0, c - 7), min(c, 7) + 1):
            macs.append((f"a{i}", f"a.v[{i}]", "v", f"b{c - i}", f"b.v[{c - i}]", "v"))
        for k in range(max(0, c - 7), min(c - 1, 7) + 1):
            j = c - k
            if 1 <= j <= 7 and P[j] != 0:
                macs.append((f"m{k}", f"m{k}", "v", f"p{j}", f"0x{P[j]:08x}u", "s"))
        s += f"    // column {c}\n" + split_blocks(macs)
        if c < 8:
            s += f"    lo = (uint32_t)ml; m{c} = 0u - lo; ml = ((ml >> 32) | ((uint64_t)hi << 32)) + (lo != 0u ? 1u : 0u); hi = 0;\n"
        else:
            s += f"    t[{c - 8}] = (uint32_t)ml; ml = (ml >> 32) | ((uint64_t)hi << 32); hi = 0;\n"
    s += "    t[7] = (uint32_t)ml; t[8] = (uint32_t)(ml >> 32);\n"
    s += f"    fr_cond_sub<{name}>(t, t[8]);\n"
    s += "    fr_t z; for (int i = 0; i < 8; ++i) z.v[i] = t[i]; return z;\n}\n\n"
    return s


def gen_wide(name, p):
    """fr_wide: 15 independent 96-bit column accumulators; wide_mac adds the 64 partial products of a*b;
    wide_reduce folds in the Montgomery terms and returns the reduced element."""
    P = limbs(p)
    s = f"// ---- {name}: Montgomery reduction of a sum of products -----------------------------------------------\n"
    s += f"template <> __device__ __forceinline__ fr_t fr_wide_reduce_dev<{name}>(const fr_wide& w) {{\n"
    s += "    uint64_t ml = 0; uint32_t hi = 0, lo; uint32_t m0, m1, m2, m3, m4, m5, m6, m7; uint32_t t[9];\n"
    for c in range(15):
        s += f"    // column {c}: r += W[{c}]\n"
        s += f"    {{ unsigned long long x; const bool cy = __builtin_uaddll_overflow(ml, w.ml[{c}], &x); hi += w.hi[{c}] + (cy ? 1u : 0u); ml = x; }}\n"
        macs = []
        for k in range(max(0, c - 7), min(c - 1, 7) + 1):
            j = c - k
            if 1 <= j <= 7 and P[j] != 0:
                macs.append((f"m{k}", f"m{k}", "v", f"p{j}", f"0x{P[j]:08x}u", "s"))
        s += split_blocks(macs)
        if c < 8:
            s += f"    lo = (uint32_t)ml; m{c} = 0u - lo; ml = ((ml >> 32) | ((uint64_t)hi << 32)) + (lo != 0u ? 1u : 0u); hi = 0;\n"
        else:
            s += f"    t[{c - 8}] = (uint32_t)ml; ml = (ml >> 32) | ((uint64_t)hi << 32); hi = 0;\n"
    s += "    t[7] = (uint32_t)ml; t[8] = (uint32_t)(ml >> 32);\n"
    s += "    // the sum of up to 32 products can exceed 2p: subtract p while needed (value < 2^32 * p by construction)\n"
    s += f"    fr_reduce_wide_tail<{name}>(t);\n"
    s += "    fr_t z; for (int i = 0; i < 8; ++i) z.v[i] = t[i]; return z;\n}\n\n"
    return s


def gen_wide_mac():
    s = "// ---- field-independent: W += a * b (64 partial products into 15 independent column accumulators) ---------\n"
    s += "__device__ __forceinline__ void fr_wide_mac(fr_wide& w, const fr_t& a, const fr_t& b) {\n"
    for i in range(8):
        text = ""
        for j in range(8):
            text += f"v_mad_u64_u32 %[ml{j}], vcc, %[a], %[b{j}], %[ml{j}]\\n\\tv_addc_co_u32 %[hi{j}], vcc, 0, %[hi{j}], vcc\\n\\t"
        outs = ", ".join(f'[ml{j}] "+v"(w.ml[{i + j}]), [hi{j}] "+v"(w.hi[{i + j}])' for j in range(8))
        ins = f'[a] "v"(a.v[{i}]), ' + ", ".join(f'[b{j}] "v"(b.v[{j}])' for j in range(8))
        s += f'    asm("{text}" : {outs} : {ins} : "vcc");\n'
    s += "}\n\n"
    return s


def main():
    out = "// GENERATED by tools/gen_fr_gfx950.py — do not edit.  gfx950 device code only (included by fr.hpp).\n\n"
    out += gen_wide_mac()
    for name, p in FIELDS.items():
        out += gen_mul(name, p)
        out += gen_wide(name, p)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stark_mlwe_amd", "csrc", "fr_gfx950.inc")
    open(path, "w").write(out)
    print("wrote", path, len(out), "bytes")


if __name__ == "__main__":
    main()
