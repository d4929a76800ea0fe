#!/usr/bin/env python3
"""Checks the output of tools/chain_row.hip with big integers: row 0 is squared, rows 1..3 multiplied by row 0, each step divided by 2^261 (mod r)."""
import sys
r = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
Rinv = pow(1 << 261, -1, r)
ins, outs, iters = {}, {}, None
for line in open(sys.argv[1]):
    w = line.split()
    if not w: continue
    if w[0] == "ITERS": iters = int(w[1])
    if w[0] in ("IN", "OUT"):
        v = sum(int(x, 16) << (29 * i) for i, x in enumerate(w[2:2 + 16]))
        (ins if w[0] == "IN" else outs)[int(w[1])] = (v, [int(x, 16) for x in w[2:18]])
    if w[0] == "PROBE": print(line.strip())
vals = [ins[i][0] for i in range(4)]
for _ in range(iters):
    x = vals[0]
    vals = [x * v * Rinv % r for v in vals]
ok = True
for i in range(4):
    got, limbs = outs[i]
    good = got % r == vals[i] and all(l == 0 for l in limbs[9:]) and max(limbs[:8]) < (1 << 30) + 16 and got < (1 << 256)
    ok &= good
    print("row", i, "OK" if good else "MISMATCH", "bits", got.bit_length())
print("CHAIN_ROW_CHECK", "PASS" if ok else "FAIL")
sys.exit(0 if ok else 1)
