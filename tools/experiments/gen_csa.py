#!/usr/bin/env python3
"""Straight-line carry-save adder network that counts N one-bit inputs per bit position (bit-sliced population count):
    python3 tools/experiments/gen_csa.py N NAME > csa_N.inc
emits a macro NAME(X, B0..Bk) where X(i) yields input word i (evaluated once each, in order) and B0..Bk receive the binary
digits of the per-bit count.  A full adder is two v_bitop3 (sum = a^b^c: 0x96, carry = majority: 0xE8)."""
import sys

n, name = int(sys.argv[1]), sys.argv[2]
levels = [[] for _ in range(8)]
lines, fresh = [], [0]


def var():
    fresh[0] += 1
    return "t%d" % fresh[0]


def settle(final=False):
    for lv in range(7):
        q = levels[lv]
        while len(q) >= 3:
            a, b, c = q.pop(0), q.pop(0), q.pop(0)
            s, cy = var(), var()
            lines.append("const uint32_t %s = __builtin_amdgcn_bitop3_b32(%s, %s, %s, 0x96), %s = __builtin_amdgcn_bitop3_b32(%s, %s, %s, 0xE8);" % (s, a, b, c, cy, a, b, c))
            q.append(s)
            levels[lv + 1].append(cy)
        if final and len(q) == 2:
            a, b = q.pop(0), q.pop(0)
            s, cy = var(), var()
            lines.append("const uint32_t %s = %s ^ %s, %s = %s & %s;" % (s, a, b, cy, a, b))
            q.append(s)
            levels[lv + 1].append(cy)


for i in range(n):
    v = var()
    lines.append("const uint32_t %s = X(%d);" % (v, i))
    levels[0].append(v)
    settle()
for _ in range(8):
    settle(final=True)
digits = n.bit_length()
out = ["#define %s(X, %s) \\" % (name, ", ".join("B%d" % d for d in range(digits)))]
for l in lines:
    out.append("    " + l + " \\")
for d in range(digits):
    out.append("    B%d = %s; \\" % (d, levels[d][0] if levels[d] else "0u"))
out.append("    (void)0")
print("\n".join(out))
sys.stderr.write("%d inputs: %d statements, %d digits\n" % (n, len(lines), digits))
