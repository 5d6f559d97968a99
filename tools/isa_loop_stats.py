"""Instruction mix of the MFMA loops of a kernel in a hipcc -S listing:
python tools/isa_loop_stats.py listing.s kernel_name_substring"""
import re, sys
L = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
st = [i for i, l in enumerate(L) if key in l and l.rstrip().endswith(":") or (key in l and ": ;" in l and l.startswith("_Z"))][0]
en = [i for i, l in enumerate(L) if i > st and l.startswith(".Lfunc_end")][0]
blocks, cur, name = [], [], "entry"
for l in L[st:en]:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append((name, cur)); name = l.strip(); cur = []
    else:
        cur.append(l)
blocks.append((name, cur))
for name, b in blocks:
    ins = [x.strip() for x in b if x.strip() and not x.strip().startswith((";", "."))]
    n = sum("v_mfma" in x for x in ins)
    if n >= 8:
        ops = {}
        for x in ins:
            ops[x.split()[0]] = ops.get(x.split()[0], 0) + 1
        valu = sum(v for k, v in ops.items() if k.startswith("v_") and "mfma" not in k)
        print(name[:14], "mfma", n, "valu", valu, f"({valu / n:.2f} per mfma)", "total", len(ins))
        print("   ", sorted(ops.items(), key=lambda kv: -kv[1])[:10])
