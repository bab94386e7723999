"""Opcode histogram of a kernel's largest basic blocks in `hipcc -S` output:
    python tools/isa_block_histogram.py march.s <mangled kernel name> <hypotheses per lane and step>
"""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
# find function
start = next(i for i, l in enumerate(src) if l.startswith(want) and l.rstrip().endswith(":") or (l.startswith(want) and ":" in l and "@" in l))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
blocks, cur, name = [], [], "entry"
for l in src[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); cur = []; name = m.group(1); continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    cur.append(t.split()[0])
blocks.append((name, cur))
big = sorted(blocks, key=lambda b: -len(b[1]))[:4]
for name, ins in big:
    valu = sum(1 for i in ins if i.startswith("v_"))
    print(name, len(ins), "instructions", valu, "VALU", "= %.2f per hypothesis" % (valu / float(sys.argv[3])))
    c = collections.Counter(ins)
    print("   ", ", ".join("%s %d" % kv for kv in c.most_common(14)))
