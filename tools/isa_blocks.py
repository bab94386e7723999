"""Basic blocks of one kernel in `hipcc -S` output, in program order: label, instructions, VALU / SALU / LDS / VMEM counts and
the branch targets -- to read a loop's per-wave instruction count off the assembly (a wave issues ~1 instruction per 5 cycles).
    python tools/isa_blocks.py march.s <mangled kernel name> [min_instructions]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 1
start = next(i for i, l in enumerate(src) if l.startswith(want + ":"))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
blocks, cur, name = [], [], "entry"
for l in src[start + 1:end + 1]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); cur = []; name = m.group(1); continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    cur.append(t)
blocks.append((name, cur))
for name, ins in blocks:
    if len(ins) < minn: continue
    ops = [i.split()[0] for i in ins]
    valu = sum(o.startswith("v_") for o in ops); salu = sum(o.startswith("s_") and not o.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_barrier", "s_nop")) for o in ops)
    lds = sum(o.startswith("ds_") for o in ops); vmem = sum(o.startswith(("global_", "buffer_", "flat_", "scratch_")) for o in ops)
    rl = sum(o in ("v_readlane_b32", "v_writelane_b32") for o in ops)
    br = [i.split()[-1] for i in ins if i.startswith(("s_cbranch", "s_branch"))]
    bar = sum(o == "s_barrier" for o in ops)
    print("%-12s %4d  valu %4d (lane-spill %3d) salu %3d lds %3d vmem %2d %s-> %s" % (name, len(ins), valu, rl, salu, lds, vmem, "BARRIER " if bar else "", ",".join(br)))
