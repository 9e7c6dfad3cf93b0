"""instruction histogram of a kernel's basic blocks from `hipcc -S` output: which block is the steady-state tile body and what it is made of.
usage: python tools/isa_hist.py file.s <substring of kernel name> [min block size]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 400
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", s):
            blocks.append((name, cur)); cur, name = [], s.split(":")[0]
        continue
    op = s.split()[0]
    cur.append(op)
    if op.startswith("s_cbranch") or op == "s_branch":
        blocks.append((name, cur)); cur, name = [], name + "+"
blocks.append((name, cur))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read:" + op
    if op.startswith("ds_"): return "ds_write:" + op
    if op.startswith("global_load") or op.startswith("buffer_load"): return "vmem_load"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "vmem_store"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("v_pk_"): return "v_pk:" + op
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_"): return "salu"
    return op
tot = collections.Counter()
for name, b in blocks:
    for op in b: tot[cls(op)] += 1
print("kernel total:", sum(tot.values()), dict(tot.most_common(12)))
for name, b in blocks:
    if len(b) < minsz: continue
    c = collections.Counter(cls(op) for op in b)
    print(f"\nblock {name}: {len(b)} instructions")
    for k, v in c.most_common(): print(f"   {k:28s} {v}")
    vc = collections.Counter(op for op in b if cls(op) == "valu")
    print("   valu by opcode:", ", ".join(f"{k} {v}" for k, v in vc.most_common(30)))
