import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if l.startswith(pat) and ":" in l and not l.startswith("\t"):
        start = i; break
assert start is not None, "function not found"
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
print("function lines", len(body))
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
# back edges
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i))
def classify(seg):
    c = collections.Counter()
    for l in seg:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith("."): continue
        op = l.split()[0]
        if op.startswith("v_fma_f64") or op.startswith("v_mul_f64") or op.startswith("v_add_f64"): c["f64 arith"] += 1
        elif op.startswith("v_fma") or op.startswith("v_mul_f32") or op.startswith("v_add_f32") or op.startswith("v_sub_f32") or op.startswith("v_mac") or op.startswith("v_pk_"): c["f32 arith"] += 1
        elif op.startswith("ds_"): c[op.split("_")[0] + "_" + op.split("_")[1]] += 1
        elif op.startswith("scratch_") or (op.startswith("buffer_") and "offen" in l or "s[0:3]" in l and op.startswith("buffer_")): c["scratch " + op.split("_")[1]] += 1
        elif op.startswith("v_readlane") or op.startswith("v_writelane"): c[op[:11]] += 1
        elif op.startswith("v_accvgpr"): c["accvgpr mov"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"): c["s_load"] += 1
        elif op.startswith("global_") or op.startswith("flat_"): c[op.split("_")[0] + " " + op.split("_")[1]] += 1
        elif op.startswith("s_waitcnt"): c["s_waitcnt"] += 1
        elif op.startswith("v_"): c["other valu"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        else: c["other"] += 1
    return c
for a, b in sorted(loops, key=lambda x: x[1] - x[0]):
    seg = body[a:b + 1]
    n = sum(1 for l in seg if l.strip() and not l.strip().startswith((";", ".")))
    print(f"loop lines {a}-{b} instrs {n}:", dict(classify(seg)))
