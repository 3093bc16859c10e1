"""instruction classes inside the loops of one kernel of a -save-temps .s file:  python tools/loop_ops.py file.s KERNEL_SUBSTRING"""
import re, sys, collections
txt = open(sys.argv[1]).read()
key = sys.argv[2]
names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", txt, re.M) if key in m.group(1)]
for name in names:
    i = txt.index(name + ":"); j = txt.index(".Lfunc_end", i)
    body = txt[i:j].split("\n")
    labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = {}
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 1 << 30) < k:
            t = labels[m.group(1)]
            loops[t] = max(loops.get(t, 0), k)
    print(name, len(body), "lines")
    for t, k in sorted(loops.items()):
        if k - t < 100: continue
        c = collections.Counter()
        for l in body[t:k + 1]:
            l = l.strip()
            if not l or l[0] in ".;": continue
            op = l.split()[0]
            kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "waitcnt" if op.startswith("s_waitcnt") else
                    "barrier" if op.startswith("s_barrier") else "branch" if "branch" in op else "salu" if op.startswith("s_") else
                    "ds_read" if op.startswith("ds_read") else "ds_write" if op.startswith("ds_") else
                    "gload" if "load" in op else "gstore" if "store" in op else op)
            c[kind] += 1
        print(f"  loop {t}..{k}:", dict(c))
