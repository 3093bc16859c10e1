"""A recorded step's HIP graph (PASSION_GRAPH_DOT=file, hipGraphDebugDotPrint): nodes in creation order with their predecessors.
python tools/graph_dot_deps.py graph.dot            -> every node: id, short name, grid, predecessor ids
python tools/graph_dot_deps.py graph.dot ID [depth] -> the ancestors of node ID (breadth first)"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
nodes = {}
for m in re.finditer(r'"graph_0_node_(\d+)"\[[^\]]*?label="\{\s*(\w+)\s*\|(.*?)\}"\];', txt, re.S):
    nid, kind, body = int(m.group(1)), m.group(2), m.group(3)
    name, grid = kind, ""
    k = re.search(r'\{ID \| \d+ \| (\S+?)\\<\\<\\<\((\d+),(\d+),(\d+)\)', body)
    if k:
        name = k.group(1); grid = str(int(k.group(2)) * int(k.group(3)) * int(k.group(4)))
    nodes[nid] = [name, grid]
edges = re.findall(r'"graph_0_node_(\d+)"\s*->\s*"graph_0_node_(\d+)"', txt)
pred = {}
for a, b in edges:
    pred.setdefault(int(b), []).append(int(a))
names = sorted({v[0] for v in nodes.values() if v[0].startswith("_Z")})
dem = {}
if names:
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    dem = dict(zip(names, out))
def short(n):
    d = dem.get(n, n)
    d = re.sub(r"\(anonymous namespace\)::|void |pconv::|at::native::", "", d)
    return re.sub(r"\(.*", "", d)[:60]
if len(sys.argv) > 2:
    start, depth = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 6
    seen, front = {start}, [start]
    for lv in range(depth):
        nxt = []
        for n in front:
            for p in pred.get(n, []):
                print(f"{'  ' * lv}{n} <- {p} {short(nodes[p][0])} wg{nodes[p][1]}")
                if p not in seen:
                    seen.add(p); nxt.append(p)
        front = nxt
else:
    for nid in sorted(nodes):
        print(f"{nid:4d} {short(nodes[nid][0]):60s} wg{nodes[nid][1]:>6s} <- {sorted(pred.get(nid, []))}")
