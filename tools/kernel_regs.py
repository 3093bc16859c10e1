"""registers / LDS / spills of every kernel in a -save-temps .s file (amdhsa metadata):  python tools/kernel_regs.py file.s [filter]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(.*", "", name).replace("(anonymous namespace)::", "").replace("void ", "")
    if flt in name:
        agpr = blk.split("\n")[0].strip()
        print(f"{name:60s} vgpr {g('vgpr_count'):>4s} agpr {agpr:>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s} spill {g('vgpr_spill_count')}")
