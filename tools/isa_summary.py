"""Resource summary of the kernels in a hipcc --save-temps .s file: name, VGPRs, SGPRs, LDS bytes,
scratch.  usage: python tools/isa_summary.py file.s [name-filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else "k_"
md = txt[txt.rfind("amdhsa.kernels:"):]
for blk in md.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if flt not in name:
        continue
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(.*", "", dem).replace("void ", "")
    print(f"{dem:40s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>7s} scratch {g('private_segment_fixed_size')}")
