#!/usr/bin/env python3
"""Register allocation of every kernel in a built library (no GPU needed): the gfx950 code objects are taken out of
the library's .hip_fatbin section and their AMDGPU metadata notes read with llvm-readelf.

    python tools/kernel_regs.py [libstb_amd/lib/libstb_amd.so] [substring ...]

prints  vgprs  agprs  sgprs  spills  lds  scratch  kernel  for every kernel whose demangled name holds one of the substrings.
`make -C libstb_amd/csrc regs` runs it; tests/test_build_regs.py pins the kernels the measurements depend on (a change of
21 registers in k_grid_hb from the shape of a wait loop cost 10 % for most of round 5: MEASUREMENTS.md section R5.5)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """the gfx950 ELF images inside lib's .hip_fatbin (one offload bundle per translation unit, laid end to end)"""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib], check=True, capture_output=True)
        blob = open(fat, "rb").read()
    out, at = [], blob.find(MAGIC)
    while at >= 0:
        (n,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(MAGIC, at + len(MAGIC))
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return r.stdout.splitlines()


def kernels(lib):
    """{demangled kernel name: {vgpr, agpr, sgpr, spill, lds, scratch}}"""
    found = {}
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk

            def num(key, blk=blk):
                m = re.search(rf"\.{key}:\s+(\d+)", blk)
                return int(m.group(1)) if m else 0
            m = re.search(r"\.name:\s+(\S+)", blk)
            if not m:
                continue
            found[m.group(1)] = dict(vgpr=num("vgpr_count"), agpr=num("agpr_count"), sgpr=num("sgpr_count"),
                                     spill=num("vgpr_spill_count"), lds=num("group_segment_fixed_size"),
                                     scratch=num("private_segment_fixed_size"))
    names = list(found)
    return dict(zip([re.sub(r"^void ", "", d) for d in demangle(names)], [found[n] for n in names]))


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else os.path.join(root, "libstb_amd", "lib", "libstb_amd.so")
    ks = kernels(lib)
    print(f"# {lib}: {len(ks)} kernels\n# vgpr agpr sgpr spill   lds scratch  kernel")
    for name in sorted(ks):
        if args and not any(a in name for a in args):
            continue
        k = ks[name]
        print(f"{k['vgpr']:6d} {k['agpr']:4d} {k['sgpr']:4d} {k['spill']:5d} {k['lds']:6d} {k['scratch']:6d}  {name[:150]}")


if __name__ == "__main__":
    main()
