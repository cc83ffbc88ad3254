"""Register / LDS / spill counts per kernel from the metadata of a device-only assembly file.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude --offload-device-only -S x.hip -o x.s
   python tools/isa_regs.py x.s [name filter]"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    # metadata entries: a YAML list under amdhsa.kernels; split at '  - .agpr_count' (first key, alphabetical)
    for blk in re.split(r"\n  - ", txt[txt.index("amdhsa.kernels:"):])[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            pass
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        if flt and flt not in name:
            continue
        print("%-70s vgpr %4s agpr %4s sgpr %4s lds %6s spill %s scratch %s" % (
            name[:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("vgpr_spill_count"),
            g("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
