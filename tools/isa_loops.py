"""Instruction mix of the MFMA loops of every kernel in a device-only assembly file (see tools/isa_regs.py for how to produce it):
scalar loads, full waits and scratch accesses inside a K loop are what to look for.   python tools/isa_loops.py x.s [name filter]"""
import re, subprocess, sys


def main():
    L = open(sys.argv[1]).read().split('\n')
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [i for i, l in enumerate(L) if re.match(r'^_Z\S+:\s*(;.*)?$', l)]
    for st in starts:
        name = L[st].split(':')[0]
        try:
            dn = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dn = name
        dn = re.sub(r"\(anonymous namespace\)::", "", dn).split("(")[0]
        if flt and flt not in dn:
            continue
        en = next((i for i in range(st, len(L)) if 's_endpgm' in L[i]), None)
        if en is None:
            continue
        lines = L[st:en]
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r'^(\.LBB\S+):', l)
            if m:
                labels[m.group(1)] = i
        best = None
        for i, l in enumerate(lines):
            m = re.search(r's_cbranch_\w+ (\.LBB\S+)|s_branch (\.LBB\S+)', l)
            if m:
                t = m.group(1) or m.group(2)
                if t in labels and labels[t] < i:
                    seg = lines[labels[t]:i + 1]
                    nm = sum('v_mfma' in x for x in seg)
                    if nm >= 4 and (best is None or len(seg) < len(best)):
                        best = seg
        if best is None:
            continue
        c = lambda pat: sum(bool(re.search(pat, x)) for x in best)
        print("%-62s loop %4d instr  mfma %3d  valu %3d  s_load %2d  lgkmcnt(0) %2d  vmcnt(0) %2d  scratch %2d  ds %3d  vmem %3d" % (
            dn[:62], len(best), c(r'v_mfma'), sum(x.strip().startswith('v_') and 'mfma' not in x for x in best), c(r's_load'), c(r'lgkmcnt\(0\)'),
            c(r'vmcnt\(0\)'), c(r'scratch_'), c(r'\bds_'), c(r'global_load|global_store|global_atomic|flat_')))


if __name__ == "__main__":
    main()
