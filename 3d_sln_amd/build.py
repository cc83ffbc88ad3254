"""Build recipe for libsln_hip.so: hipcc, gfx950 only, in-tree (so the .so travels with gpurun snapshots).

    python -m 3d_sln_amd.build          (or __graft_entry__.build())
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsln_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# kernarg preload: the first 14 dwords of a kernel's leading scalar / pointer arguments arrive in SGPRs with the wavefront instead of
# through s_load (~400 clocks of scalar-cache miss at every kernel start, tools/lab/kernarg_lat.hip): edge kernels -1.5 %, the step
# -0.3 %.  (Struct arguments are passed by reference and are not preloaded; hipcc emits the fallback loads for older firmware itself.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value",
         "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "-I" + os.path.join(os.path.dirname(HERE), "include")] + os.environ.get("SLN_HIPCC_EXTRA", "").split()   # lab: -DSLN_NT_SCHED=0 ...
# raster kernels: bit-exact agreement with the CPU restatement needs contraction off (see raster.hip)
# placement: the torch expression it replaces rounds after every elementwise op
PER_FILE = {"raster.hip": ["-ffp-contract=off"], "graph_build.hip": ["-ffp-contract=off"], "placement.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(src, flags):
    h = hashlib.sha1()
    h.update(" ".join(flags).encode())
    for f in [src] + sorted(os.path.join(CSRC, x) for x in os.listdir(CSRC) if x.endswith(".h")) + \
            [os.path.join(os.path.dirname(HERE), "include", "sln_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(name):
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJ, name + ".o")
    flags = FLAGS + PER_FILE.get(name, [])
    stamp = _stamp(src, flags)
    sfile = obj + ".stamp"
    if os.path.exists(obj) and os.path.exists(sfile) and open(sfile).read() == stamp:
        return obj, False
    r = subprocess.run([HIPCC] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (name, r.stderr[-4000:]))
    with open(sfile, "w") as fh:
        fh.write(stamp)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    names = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(names))) as ex:
        res = list(ex.map(_compile, names))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(OUT):
        # -no-hip-rt: do NOT link /opt/rocm's libamdhip64.so.7.  The process already holds the HIP
        # runtime PyTorch ships (SONAME libamdhip64.so); a second runtime in the same process cannot
        # share streams/devices with it.  _lib.py puts torch's runtime into the global symbol scope
        # before loading this library, which then binds to it.
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt"] + objs + ["-o", OUT],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    if verbose:
        print("built", OUT, "(%d objects, %d recompiled)" % (len(objs), sum(ch for _, ch in res)))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
