"""Build libcslgan_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcslgan_hip.so")
SOURCES = ["clip_kernels.hip", "igemm_kc.hip", "igemm_halo.hip", "igemm_skinny.hip", "igemm_mc.hip", "igemm_wgh.hip", "igemm_bf16.hip", "igemm_x3.hip", "igemm_bf16s.hip", "conv_c3.hip", "linear_k1.hip", "conv1x1.hip", "gram_norm.hip", "pointwise_kernels.hip", "step_kernels.hip"]
HEADERS = ["common.h", "igemm.h", os.path.join("..", "..", "include", "cslgan.h")]


def _newer_than(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the C-ABI shared library.

    Incremental by modification time per object (a source or any header newer than its object recompiles it; the library is
    re-linked when an object is newer).  force=True — or CSLGAN_FORCE_BUILD=1 in the environment — recompiles EVERY source and
    re-links, whatever is on disk: the way to prove on a box that received a prebuilt .so that the tree still builds."""
    force = force or os.environ.get("CSLGAN_FORCE_BUILD", "0") == "1"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    jobs = int(os.environ.get("CSLGAN_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))
    pending = []
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer_than(o, [os.path.join(CSRC, s)] + headers):
            pending.append((s, o))
    while pending or procs:
        while pending and len(procs) < jobs:
            s, o = pending.pop(0)
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
                   "-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        s, pr = procs.pop(0)
        out, _ = pr.communicate()
        if pr.returncode != 0:
            for _, other in procs:
                other.kill()
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode()))
        if verbose and out:
            print(out.decode())
    if not force and not _newer_than(LIB, objs):
        return LIB
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
