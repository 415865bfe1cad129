"""Build recipe for libscvx_hip.so (hipcc, gfx950 only, in-tree so the .so travels with gpurun)."""
import fcntl
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libscvx_hip.so")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "scvx.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    # Several ranks of one job may find the library stale at the same moment: one of them builds, the others wait on the lock and
    # then find it fresh.  Objects go to a directory of this build's own, the link to a temporary name that is renamed onto LIB in
    # one step -- a process that has the old library mapped keeps its (unlinked) copy.
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    with open(os.path.join(ROOT, "build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not stale():
            return LIB
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
        if verbose:
            flags.insert(0, "-Rpass-analysis=kernel-resource-usage")
        objdir = tempfile.mkdtemp(prefix="obj.", dir=os.path.join(ROOT, "build"))
        try:
            # one hipcc per translation unit, side by side (the conic solver's file alone takes ~70 s), then one link
            jobs = []
            for src in sources():
                obj = os.path.join(objdir, os.path.basename(src) + ".o")
                jobs.append((src, obj, subprocess.Popen([hipcc] + flags + ["-c", src, "-o", obj])))
            failed = [src for src, _, pr in jobs if pr.wait() != 0]
            if failed:
                raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
            tmp = os.path.join(objdir, "libscvx_hip.so")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [obj for _, obj, _ in jobs])
            os.replace(tmp, LIB)
            # the objects of THIS build, and only these, for tools/build_variant.sh (A/B variants relink against them)
            keep = os.path.join(ROOT, "build", "obj")
            shutil.rmtree(keep, ignore_errors=True)
            os.makedirs(keep)
            for _, obj, _ in jobs:
                os.replace(obj, os.path.join(keep, os.path.basename(obj)))
        finally:
            shutil.rmtree(objdir, ignore_errors=True)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
