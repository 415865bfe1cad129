"""Identity of the library a measurement was taken on: sha256 over the sources libscvx_hip.so is built from (csrc/*, include/scvx.h),
first 16 hex digits, plus the git commit when the tree has one.  `python tools/lib_hash.py` prints it; bench.py and
tools/final_profiles.sh import it so a calibration file and the running tree can be told apart."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_source_hash():
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "successiveconvexification_amd", "csrc")
    for f in sorted(os.listdir(cs)) + ["../../include/scvx.h"]:
        p = os.path.normpath(os.path.join(cs, f))
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def lib_build_switches(path=None):
    """the compile-time switches the LOADED library reports (scvx_debug_build_switches): two builds of the same sources with different -D
    flags differ here (the binary itself is not reproducible byte for byte: its hash is no identity)"""
    import ctypes
    path = path or os.path.join(ROOT, "successiveconvexification_amd", "libscvx_hip.so")
    try:
        L = ctypes.CDLL(path)
        buf = ctypes.create_string_buffer(512)
        L.scvx_debug_build_switches(buf, 512)
        return buf.value.decode()
    except Exception:
        return None


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


if __name__ == "__main__":
    print(lib_source_hash(), git_head(), lib_build_switches())
