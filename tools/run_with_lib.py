import sys, os, runpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from successiveconvexification_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
