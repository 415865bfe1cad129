for L in variants/libscvx_k0base.so successiveconvexification_amd/libscvx_hip.so $EXTRA; do echo "== $L"; timeout -k 10 300 python -c "
import sys, os, runpy
sys.path.insert(0, '.')
from successiveconvexification_amd import _lib
_lib.LIB_PATH = os.path.abspath('$L')
sys.argv = ['tools/threedof_bench.py', '--B', '1,256,2048,8192']
runpy.run_path('tools/threedof_bench.py', run_name='__main__')
" 2>&1 | grep -v "^$" || exit 1; done
