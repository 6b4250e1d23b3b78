"""tools only: per-workgroup hop timeline of the dense triangular solves (k_trsv_tiles) -- a library built with
-DDLG_TRSV_PROFILE (tools/variant_lib.sh build dense_diag.hip -DDLG_TRSV_PROFILE; cp tools/micro/libvar.so tools/micro/libtrsvprof.so)."""
import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo') else os.getcwd())
sys.argv = ["bench.py", "--no-cpu-baseline", "--workload", "dense-50k", "--steps", "5", "--warmup", "2"]
from libdogleg_amd import capi
capi.LIB_PATH = os.path.abspath(os.environ.get("DLG_PROF_LIB", "tools/micro/libvar.so"))
import bench, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
capi.lib().dlg_trsv_profile_dump(32)
