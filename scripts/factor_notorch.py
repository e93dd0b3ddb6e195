"""Development aid: factor a generated problem through the C-ABI alone (no torch: usable under the host sanitizer build)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cholesky_amd._lib import load, check  # noqa: E402
from cholesky_amd.plan import Problem

dims = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "40,40,40,6,64").split(",")]
L = load()
plan = Problem(*dims).plan()
h = C.c_void_p()
check(L.cholamd_device_create(plan.h, 0, C.byref(h)), "create")
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    check(L.cholamd_device_set_option(h, k.encode(), int(v)), "option")
a = C.c_void_p()
check(L.cholamd_device_alloc(h, plan.arena_doubles, C.byref(a)), "alloc")
for it in range(2):
    check(L.cholamd_device_fill(h, a, None), "fill")
    check(L.cholamd_factor(h, a, None), "factor")
    check(L.cholamd_device_sync(h, None), "sync")
    sep = C.c_int(0)
    print("factor", it, "info", L.cholamd_factor_info(h, C.byref(sep)), flush=True)
