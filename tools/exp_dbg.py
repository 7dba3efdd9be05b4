"""Phase timestamps of the winner kernel (temporary instrumentation; 100 MHz wall clock)."""
import os, sys, ctypes, subprocess
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
import bench
bench.main()
from semantic_search_amd import _capi
lib = _capi.load() if hasattr(_capi, "load") else _capi.lib
buf = (ctypes.c_ulonglong * (4096 * 8))()
lib.ssmap_debug_read.argtypes = [ctypes.c_void_p]
print("rc", lib.ssmap_debug_read(buf))
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8)[:256].astype(np.int64)
t0 = a[:, 0].min()
rel = (a[:, :6] - t0) / 100.0   # us
print("per-phase mean start (us since first block start):", rel.mean(0).round(2))
print("per-phase max:", rel.max(0).round(2))
d = np.diff(a[:, :6], axis=1) / 100.0
print("phase durations mean:", d.mean(0).round(2), "max:", d.max(0).round(2))
