"""Stage-1 time at 150 classes with the ADE goal pools against a table without goal classes."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_search_amd as ssa
from tests import scenes

H, W = 480, 640
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

dev = torch.device("cuda:0")
E = 16
for C in (150, 40):
    logits = torch.randn((E, C, H, W), device=dev) * 3
    for name, lut in (("pools", scenes.schema_lut(C)), ("no goals", np.full(C, 2))):
        eng = ssa.MapEngine(E, (H, W), depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
                            prob_aggregate="height_map_log_odds", seg_model_n_classes=C, max_map_hw=(480, 480), device=dev)
        eng.set_schema(lut)
        ms = timeit(lambda: eng.stage1(logits, 2.5))
        print(f"C={C} stage1 {name:9s}: {ms*1e3:8.1f} us  {E*C*H*W*4/1e9/ms*1e3:7.1f} GB/s")
        eng.close()
    del logits
