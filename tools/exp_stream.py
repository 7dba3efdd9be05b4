"""Micro-experiments on the streaming kernel (run on the GPU box)."""
import sys, os, time
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

def main():
    dev = torch.device("cuda:0")
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for C in (40, 150, 48):
        eng = ssa.MapEngine(E, (H, W), depth_hfov=59.0, depth_range=(0.5, 5.0), probablistic_mapping=True,
                            prob_aggregate="height_map_log_odds", seg_model_n_classes=C, max_map_hw=(480, 480), device=dev)
        eng.set_schema(scenes.schema_lut(C) if C in (40,150) else np.full(C, 2))
        logits = torch.randn((E, C, H, W), device=dev) * 3
        ms = timeit(lambda: eng.stage1(logits, 3.2))
        gb = E * C * H * W * 4 / 1e9
        print(f"stage1 only  C={C:4d} E={E}: {ms*1e3:8.1f} us  {gb/ms*1e3:7.1f} GB/s (reads only)")
        ms = timeit(lambda: eng.stage1(logits, 3.2, want_probs=True))
        print(f"stage1+probs C={C:4d} E={E}: {ms*1e3:8.1f} us  {2*gb/ms*1e3:7.1f} GB/s (r+w)")
        # plain torch copy as the streaming reference on this box
        dst = torch.empty_like(logits)
        ms = timeit(lambda: dst.copy_(logits))
        print(f"torch copy   C={C:4d} E={E}: {ms*1e3:8.1f} us  {2*gb/ms*1e3:7.1f} GB/s (r+w)")
        ms = timeit(lambda: torch.softmax(logits, dim=1))
        print(f"torch softmax C={C:4d} E={E}: {ms*1e3:8.1f} us")
        eng.close(); del logits, dst
main()
