"""pita_quantile_clamp at 65 536 values: chunks of 512 (rank counting) and one chunk (radix select)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pita_amd
L, sp = pita_amd._lib.lib(), pita_amd._lib.stream_ptr()
a = torch.randn(65536, device="cuda")
for chunk in (512, 1024, 2048, 65536):
    for _ in range(5): L.pita_quantile_clamp(a.data_ptr(), 65536, chunk, 0.9, sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): L.pita_quantile_clamp(a.data_ptr(), 65536, chunk, 0.9, sp)
    e1.record(); torch.cuda.synchronize()
    print(f"chunk {chunk}: {e0.elapsed_time(e1) * 20:.1f} us per call")
