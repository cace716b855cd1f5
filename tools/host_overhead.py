import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sageattention_amd as sa
q = torch.randn(1, 2, 128, 64, dtype=torch.float16, device="cuda")
for name, fn in (("fp16", sa.sageattn_qk_int8_pv_fp16_cuda), ("fp8", sa.sageattn_qk_int8_pv_fp8_cuda), ("sageattn", sa.sageattn)):
    for _ in range(50): fn(q, q, q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): fn(q, q, q)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(name, "host us/call", (t1 - t0) / 2000 * 1e6, "incl. drain", (t2 - t0) / 2000 * 1e6)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(500): sa.sageattn_qk_int8_pv_fp16_cuda(q, q, q)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
