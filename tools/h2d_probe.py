"""developer helper: host-to-device copy rate of pinned memory on this box (what the upload of a picture's records can reach)"""
import time, torch
for mb in (1, 4, 12, 48):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): d.copy_(h, non_blocking=True)
        s.synchronize(); t0 = time.perf_counter()
        for _ in range(20): d.copy_(h, non_blocking=True)
        s.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("H2D %2d MB pinned: %.3f ms  %.1f GB/s" % (mb, dt * 1e3, mb / 1024 / dt))
