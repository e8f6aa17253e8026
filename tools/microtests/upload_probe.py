"""Where the time of a NumPy -> device upload goes (round 5): pageable .cuda() against a pinned staging block."""
import time, numpy as np, torch
a = np.random.default_rng(0).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
t = torch.from_numpy(a)
def bench(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print("%-44s %.3f ms" % (name, (time.perf_counter() - t0) / n * 1e3), flush=True)
bench("pageable .cuda()", lambda: t.cuda())
h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
bench("torch.empty(pin_memory=True) + drop", lambda: torch.empty(t.shape, dtype=t.dtype, pin_memory=True))
bench("h.copy_(t) into a pinned block", lambda: h.copy_(t))
hn = h.numpy()
bench("np.copyto(pinned view, a)", lambda: np.copyto(hn, a))
b = np.empty_like(a)
bench("np.copyto(pageable, a)", lambda: np.copyto(b, a))
bench("h.to(cuda, non_blocking)", lambda: h.to("cuda", non_blocking=True))
d = torch.empty(t.shape, dtype=t.dtype, device="cuda")
bench("d.copy_(h, non_blocking)", lambda: d.copy_(h, non_blocking=True))
bench("d.copy_(t pageable, non_blocking)", lambda: d.copy_(t, non_blocking=True))
def up():
    hh = torch.empty(t.shape, dtype=t.dtype, pin_memory=True); hh.copy_(t); return hh.to("cuda", non_blocking=True)
bench("upload = empty(pin) + copy_ + to(cuda)", up)
def up2():
    x = up(); y = up(); return x, y
bench("two uploads back to back", up2)
import ctypes
torch.set_num_threads(1)
bench("h.copy_(t) 1 thread", lambda: h.copy_(t))
