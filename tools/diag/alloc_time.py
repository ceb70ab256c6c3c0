import time, torch
def t(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    r = (time.perf_counter() - t0) / n * 1e6; torch.cuda.synchronize(); return r
z = torch.zeros(1024, device="cuda:0")
for shape in ((64, 82), (64, 84), (256, 82), (256, 1276), (64, 81), (3, 82)):
    def f1():
        a = torch.empty(*shape, dtype=torch.float32, device="cuda:0")
    def f2():
        a = torch.empty(*shape, dtype=torch.float32, device="cuda:0"); b = torch.empty_like(a); return a, b
    def f3():
        a = torch.empty(*shape, dtype=torch.float32, device="cuda:0"); a.zero_(); return a
    dev = torch.device("cuda:0")
    def f4():
        a = torch.empty(*shape, dtype=torch.float32, device=dev); return a
    def f5():
        a = torch.empty(shape, dtype=torch.float32, device=dev); return a
    print(shape, "empty %.1f us, two %.1f us, empty+kernel %.1f us, device obj %.1f, tuple %.1f" % (t(f1), t(f2), t(f3), t(f4), t(f5)), flush=True)
