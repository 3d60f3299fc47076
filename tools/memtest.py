import torch, time
x = torch.empty(32768*1040, dtype=torch.uint8, device='cuda')
y = torch.empty_like(x)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
print("zero_ 34MB us", t(lambda: x.zero_()))
print("copy_ 34MB us", t(lambda: y.copy_(x)))
z = torch.empty(16, dtype=torch.uint8, device='cuda')
print("zero_ 16B us", t(lambda: z.zero_()))
big = torch.empty(8*32768*1040, dtype=torch.uint8, device='cuda')
print("zero_ 272MB us", t(lambda: big.zero_(), 500))
