import sys, time, torch
sys.path.insert(0, "/root/repo")
lib = sys.argv[1]
try:
    torch.backends.cuda.preferred_blas_library(lib)
except Exception as e:
    print("cannot set", lib, e)
dev = torch.device("cuda:0")
def bench(a, b, n=20):
    for _ in range(3): torch.matmul(a, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): torch.matmul(a, b)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
g = torch.Generator().manual_seed(0)
cases = {"ffn1 [69120x256]x[256x1024]": (torch.randn(2, 34560, 256, device=dev), torch.randn(256, 1024, device=dev)),
         "ffn2 [69120x1024]x[1024x128]": (torch.randn(2, 34560, 1024, device=dev), torch.randn(1024, 128, device=dev)),
         "proj [69120x128]x[128x128]": (torch.randn(2, 34560, 128, device=dev), torch.randn(128, 128, device=dev)),
         "qk win [128,540,128]x[128,128,540]": (torch.randn(128, 540, 128, device=dev), torch.randn(128, 128, 540, device=dev)),
         "pv win [128,540,540]x[128,540,128]": (torch.randn(128, 540, 540, device=dev), torch.randn(128, 540, 128, device=dev)),
         "qk 1/8 [8,2160,128]x[8,128,2160]": (torch.randn(8, 2160, 128, device=dev), torch.randn(8, 128, 2160, device=dev))}
for k, (a, b) in cases.items():
    us = bench(a, b)
    fl = 2.0 * a.shape[-2] * a.shape[-1] * b.shape[-1] * (a.shape[0] if a.dim() == 3 else 1)
    print(f"{lib:10s} {k:40s} {us:8.1f} us {fl / us / 1e6:6.1f} TF/s")
