"""lab: what an idle device synchronise costs in this process (it sits inside every timed region), by number of live streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gr-doa_amd", "python")]
import torch
def idle_sync(n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
x = torch.zeros(1024, device="cuda"); torch.cuda.synchronize()
print(f"torch only: idle device sync {idle_sync():.2f} us")
import doa
pipe = doa.music_pipeline(4, 1024, 0, 0, 0.5, 1, 1024, 4096)          # creates the library's streams (pool primed at first handle)
print(f"+ libdoa handle (4 primed streams): {idle_sync():.2f} us")
extra = [torch.cuda.Stream() for _ in range(4)]
for s in extra:
    with torch.cuda.stream(s): x.add_(1)
torch.cuda.synchronize()
print(f"+ 4 used torch streams: {idle_sync():.2f} us")
more = [torch.cuda.Stream() for _ in range(16)]
for s in more:
    with torch.cuda.stream(s): x.add_(1)
torch.cuda.synchronize()
print(f"+ 16 more used streams: {idle_sync():.2f} us")
# one tiny kernel then sync: launch + completion + wake-up
def one_kernel(n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        x.add_(1); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print(f"one tiny kernel + sync: {one_kernel():.2f} us")
t0 = time.perf_counter()
for _ in range(200): pipe.synchronize()
print(f"pipe.synchronize idle: {(time.perf_counter()-t0)/200*1e6:.2f} us")
