"""Lab: which torch streams run concurrently?  Every pair (i, j) of N freshly created streams gets one single-workgroup spin
kernel each (torch.cuda._sleep); elapsed ~ 1x = concurrent (different hardware queues), ~ 2x = serialised.
python tools/stream_queue_lab.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rdmnet_amd  # (sets GPU_MAX_HW_QUEUES as the product does)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))
streams = [torch.cuda.Stream() for _ in range(n)]
cycles = 20_000_000


def timed(ss):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in ss:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


timed(streams[:1])
one = timed(streams[:1])
print(f'one stream: {one:.2f} ms')
for i in range(n):
    print(' '.join(f'{timed([streams[i], streams[j]]) / one:4.1f}' if j != i else '  - ' for j in range(n)))
for k in (2, 3, 4, 5, 6, 8):
    if k <= n:
        print(f'{k} streams at once: {timed(streams[:k]) / one:.2f}x')
