"""Experiment (GPU box): host cost of 30 dependent tiny launches, eager vs hipGraph replay (torch.cuda.CUDAGraph)."""
import time, torch
dev = "cuda:0"
x = torch.zeros(4096, device=dev)
K = 30
def body():
    for _ in range(K):
        x.add_(1.0)
for _ in range(3): body()
torch.cuda.synchronize()
for reps in (200,):
    t0 = time.perf_counter()
    for _ in range(reps): body()
    t1 = time.perf_counter()           # host issue time
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("eager: host issue %.1f us / %d-kernel sequence (%.2f us/launch), wall %.1f us" % ((t1 - t0) / reps * 1e6, K, (t1 - t0) / reps / K * 1e6, (t2 - t0) / reps * 1e6))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    body()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        body()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
reps = 200
t0 = time.perf_counter()
for _ in range(reps): g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("graph: host issue %.1f us / replay, wall %.1f us / replay (%.2f us/node)" % ((t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6, (t2 - t0) / reps / K * 1e6))
# 4 graphs on 4 streams
streams = [torch.cuda.Stream() for _ in range(4)]
xs = [torch.zeros(4096, device=dev) for _ in range(4)]
graphs = []
for st, xx in zip(streams, xs):
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        for _ in range(K): xx.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(gg, stream=st):
            for _ in range(K): xx.add_(1.0)
    graphs.append(gg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps // 4):
    for st, gg in zip(streams, graphs):
        with torch.cuda.stream(st):
            gg.replay()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("4 graphs on 4 streams: wall %.1f us / replay" % ((t2 - t0) / reps * 1e6))
t0 = time.perf_counter()
for _ in range(reps // 4):
    for st, xx in zip(streams, xs):
        with torch.cuda.stream(st):
            for _ in range(K): xx.add_(1.0)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("eager on 4 streams: wall %.1f us / sequence" % ((t2 - t0) / reps * 1e6))
