"""Is a hipGraph replay ordered after kernels enqueued before it on the same stream?"""
import torch
dev = torch.device('cuda:0')
x = torch.zeros(1 << 20, device=dev)
big = torch.randn(4096, 4096, device=dev)
def body():
    return x * 2.0 + 1.0
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        y = body()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = body()
for label, stream in (('default stream', torch.cuda.current_stream()), ('side stream', torch.cuda.Stream())):
    bad = 0
    with torch.cuda.stream(stream):
        for i in range(200):
            src = torch.full((1 << 20,), float(i), device=dev)
            for _ in range(3):
                big @ big                      # keep the stream busy so that the copy below is still queued
            x.copy_(src)
            g.replay()
            got = y.clone()
            if float((got - (2.0 * i + 1.0)).abs().max()) != 0.0:
                bad += 1
    torch.cuda.synchronize()
    print(label, 'wrong replays:', bad, 'of 200')
