import torch, sys
x = torch.ones(1024, device="cuda")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = x * 2 if sys.argv[1] == "norng" else x * 2 + torch.rand(1024, device="cuda")
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print("done", sys.argv[1])
