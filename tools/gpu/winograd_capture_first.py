import torch, sys
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
from openpifpaf_amd import winograd
x=torch.randn((2,64,40,40),device='cuda').contiguous(memory_format=torch.channels_last)
w=torch.randn((64,64,3,3),device='cuda')*0.05
u=winograd.transform_filter(w,2)
out=torch.empty_like(x)
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        winograd.conv3x3(x,u,64,variant=2,out=out)      # FIRST call of the process happens inside the capture
    g.replay()
torch.cuda.synchronize()
ref=torch.nn.functional.conv2d(x,w,padding=1)
print('capture-first ok, err', float((out-ref).abs().max()/ref.abs().max()))
