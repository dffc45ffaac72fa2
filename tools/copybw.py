import sys,os
sys.path.insert(0,os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import torch
from tools import spc_tools
src=torch.empty(1<<30,dtype=torch.uint8,device="cuda"); dst=torch.empty_like(src)
s=torch.cuda.current_stream()
for _ in range(3): spc_tools.stream_copy(dst,src,s)
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record(s)
for _ in range(10): spc_tools.stream_copy(dst,src,s)
b.record(s); torch.cuda.synchronize()
print("grid",os.environ.get("SPC_COPY_GRID","2048"),"copy GB/s (r+w):", 2*src.numel()*10/(a.elapsed_time(b)*1e-3)/1e9)
a.record(s)
for _ in range(10): dst.copy_(src)
b.record(s); torch.cuda.synchronize()
print("torch copy_ GB/s (r+w):", 2*src.numel()*10/(a.elapsed_time(b)*1e-3)/1e9)
