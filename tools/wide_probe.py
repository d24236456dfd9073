import sys, os, json, time
sys.path.insert(0, os.getcwd())
import torch
from sdf_tools_amd import capi, synth
n=512; shape=(n,n,n); dev=torch.device("cuda",0)
out=torch.empty(shape,dtype=torch.float32,device=dev); s=torch.cuda.current_stream().cuda_stream
names = ["pack", "ball", "z", "y", "env_y", "x", "env_x"]
for p in (0.05, 0.04, 0.03, 0.02):
    masks=[synth.bernoulli_mask_torch(shape,p,1+k,device=dev) for k in range(2)]
    for wf in (16, 1000):
        ctx=capi.SdfGpu(0); ctx.set_option("wide_y_from", wf); ctx.set_option("dense", 0)
        for i in range(12):
            ctx.build_device(masks[i%2].data_ptr(),shape,out.data_ptr(),0.01,False,s); torch.cuda.synchronize()
        ctx.get_stage_times(); ctx.set_profiling(1)
        for i in range(8):
            ctx.build_device(masks[i%2].data_ptr(),shape,out.data_ptr(),0.01,False,s)
        torch.cuda.synchronize(); st,b=ctx.get_stage_times(); ctx.set_profiling(0)
        print(p, "wide_from", wf, {k: round(v/max(b,1),3) for k,v in zip(names,st) if v>0}, ctx.get_extrema())
        ctx.close()
