import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sdf_tools_amd import capi, synth
n=512; dev=torch.device("cuda",0)
ctx=capi.SdfGpu(0); s=torch.cuda.current_stream().cuda_stream
out=torch.empty((n,n,n),dtype=torch.float32,device=dev)
m=synth.bernoulli_mask_torch((n,n,n),0.0005,1,device=dev); m[:,:,:2]=1
ctx.set_option("dense",0); ctx.set_option("far_predict",2)
for i in range(6):
    ctx.build_device(m.data_ptr(),(n,n,n),out.data_ptr(),0.01,False,s)
    print("noisy", i, ctx.debug_flat_habit(), ctx.last_build_info()["far_predicted"])
r=synth.room_mask_torch((n,n,n),dev)
for i in range(4):
    ctx.build_device(r.data_ptr(),(n,n,n),out.data_ptr(),0.01,False,s)
    print("room", i, ctx.debug_flat_habit())
