import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from sdf_tools_amd import capi, synth
n=512; res=0.01
ctx=capi.SdfGpu(0); dev=torch.device("cuda",0)
pts=torch.from_numpy(synth.two_box_points(200000, seed=0, scale=n*res)).to(dev)
mask=torch.zeros((n,n,n),dtype=torch.uint8,device=dev); out=torch.empty((n,n,n),dtype=torch.float32,device=dev)
s=torch.cuda.current_stream().cuda_stream
ctx.voxelize_points_device(pts.data_ptr(), pts.shape[0], (0.0,0.0,0.0), res, (n,n,n), mask.data_ptr(), True, s)
names=["pack_bits","dense_ball","sweep_z","sweep_y","envelope_y","sweep_x","envelope_x"]
for fm in (0,1):
    for rep in range(3):
        ctx.set_option("policy_reset",1); ctx.set_option("fixup_mode",fm)
        torch.cuda.synchronize(); ctx.get_stage_times(); ctx.set_profiling(1)
        ctx.build_device(mask.data_ptr(), (n,n,n), out.data_ptr(), res, False, s)
        torch.cuda.synchronize(); ms,b=ctx.get_stage_times(); ctx.set_profiling(0)
        print("fix_mode",fm,{k:round(v,4) for k,v in zip(names,ms) if v>0})
