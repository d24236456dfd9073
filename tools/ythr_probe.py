import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from sdf_tools_amd import capi, synth
n=512; shape=(n,n,n); dev=torch.device("cuda",0)
out=torch.empty(shape,dtype=torch.float32,device=dev); s=torch.cuda.current_stream().cuda_stream
for p in (0.03, 0.025, 0.02, 0.015):
    masks=[synth.bernoulli_mask_torch(shape,p,1+k,device=dev) for k in range(2)]
    for thr in (36,):
        ctx=capi.SdfGpu(0); ctx.set_option("far_threshold_y", thr)
        for i in range(20):
            ctx.build_device(masks[i%2].data_ptr(),shape,out.data_ptr(),0.01,False,s); torch.cuda.synchronize()
        t0=time.perf_counter()
        for i in range(40): ctx.build_device(masks[i%2].data_ptr(),shape,out.data_ptr(),0.01,False,s)
        torch.cuda.synchronize(); print(p, "thr_y", thr, round((time.perf_counter()-t0)/40*1e3,3), ctx.last_path())
        ctx.close()
