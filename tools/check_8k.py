import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np, time
import oracle, vmaf_amd
from vmaf_amd import synth
w,h=7680,4320
r,d=synth.make_clip(w,h,2,seed=5)
ref,dis=synth.to_luma_f32(r),synth.to_luma_f32(d)
with vmaf_amd.Engine(w,h,max_batch=2) as e:
    e.enable_metrics()
    t=time.time(); got=e.extract(ref,dis); print('gpu',time.time()-t); m=e.metrics.copy()
    got2=e.extract_raw(r,d); assert np.array_equal(got,got2)
t=time.time(); want=oracle.clip_atoms(ref[:1],dis[:1],oracle.ORDER_AVX); print('oracle',time.time()-t)
rel=np.abs(got[0]-want[0])/np.maximum(1,np.abs(want[0])); print('max rel',rel.max())
print(m[0], oracle.psnr(ref[0],dis[0]), oracle.ssim(ref[0],dis[0])[0], oracle.ms_ssim(ref[0],dis[0])[0])
