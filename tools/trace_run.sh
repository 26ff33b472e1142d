python3 - <<'PY'
import os,sys,subprocess,tempfile,time
sys.path.insert(0,".")
from vmaf_amd import synth
exe="vmaf_amd/vmafossexec"; model="model/vmaf_v0.6.1.pkl"
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    r,d=synth.make_clip(1920,1080,30,seed=1234)
    rp,dp=tmp+"/r.yuv",tmp+"/d.yuv"
    synth.write_yuv(rp,r,"yuv420p"); synth.write_yuv(dp,d,"yuv420p")
    for i in range(3):
        t0=time.perf_counter()
        p=subprocess.run([exe,"yuv420p","1920","1080",rp,dp,model],capture_output=True,text=True,env=dict(os.environ,VMAF_HIP_TRACE="1"))
        print("wall %.3f"%(time.perf_counter()-t0)); 
    print(p.stderr)
PY
