python3 - <<'PY'
# Where a short vmafossexec run spends its wall time: parent-side clock around the child + the child's own stamps
# (VMAF_HIP_TRACE=1).  30- and 600-frame 1080p pairs in /dev/shm; third run of each (binary, libraries, page cache warm).
import os,sys,subprocess,tempfile,time,re
sys.path.insert(0,".")
from vmaf_amd import synth
exe="vmaf_amd/vmafossexec"; model="model/vmaf_v0.6.1.pkl"
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    r,d=synth.make_clip(1920,1080,30,seed=1234)
    rp,dp=tmp+"/r.yuv",tmp+"/d.yuv"
    synth.write_yuv(rp,r,"yuv420p"); synth.write_yuv(dp,d,"yuv420p")
    one_r,one_d=open(rp,"rb").read(),open(dp,"rb").read()
    for frames in (30,600):
        if frames>30:
            with open(rp,"ab") as f, open(dp,"ab") as g:
                for _ in range(frames//30-1): f.write(one_r); g.write(one_d)
        for i in range(3):
            t0=time.time()
            p=subprocess.run([exe,"yuv420p","1920","1080",rp,dp,model],capture_output=True,text=True,env=dict(os.environ,VMAF_HIP_TRACE="1"))
            t1=time.time()
        st={m.group(2):float(m.group(1)) for m in re.finditer(r"\[vmafossexec ([0-9.]+)\] (.*)",p.stderr)}
        print(f"== {frames} frames: wall {1e3*(t1-t0):.1f} ms; exec -> main {1e3*(st['main entered']-t0):.1f} ms; main {1e3*(st['main returns']-st['main entered']):.1f} ms; return -> parent sees exit {1e3*(t1-st['main returns']):.1f} ms")
        print("\n".join(l for l in p.stderr.split("\n") if "vmaf_hip" in l))
PY
