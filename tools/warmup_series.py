import os, sys, json, ctypes
sys.path.insert(0, os.getcwd())
sys.argv=['bench.py','--steps','60','--warmup','0','--no-cpu-baseline']
import bench, torch
args=bench.parse()
import binaural_audio_synthesis_amd as bas
dev=torch.device('cuda',0); torch.cuda.set_device(0)
host=bas.synth.make_table("consistent",0).truncated(128)
tbl=bas.irs_and_delaydiffs(host.upsampling,host.diffs_left,host.diffs_right,host.irs_left,host.irs_right,device=dev)
sc=bench.Scene(args,bas,dev,1,0,"strong",tbl,8)
ev=bench.HipEvents(60)
import time
torch.cuda.synchronize()
ts=[]
for i in range(60):
    t0=time.perf_counter(); sc.render_into(sc.y, ev.pairs[i]); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
print("fir ms:", " ".join(f"{ev.elapsed_ms(i):.3f}" for i in range(60)))
print("step ms (synced):", " ".join(f"{t:.3f}" for t in ts))
