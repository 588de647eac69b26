import sys, os, json
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "hm-opencl_amd"))
import torch
from hmme import api, synth, sequence, shard
dev=torch.device("cuda",0)
eng=api.Engine(0,128); eng.set_lambda(57.9)
w,h,n_frames,sr=3840,2160,64,64
src=synth.Sequence(w,h,n_frames,seed=777,bit_depth=8)
pairs=shard.gop_pairs(n_frames,"randomaccess")
for refine in (False, True):
    for ppl in (1,2,4,8):
        res=None
        for _ in range(2):
            res=None
            res=sequence.run_rank(eng,src,pairs,w,h,8,sr,stream_mode=True,pairs_per_launch=ppl,device=dev,refine=refine)
        print(json.dumps({"refine":refine,"pairs_per_launch":ppl,"seconds":round(res["seconds"],4),"pairs_per_s":round(len(pairs)/res["seconds"],1),"stages":res["stages"]}))
eng.close()
