"""The 1-D rows of tools/config_bench.py alone (training step under hipGraph replay, eager forward)."""
import sys; sys.path[:0]=["/root/repo","/root/repo/tools","/root/repo/tests"]
import torch, config_bench as CB
dev=torch.device("cuda",0)
for d,K,B in ((6,5,1024),(6,5,65536),(63,10,1024),(63,10,65536)):
    r=CB.tab_case(d,K,32,B,dev); print(r["config"],B,"train ms",round(r["train_ms_per_step"],3),"fwd Mrows/s",round(r["forward_rows_per_s"]/1e6,1),flush=True)
r=CB.tab_case(63,10,32,65536,dev,made=True); print(r["config"],"train ms",round(r["train_ms_per_step"],3),"fwd Mrows/s",round(r["forward_rows_per_s"]/1e6,1),flush=True)
