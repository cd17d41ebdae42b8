import sys; sys.path[:0]=["/root/repo","/root/repo/tools","/root/repo/tests"]
import torch, config_bench as CB
dev=torch.device("cuda",0)
r=CB.tab_case(63,10,32,65536,dev); print(r["config"],"train ms",round(r["train_ms_per_step"],3),"fwd Mrows/s",round(r["forward_rows_per_s"]/1e6,1),flush=True)
