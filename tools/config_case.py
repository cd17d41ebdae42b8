"""One row of tools/config_bench.py (for kernel traces): python tools/config_case.py {distill|c4}"""
import sys; sys.path[:0]=["/root/repo","/root/repo/tools","/root/repo/tests"]
import torch, config_bench as CB
dev=torch.device("cuda",0)
which = sys.argv[1] if len(sys.argv) > 1 else "distill"
if which == "distill":
    r=CB.image_case("celeba distill", CB.glow_cfg([64,64,3],16,3,256), CB.glow_cfg([64,64,3],32,3,512), 16, dev, steps=10)
else:
    r=CB.image_case("config4", CB.glow_cfg([64,64,3],32,4,512), None, 16, dev, steps=10)
print(r["config"], round(r["images_per_s"]), round(r["ms_per_step"],3), r["student_steps_on_fused_forward"], flush=True)
