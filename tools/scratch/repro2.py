import sys, os, faulthandler
faulthandler.enable()
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import test_gpu_fullsize as T
pre = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else []
if os.environ.get("REPRO_GC") == "1":
    import gc
    _orig = T._distill_pair
    def _dp(g, dt):
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
        return _orig(g, dt)
    T._distill_pair = _dp
for s in pre:
    T.test_full_size_distillation_two_steps_fp32(s)
    print("two_steps", s, "ok", flush=True)
T.test_full_size_schedules_are_bit_identical_fp32_and_bf16()
print("done", flush=True)
