import sys, os, faulthandler
faulthandler.enable()
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import test_gpu_fullsize as T
pre = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else []
for s in pre:
    T.test_full_size_distillation_two_steps_fp32(s)
    print("two_steps", s, "ok", flush=True)
T.test_full_size_schedules_are_bit_identical_fp32_and_bf16()
print("done", flush=True)
