import sys, os
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import test_gpu_level_bwd as T
for variant in (dict(flow_permutation="shuffle"), dict(LU_decomposed=False)):
    cfg = T._cfg(2, **variant)
    ref = T._grads(cfg, 6, 77, "steps", 5)
    for mode in ("level", "level_multi"):
        got = T._grads(cfg, 6, 77, mode, 5)
        errs = []
        for k in ref:
            sc = float(ref[k].abs().max())
            e = float((got[k] - ref[k]).abs().max()) / sc if sc > 0 else float(got[k].abs().max())
            errs.append((e, k))
        errs.sort(reverse=True)
        print(variant, mode, [(f"{e:.2e}", k) for e, k in errs[:12]], flush=True)
