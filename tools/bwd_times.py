"""Student forward + backward alone on the chip (bf16, CIFAR K=8, B=64): ms per pass for the backward forms.
usage: python tools/bwd_times.py [reps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import nf_distillation_amd as nfa
from nf_distillation_amd import capi
from param_util import randomize_state_dict, shapes_of, synthetic_images
from oracle_check import cifar_cfg

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else None   # run one form only (for rocprofv3): "steps", "level", "runs", "aux"
m = nfa.create_glow_model(cifar_cfg(8))
m.load_state_dict(randomize_state_dict(shapes_of(m.state_dict()), 11))
m = m.cuda().set_compute_dtype("bf16")
eng = m.flow.engine()
x, noise = synthetic_images(64, seed=3)
x, noise = x.cuda(), noise.cuda()
side = torch.cuda.Stream()
aux = capi.lib().nfk_aux_create(side.cuda_stream)
for tag, name, lb, lm, ax in (("steps", "per-step chain", False, False, False), ("level", "level, one launch per step", True, False, False),
                              ("runs", "level, runs", True, True, False), ("aux", "level, runs, wgrad on aux", True, True, True)):
    if only and tag != only:
        continue
    eng.level_backward, eng.level_multi = lb, lm
    eng.aux, eng.level_wgrad_aux = (aux if ax else None), ax

    def step():
        if grads:
            torch._foreach_zero_(grads)
        _, bpd, _ = m(x.clone(), None, noise=noise)
        bpd.mean().backward()
    grads = []
    step()
    grads = [p.grad for p in m.parameters() if p.grad is not None]
    for _ in range(3):
        step()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name:32s} {1e3 * (time.perf_counter() - t0) / reps:7.3f} ms per forward + backward (hipGraph replay)", flush=True)
