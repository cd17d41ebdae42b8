"""Reproducer of the HIP runtime fault recorded in DESIGN.md section 5: a captured graph with three branches (teacher || student ||
the auxiliary stream's level weight gradients, engine.level_wgrad_aux = True) segfaults inside hip::Graph::UpdateStreams at its
first hipGraphLaunch when earlier graphs that used the same auxiliary stream were captured AND destroyed in this process.
    python tools/probes/graph_update_streams_repro.py all      (on a GPU box; `rocgdb -x tools/probes/graph_update_streams_gdb.txt`)
Not part of any test: the product keeps the option off by default and bench.py keeps its graphs alive."""
import sys, os, faulthandler
faulthandler.enable()
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import test_gpu_fullsize as T
from golden_util import Golden
from nf_distillation_amd.pipeline import PipelinedStep
from nf_distillation_amd.training import CapturedStep

which = sys.argv[1] if len(sys.argv) > 1 else "all"
g = Golden("distill_full")
batches = T._distill_batches(g)

def pipelined(dt):
    step = T._distill_pair(g, dt)
    step.enable_concurrency()
    runner = PipelinedStep(step, batches[0][0], use_graph=True, warmup=2, with_noise=True)
    cur = torch.cuda.current_stream().cuda_stream
    print(dt, "streams: current", hex(cur), "runner.side", hex(runner.side.cuda_stream), "step._side", hex(step._side.cuda_stream),
          "step._aux", hex(step._aux.cuda_stream), flush=True)
    runner.prepare(*batches[0])
    torch.cuda.synchronize()
    print(dt, "pipelined prepared ok", flush=True)
    step.enable_concurrency(False)

def graph(dt):
    step = T._distill_pair(g, dt)
    step.enable_concurrency()
    runner = CapturedStep(step, batches[0][0], warmup=2, with_noise=True)
    for _ in range(3):
        runner.training_step(batches[0][0].clone(), batches[0][1], batches[0][2])
    torch.cuda.synchronize()
    print(dt, "graph ok", hex(step._aux.cuda_stream), flush=True)
    step.enable_concurrency(False)

seq = {"all": ["g32", "p32", "g32", "p32", "g16", "p16"], "short": ["p32", "p16"], "only": ["p16"], "gp": ["g16", "p16"],
       "pp": ["p16", "p16"], "ggp": ["g32", "g16", "p16"]}[which]
for s in seq:
    (graph if s[0] == "g" else pipelined)("f32" if s.endswith("32") else "bf16")
print("done", flush=True)
