"""Probe: nfk_event_record_external inside torch.cuda.graph capture, from the main thread and from an autograd backward."""
import os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nf_distillation_amd import capi
lib = capi.lib()

class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x * 2
    @staticmethod
    def backward(ctx, g):
        ev = lib.nfk_event_create()
        rc = lib.nfk_event_record_external(ev, capi.stream())
        print("  in backward: thread", threading.current_thread().name, "rc", rc, "stream", hex(capi.stream() or 0), flush=True)
        return g * 2

x = torch.ones(4, device="cuda", requires_grad=True)
for mt in (True, False):
    for mode in ("global", "thread_local", "relaxed"):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.autograd.set_multithreading_enabled(mt):
                with torch.cuda.graph(g, capture_error_mode=mode):
                    ev = lib.nfk_event_create()
                    rc = lib.nfk_event_record_external(ev, capi.stream())
                    print(f"multithreading={mt} mode={mode}: main thread rc {rc} stream {hex(capi.stream() or 0)}", flush=True)
                    y = F.apply(x).sum()
                    y.backward()
            print("  capture ok")
        except Exception as e:
            print("  capture failed:", repr(e)[:200])
            torch.cuda.synchronize()
