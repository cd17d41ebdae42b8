"""Software-pipelined distillation step: teacher(t+1) runs concurrently with student(t).

In the reference's step (pl_module.py:198-382) the critical path is
    teacher forward  ->  loss  ->  student backward  ->  optimiser,
but the teacher is frozen: its outputs depend only on the batch.  `PipelinedStep` therefore computes the
teacher's hint tensors for batch t+1 on a side stream while the student runs forward / backward / Adam on
batch t with the hints produced one call earlier.  Every call still performs exactly the work of one reference
step (one teacher forward, one student forward + backward, one optimiser step), on the same values:

  * batch t+1 is dequantised for the student (x += n_s) and that tensor is kept for the next call, then the
    teacher re-dequantises the same buffer in place (x += n_t), so the teacher still sees student noise +
    its own (models/utils.py:38 + pl_module.py:216-225) and the RNG draws come in the reference's order
    n_s(0), n_t(0), n_s(1), n_t(1), ...
  * the student consumes the hint tensors through static buffers, so the whole call (both branches) can be
    captured into one hipGraph and replayed.  Both branches are plain chains of launches (the backward runs each
    conv's dgrad and weight-gradient GEMM as one launch), which is what lets them share the chip cleanly: the
    teacher's deep-level steps fill 16-64 CUs, the student's chain takes the rest.

`training_step(x_next)` trains on the batch given to the PREVIOUS call and returns its losses; `prime(x0)`
starts the pipeline and `drain()` trains on the last batch.  Results are bit-identical to the sequential
`DistillationStep` fed with the same noise (tests/test_gpu_pipeline.py).

Multi-GPU: as in `training.CapturedStep` the collective is never captured: graph A = both branches up to the end
of the backward, then the eager bucketed all-reduce of the flat gradient arena, then graph B = clip + Adam.
"""
import os

import torch

from .training import DistillationStep, distillation_loss


class PipelinedStep:
    def __init__(self, step: DistillationStep, example_x, use_graph=True, warmup=2, with_noise=False, strict=False):
        self.strict = strict    # True: a failed capture raises instead of degrading to eager execution
        if step.teacher is None or step.w_kd <= 0:
            raise ValueError("pipelining needs a teacher with kd_weight > 0")
        if getattr(step.student, "y_condition", False) or step.w_perc > 0:
            # conditional / tabular recipes are launch-bound 1-D flows; they run through DistillationStep / CapturedStep
            raise ValueError("the software pipeline covers the unconditional image recipe (no y_onehot, no perceptual branch)")
        self.step = step
        self.x_in = torch.empty_like(example_x)        # incoming batch t+1
        self.x_stu_next = torch.empty_like(example_x)  # batch t+1 + student noise (student input of the next call)
        self.x_cur = torch.empty_like(example_x)       # student input of this call
        self.zeros = torch.zeros_like(example_x)       # "noise" for the already dequantised student input
        # optional caller-supplied dequantisation noise for batch t+1 (parity tests); otherwise drawn on the device
        self.ns_in = torch.zeros_like(example_x) if with_noise else None
        self.nt_in = torch.zeros_like(example_x) if with_noise else None
        self.t_next = None   # static hint buffers written by the teacher branch
        self.t_cur = None    # ... and read by the student branch
        self.side = torch.cuda.Stream()
        self.use_graph = use_graph
        self.split = step.world > 1
        if self.split and use_graph:
            step.set_allreduce_overlap(False)   # graph mode: one all-reduce after the replayed backward
        self.graph = self.graph_b = None
        self.out, self._warm = None, warmup
        self.primed = False
        self._one = None
        # The teacher's chain runs BESIDE the student's and has slack: of its deep-level fused steps only those that would
        # fill fewer than 32 CUs are split over more workgroups (level 3: 16 -> 64); level 2 stays on its 64 CUs instead
        # of taking all 256 from the student's chain (+3.4 % step throughput against splitting both).
        self.teacher_split = "frugal"
        # The student's forward, too, runs beside the other chain here: its level-2 steps stay on 64 CUs (28.8 us) instead
        # of spreading over 256 for 19.8 + 5 us (split + finishing launch); +2 % step throughput.  Alone on the chip (the
        # sequential schedule) the 4-way split of both levels is the faster one and stays the engine's default.
        self.student_split = "frugal"
        # Tried and dropped, with the measurements (profiles/r02_*, profiles/r03_schedule_experiments.json): forking the teacher's
        # branch after the student's first or second level (-2.6 % / -7.5 %), a lock-step of the two chains' level-1 phases
        # (-6 %), creating the student's graph nodes before the teacher's (no change).

    # ------------------------------------------------------------------ the two branches
    def _teacher_branch(self):
        """dequantise x_in for the student, keep that copy, run the teacher on the same buffer, keep its hints"""
        st = self.step
        n_s = self.ns_in if self.ns_in is not None else torch.empty_like(self.x_in).uniform_(0, 1.0 / 256)
        self.x_in.add_(n_s)                                                    # n_s(t+1)
        self.x_stu_next.copy_(self.x_in)
        teng = st.teacher.flow.engine() if hasattr(st.teacher, "flow") else None
        prev = teng.fused_split_inference if teng is not None else None
        if teng is not None:
            teng.fused_split_inference = self.teacher_split
        try:
            with torch.no_grad():
                t_z, _, _ = st.teacher(self.x_in, None, noise=self.nt_in, keep=st.t_idx, hints_only=True)   # adds n_t(t+1) in place
        finally:
            if teng is not None:
                teng.fused_split_inference = prev
        hints = [t_z[i] for i in st.t_idx]
        if self.t_next is None:
            self.t_next = [torch.empty_like(h) for h in hints]
            self.t_cur = [torch.empty_like(h) for h in hints]
        torch._foreach_copy_(self.t_next, hints)   # one launch for all hint tensors

    def _student_fwd_bwd(self, zero=True, before_loss=None):
        """zero=False: the caller clears the gradient arena itself (on another stream); before_loss(): called between the
        student's forward and the loss (the staged pipeline waits there for the hint tensors' hand-over)."""
        st = self.step
        if zero:
            st.flat.zero_grad()
        else:
            st.flat.attach_grads()
        st.reducer.reset()
        seng = st.student.flow.engine() if hasattr(st.student, "flow") else None
        prev = seng.fused_split_training if seng is not None else None
        if seng is not None:
            seng.fused_split_training = self.student_split
        try:
            s_z, s_nll, _ = st.student(self.x_cur, None, noise=self.zeros)
        finally:
            if seng is not None:
                seng.fused_split_training = prev
        if before_loss is not None:
            before_loss()
        out = distillation_loss(s_nll, [s_z[i] for i in st.s_idx], self.t_cur, st.w_nll, st.w_kd)
        if self._one is None:
            self._one = torch.ones((), device=out["result_loss"].device, dtype=out["result_loss"].dtype)
        out["result_loss"].backward(self._one)   # (the seed: a resident 1 instead of a fill launch per step)
        return {k: v.detach() for k, v in out.items()}

    def _handover(self):
        # one launch (multi-tensor copy) instead of one per tensor: these sit at the head of both chains
        torch._foreach_copy_(self.t_cur + [self.x_cur], self.t_next + [self.x_stu_next])

    def _body_a(self):
        """both branches up to the end of the backward"""
        main = torch.cuda.current_stream()
        self._handover()   # what the previous call prepared
        fork = torch.cuda.Event()
        fork.record(main)
        self.side.wait_event(fork)
        with torch.cuda.stream(self.side):
            self._teacher_branch()
            join = torch.cuda.Event()
            join.record(self.side)
        out = self._student_fwd_bwd()
        main.wait_event(join)
        return out

    def _body(self):
        out = self._body_a()
        self.step.allreduce()
        self.step.optimize()
        return out

    # ------------------------------------------------------------------ public API
    def _load(self, x, noise_s, noise_t):
        self.x_in.copy_(x)
        if self.ns_in is not None:
            self.ns_in.copy_(noise_s)
            self.nt_in.copy_(noise_t)

    def prime(self, x0, noise_s=None, noise_t=None):
        """Start the pipeline with the first batch (runs one teacher forward)."""
        self._load(x0, noise_s, noise_t)
        self._teacher_branch()
        self.primed = True

    def _capture(self):
        torch.cuda.synchronize()
        if not self.split:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.out = self._body()
            self.graph = g
            return
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            self.out = self._body_a()
        with torch.cuda.graph(gb, pool=ga.pool()):
            self.step.optimize()
        self.graph, self.graph_b = ga, gb

    def _replay(self):
        self.graph.replay()
        if self.split:
            self.step.reducer.reset()
            self.step.allreduce()
            self.graph_b.replay()
        return self.out

    def training_step(self, x_next, noise_s=None, noise_t=None):
        """Train on the previously supplied batch while the teacher processes `x_next`."""
        if not self.primed:
            raise RuntimeError("call prime(first_batch) first")
        self._load(x_next, noise_s, noise_t)
        if not self.use_graph:
            return self._body()
        if self.graph is None:
            if self._warm > 0:
                self._warm -= 1
                return self._body()
            try:
                self._capture()
            except Exception as exc:  # pragma: no cover - depends on the runtime
                if self.strict:
                    raise
                import warnings
                warnings.warn(f"hipGraph capture of the pipelined step failed ({exc!r}); running eagerly")
                self.use_graph = False
                self.graph = self.graph_b = None
                if self.split:
                    self.step.set_allreduce_overlap(True)
                torch.cuda.synchronize()
                return self._body()
        return self._replay()

    def prepare(self, x0, noise_s=None, noise_t=None):
        """prime + eager warm-up + capture, so that no timed call pays for it (consumes warm-up batches = x0)."""
        self.prime(x0, noise_s, noise_t)
        while self.use_graph and self.graph is None:
            self.training_step(x0, noise_s, noise_t)

    def drain(self):
        """Train on the last supplied batch (no new teacher work)."""
        self._handover()
        self.primed = False
        out = self._student_fwd_bwd()
        self.step.allreduce()
        self.step.optimize()
        return out


class AheadPipelinedStep(PipelinedStep):
    """The software pipeline with the teacher TWO batches ahead, so that the gradient all-reduce is overlapped in graph mode.

    Several ranks replay `student fwd+bwd -> (eager all-reduce) -> clip + Adam`; during the all-reduce the GPU has nothing to
    compute, and cutting the student's backward into bucket-sized graphs is not possible from inside `loss.backward()` (the
    engine's backward runs on autograd's device thread; a capture must end on the thread that began it).  The frozen teacher
    offers the work instead: its forward for batch t+2 is LAUNCHED right after the student's backward of batch t, as a graph
    of its own on the side stream, so it runs under the all-reduce and the optimiser of batch t (and then beside the student
    of batch t+1, as before).  What the student's chain no longer has to share the chip with is as long as the collective.
    Every call still does one teacher forward, one student forward + backward, one all-reduce and one optimiser step, on the
    same values in the same order (bit-identical parameters: tests/test_gpu_pipeline.py, tests/test_gpu_ddp.py).

    Call pattern: `prime(x0, x1)`, then `training_step(x_{t+2})` trains on batch t and returns its losses, `drain()` (twice
    the work of the base class: two batches are in flight) returns the losses of the last two batches.  Opt-in
    (`bench.py --teacher-ahead 2`): no RCCL run has been possible on this repo's one-GPU boxes, and the plain
    `PipelinedStep` keeps the N-rank path as simple as it can be until one has.
    """

    def __init__(self, step, example_x, use_graph=True, warmup=2, with_noise=False, strict=False):
        super().__init__(step, example_x, use_graph, warmup, with_noise, strict)
        self.main = torch.cuda.Stream()
        self.ev_t, self.ev_h = torch.cuda.Event(), torch.cuda.Event()
        self.ev_load = torch.cuda.Event()   # the caller's batch has been copied into the static buffers (side stream)
        self.graph_t = None
        if step.world > 1:
            step.set_allreduce_overlap(False)   # one all-reduce after the backward, under the teacher's next forward

    # ---- building blocks (each runs on the stream named in its comment)
    def _teacher_async(self, x, noise_s, noise_t):
        """side stream: wait for the handover, load batch `x`, run (or replay) the teacher branch"""
        self.side.wait_event(self.ev_h)
        with torch.cuda.stream(self.side):
            self._load(x, noise_s, noise_t)
            self.ev_load.record(self.side)
            if self.graph_t is not None:
                self.graph_t.replay()
            else:
                self._teacher_branch()
            self.ev_t.record(self.side)

    def _handover_main(self):
        """main stream: the teacher's outputs of the batch the student trains on next become current"""
        self.main.wait_event(self.ev_t)
        self._handover()
        self.ev_h.record(self.main)

    def _student(self):
        if self.graph is not None:
            self.graph.replay()
            return self.out
        return self._student_fwd_bwd()

    def _finish(self):
        if self.graph is not None:
            self.step.reducer.reset()   # (a replayed backward does not run the Python that resets the bucket state)
        self.step.allreduce()
        if self.graph_b is not None:
            self.graph_b.replay()
        else:
            self.step.optimize()

    def _clone_out(self, out):
        return {k: v.clone() for k, v in out.items()}

    # ---- public API
    def prime(self, x0, x1=None, noise_s=None, noise_t=None, noise_s1=None, noise_t1=None):
        """teacher(x0) -> handover -> teacher(x1) in flight.  x1 defaults to x0 (benchmarks that cycle one batch)."""
        x1 = x0 if x1 is None else x1
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        self.main.wait_stream(cur)
        with torch.cuda.stream(self.main):
            self.ev_h.record(self.main)
        self._teacher_async(x0, noise_s, noise_t)
        with torch.cuda.stream(self.main):
            self._handover_main()
        self._teacher_async(x1, noise_s1 if noise_s1 is not None else noise_s, noise_t1 if noise_t1 is not None else noise_t)
        cur.wait_event(self.ev_load)
        cur.wait_stream(self.main)
        self.primed = True

    def _capture(self):
        torch.cuda.synchronize()
        gt, ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gt, stream=self.side):
            self._teacher_branch()
        with torch.cuda.graph(ga, stream=self.main):
            self.out = self._student_fwd_bwd()
        with torch.cuda.graph(gb, stream=self.main):
            self.step.optimize()
        self.graph_t, self.graph, self.graph_b = gt, ga, gb

    def _iteration(self, x_new, noise_s, noise_t):
        """student(t) || teacher(t+1) [in flight] -> handover -> teacher(t+2) launched -> all-reduce(t) -> optimiser(t)"""
        cur = torch.cuda.current_stream()
        self.main.wait_stream(cur)
        self.side.wait_stream(cur)   # (x_new was produced on the caller's stream)
        with torch.cuda.stream(self.main):
            out = self._student()
            if x_new is not None:
                self._handover_main()
        if x_new is not None:
            self._teacher_async(x_new, noise_s, noise_t)
            # the caller may free or overwrite x_new / the noise tensors once this call returns: its stream must not run
            # ahead of the side stream's copy of them (queued behind the handover event)
            cur.wait_event(self.ev_load)
        with torch.cuda.stream(self.main):
            self._finish()
        cur.wait_stream(self.main)
        return out

    def training_step(self, x_next, noise_s=None, noise_t=None):
        if not self.primed:
            raise RuntimeError("call prime(first_batch, second_batch) first")
        if self.use_graph and self.graph is None:
            if self._warm > 0:
                self._warm -= 1
            else:
                # the capture itself consumes nothing: it records the launches of one teacher branch / one student pass /
                # one optimiser step without running them (the buffers they touch are the static ones)
                torch.cuda.synchronize()
                try:
                    self._capture_guarded()
                except Exception as exc:  # pragma: no cover - depends on the runtime
                    if self.strict:
                        raise
                    import warnings
                    warnings.warn(f"hipGraph capture of the pipelined step failed ({exc!r}); running eagerly")
                    self.use_graph = False
                    self.graph = self.graph_b = self.graph_t = None
                    torch.cuda.synchronize()
        return self._iteration(x_next, noise_s, noise_t)

    def _capture_guarded(self):
        """Capture must not disturb the values in flight: the teacher's outputs for the next batch sit in t_next / x_stu_next
        and the batch after it in x_in.  Nothing executes during capture, but the RNG offsets of the captured uniform draws
        are registered with the graph-safe generator, and zero_grad / the reducer are reset by the replays themselves."""
        self._capture()

    def prepare(self, x0, noise_s=None, noise_t=None):
        self.prime(x0, x0, noise_s, noise_t)
        while self.use_graph and self.graph is None:
            self.training_step(x0, noise_s, noise_t)

    def drain(self):
        """Train on the two batches still in flight (no new teacher work); returns the losses of both, oldest first."""
        first = self._clone_out(self._iteration(None, None, None))
        cur = torch.cuda.current_stream()
        self.main.wait_stream(cur)
        with torch.cuda.stream(self.main):
            self._handover_main()
            second = self._student()
            self._finish()
        cur.wait_stream(self.main)
        self.primed = False
        return [first, second]


class StagedPipelinedStep(PipelinedStep):
    """The software pipeline with the frozen teacher cut into TWO stages that run beside each other (and beside the student):

        stage A   level 1 of the teacher on batch t+2         (32 launches that each want the whole chip)
        stage B   the teacher's deeper levels on batch t+1    (two long launches on 64 CUs each: one per level)
        student   forward / backward / optimiser on batch t

    In `PipelinedStep` the teacher's forward is one chain of ~2 ms that has to fit beside the student's ~2.9 ms every call --
    it is co-critical: shortening it by 70 us (the hints-only encode) shortened the step by 100 us.  Cut after level 1, its
    longest stage is 1.1 ms and never on the critical path.  Every call still does one whole teacher forward's work (stage A
    of one batch, stage B of another), one student forward + backward and one optimiser step, on the same values in the same
    RNG order: parameters bit-identical to the sequential step (tests/test_gpu_pipeline.py).

    Call pattern as `AheadPipelinedStep`: `prime(x0, x1)`, then `training_step(x_{t+2})` trains on batch t; `drain()` returns the
    losses of the two batches still in flight.  The stages are two FlowEngines over slices of the teacher's layer list (shared,
    frozen parameters; hints-only passes: engine.encode(..., keep, hints_only=True))."""

    def __init__(self, step, example_x, use_graph=True, warmup=2, with_noise=False, strict=False):
        super().__init__(step, example_x, use_graph, warmup, with_noise, strict)
        from .engine import FlowEngine
        from .modules import Split2d
        layers = list(step.teacher.flow.layers)
        cut = next((i for i, l in enumerate(layers) if isinstance(l, Split2d)), None)
        if cut is None or getattr(step.teacher, "is_1d", False):
            raise ValueError("the staged pipeline needs a multi-level image teacher")
        self.cut = cut                                   # stage A = layers[:cut] (squeeze + level-1 steps), B = the rest
        teng = step.teacher.flow.engine()
        self.eng_a, self.eng_b = FlowEngine(layers[:cut]), FlowEngine(layers[cut:])
        for e in (self.eng_a, self.eng_b):
            e.set_compute_dtype("bf16" if teng.bf16 else "f32")
            e.fused, e.fused_runs = teng.fused, teng.fused_runs
        n = len(layers)
        self.keep_a = [i for i in step.t_idx if i < cut]
        self.keep_b = [(i % n) - cut for i in step.t_idx if (i % n) >= cut]
        self.side_b = torch.cuda.Stream()
        self.xs_a = torch.empty_like(example_x)          # student input of the batch stage A works on (age 0)
        self.xs_b = torch.empty_like(example_x)          # ... of the batch stage B works on (age 1); x_cur is age 2
        self.h_a = self.h_b = self.h_cur = None          # stage A's hint tensors at ages 0 / 1 / 2
        self.za_a = self.za_cur = None                   # stage A's output: written (age 0) / read by stage B (age 1)
        self.hb_a = self.hb_cur = None                   # stage B's hint tensors: written (age 1) / read by the student (age 2)
        self.in_flight = 0
        self.student_first = os.environ.get("NFK_STAGED_STUDENT_FIRST", "1") == "1"   # (see _body_a)

    # ---- the two stages (each on the stream that is current when it is called)
    def _stage_a(self):
        st = self.step
        n_s = self.ns_in if self.ns_in is not None else torch.empty_like(self.x_in).uniform_(0, 1.0 / 256)
        self.x_in.add_(n_s)
        self.xs_a.copy_(self.x_in)
        n_t = self.nt_in if self.nt_in is not None else torch.empty_like(self.x_in).uniform_(0, 1.0 / 256)
        self.eng_a.fused_split_inference = self.teacher_split
        with torch.no_grad():
            outs, _ = self.eng_a.encode(self.x_in, None, 0.0, n_t, keep=self.keep_a, hints_only=True)
        hints, za = [outs[i] for i in self.keep_a], outs[-1]
        if self.h_a is None:
            self.h_a, self.h_b, self.h_cur = ([torch.empty_like(h) for h in hints] for _ in range(3))
            self.za_a, self.za_cur = torch.empty_like(za), torch.empty_like(za)
        torch._foreach_copy_(self.h_a + [self.za_a], hints + [za])

    def _stage_b(self):
        self.eng_b.fused_split_inference = self.teacher_split
        with torch.no_grad():
            outs, _ = self.eng_b.encode(self.za_cur, None, 0.0, None, keep=self.keep_b, hints_only=True)
        hints = [outs[i] for i in self.keep_b]
        if self.hb_a is None:
            self.hb_a, self.hb_cur = [torch.empty_like(h) for h in hints], [torch.empty_like(h) for h in hints]
        torch._foreach_copy_(self.hb_a, hints)

    def _shift(self, full=True, x_done=False):
        """Everything moves one age on: what the student reads first (two multi-tensor copies: the second one's sources are
        the first one's destinations' sources).  x_done: the caller has copied the student's input already."""
        if full:
            torch._foreach_copy_(self.h_cur + self.hb_cur + ([] if x_done else [self.x_cur]) + [self.za_cur],
                                 self.h_b + self.hb_a + ([] if x_done else [self.xs_b]) + [self.za_a])
            self.t_cur = self.h_cur + self.hb_cur
        else:   # priming: no stage-B results yet
            self.za_cur.copy_(self.za_a)
        torch._foreach_copy_(self.h_b + [self.xs_b], self.h_a + [self.xs_a])

    def _fork_stages(self, main, run_a=True, run_b=True):
        fork = torch.cuda.Event()
        fork.record(main)
        joins = []
        for stream, fn, on in ((self.side, self._stage_a, run_a), (self.side_b, self._stage_b, run_b)):
            if not on:
                continue
            stream.wait_event(fork)
            with torch.cuda.stream(stream):
                fn()
                ev = torch.cuda.Event()
                ev.record(stream)
                joins.append(ev)
        return joins

    def _body_a(self):
        """The student's chain is the step: of the hand-over only its input is copied on its stream; the hint tensors move on
        stage A's stream (the loss waits for them), the gradient arena is cleared on stage B's (the loss's backward waits).
        student_first: the student's launches are enqueued -- in a captured graph: its nodes created -- before the stages'.
        A replayed hipGraph reaches the hardware queues node by node in creation order at a rate the host sets; the stages
        have slack, the student's chain has none."""
        main = torch.cuda.current_stream()
        self.x_cur.copy_(self.xs_b)
        fork = torch.cuda.Event()
        fork.record(main)
        self.side.wait_event(fork)
        self.side_b.wait_event(fork)
        with torch.cuda.stream(self.side):
            self._shift(x_done=True)
            shifted = torch.cuda.Event()
            shifted.record(self.side)
        with torch.cuda.stream(self.side_b):
            self.step.flat.grad.zero_()
            zeroed = torch.cuda.Event()
            zeroed.record(self.side_b)
            self.side_b.wait_event(shifted)

        def hints_ready():
            main.wait_event(shifted)
            main.wait_event(zeroed)

        def stages():
            with torch.cuda.stream(self.side):
                self._stage_a()
                join_a = torch.cuda.Event()
                join_a.record(self.side)
            with torch.cuda.stream(self.side_b):
                self._stage_b()
                join_b = torch.cuda.Event()
                join_b.record(self.side_b)
            return join_a, join_b
        if self.student_first:
            out = self._student_fwd_bwd(zero=False, before_loss=hints_ready)
            joins = stages()
        else:
            joins = stages()
            out = self._student_fwd_bwd(zero=False, before_loss=hints_ready)
        for ev in joins:
            main.wait_event(ev)
        return out

    # ---- public API
    def prime(self, x0, x1=None, noise_s=None, noise_t=None, noise_s1=None, noise_t1=None):
        """teacher(x0) -> handover -> teacher(x1) in flight.  x1 defaults to x0 (benchmarks that cycle one batch)."""
        x1 = x0 if x1 is None else x1
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        self.main.wait_stream(cur)
        with torch.cuda.stream(self.main):
            self.ev_h.record(self.main)
        self._teacher_async(x0, noise_s, noise_t)
        with torch.cuda.stream(self.main):
            self._handover_main()
        self._teacher_async(x1, noise_s1 if noise_s1 is not None else noise_s, noise_t1 if noise_t1 is not None else noise_t)
        cur.wait_event(self.ev_load)
        cur.wait_stream(self.main)
        self.primed = True

    def _capture(self):
        torch.cuda.synchronize()
        gt, ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gt, stream=self.side):
            self._teacher_branch()
        with torch.cuda.graph(ga, stream=self.main):
            self.out = self._student_fwd_bwd()
        with torch.cuda.graph(gb, stream=self.main):
            self.step.optimize()
        self.graph_t, self.graph, self.graph_b = gt, ga, gb

    def _iteration(self, x_new, noise_s, noise_t):
        """student(t) || teacher(t+1) [in flight] -> handover -> teacher(t+2) launched -> all-reduce(t) -> optimiser(t)"""
        cur = torch.cuda.current_stream()
        self.main.wait_stream(cur)
        self.side.wait_stream(cur)   # (x_new was produced on the caller's stream)
        with torch.cuda.stream(self.main):
            out = self._student()
            if x_new is not None:
                self._handover_main()
        if x_new is not None:
            self._teacher_async(x_new, noise_s, noise_t)
            # the caller may free or overwrite x_new / the noise tensors once this call returns: its stream must not run
            # ahead of the side stream's copy of them (queued behind the handover event)
            cur.wait_event(self.ev_load)
        with torch.cuda.stream(self.main):
            self._finish()
        cur.wait_stream(self.main)
        return out

    def training_step(self, x_next, noise_s=None, noise_t=None):
        if not self.primed:
            raise RuntimeError("call prime(first_batch, second_batch) first")
        if self.use_graph and self.graph is None:
            if self._warm > 0:
                self._warm -= 1
            else:
                # the capture itself consumes nothing: it records the launches of one teacher branch / one student pass /
                # one optimiser step without running them (the buffers they touch are the static ones)
                torch.cuda.synchronize()
                try:
                    self._capture_guarded()
                except Exception as exc:  # pragma: no cover - depends on the runtime
                    if self.strict:
                        raise
                    import warnings
                    warnings.warn(f"hipGraph capture of the pipelined step failed ({exc!r}); running eagerly")
                    self.use_graph = False
                    self.graph = self.graph_b = self.graph_t = None
                    torch.cuda.synchronize()
        return self._iteration(x_next, noise_s, noise_t)

    def _capture_guarded(self):
        """Capture must not disturb the values in flight: the teacher's outputs for the next batch sit in t_next / x_stu_next
        and the batch after it in x_in.  Nothing executes during capture, but the RNG offsets of the captured uniform draws
        are registered with the graph-safe generator, and zero_grad / the reducer are reset by the replays themselves."""
        self._capture()

    def prepare(self, x0, noise_s=None, noise_t=None):
        self.prime(x0, x0, noise_s, noise_t)
        while self.use_graph and self.graph is None:
            self.training_step(x0, noise_s, noise_t)

    def drain(self):
        """Train on the two batches still in flight (no new teacher work); returns the losses of both, oldest first."""
        first = self._clone_out(self._iteration(None, None, None))
        cur = torch.cuda.current_stream()
        self.main.wait_stream(cur)
        with torch.cuda.stream(self.main):
            self._handover_main()
            second = self._student()
            self._finish()
        cur.wait_stream(self.main)
        self.primed = False
        return [first, second]


class StagedPipelinedStep(PipelinedStep):
    """The software pipeline with the frozen teacher cut into TWO stages that run beside each other (and beside the student):

        stage A   level 1 of the teacher on batch t+2         (32 launches that each want the whole chip)
        stage B   the teacher's deeper levels on batch t+1    (two long launches on 64 CUs each: one per level)
        student   forward / backward / optimiser on batch t

    In `PipelinedStep` the teacher's forward is one chain of ~2 ms that has to fit beside the student's ~2.9 ms every call --
    it is co-critical: shortening it by 70 us (the hints-only encode) shortened the step by 100 us.  Cut after level 1, its
    longest stage is 1.1 ms and never on the critical path.  Every call still does one whole teacher forward's work (stage A
    of one batch, stage B of another), one student forward + backward and one optimiser step, on the same values in the same
    RNG order: parameters bit-identical to the sequential step (tests/test_gpu_pipeline.py).

    Call pattern as `AheadPipelinedStep`: `prime(x0, x1)`, then `training_step(x_{t+2})` trains on batch t; `drain()` returns the
    losses of the two batches still in flight.  The stages are two FlowEngines over slices of the teacher's layer list (shared,
    frozen parameters; hints-only passes: engine.encode(..., keep, hints_only=True))."""

    def __init__(self, step, example_x, use_graph=True, warmup=2, with_noise=False, strict=False):
        super().__init__(step, example_x, use_graph, warmup, with_noise, strict)
        from .engine import FlowEngine
        from .modules import Split2d
        layers = list(step.teacher.flow.layers)
        cut = next((i for i, l in enumerate(layers) if isinstance(l, Split2d)), None)
        if cut is None or getattr(step.teacher, "is_1d", False):
            raise ValueError("the staged pipeline needs a multi-level image teacher")
        self.cut = cut                                   # stage A = layers[:cut] (squeeze + level-1 steps), B = the rest
        teng = step.teacher.flow.engine()
        self.eng_a, self.eng_b = FlowEngine(layers[:cut]), FlowEngine(layers[cut:])
        for e in (self.eng_a, self.eng_b):
            e.set_compute_dtype("bf16" if teng.bf16 else "f32")
            e.fused, e.fused_runs = teng.fused, teng.fused_runs
        n = len(layers)
        self.keep_a = [i for i in step.t_idx if i < cut]
        self.keep_b = [(i % n) - cut for i in step.t_idx if (i % n) >= cut]
        self.side_b = torch.cuda.Stream()
        self.xs_a = torch.empty_like(example_x)          # student input of the batch stage A works on (age 0)
        self.xs_b = torch.empty_like(example_x)          # ... of the batch stage B works on (age 1); x_cur is age 2
        self.h_a = self.h_b = self.h_cur = None          # stage A's hint tensors at ages 0 / 1 / 2
        self.za_a = self.za_cur = None                   # stage A's output: written (age 0) / read by stage B (age 1)
        self.hb_a = self.hb_cur = None                   # stage B's hint tensors: written (age 1) / read by the student (age 2)
        self.in_flight = 0
        self.student_first = os.environ.get("NFK_STAGED_STUDENT_FIRST", "1") == "1"   # (see _body_a)

    # ---- the two stages (each on the stream that is current when it is called)
    def _stage_a(self):
        st = self.step
        n_s = self.ns_in if self.ns_in is not None else torch.empty_like(self.x_in).uniform_(0, 1.0 / 256)
        self.x_in.add_(n_s)
        self.xs_a.copy_(self.x_in)
        n_t = self.nt_in if self.nt_in is not None else torch.empty_like(self.x_in).uniform_(0, 1.0 / 256)
        self.eng_a.fused_split_inference = self.teacher_split
        with torch.no_grad():
            outs, _ = self.eng_a.encode(self.x_in, None, 0.0, n_t, keep=self.keep_a, hints_only=True)
        hints, za = [outs[i] for i in self.keep_a], outs[-1]
        if self.h_a is None:
            self.h_a, self.h_b, self.h_cur = ([torch.empty_like(h) for h in hints] for _ in range(3))
            self.za_a, self.za_cur = torch.empty_like(za), torch.empty_like(za)
        torch._foreach_copy_(self.h_a + [self.za_a], hints + [za])

    def _stage_b(self):
        self.eng_b.fused_split_inference = self.teacher_split
        with torch.no_grad():
            outs, _ = self.eng_b.encode(self.za_cur, None, 0.0, None, keep=self.keep_b, hints_only=True)
        hints = [outs[i] for i in self.keep_b]
        if self.hb_a is None:
            self.hb_a, self.hb_cur = [torch.empty_like(h) for h in hints], [torch.empty_like(h) for h in hints]
        torch._foreach_copy_(self.hb_a, hints)

    def _shift(self, full=True, x_done=False):
        """Everything moves one age on: what the student reads first (two multi-tensor copies: the second one's sources are
        the first one's destinations' sources).  x_done: the caller has copied the student's input already."""
        if full:
            torch._foreach_copy_(self.h_cur + self.hb_cur + ([] if x_done else [self.x_cur]) + [self.za_cur],
                                 self.h_b + self.hb_a + ([] if x_done else [self.xs_b]) + [self.za_a])
            self.t_cur = self.h_cur + self.hb_cur
        else:   # priming: no stage-B results yet
            self.za_cur.copy_(self.za_a)
        torch._foreach_copy_(self.h_b + [self.xs_b], self.h_a + [self.xs_a])

    def _fork_stages(self, main, run_a=True, run_b=True):
        fork = torch.cuda.Event()
        fork.record(main)
        joins = []
        for stream, fn, on in ((self.side, self._stage_a, run_a), (self.side_b, self._stage_b, run_b)):
            if not on:
                continue
            stream.wait_event(fork)
            with torch.cuda.stream(stream):
                fn()
                ev = torch.cuda.Event()
                ev.record(stream)
                joins.append(ev)
        return joins

    def _body_a(self):
        """The student's chain is the step: of the hand-over only its input is copied on its stream; the hint tensors move on
        stage A's stream (the loss waits for them), the gradient arena is cleared on stage B's (the loss's backward waits)."""
        main = torch.cuda.current_stream()
        self.x_cur.copy_(self.xs_b)
        fork = torch.cuda.Event()
        fork.record(main)
        self.side.wait_event(fork)
        self.side_b.wait_event(fork)
        with torch.cuda.stream(self.side):
            self._shift(x_done=True)
            shifted = torch.cuda.Event()
            shifted.record(self.side)
            self._stage_a()
            join_a = torch.cuda.Event()
            join_a.record(self.side)
        with torch.cuda.stream(self.side_b):
            self.step.flat.grad.zero_()
            zeroed = torch.cuda.Event()
            zeroed.record(self.side_b)
            self.side_b.wait_event(shifted)
            self._stage_b()
            join_b = torch.cuda.Event()
            join_b.record(self.side_b)

        def hints_ready():
            main.wait_event(shifted)
            main.wait_event(zeroed)
        out = self._student_fwd_bwd(zero=False, before_loss=hints_ready)
        main.wait_event(join_a)
        main.wait_event(join_b)
        return out

    # ---- public API
    def prime(self, x0, x1=None, noise_s=None, noise_t=None, noise_s1=None, noise_t1=None):
        x1 = x0 if x1 is None else x1
        main = torch.cuda.current_stream()
        self._load(x0, noise_s, noise_t)
        self._stage_a()                                   # batch 0 through stage A
        self._shift(full=False)
        self._load(x1, noise_s1 if noise_s1 is not None else noise_s, noise_t1 if noise_t1 is not None else noise_t)
        for ev in self._fork_stages(main):                # batch 1 through stage A beside batch 0 through stage B
            main.wait_event(ev)
        self.primed, self.in_flight = True, 2

    def prepare(self, x0, noise_s=None, noise_t=None):
        self.prime(x0, x0, noise_s, noise_t)
        while self.use_graph and self.graph is None:
            self.training_step(x0, noise_s, noise_t)

    def drain(self):
        """Train on the two batches still in flight (no new teacher input); returns the losses of both, oldest first."""
        main = torch.cuda.current_stream()
        outs = []
        for left in (2, 1):
            self._shift()
            joins = self._fork_stages(main, run_a=False, run_b=left == 2)
            out = self._student_fwd_bwd()
            for ev in joins:
                main.wait_event(ev)
            self.step.allreduce()
            self.step.optimize()
            outs.append({k: v.clone() for k, v in out.items()})
        self.primed = False
        return outs
