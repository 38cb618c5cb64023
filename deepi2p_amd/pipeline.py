"""Stream / graph executor of the registration hot path: the replacement of the reference's evaluation driver
(evaluation/visualize_and_save_data.py:69-97 -> evaluation/registration_lsq.py:291-345: per batch `set_input`, `inference_pass`,
then per frame `get_initial_guess` + the 60-restart `solve_P_random_perturb` fan-out over worker processes).

One STEP = one batch of frames through  H2D copies -> classifier -> argmax -> initial guess / front filter -> R-restart pose solve ->
argmin.  Steps of different batches are independent, and a lone step cannot fill the chip (the pose solve is a chain of dependent
sweeps with a heavy tail, several classifier kernels have fewer workgroups than CUs), so the executor keeps S steps in flight:

  * S HIP streams, one SLOT per stream: device input tensors, pinned host staging buffers, device outputs;
  * the ~150 launches of a step are captured ONCE per slot into a hipGraph (every entry point of the C-ABI library is
    capture-safe: no allocation, no synchronisation) and replayed; the H2D copies read the slot's pinned buffers on a second stream
    per slot (SDMA engines; as memcpy nodes of the graph they would run as blit kernels on the CUs), so `submit(host_batch)` is:
    memcpy into pinned memory + ONE async copy of the slot's flat staging buffer (all seven inputs) + one graph launch;
  * results stay on the device until `result()` is asked for them (one event per slot).

`bench.py` is a thin caller of this class.  Environment: more than 3 streams need GPU_MAX_HW_QUEUES >= S (set before the HIP
runtime starts; with the default of 4 hardware queues the streams share queues and 4 are slower than 3 -- DESIGN.md section 4).
"""
import time

import torch

from . import ops

INPUT_NAMES = ("pc", "intensity", "sn", "node_a", "node_b", "img")
# per-batch camera matrices (the reference's loader hands K with every batch: visualize_and_save_data.py:81-90, set_input(..., P, img, K)).
# Optional in a host batch: when present it is staged, copied and read by the step like the other six; when absent the slot keeps the
# K it was constructed with / last given.
K_NAME = "K"


class Slot:
    """Per-stream state: device inputs, pinned staging, the captured graphs and their (static) outputs."""

    def __init__(self, index, stream):
        self.index, self.stream = index, stream
        self.devs, self.cur, self.host = [{}], 0, {}      # device input sets (two with double_buffer) and the one holding the newest batch
        self.graphs = {}          # (with_h2d, input set) -> (hipGraph, outputs dict)
        self.set_done = [torch.cuda.Event(), torch.cuda.Event()]      # last step that read input set j
        self.done = torch.cuda.Event(enable_timing=True)
        self.start = torch.cuda.Event(enable_timing=True)
        self.outputs = None
        self.busy = False
        self.host_flat, self.dev_flats = None, []         # ONE pinned staging buffer / ONE device buffer per input set: host[k] / devs[j][k] are views

    @property
    def dev(self):
        return self.devs[self.cur]

    def copy_in(self, j=None):
        """The step's host->device transfer of ALL inputs of the slot: one asynchronous copy of the flat staging buffer (seven tensors, one
        DMA command) on the current stream."""
        self.dev_flats[self.cur if j is None else j].copy_(self.host_flat, non_blocking=True)


def _flat_views(flat, layout):
    """Typed views into a flat uint8 buffer: layout = [(name, offset, shape, dtype)]."""
    out = {}
    for name, off, shape, dtype in layout:
        n = 1
        for v in shape:
            n *= int(v)
        out[name] = flat[off:off + n * torch.empty((), dtype=dtype).element_size()].view(dtype).view(shape)
    return out


class RegistrationExecutor:
    """mm: a deepi2p_amd.networks.MMClassifer(Coarse) with weights loaded; pipe: a deepi2p_amd.registration.RegistrationPipeline.

    executor = RegistrationExecutor(mm, pipe, K, example_batch, n_streams=8)
    ticket = executor.submit(host_batch)          # dict of CPU tensors pc/intensity/sn/node_a/node_b/img [+ K]; returns at once
    out = executor.result(ticket)                 # dict: pred i32[B,N], P f64[B,4,4], cost, best, iters, ... (device tensors of the slot)

    Fixed per executor (they are baked into the captured graphs): the batch SHAPE (every host batch must have the example's shapes --
    pad a short last batch; a mismatch raises), the restart list (``restarts``: one draw for all steps; pass your own, or build one
    executor per list) and the solver settings of ``pipe``.  Per batch: the six network inputs and the camera matrices ``K``
    (f32 / f64 [B,3,3]; a per-slot device buffer the graph reads, so datasets with per-sequence or per-frame intrinsics are solved with
    THEIR K).  The executor follows the classifier's weights: when they change (load_state_dict, .to(), an optimiser step) the next
    submit re-packs the kernel operands and re-captures the graphs instead of replaying against freed or stale operands.

    labels_override: i32[B,N] device tensor fed to the solver INSTEAD of the network's argmax (the benchmark's synthetic labels,
    SURVEY.md 8d: random-init weights predict nothing); default None = the network's own prediction, as the reference does."""

    def __init__(self, mm, pipe, K, example_batch, n_streams=8, use_graph=True, restarts=None, labels_override=None, step_fn=None,
                 post_fn=None, h2d_mode="copy_stream", split_solver=False, double_buffer=False):
        self.mm, self.pipe = mm, pipe
        self.device = mm.device
        self.n_streams = max(1, int(n_streams))
        self.use_graph = bool(use_graph)
        self.labels_override = labels_override
        B = int(example_batch["pc"].shape[0])
        self.K64 = K.to(self.device, torch.float64).contiguous()
        if self.K64.dim() == 2:
            self.K64 = self.K64.unsqueeze(0).expand(B, 3, 3).contiguous()
        if tuple(self.K64.shape) != (B, 3, 3):
            raise ValueError("K must be [3,3] or [B,3,3] with B = %d frames, got %s" % (B, tuple(K.shape)))
        self.restarts = restarts if restarts is not None else pipe.draw(B, self.device)
        self.step_fn = step_fn            # custom graph-capturable step: step_fn(slot, device_inputs) -> outputs dict
        self.post_fn = post_fn            # launched EAGERLY on the slot's stream after the step (work that cannot be captured, e.g. a
                                          # torch.distributed collective): post_fn(slot, outputs) -> outputs dict
        self.graph_error = None
        # where the host->device copies of a step run: "copy_stream" (default) = hipMemcpyAsync on a second stream per slot + an event the step
        # waits for (the copies go through the SDMA engines and overlap the slot's own previous step: 0.96-0.97 of the resident rate);
        # "eager" = hipMemcpyAsync on the slot's stream ahead of the replay (0.95-0.96); "graph" = memcpy nodes of the step's graph, which
        # the runtime executes as blit KERNELS on the CUs (0.91-0.95; tools/sweep_h2d_mode.sh, tools/probe_h2d.sh)
        self.h2d_mode = h2d_mode
        # double_buffer (option; h2d_mode "copy_stream" with graphs): TWO sets of device inputs per slot and one graph per set (sharing one
        # memory pool), so the copies of a slot's NEXT batch do not wait for its running step to stop reading the inputs.  Measured
        # (tools/oneoff/r04_dbuf.sh): the H2D-inclusive latency of a batch drops 51.5 -> 48.5 ms at 8 batches in flight, the H2D-inclusive RATE
        # does not move (0.95 of the resident rate either way: the copies do not cost the step its inputs' wait) -- off by default
        self.double_buffer = bool(double_buffer) and h2d_mode == "copy_stream" and self.use_graph
        # split_solver (experiment, graphs only): the classifier and the pose solve of a step as TWO graphs on two streams of different
        # priority -- the classifier's ~90 short kernels on a high-priority queue, the solver's long-lived workgroups on a normal one, so
        # that a freed compute unit goes to a waiting classifier kernel first (tools/oneoff/r04_split.sh)
        self.split_solver = bool(split_solver) and self.use_graph and step_fn is None
        mm.detector.prepack()                 # derive the kernel operands now, on the current stream, before other streams use them
        self._weights_version = mm.detector.weights_version
        self._packed_refs = mm.detector.packed_operands()     # the graphs hold raw pointers into these: keep them alive
        self._x3_refs = []                                    # ... and into their split (bf16x3) copies: accumulated at every capture
        self._K64_host = self.K64.cpu()
        torch.cuda.synchronize(self.device)
        self.slots = []
        for i in range(self.n_streams):
            s = Slot(i, torch.cuda.Stream(device=self.device, priority=-1 if self.split_solver else 0))
            s.solver_stream = torch.cuda.Stream(device=self.device) if self.split_solver else None
            s.net_done = torch.cuda.Event()
            s.copy_stream = torch.cuda.Stream(device=self.device) if h2d_mode == "copy_stream" else None
            s.copied = torch.cuda.Event()
            # one flat pinned buffer and one flat device buffer hold all seven inputs (256-byte aligned pieces): a step's transfer is ONE copy
            layout, off = [], 0
            for k in INPUT_NAMES + (K_NAME,):
                shape, dtype = ((B, 3, 3), torch.float64) if k == K_NAME else (tuple(example_batch[k].shape), example_batch[k].dtype)
                n = 1
                for v in shape:
                    n *= int(v)
                layout.append((k, off, shape, dtype))
                off = (off + n * torch.empty((), dtype=dtype).element_size() + 255) // 256 * 256
            s.host_flat = torch.empty((off,), dtype=torch.uint8).pin_memory()
            s.host = _flat_views(s.host_flat, layout)
            for k in INPUT_NAMES:
                s.host[k].copy_(example_batch[k])
            s.host[K_NAME].copy_(self._K64_host)
            n_sets = 2 if self.double_buffer else 1
            s.devs = []
            for _ in range(n_sets):
                flat = s.host_flat.to(self.device, non_blocking=False)
                s.dev_flats.append(flat)
                s.devs.append(_flat_views(flat, layout))
            self.slots.append(s)
        self._next = 0
        self._h2d_warm = False
        self._replayed = set()
        self._warmed = set()
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------------------------------------------------ one step
    def _step(self, slot, with_h2d):
        """Enqueue one step on the CURRENT stream (the slot's stream, or the capturing stream)."""
        if with_h2d:        # a1: MMClassifer.set_input's copies (multimodal_classifier.py:82-93), from the slot's pinned buffers
            slot.copy_in()
        d = slot.dev
        if self.step_fn is not None:
            return self.step_fn(slot, d)
        return self._solve_part(slot, self._net_part(slot))

    def _net_part(self, slot):
        d = slot.dev
        logits = self.mm.detector(d["pc"], d["intensity"], d["sn"], d["node_a"], d["node_b"], d["img"])
        coarse = logits[0] if isinstance(logits, tuple) else logits
        net = {"pred": ops.argmax_channels(coarse)}                          # inference_pass (:100-117): i32 [B,N]
        if isinstance(logits, tuple):
            net["fine_pred"] = ops.argmax_channels(logits[1])
        return net

    def _solve_part(self, slot, net):
        d = slot.dev
        labels = self.labels_override if self.labels_override is not None else net["pred"]
        out = self.pipe(d["pc"], labels, d[K_NAME], self.restarts)           # same stream: the pose solve follows its classification
        out.update(net)
        return out

    def _capture(self, slot, with_h2d):
        """Capture the step once per input set of the slot (the sets' graphs share one memory pool: they never run at the same time)."""
        keep = slot.cur
        pool = None
        for j in range(len(slot.devs)):
            slot.cur = j
            self._capture_set(slot, with_h2d, j, pool)
            g = slot.graphs[(with_h2d, j)][0]
            pool = (g[0] if isinstance(g, tuple) else g).pool()
        slot.cur = keep

    def _capture_set(self, slot, with_h2d, j, pool):
        kw = {} if pool is None else {"pool": pool}
        with torch.cuda.stream(slot.stream):
            self._step(slot, with_h2d)                                       # eager once: lazily created constants, allocator warm-up
        slot.stream.synchronize()
        # the graph will hold raw pointers to the split (bf16x3) weights too: ACCUMULATE them (a later capture must not drop what an
        # earlier graph points to); the list is emptied only when every graph is (_follow_weights)
        have = {id(t) for t in self._x3_refs}
        self._x3_refs += [t for t in ops.x3_live_operands(self._packed_refs) if id(t) not in have]      # this model's splits only
        if self.split_solver:
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, stream=slot.stream, **kw):
                if with_h2d:
                    slot.copy_in()
                net = self._net_part(slot)
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(gb, stream=slot.solver_stream, pool=ga.pool()):
                out = self._solve_part(slot, net)
            slot.graphs[(with_h2d, j)] = ((ga, gb), out)
            return
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=slot.stream, **kw):
            out = self._step(slot, with_h2d)
        slot.graphs[(with_h2d, j)] = (g, out)

    def _replay(self, slot, g):
        """Replay a slot's step on the CURRENT stream (= slot.stream); with split_solver the pose solve follows on the solver stream."""
        if not isinstance(g, tuple):
            g.replay()
            return
        slot.stream.wait_event(slot.done)             # the slot's previous solve still reads the labels this classifier pass overwrites
        g[0].replay()
        slot.net_done.record()
        with torch.cuda.stream(slot.solver_stream):
            slot.solver_stream.wait_event(slot.net_done)
            g[1].replay()

    def warm_up(self, with_h2d=True):
        """Capture (or run once) every slot's step so that the first timed submit pays nothing extra.  A failed capture switches the
        executor to eager launches (and remembers why in `graph_error`)."""
        self._follow_weights()
        want_h2d = bool(with_h2d)
        with_h2d = with_h2d and self.h2d_mode == "graph"
        for slot in self.slots:
            if self.use_graph and (with_h2d, 0) not in slot.graphs:
                try:
                    self._capture(slot, with_h2d)
                except Exception as exc:          # noqa: BLE001 -- any capture problem: run eagerly, keep the reason
                    self.graph_error = "%s: %s" % (type(exc).__name__, exc)
                    self.use_graph = False
                    for s in self.slots:
                        s.graphs.clear()
            if not self.use_graph:
                with torch.cuda.stream(slot.stream):
                    slot.outputs = self._step(slot, with_h2d)
        if self.use_graph:
            # likewise the first replay of a captured graph (the runtime uploads it then): once per slot and graph here
            for slot in self.slots:
                if (with_h2d, 0) in slot.graphs and (slot.index, with_h2d) not in self._replayed:
                    with torch.cuda.stream(slot.stream):
                        for j in range(len(slot.devs)):
                            self._replay(slot, slot.graphs[(with_h2d, j)][0])
                    self._replayed.add((slot.index, with_h2d))
        if want_h2d and not self._h2d_warm:
            # the first copy on a stream pays one-time costs (DMA queue set-up, first touch of the pinned buffers by the engine): once per
            # slot here, like the graph capture above, not inside somebody's first timed steps
            for slot in self.slots:
                with torch.cuda.stream(slot.copy_stream if slot.copy_stream is not None else slot.stream):
                    for j in range(len(slot.devs)):
                        slot.copy_in(j)
            self._h2d_warm = True
        self._warmed.add(want_h2d)
        torch.cuda.synchronize(self.device)

    def step_eager(self, slot_index=0, with_h2d=False):
        """One step launched eagerly on the CURRENT stream with slot `slot_index`'s buffers (profiling passes: events around every
        C-ABI call need eager launches on one stream)."""
        slot = self.slots[slot_index]
        out = self._step(slot, with_h2d)
        if self.post_fn is not None:
            out = self.post_fn(slot, out)
        return out

    # ------------------------------------------------------------------------------------------------------------ submit / result
    def submit(self, host_batch=None, with_h2d=None):
        """Start one step on the next slot (round robin) and return its ticket.  host_batch: dict of CPU tensors (copied into the slot's
        pinned buffers, then H2D inside the step); None: the slot's resident device inputs are used as they are (with_h2d=False) or
        re-sent from its pinned buffers (with_h2d=True)."""
        if with_h2d is None:
            with_h2d = host_batch is not None
        self._follow_weights()
        if bool(with_h2d) not in self._warmed:
            self.warm_up(with_h2d)                # first use: captures, first replays, first copies (synchronises the device once)
        slot = self.slots[self._next]
        if host_batch is not None:                # validate BEFORE the slot position advances: a rejected submit consumes nothing
            for k in INPUT_NAMES + ((K_NAME,) if K_NAME in host_batch else ()):
                if tuple(host_batch[k].shape) != tuple(slot.host[k].shape):
                    raise ValueError("host batch %r has shape %s, this executor was built (and its graphs captured) for %s -- pad the batch "
                                     "or build an executor for that shape" % (k, tuple(host_batch[k].shape), tuple(slot.host[k].shape)))
            if not with_h2d:
                raise ValueError("a host batch needs with_h2d=True (its copies are part of the step)")
        self._next = (self._next + 1) % self.n_streams
        if slot.busy and host_batch is not None:
            # new host data: the slot's previous H2D copies must have read the pinned buffers (their own event when they run on the copy
            # stream; otherwise they are part of the step)
            (slot.copied if self.h2d_mode == "copy_stream" and self.double_buffer else slot.done).synchronize()
        if host_batch is not None:
            for k in INPUT_NAMES:
                slot.host[k].copy_(host_batch[k])
            if K_NAME in host_batch:
                slot.host[K_NAME].copy_(host_batch[K_NAME])       # f32 -> f64 on the way into the pinned buffer
            else:
                slot.host[K_NAME].copy_(self._K64_host)           # no K in this batch: the constructor's, not what the slot last held
        names = INPUT_NAMES + (K_NAME,)
        in_step = with_h2d and self.h2d_mode == "graph"
        if with_h2d and self.h2d_mode == "copy_stream":
            if self.double_buffer:
                slot.cur ^= 1                                     # the set the slot's step BEFORE the previous one read
            with torch.cuda.stream(slot.copy_stream):
                slot.copy_stream.wait_event(slot.set_done[slot.cur])      # the last step that read this input set
                slot.copy_in()
                slot.copied.record()
        with torch.cuda.stream(slot.stream):
            slot.start.record()
            if with_h2d and self.h2d_mode == "eager":
                slot.copy_in()
            elif with_h2d and self.h2d_mode == "copy_stream":
                slot.stream.wait_event(slot.copied)
            if self.use_graph:
                if (in_step, slot.cur) not in slot.graphs:
                    self._capture(slot, in_step)
                g, out = slot.graphs[(in_step, slot.cur)]
                self._replay(slot, g)
                slot.outputs = out
            else:
                slot.outputs = self._step(slot, in_step)
            if self.post_fn is not None:
                slot.outputs = self.post_fn(slot, slot.outputs)
            if self.split_solver and self.use_graph:
                with torch.cuda.stream(slot.solver_stream):
                    slot.done.record()
                    slot.set_done[slot.cur].record()
            else:
                slot.done.record()
                slot.set_done[slot.cur].record()
        slot.busy = True
        return slot.index

    def _follow_weights(self):
        """The captured graphs hold raw pointers to the classifier's PACKED operands, which the module frees and re-derives whenever its
        weights change (load_state_dict, .to(), ClassifierTrainer.optimize).  Re-pack and re-capture then, instead of replaying against
        freed or stale operands."""
        v = self.mm.detector.weights_version
        if v == self._weights_version:
            return
        torch.cuda.synchronize(self.device)           # nothing of the old graphs may still be running
        for s in self.slots:
            s.graphs.clear()
            s.busy = False
        self._replayed.clear()
        self._warmed.clear()
        self._x3_refs = []
        self.mm.detector.prepack()
        self._packed_refs = self.mm.detector.packed_operands()
        self._weights_version = self.mm.detector.weights_version
        self.weights_refreshes = getattr(self, "weights_refreshes", 0) + 1

    def result(self, ticket, wait=True):
        """Outputs of the step last submitted on slot `ticket` (device tensors owned by the slot: valid until it is reused)."""
        slot = self.slots[ticket]
        if wait:
            slot.done.synchronize()
            slot.busy = False
        return slot.outputs

    def latency_ms(self, ticket):
        """Device time of the slot's last step, first launch to last (valid after result(ticket))."""
        slot = self.slots[ticket]
        return slot.start.elapsed_time(slot.done)

    def synchronize(self):
        torch.cuda.synchronize(self.device)
        for s in self.slots:
            s.busy = False

    # ------------------------------------------------------------------------------------------------------------ convenience
    def run(self, batches, with_h2d=True):
        """Push an iterable of host batches through the executor; yields (index, outputs) in submission order, each as soon as its
        slot is needed again or the input is exhausted (outputs are cloned so they survive the slot's reuse)."""
        def take(t):
            # the clones run on the SLOT's stream: the replay that re-uses the slot is enqueued behind them (on the caller's stream nothing
            # would order the slot's next replay after the clone kernels); the caller's stream then waits for the clones
            out = self.result(t)
            slot = self.slots[t]
            with torch.cuda.stream(slot.stream):
                copy = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}
                ev = torch.cuda.Event()
                ev.record()
            torch.cuda.current_stream(self.device).wait_event(ev)
            return copy

        pending = []
        for i, b in enumerate(batches):
            if len(pending) == self.n_streams:
                j, t = pending.pop(0)
                yield j, take(t)
            pending.append((i, self.submit(b, with_h2d=with_h2d)))
        for j, t in pending:
            yield j, take(t)

    def throughput(self, steps, warmup, with_h2d, barrier=None):
        """Timed loop of the benchmark contract: `warmup` untimed steps, then exactly `steps` steps between two full synchronisations.
        -> (seconds, outputs of the last step, per-step device latencies of the timed steps in ms)."""
        self.warm_up(with_h2d)
        for _ in range(warmup):
            self.submit(None, with_h2d=with_h2d)
        if barrier is not None:
            barrier()
        self.synchronize()
        t0 = time.perf_counter()
        tickets = [self.submit(None, with_h2d=with_h2d) for _ in range(steps)]
        if barrier is not None:
            barrier()
        self.synchronize()
        dt = time.perf_counter() - t0
        last = self.slots[tickets[-1]].outputs
        lat = [self.latency_ms(t) for t in sorted(set(tickets[-self.n_streams:]))]
        return dt, last, lat
