"""Two clips per GPU, step by step on two HIP streams (independent trajectories; review r4 item 7).

Every kernel of the denoising graph fills the MI355X by itself, so one stream leaves idle what a kernel cannot use - the partial last
round of a persistent GEMM, the launch gaps, the matrix pipe under an HBM-bound normalisation.  A second, INDEPENDENT trajectory on a
second stream can fill it - or get in its way: the builder's boxes measured +3.0 ... +8.6 % aggregate DDIM steps/s for two 576x1024x25
clips (tools/two_stream_ab.py, profiles/r05q, r05ac, r05al, r05at), the driver's box of round 5 MINUS 8.7 % (BENCH_r05.json
extra.two_clips_per_gpu: 5.329 steps/s one after the other, 4.865 on two streams, gain 0.913).  The sign depends on the box, so the mode
is OPT-IN (VCX_CLIPS_PER_GPU=2; default 1 = one clip after the other); outputs are bit-identical either way.  Within one trajectory
there is nothing independent to run (profiles/r05_experiments.md section 4).

How: each clip runs the UNMODIFIED driver code (image_guided_synthesis: encoders, DDIM loop, decode) in its own host thread under its own
`torch.cuda.stream`; a baton makes the threads take turns, and the sampler hands the baton on after every DDIM step (`step_yield`), so the
host queues step i of clip A on stream A, then step i of clip B on stream B, ... and the GPU always has two streams to draw from.  The
global random generators (CPU and current CUDA device) are part of a lane's context: saved when it gives the baton away, restored when it
gets it back, so every clip sees exactly the sequence of draws it would see running alone after `torch.manual_seed(seed + index)` - the
results do not depend on whether, or with whom, a clip shared the GPU.

Shared model state: the lanes run the SAME model, whose kernel-layout weight packs are built lazily by the first forward that needs them
(PackedModule.packed()).  `prepack()` builds them all on the caller's stream before any lane starts (run_sharded does that when it is
given the model); independently of that, a lane that takes the baton for the first time - and every lane that takes it from a lane that
has just finished (its VAE decode may have built packs) - makes its stream wait for an event recorded on the yielding lane's stream, so
a pack written on stream A is never read on stream B before the kernels that wrote it have run."""
import threading

import torch

_tls = threading.local()


def step_yield():
    """Called by the samplers after every DDIM step: hands the baton to the other lane (no-op outside run_interleaved)."""
    il = getattr(_tls, "interleaver", None)
    if il is not None:
        il._switch(_tls.lane)


class _Interleaver:
    def __init__(self, n_lanes, streams=None):
        self.cv = threading.Condition()
        self.streams = streams        # the lanes' HIP streams (None on a GPU-less host)
        self.turn = 0
        self.alive = [True] * n_lanes
        self.rng = [None] * n_lanes
        self.cuda = torch.cuda.is_available() and streams is not None and streams[0] is not None
        self.handoff = None           # event recorded on the yielding lane's stream at its last hand-over
        self.sync_next = True         # the next lane to take the baton waits for it (first hand-over to each lane, hand-over from a finished lane)
        self.seen = [False] * n_lanes

    def _record(self, k, retiring):
        """(under self.cv) the yielding lane k marks where its stream stands"""
        if self.cuda:
            self.handoff = torch.cuda.Event()
            self.handoff.record(self.streams[k])
            if retiring:
                self.sync_next = True

    def _wait_handoff(self, k):
        """(under self.cv) lane k has the baton: order its stream behind the yielding lane's work where packs may have been built"""
        if self.cuda and self.handoff is not None and (self.sync_next or not self.seen[k]):
            self.streams[k].wait_event(self.handoff)
        self.sync_next = False
        self.seen[k] = True

    def _save(self, k):
        self.rng[k] = (torch.random.get_rng_state(), torch.cuda.get_rng_state() if self.cuda else None)

    def _load(self, k):
        if self.rng[k] is not None:
            torch.random.set_rng_state(self.rng[k][0])
            if self.cuda:
                torch.cuda.set_rng_state(self.rng[k][1])

    def _next_alive(self, k):
        n = len(self.alive)
        for d in range(1, n + 1):
            j = (k + d) % n
            if self.alive[j]:
                return j
        return k

    def _acquire(self, k):          # first entry of lane k: wait for the baton
        with self.cv:
            while self.turn != k:
                self.cv.wait()
            self._load(k)
            self._wait_handoff(k)

    def _switch(self, k):
        with self.cv:
            nxt = self._next_alive(k)
            if nxt == k:
                return
            self._save(k)
            self._record(k, False)
            self.turn = nxt
            self.cv.notify_all()
            while self.turn != k:
                self.cv.wait()
            self._load(k)
            self._wait_handoff(k)

    def _retire(self, k):
        with self.cv:
            self._save(k)
            self._record(k, True)
            self.alive[k] = False
            self.turn = self._next_alive(k)
            self.cv.notify_all()


def prepack(model):
    """Build every lazily built kernel-layout pack below `model` now, on the current stream (PackedModule.packed()), so that no lane of
    run_interleaved is the first to need one.  Modules whose pack depends on call-time information keep building it on demand."""
    from .lvdm.modules.attention import PackedModule
    if model is None:
        return 0
    n = 0
    for m in model.modules():
        if isinstance(m, PackedModule):
            try:
                m.packed()
                n += 1
            except NotImplementedError:
                pass
    return n


def run_interleaved(fn, items, n_lanes=2):
    """[fn(item, index) for (index, item) in items], `n_lanes` of them in flight at a time on their own streams, taking turns at every
    `step_yield()`.  `items` = list of (index, item).  Results in the order of `items`; the first exception of a lane is re-raised.
    The global generators are left as after the LAST item of the list, as a plain loop would leave them."""
    items = list(items)
    out = [None] * len(items)
    if n_lanes < 2 or len(items) < 2:
        for pos, (index, item) in enumerate(items):
            out[pos] = fn(item, index)
        return out
    cuda = torch.cuda.is_available()
    main = torch.cuda.current_stream() if cuda else None
    final_rng = None
    for base in range(0, len(items), n_lanes):
        group = items[base:base + n_lanes]
        errors = [None] * len(group)
        streams = [torch.cuda.Stream() for _ in group] if cuda else [None] * len(group)
        il = _Interleaver(len(group), streams)
        device = torch.cuda.current_device() if cuda else None
        # thread-local torch state of the caller that a new thread does not inherit (a caller's `with torch.no_grad()` must hold in the lanes)
        grad_on, infer_on = torch.is_grad_enabled(), torch.is_inference_mode_enabled()

        def lane(k):
            _tls.interleaver, _tls.lane = il, k
            try:
                with torch.inference_mode(infer_on), torch.set_grad_enabled(grad_on):
                    if cuda:
                        torch.cuda.set_device(device)
                        streams[k].wait_stream(main)
                        with torch.cuda.stream(streams[k]):
                            il._acquire(k)
                            index, item = group[k]
                            out[base + k] = fn(item, index)
                    else:
                        il._acquire(k)
                        index, item = group[k]
                        out[base + k] = fn(item, index)
            except BaseException as e:      # noqa: BLE001 - re-raised by the caller
                errors[k] = e
            finally:
                _tls.interleaver = None
                il._retire(k)
        threads = [threading.Thread(target=lane, args=(k,), name=f"vcx-clip-{group[k][0]}") for k in range(len(group))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        for e in errors:
            if e is not None:
                raise e
        if cuda:
            for k, s in enumerate(streams):
                main.wait_stream(s)
                r = out[base + k]
                for t in (r if isinstance(r, (list, tuple)) else [r]):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(main)
        final_rng = il.rng[len(group) - 1]
    if final_rng is not None:       # what a plain loop leaves behind: the state after the last item
        torch.random.set_rng_state(final_rng[0])
        if cuda and final_rng[1] is not None:
            torch.cuda.set_rng_state(final_rng[1])
    return out
