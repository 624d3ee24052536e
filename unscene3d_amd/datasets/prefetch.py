"""The next batch's coordinate work on a side stream, while the current step's backward runs.

Voxelisation, the coordinate hash, the strided maps of the U-Net pyramid, the neighbour tables and their row orders
depend only on the scene's coordinates, and each of them ends in a count read-back.  Issued at the start of a step
on the compute stream, every read-back waits for whatever the previous step still has queued (backward, optimizer)
and the host cannot start issuing the forward pass — with host issue time and device time level that
serialises the two.  The reference hides this work in DataLoader workers (datasets/utils.py:181-219 runs on the CPU,
concurrently with the GPU step); here it runs on the GPU, on its own HIP stream: its read-backs only wait for its own
short kernels, and the step that consumes the batch finds the maps cached in the SparseTensor's coordinate manager.

Tensors allocated on the side stream are handed to the compute stream with `record_stream` (the caching allocator
then keeps a freed block away from the side stream until the compute stream is past it), and every prefetched batch
is kept referenced until the batch after it has been consumed."""
from __future__ import annotations

import atexit
import collections
import queue
import threading
import weakref

import torch

from .. import MinkowskiEngine as ME


def _record_streams(obj, stream, seen, collect=None):
    """record_stream(stream) on every HIP tensor reachable from obj (containers, object attributes, attributes hung
    on tensors such as the cached row orders of a neighbour table).  collect (a list): the tensors are appended to it
    INSTEAD (bounded_lifetime: they are kept referenced until the device has finished their step — also the ones the
    step itself lets go of early, like the decoder's key samples, which the model pops from the geometry once used)."""
    if obj is None or isinstance(obj, (int, float, str, bool, bytes, slice)) or id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            if collect is not None:
                collect.append(obj)
            else:
                obj.record_stream(stream)
        d = getattr(obj, "__dict__", None)
        if d:
            _record_streams(d, stream, seen, collect)
        return
    if isinstance(obj, dict):
        for v in obj.values():
            _record_streams(v, stream, seen, collect)
        return
    if isinstance(obj, (list, tuple, set, collections.deque)):
        for v in obj:
            _record_streams(v, stream, seen, collect)
        return
    d = getattr(obj, "__dict__", None)
    if d:
        _record_streams(d, stream, seen, collect)
    for name in getattr(type(obj), "__slots__", ()):
        _record_streams(getattr(obj, name, None), stream, seen, collect)


class ScenePrefetcher:
    """prefetch = ScenePrefetcher(collate, add_raw_coordinates, n_down);  prefetch.submit(samples) issues the work for
    the next batch (several may be in flight: take() returns them in submission order — the reference's DataLoader
    keeps prefetch_factor = 2 batches per worker ahead); prefetch.take() -> (data, target, names) with `data.sparse_tensor` (maps prepared) and
    `data.raw_coordinates` attached, ready for `InstanceSegmentation.training_step`."""

    def __init__(self, collate, add_raw_coordinates: bool = True, n_down: int = 4, ksize: int = 3, device="cuda",
                 precompute=None, threaded: bool = False, bounded_lifetime: bool = False):
        """precompute: optional `Mask3D.precompute_geometry` (bound method): the parameter-free, geometry-only part
        of the model's forward pass is then issued here as well.
        threaded: issue the batch from a worker thread (the reference's DataLoader workers, conf/data/indoor.yaml:24).
        Random draws made while issuing (the decoder's key samples, `Mask3D._draw_key_samples` -> torch.randperm on the
        default generator) then happen on the worker: they stay in batch order — one batch is issued at a time — but a
        main thread that ALSO draws from the default HIP generator during the step (dropout > 0, random query
        initialisation) interleaves with them nondeterministically: seeded runs that need the reference's draw order
        use threaded=False (or hand the model its own `randperm` source).
        The ~6 ms of host time a 150 k-voxel batch takes to issue — about a third of it blocked in the count read-backs
        of the voxel unique / coordinate maps, which release the interpreter lock — then overlap with the main thread
        issuing the step instead of extending it; submit() returns at once, take() joins.
        bounded_lifetime: the caller promises to call `retire(event)` after every step with an event that completes when
        the DEVICE has finished the step on every stream (the mark of `trainer.StepsInFlight`: recorded on the compute
        stream behind the end-of-backward joins of the lane and the key-preparation stream).  A batch is then kept
        referenced — with every tensor reachable from it at issue time, also those the step lets go of early — until its step's event has completed, and no tensor of it is `record_stream`ed: a block freed after
        that point is not in use anywhere, so it may go straight back to the prefetch stream's pool.  What that saves per
        step: the walk over the batch's ~1 800 objects on the worker (1 ms), and on the main thread the release of the
        batch of two steps ago, whose every marked block costs the allocator an event (0.6 ms inside take(), seen on
        the compute stream as a wait at every step's start on a host-bound box; `profiles/r06_step_sections.txt`)."""
        self.collate, self.add_raw, self.n_down, self.ksize = collate, add_raw_coordinates, n_down, ksize
        self.precompute = precompute
        self.device = torch.device(device)
        # a stream measured to run beside the compute stream (streams.py: two HIP streams may share a hardware queue)
        from .. import streams
        self.side = streams.pick(self.device, "prefetch")
        self.consumer = torch.cuda.default_stream(self.device)      # the stream take() is normally called on
        self._pending = collections.deque()        # batches submitted and not yet taken, oldest first
        self._keep = collections.deque(maxlen=2)
        self.bounded = bool(bounded_lifetime)
        self._live = collections.deque()           # bounded_lifetime: [batch, event of its step or None] in take() order
        self._jobs = self._worker = None
        if threaded:
            self._jobs = queue.Queue()
            self._worker = threading.Thread(target=self._serve, name="usc3d-scene-prefetch", daemon=True)
            self._worker.start()
            # a daemon thread still inside the runtime when the interpreter tears down aborts the process
            # ("terminate called without an active exception"): stop it first
            ref = weakref.ref(self)
            atexit.register(lambda: ref() is not None and ref().close())

    def _serve(self):
        torch.cuda.set_device(self.device)
        while True:
            job = self._jobs.get()
            if job is None:
                return
            samples, box, gate = job
            try:
                box["result"] = self._issue(samples, gate)
            except BaseException as err:                     # handed to the thread that calls take()
                box["error"] = err
            box["done"].set()

    def close(self):
        """Stop the worker thread (pending work is finished first); submit() falls back to the calling thread."""
        worker, self._worker = self._worker, None
        if worker is not None:
            self._jobs.put(None)
            worker.join(timeout=30)
            if worker.is_alive():
                # still inside a native call (a count read-back behind a hung stream): say so instead of pretending the
                # prefetcher is closed — the daemon thread dies with the process
                import warnings
                warnings.warn("ScenePrefetcher.close(): the worker thread did not finish within 30 s and is left "
                              "running (daemon); a batch may still be in flight on the side stream")

    def submit(self, samples, gate=None):
        """gate: an event (of the compute stream, say) that the batch's device work waits for."""
        if self._worker is not None:
            box = {"done": threading.Event()}
            self._jobs.put((samples, box, gate))
            self._pending.append(box)
            return
        self._pending.append(self._issue(samples, gate))

    @property
    def in_flight(self) -> int:
        return len(self._pending)

    def _issue(self, samples, gate=None):
        with torch.cuda.stream(self.side):
            if gate is not None:
                self.side.wait_event(gate)
            data, target, names = self.collate(samples)
            feats, raw = data.features, None
            if self.add_raw:
                raw = feats[:, -3:].contiguous()
                feats = feats[:, :-3].contiguous()
            x = ME.SparseTensor(coordinates=data.coordinates, features=feats, device=self.device)
            x.coordinate_manager.single_scene = len(samples) == 1
            x.coordinate_manager.prepare(x.tensor_stride[0], n_down=self.n_down, ksize=self.ksize)
            if self.precompute is not None and raw is not None and len(target) > 0:
                ns = [t.get("num_segments") for t in target]
                self.precompute(x, raw, [t["point2segment"] for t in target],
                                None if any(n is None or n.is_cuda for n in ns) else ns, n_levels=self.n_down + 1)
            data.sparse_tensor, data.raw_coordinates = x, raw
            done = torch.cuda.Event()
            done.record(self.side)
        batch = (data, target, names)
        # the allocator must know that the consumer's stream reads these tensors too; done HERE (on the worker thread when
        # there is one: the walk over ~1 800 objects is 1 ms of the step's host time) for the stream the consumer normally
        # is; take() repeats it only for another stream
        if self.bounded:
            held = []
            _record_streams(batch, None, set(), held)
            batch[0]._usc_held_tensors = held          # travels with the batch; dropped with it (take() / retire())
        else:
            _record_streams(batch, self.consumer, set())
        return batch, done, self.consumer.cuda_stream

    def take(self):
        if not self._pending:
            raise RuntimeError("ScenePrefetcher.take() without a submit()")
        pending = self._pending.popleft()
        if isinstance(pending, dict):                        # issued by the worker thread
            pending["done"].wait()
            if "error" in pending:
                raise pending["error"]
            pending = pending["result"]
        batch, done, recorded = pending
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        if self.bounded:
            self._drop_finished()
            if len(self._live) >= 6:
                raise RuntimeError("ScenePrefetcher(bounded_lifetime=True): retire(event) has not been called for the "
                                   "last batches — their memory cannot be released safely")
            self._live.append([batch, None])
            return batch
        if main.cuda_stream != recorded:
            _record_streams(batch, main, set())
        self._keep.append(batch)
        return batch

    def drain(self):
        """Take and drop every batch still in flight (a change of configuration between measurement loops): nothing
        consumed them, so they may go as soon as the prefetch stream has finished them."""
        while self._pending:
            self.take()
            if self.bounded:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))      # behind take()'s wait for the batch's own event
                self.retire(ev)

    def retire(self, event):
        """bounded_lifetime: `event` completes when the device has finished, on every stream, the step that consumed the
        batch handed out by the last take()."""
        for ent in self._live:
            if ent[1] is None:
                ent[1] = event
        self._drop_finished()

    def _drop_finished(self):
        while self._live and self._live[0][1] is not None and self._live[0][1].query():
            self._live.popleft()
