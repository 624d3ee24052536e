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


def _record_streams(obj, stream, seen):
    """record_stream(stream) on every HIP tensor reachable from obj (containers, object attributes, attributes hung
    on tensors such as the cached row orders of a neighbour table)."""
    if obj is None or isinstance(obj, (int, float, str, bool, bytes, slice)) or id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
        d = getattr(obj, "__dict__", None)
        if d:
            _record_streams(d, stream, seen)
        return
    if isinstance(obj, dict):
        for v in obj.values():
            _record_streams(v, stream, seen)
        return
    if isinstance(obj, (list, tuple, set, collections.deque)):
        for v in obj:
            _record_streams(v, stream, seen)
        return
    d = getattr(obj, "__dict__", None)
    if d:
        _record_streams(d, stream, seen)
    for name in getattr(type(obj), "__slots__", ()):
        _record_streams(getattr(obj, name, None), stream, seen)


class ScenePrefetcher:
    """prefetch = ScenePrefetcher(collate, add_raw_coordinates, n_down);  prefetch.submit(samples) issues the work for
    the next batch (several may be in flight: take() returns them in submission order — the reference's DataLoader
    keeps prefetch_factor = 2 batches per worker ahead); prefetch.take() -> (data, target, names) with `data.sparse_tensor` (maps prepared) and
    `data.raw_coordinates` attached, ready for `InstanceSegmentation.training_step`."""

    def __init__(self, collate, add_raw_coordinates: bool = True, n_down: int = 4, ksize: int = 3, device="cuda",
                 precompute=None, threaded: bool = False):
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
        issuing the step instead of extending it; submit() returns at once, take() joins."""
        self.collate, self.add_raw, self.n_down, self.ksize = collate, add_raw_coordinates, n_down, ksize
        self.precompute = precompute
        self.device = torch.device(device)
        # a stream measured to run beside the compute stream (streams.py: two HIP streams may share a hardware queue)
        from .. import streams
        self.side = streams.pick(self.device, "prefetch")
        self.consumer = torch.cuda.default_stream(self.device)      # the stream take() is normally called on
        self._pending = collections.deque()        # batches submitted and not yet taken, oldest first
        self._keep = collections.deque(maxlen=2)
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
        if main.cuda_stream != recorded:
            _record_streams(batch, main, set())
        self._keep.append(batch)
        return batch
