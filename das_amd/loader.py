"""Batches prepared ahead of the trainer.

The reference trains with `workers_per_gpu=4` (configs/das/exp_panoptic.py:159-160 -> mmdet `build_dataloader`: worker
processes decode and augment on the CPU). Here augmentation runs on the GPU (das_amd/pipelines.py), so what is left on
the host is the JPEG decode and the annotation arithmetic. Two loaders: `ProcessLoader` (further down; what
tools/train.py uses by default) keeps the reference's worker processes for the host work and replays the image ops in
the trainer; `PrefetchLoader` does everything in threads of the trainer's process:

  * `workers` background THREADS (PIL's decoder and the host-to-device copy release the GIL) build whole batches, each
    on its own HIP stream: `dataset[i]` for every index of the batch — the pipeline's GPU kernels are queued on that
    stream and run beside the training step —, `collate`, then an event;
  * batches are handed over strictly in order; the consumer's current stream waits for the batch's event and the
    tensors are registered with it (`record_stream`), so that the caching allocator cannot recycle them early;
  * a sample the pipeline dropped (no valid person left after augmentation -> None) is replaced by the next index, as
    mmdet's `CustomDataset._rand_another` re-draws;
  * an exception in a worker is re-raised by the consumer at the batch it belongs to.

With `workers=0` the same batches are built synchronously on the calling thread (what tools/train.py did before).
Random draws come from the process-wide numpy / torch generators: with one worker the sample sequence is the
synchronous one, with several it depends on thread timing (as it depends on worker seeds in the reference).
"""
import os
import queue
import sys
import threading

import torch


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


class PrefetchLoader:
    def __init__(self, dataset, batches, collate_fn, device='cuda', workers=1, depth=2, max_redraws=100,
                 switch_interval=5e-5):
        self.switch_interval = switch_interval
        self.dataset, self.batches, self.collate_fn = dataset, [list(b) for b in batches], collate_fn
        self.device = torch.device(device)
        self.workers, self.depth, self.max_redraws = max(0, int(workers)), max(1, int(depth)), max_redraws

    def __len__(self):
        return len(self.batches)

    def _build(self, indices):
        samples = []
        for i in indices:
            smp, tries = self.dataset[i], 0
            while smp is None and tries < self.max_redraws:
                i = (i + 1) % len(self.dataset)
                smp, tries = self.dataset[i], tries + 1
            if smp is None:
                raise RuntimeError(f'no valid sample within {self.max_redraws} re-draws after index {i}')
            samples.append(smp)
        return self.collate_fn(samples, device=self.device)

    def __iter__(self):
        if self.workers == 0:
            for b in self.batches:
                yield self._build(b)
            return
        cuda = self.device.type == 'cuda'
        if cuda and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        n = len(self.batches)
        fatal = []          # an exception that killed a worker outside a batch (consumer re-raises it)
        slots = [queue.Queue(maxsize=1) for _ in range(n)]       # one hand-over cell per batch: order is by index
        ticket = iter(range(n))
        lock, stop = threading.Lock(), threading.Event()
        window = threading.Semaphore(self.workers + self.depth)   # batches built but not yet consumed

        def work():
            try:
                run()
            except BaseException as e:   # noqa: BLE001 — a dead worker must not leave the consumer waiting for ever
                fatal.append(e)

        def run():
            if cuda:
                torch.cuda.set_device(self.device)
            stream = torch.cuda.Stream(device=self.device) if cuda else None
            while not stop.is_set():
                window.acquire()
                if stop.is_set():
                    return
                with lock:
                    k = next(ticket, None)
                if k is None:
                    return
                try:
                    if cuda:
                        with torch.cuda.stream(stream):
                            data = self._build(self.batches[k])
                            ready = torch.cuda.Event()
                            ready.record(stream)
                    else:
                        data, ready = self._build(self.batches[k]), None
                    slots[k].put((data, ready, None))
                except BaseException as e:   # noqa: BLE001 — handed to the consumer, which re-raises it
                    slots[k].put((None, None, e))

        threads = [threading.Thread(target=work, daemon=True, name=f'das-prefetch-{t}') for t in range(self.workers)]
        # The trainer gives the GIL up at every kernel launch (ctypes / torch release it around the call) and must get it
        # back a few microseconds later; with the interpreter's default switch interval (5 ms) a worker that is busy in
        # pure-Python / numpy code keeps it that long each time, and ~700 eager launches per step queue up behind the
        # workers (measured: 160 img/s alone -> 110 img/s beside 2...8 workers). A short interval hands it back at once.
        interval = sys.getswitchinterval()
        sys.setswitchinterval(min(interval, self.switch_interval))
        for t in threads:
            t.start()
        try:
            for k in range(n):
                while True:
                    try:
                        data, ready, err = slots[k].get(timeout=0.2)
                        break
                    except queue.Empty:
                        if fatal:
                            raise RuntimeError('a prefetch worker died') from fatal[0]
                        if not any(t.is_alive() for t in threads) and slots[k].empty():
                            raise RuntimeError(f'all prefetch workers exited before batch {k} was built')
                window.release()
                if err is not None:
                    raise err
                if ready is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ready)
                    for t in _tensors(data):
                        if t.is_cuda:
                            t.record_stream(cur)
                yield data
        finally:
            stop.set()
            for _ in threads:
                window.release()
            sys.setswitchinterval(interval)


# ---------------------------------------------------------------------------------------------------------------------
# Worker PROCESSES: the reference's dataloader model (mmdet build_dataloader -> torch DataLoader workers).
#
# What the measurements say (tools/dev/loader_bench.py, tools/dev/scripts/loader_prof.sh, profiles/r03_loader_bench.txt):
#   * the data path needs 0.2 ms of GPU time per 1920 x 1080 frame (upload included) and 13 ms of CPU time, of which
#     10 ms JPEG decode (no interpreter lock) and 2.5 ms Python / numpy (annotation arithmetic, stage plumbing);
#   * loader THREADS beside the trainer: 100-120 img/s against 162 img/s for either alone — with 3-4 of 16 CPUs busy,
#     no cgroup throttling, page-locked or pageable uploads alike: the trainer holds the interpreter lock for 40-60 ms
#     of a 90 ms step, the loader's 16 x 2.5 ms come on top, and every hand-over of the lock costs both sides;
#   * worker processes that run the WHOLE pipeline, GPU ops included, each in its own HIP context: 130-200 img/s alone,
#     slower with 8 workers than with 4 (the GPU's queues are time-sliced between processes), 125 img/s beside the
#     trainer — whether a batch travels as ~100 IPC tensors or through a shared ring.
# So the split is by resource, not by stage: worker processes do everything that needs a CPU and NO GPU — decode, the
# random draws, the annotation arithmetic of every stage — and never create a HIP context; the image ops of the stages
# are RECORDED (das_amd.image_ops.FramePlan: the same stage code, `pipelines.DEFER_IMAGE_OPS`) and replayed by the
# trainer on a side stream, one batch ahead of the step that consumes it (~130 launches, ~3 ms of GPU time per batch).
# Decoded frames travel through a ring of shared host memory per worker (a file in /dev/shm mapped by both sides and
# page-locked once by the trainer: the upload is an asynchronous DMA straight out of it, 0.12 ms per frame); annotations
# travel as numpy arrays inside the queue message and go up in one copy (`datasets.pack_to_device`).
class _FrameRing:
    """`slots` frames of `slot_bytes` in shared host memory; `create` (trainer: also page-locks it) / `attach` (worker)."""

    def __init__(self, path, slots, slot_bytes, owner):
        import mmap
        import numpy as np
        self.path, self.slots, self.slot_bytes, self.owner = path, slots, slot_bytes, owner
        self._fd = os.open(path, os.O_RDWR)
        self._mm = mmap.mmap(self._fd, slots * slot_bytes)
        self.bytes = np.frombuffer(self._mm, dtype=np.uint8)
        self.registered = False

    @classmethod
    def create(cls, slots, slot_bytes, pin, min_slots=8):
        """A ring file of `slots` x `slot_bytes` in /dev/shm (RAM-backed; page-locking it later allocates every page).
        The capacity is checked first: on a small /dev/shm (64 MiB in a default container) the ring shrinks to what
        fits — keeping a quarter of the free space for others — down to `min_slots`; below that the file goes to the
        ordinary temp directory (the page cache: uploads are then staged copies, slower but correct). Never a SIGBUS in a
        worker on its first frame write."""
        import tempfile
        import warnings
        where = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None
        if where is not None:
            try:
                st = os.statvfs(where)
                free = st.f_bavail * st.f_frsize
            except OSError:
                free = 0
            fit = int(free * 3 // 4 // slot_bytes)
            if fit < slots:
                if fit >= min_slots:
                    warnings.warn(f'das_amd.loader: /dev/shm has {free >> 20} MiB free, frame ring shrunk from {slots} to {fit} '
                                  f'slots of {slot_bytes >> 20} MiB (fewer frames in flight per worker)')
                    slots = fit
                else:
                    warnings.warn(f'das_amd.loader: /dev/shm has {free >> 20} MiB free, too little for a frame ring of even '
                                  f'{min_slots} x {slot_bytes >> 20} MiB: the ring lives in {tempfile.gettempdir()} instead')
                    where, slots = None, max(min_slots, min(slots, 16))
        fd, path = tempfile.mkstemp(prefix='das_frames_', dir=where)
        try:
            os.posix_fallocate(fd, 0, slots * slot_bytes)    # (fails here, with ENOSPC, instead of a SIGBUS at first touch)
        except OSError:
            os.close(fd)
            os.unlink(path)
            raise
        os.close(fd)
        ring = cls(path, slots, slot_bytes, True)
        if pin:
            ring.pin()
        return ring

    def pin(self):
        """Page-lock the ring for asynchronous uploads (needs a GPU context: called when the loader is bound to its device)."""
        if not self.registered:
            t = torch.from_numpy(self.bytes)
            self.registered = int(torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel(), 0)) == 0
            if not self.registered:
                import warnings
                warnings.warn('das_amd.loader: could not page-lock the frame ring; uploads fall back to staged copies')
        return self.registered

    def unlink(self):
        """Remove the backing file's name (both sides keep their mappings): it can no longer outlive the processes."""
        if self.owner and self.path is not None:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            self.path = None

    def slot(self, i, nbytes):
        return self.bytes[i * self.slot_bytes:i * self.slot_bytes + nbytes]

    def close(self):
        if self._mm is None:
            return
        if self.registered:
            try:
                torch.cuda.cudart().cudaHostUnregister(torch.from_numpy(self.bytes).data_ptr())
            except Exception:   # noqa: BLE001
                pass
        self.bytes = None
        try:
            self._mm.close()
        except BufferError:     # (a numpy view is still alive somewhere: the mapping goes with the process)
            pass
        self._mm = None
        os.close(self._fd)
        self.unlink()


def _process_worker(wid, dataset_cfg, seed, max_redraws, ring_spec, task_q, free_q, result_q):
    import queue as _q
    import traceback
    import numpy as np
    try:
        from . import pipelines
        from .datasets import ANNOTATION_KEYS, build_dataset
        from .image_ops import FramePlan
        pipelines.DEFER_IMAGE_OPS = True          # (this process never touches the GPU)
        np.random.seed(seed + wid)
        torch.manual_seed(seed + wid)
        dataset = build_dataset(dataset_cfg)
        ring = _FrameRing(*ring_spec, owner=False)
        free = list(range(ring.slots))
        builder = PrefetchLoader(dataset, [], lambda smp, device=None: smp, device='cpu', workers=0, max_redraws=max_redraws)
        result_q.put(('ready', wid, len(dataset)))
    except BaseException:   # noqa: BLE001
        result_q.put(('fatal', wid, traceback.format_exc()))
        return

    def take_slot():
        while not free:                           # every slot still waits for its upload: block until one comes back
            free.extend(free_q.get())
        return free.pop()

    while True:
        task = task_q.get()
        if task is None:
            ring.close()
            return
        call, k, indices = task
        taken = []
        try:
            while True:                           # slots handed back in the meantime
                try:
                    free.extend(free_q.get_nowait())
                except _q.Empty:
                    break
            out = []
            for smp in builder._build(indices):
                smp = dict(smp)
                img = smp['img']
                if isinstance(img, FramePlan):
                    frame = np.ascontiguousarray(img.frame)
                    if frame.nbytes <= ring.slot_bytes and len(indices) <= ring.slots:
                        i = take_slot()
                        taken.append(i)
                        ring.slot(i, frame.nbytes)[:] = frame.reshape(-1).view(np.uint8)
                        smp['img'] = ('ring', i, frame.shape, frame.dtype.str, img.with_frame(None))
                    else:                         # (a frame larger than a slot travels inside the message)
                        smp['img'] = ('plan', img)
                elif torch.is_tensor(img):
                    smp['img'] = ('array', img.numpy())
                else:
                    smp['img'] = ('array', np.asarray(img))
                for key in ANNOTATION_KEYS:
                    if key in smp and torch.is_tensor(smp[key]):
                        smp[key] = smp[key].numpy()
                out.append(smp)
            result_q.put(('ok', (call, k), (wid, out)))
        except BaseException:   # noqa: BLE001 — reported to the consumer, which raises at that batch
            free.extend(taken)
            result_q.put(('err', (call, k), traceback.format_exc()))


class ProcessLoader:
    """`workers` CPU-only processes preparing batches of `build_dataset(dataset_cfg)`; `batches(list of index lists)`
    yields the collated batches on GPU `device`, in order, each prepared one step ahead. The pool lives until `close()`
    (start-up costs a second or two per worker: interpreter, imports), so one loader serves every epoch.
    `max_frame_bytes` x `ring_slots` of shared, page-locked host memory per worker hold the decoded frames in flight
    (shrunk when /dev/shm is small: `_FrameRing.create`).
    `device=None`: start the worker processes NOW and touch no GPU — a trainer creates the loader before its first GPU
    call (starting processes out of a GPU-initialised parent is the one process-creation pattern to avoid on some
    hosts) and calls `bind(device)` once the model is on the device; with a device given, both happen here."""

    def __init__(self, dataset_cfg, device='cuda', workers=4, depth=2, seed=0, max_redraws=100, start_timeout=300,
                 max_frame_bytes=8 << 20, ring_slots=48):
        import torch.multiprocessing as mp
        self.device, self.side = None, None
        self.workers, self.depth = max(1, int(workers)), max(1, int(depth))
        self.rings = [_FrameRing.create(ring_slots, max_frame_bytes, pin=False) for _ in range(self.workers)]
        ring_slots = min(r.slots for r in self.rings)
        ctx = mp.get_context('spawn')
        self.task_q, self.result_q = ctx.Queue(), ctx.Queue()
        self.free_qs = [ctx.Queue() for _ in range(self.workers)]
        self.procs = [ctx.Process(target=_process_worker, daemon=True, name=f'das-loader-{w}',
                                  args=(w, dataset_cfg, seed, max_redraws,
                                        (self.rings[w].path, self.rings[w].slots, max_frame_bytes), self.task_q, self.free_qs[w],
                                        self.result_q))
                      for w in range(self.workers)]
        for p in self.procs:
            p.start()
        self.length = None
        self.host_seconds = 0.0  # time this (the trainer's) thread spent uploading / replaying image ops (diagnostic)
        self.wait_seconds = 0.0  # ... and blocked because the batch it needed had not arrived yet
        self.late_batches = 0    # batches that could not be prepared one step ahead
        self._lent = []         # (worker, [slots], event): frames whose upload may still be running
        self._call = 0          # batches() call counter: results of an abandoned call are recognised and dropped
        for _ in self.procs:
            kind, wid, info = self._get(start_timeout)
            if kind != 'ready':
                self.close()
                raise RuntimeError(f'loader worker {wid} failed to start:\n{info}')
            self.length = info
        for r in self.rings:     # every worker has mapped its ring: the files' names can go (a crashed trainer leaks nothing)
            r.unlink()
        if device is not None:
            self.bind(device)

    def bind(self, device='cuda'):
        """Attach the loader to its GPU: side stream for the uploads / recorded image ops, page-locked rings."""
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        with torch.cuda.device(self.device):
            self.side = torch.cuda.Stream()
            for r in self.rings:
                r.pin()
        return self

    def _get(self, timeout, block=True):
        import queue as _q
        waited = 0.0
        while True:
            try:
                return self.result_q.get(timeout=0.05) if block else self.result_q.get_nowait()
            except _q.Empty:
                if not block:
                    return None
                waited += 0.05
                self._return_slots()       # (a worker may be waiting for exactly these)
                dead = [p.name for p in self.procs if not p.is_alive()]
                if dead:
                    raise RuntimeError(f'loader worker process(es) died: {dead}')
                if waited >= timeout:
                    raise RuntimeError(f'no batch from the loader workers within {timeout} s')

    def _return_slots(self, wait=False):
        keep = []
        for wid, slots, ev in self._lent:
            if wait:
                ev.synchronize()
            if ev.query():
                self.free_qs[wid].put(slots)
            else:
                keep.append((wid, slots, ev))
        self._lent = keep

    @staticmethod
    def _slots_of(msg):
        return [s['img'][1] for s in msg[1] if s['img'][0] == 'ring']

    def _materialize(self, msg):
        """Uploads + the recorded image ops of one batch on the side stream; returns (batch dict, event)."""
        import time
        import numpy as np
        from .datasets import ANNOTATION_KEYS, pack_to_device
        from .image_ops import FramePlan
        t_start = time.perf_counter()
        wid, samples = msg
        ring = self.rings[wid]
        plans = []
        for s in samples:
            kind = s['img'][0]
            if kind == 'ring':
                _, i, shape, dt, plan = s['img']
                n = int(np.prod(shape)) * np.dtype(dt).itemsize
                plans.append(plan.with_frame(ring.slot(i, n).view(np.dtype(dt)).reshape(shape)))
            elif kind == 'plan':
                plans.append(s['img'][1])
            else:
                plans.append(s['img'][1])
        shapes = [tuple(p.shape) for p in plans]
        hm, wm = max(sh[-2] for sh in shapes), max(sh[-1] for sh in shapes)
        same = len(set(shapes)) == 1
        with torch.cuda.device(self.device), torch.cuda.stream(self.side):
            img = None
            for b, p in enumerate(plans):
                if isinstance(p, FramePlan):
                    direct = same and bool(p.ops) and p.ops[-1][0] == 'normalize_pad_chw'
                    if direct and img is None:
                        img = torch.empty((len(plans),) + shapes[0], dtype=torch.float32, device=self.device)
                    t = p.run(self.device, out=img[b] if direct else None)
                    if direct:
                        continue
                else:
                    t = torch.from_numpy(p).to(self.device, non_blocking=True)
                if img is None:
                    img = (torch.empty if same else torch.zeros)((len(plans), t.shape[0], hm, wm), dtype=t.dtype,
                                                                 device=self.device)
                img[b, :, :t.shape[-2], :t.shape[-1]] = t
            keys = [k for k in ANNOTATION_KEYS if k in samples[0]]
            flat = pack_to_device([np.asarray(s[k]) for k in keys for s in samples], self.device)
            ev = torch.cuda.Event()
            ev.record(self.side)
        slots = self._slots_of(msg)
        if slots:
            self._lent.append((wid, slots, ev))
        out = dict(img=img, img_metas=[s['img_metas'] for s in samples])
        for j, k in enumerate(keys):
            out[k] = flat[j * len(samples):(j + 1) * len(samples)]
        self.host_seconds += time.perf_counter() - t_start
        return out, ev

    def batches(self, batches, timeout=600):
        if self.side is None:
            self.bind('cuda')
        batches = [list(b) for b in batches]
        n, sent, ready, prepared = len(batches), 0, {}, {}
        window = self.workers + self.depth
        self._call += 1
        call = self._call

        def take(block):
            got = self._get(timeout, block)
            if got is None:
                return False
            kind, kk, data = got
            if kind == 'fatal':
                raise RuntimeError(f'loader worker died:\n{data}')
            if kk[0] != call:                     # left over from a call that was abandoned half way
                if kind == 'ok' and self._slots_of(data):
                    self.free_qs[data[0]].put(self._slots_of(data))
                return True
            if kind == 'err':
                raise RuntimeError(f'loader worker failed on batch {kk[1]}:\n{data}')
            ready[kk[1]] = data
            return True

        try:
            for k in range(n):
                while sent < n and sent < k + window:
                    self.task_q.put((call, sent, batches[sent]))
                    sent += 1
                if k not in prepared:
                    import time
                    t_wait = time.perf_counter()
                    while k not in ready:
                        take(True)
                    self.wait_seconds += time.perf_counter() - t_wait
                    self.late_batches += 1
                    prepared[k] = self._materialize(ready.pop(k))
                while take(False):                # whatever else has arrived
                    pass
                if k + 1 < n and k + 1 in ready:  # the next batch's GPU work is queued BEFORE this step's
                    prepared[k + 1] = self._materialize(ready.pop(k + 1))
                self._return_slots()
                data, ev = prepared.pop(k)
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for t in _tensors(data):
                    if t.is_cuda:
                        t.record_stream(cur)
                yield data
        finally:
            # batches received but not consumed (early exit, an error): their frames go back unread
            for msg in ready.values():
                if self._slots_of(msg):
                    self.free_qs[msg[0]].put(self._slots_of(msg))
            self._return_slots(wait=True)

    def close(self):
        for _ in self.procs:
            try:
                self.task_q.put(None)
            except Exception:   # noqa: BLE001
                pass
        for p in self.procs:
            p.join(5)
            if p.is_alive():
                p.terminate()
        self.procs = []
        for r in getattr(self, 'rings', []):
            r.close()
        self.rings = []

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
