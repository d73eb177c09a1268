"""Batches prepared ahead of the trainer.

The reference trains with `workers_per_gpu=4` (configs/das/exp_panoptic.py:159-160 -> mmdet `build_dataloader`: worker
processes decode and augment on the CPU). Here augmentation already runs on the GPU (das_amd/pipelines.py), so what is
left on the host is the JPEG decode and the annotation arithmetic; `PrefetchLoader` moves both off the training thread:

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
# Worker PROCESSES: the reference's dataloader model (mmdet build_dataloader -> torch DataLoader workers), with the
# augmentation on the GPU. Threads share the trainer's interpreter lock: ~2.5 ms of Python per sample in the pipeline's
# stages, 16 samples per step, beside a trainer that has to stay ahead of the GPU through ~700 eager launches per step —
# measured: the trainer alone 162 img/s, beside 2-8 loader threads 105-120 img/s (tools/dev/loader_bench.py --probe: the
# same threads doing nothing cost 0 %, decode only 9 %, the full pipeline 27 %). Each worker process has its own
# interpreter and its own HIP context on the same GPU, builds whole batches there and hands the collated CUDA tensors over
# by IPC handle (torch.multiprocessing reductions; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver).
def _process_worker(wid, dataset_cfg, device_index, seed, max_redraws, task_q, result_q):
    import traceback
    import numpy as np
    try:
        torch.cuda.set_device(device_index)
        from .datasets import build_dataset, collate
        np.random.seed(seed + wid)
        torch.manual_seed(seed + wid)
        dataset = build_dataset(dataset_cfg)
        builder = PrefetchLoader(dataset, [], collate, device=f'cuda:{device_index}', workers=0, max_redraws=max_redraws)
        result_q.put(('ready', wid, len(dataset)))
    except BaseException:   # noqa: BLE001
        result_q.put(('fatal', wid, traceback.format_exc()))
        return
    held = []            # the last batches sent: kept alive here until the consumer had time to map them
    while True:
        task = task_q.get()
        if task is None:
            return
        k, indices = task
        try:
            data = builder._build(indices)
            torch.cuda.synchronize()          # the consumer's stream knows nothing of this process's streams
            result_q.put(('ok', k, data))
            held.append(data)
            del held[:-4]
        except BaseException:   # noqa: BLE001 — reported to the consumer, which raises at that batch
            result_q.put(('err', k, traceback.format_exc()))


class ProcessLoader:
    """`workers` processes building batches of `build_dataset(dataset_cfg)` on GPU `device`; `batches(list of index
    lists)` yields the collated batches in order. The pool lives until `close()` (start-up costs a few seconds per
    worker: interpreter, torch import, HIP context), so one loader serves every epoch."""

    def __init__(self, dataset_cfg, device='cuda', workers=4, depth=2, seed=0, max_redraws=100, start_timeout=300):
        import torch.multiprocessing as mp
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.workers, self.depth = max(1, int(workers)), max(1, int(depth))
        ctx = mp.get_context('spawn')
        self.task_q, self.result_q = ctx.Queue(), ctx.Queue()
        self.procs = [ctx.Process(target=_process_worker, daemon=True, name=f'das-loader-{w}',
                                  args=(w, dataset_cfg, self.device.index, seed, max_redraws, self.task_q, self.result_q))
                      for w in range(self.workers)]
        for p in self.procs:
            p.start()
        self.length = None
        for _ in self.procs:
            kind, wid, info = self._get(start_timeout)
            if kind != 'ready':
                self.close()
                raise RuntimeError(f'loader worker {wid} failed to start:\n{info}')
            self.length = info

    def _get(self, timeout):
        import queue as _q
        waited = 0.0
        while True:
            try:
                return self.result_q.get(timeout=0.5)
            except _q.Empty:
                waited += 0.5
                dead = [p.name for p in self.procs if not p.is_alive()]
                if dead:
                    raise RuntimeError(f'loader worker process(es) died: {dead}')
                if waited >= timeout:
                    raise RuntimeError(f'no batch from the loader workers within {timeout} s')

    def batches(self, batches, timeout=600):
        batches = [list(b) for b in batches]
        n, sent, ready = len(batches), 0, {}
        window = self.workers + self.depth
        for k in range(n):
            while sent < n and sent < k + window:
                self.task_q.put((sent, batches[sent]))
                sent += 1
            while k not in ready:
                kind, kk, data = self._get(timeout)
                if kind == 'err':
                    raise RuntimeError(f'loader worker failed on batch {kk}:\n{data}')
                if kind == 'fatal':
                    raise RuntimeError(f'loader worker died:\n{data}')
                ready[kk] = data
            yield ready.pop(k)

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

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
