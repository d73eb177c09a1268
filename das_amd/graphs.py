"""HIP graphs for the static part of the training step.

The train step issues ~2700 kernel launches; the host needs 60-90 ms to queue them, about as long as the GPU needs to
run them, and every Python thread beside the trainer (the prefetching loader) makes it slower. Two thirds of those
launches belong to the backbone + neck, whose shapes, addresses and launch parameters do not depend on the data: the
forward pass of that trunk is captured once into one hipGraph and its backward pass into a second one (the same
kernels, launched by the same code — capture only records them), and a step then replays two graphs instead of
queueing ~2000 launches. The head, the losses and their backward depend on the number of positive locations and
stay eager; their launches are queued while the trunk's graph runs.

What makes the capture legal:
  * nothing in the trunk synchronises with the host or allocates device memory through HIP once warm: workspaces
    (per stream: the warm-up runs on the capture stream), kernel attributes, weight-gradient schedules are set up by
    two eager warm-up iterations; tensors come from the graph's private pool of the caching allocator;
  * Python-side state that decides WHAT is launched is pinned for the capture: the per-step caches of packed weights
    are invalidated first (so the packing launches are part of the graph and run at every replay), the zero-filled
    statistics arena is a private one whose memset is the graph's first node, the weight gradients' side stream is
    off (everything on the capture stream);
  * gradients of the trunk's parameters are added into the optimizer's flat gradient buffer by the captured kernels
    themselves (fixed addresses). The Python completion hooks that drive the overlapped all-reduce do not run at
    replay; with several ranks `_Replay.backward` reports the trunk's parameters complete right after it has queued
    the backward graph (`FlatSGD.mark_complete`), so the head's buckets (complete before the trunk's backward starts)
    go out while the graph runs and the trunk's buckets right behind it, in the same fixed order on every rank. What
    the graph gives up is the overlap of the TRUNK's buckets with the trunk's own backward (~0.5 GB of gradients:
    a few ms on xGMI); a backward captured in bucket-aligned segments would keep it (not built: the four MSPN stages
    exchange three skip tensors per level). SyncBN layers all-reduce inside the trunk: a trunk that contains one is
    not captured with several ranks (the collective would have to be part of the graph).
The warm-up's side effects (BatchNorm running statistics, gradients) are undone after the capture — also when the
capture is refused (the model is left exactly as it was)."""
import torch

from . import autograd as ag
from . import nn as dnn


# Capture the weight gradients on their side streams (forked from / joined to the capture stream by events: the graph
# then has parallel branches, as the launch-by-launch path has) instead of serially on the capture stream.
CAPTURE_SIDE_STREAM = False


class _Replay(torch.autograd.Function):
    """Forward graph now, backward graph when the feature maps' gradients arrive. `anchor` is a dummy input that
    requires grad: the image does not, and without a differentiable input autograd would not call backward."""

    @staticmethod
    def forward(ctx, img, anchor, trunk):
        ctx.trunk = trunk
        trunk.x.copy_(img)
        trunk.fwd.replay()
        trunk.mark_packed()
        return tuple(o.detach() for o in trunk.outs)

    @staticmethod
    def backward(ctx, *gouts):
        trunk = ctx.trunk
        for dst, g in zip(trunk.gouts, gouts):
            if g is None:
                dst.zero_()
            else:
                dst.copy_(g)
        trunk.bwd.replay()
        trunk.opt.mark_complete(trunk.slots)     # (several ranks: the trunk's gradient buckets may go out now)
        return None, None, None


class GraphedTrunk:
    """model.extract_feat (backbone + neck) in training mode as two hipGraphs. `trunk(img)` returns the feature maps
    (autograd-connected: their gradients trigger the backward graph); None if `img` does not match the captured
    batch (the caller runs the eager path)."""

    def __init__(self, model, optimizer, img, warmup=2):
        assert model.training and img.is_cuda
        world = torch.distributed.get_world_size() if torch.distributed.is_available() and \
            torch.distributed.is_initialized() else 1
        if world != 1 and any(getattr(m, '_das_sync', False) for t in (model.backbone, model.neck) if t is not None
                              for m in t.modules()):
            raise RuntimeError('GraphedTrunk: the trunk holds SyncBN layers, whose statistics all-reduce cannot be part of '
                               'the captured graph; use the eager path (or norm_cfg type BN) with several ranks')
        self.model, self.opt = model, optimizer
        self.shape, self.dtype = tuple(img.shape), img.dtype
        dev = img.device
        params = [p for m in (model.backbone, model.neck) if m is not None for p in m.parameters() if p.requires_grad]
        self.slots = [p._das_slot for p in params if getattr(p, '_das_slot', None) is not None]
        # state the warm-up iterations would leave behind
        buffers = [b for m in (model.backbone, model.neck) if m is not None for b in m.buffers()]
        keep_buf = [b.detach().clone() for b in buffers]
        keep_g = optimizer.flat_g.detach().clone()
        side_was = ag.WGRAD_SIDE_STREAM
        if not CAPTURE_SIDE_STREAM:
            ag.WGRAD_SIDE_STREAM = False
        overlap_was, optimizer.overlap = optimizer.overlap, False   # (no collective from the warm-up / capture passes)
        ok = False
        arena_was = dnn._STATS_ARENA
        self.arena = dnn._ZeroArena()
        self.x = img.detach().clone()
        self.anchor = torch.zeros(1, device=dev, requires_grad=True)
        self.stream = torch.cuda.Stream(device=dev)
        try:
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            import warnings
            with torch.cuda.stream(self.stream), warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter('always')
                for _ in range(warmup):
                    outs = model.extract_feat(self.x)
                    self._backward(outs, [torch.full_like(o, 1e-3) for o in outs], params)
                    del outs
            if any('AccumulateGrad' in str(w.message) for w in caught):
                # An autograd graph of an earlier step is still referenced somewhere (a kept loss tensor is enough): the
                # parameters' gradient accumulators then stay bound to the training stream, backward on the capture
                # stream has to synchronise with it, and that cannot be captured (the HIP runtime crashes at end-capture).
                raise RuntimeError('GraphedTrunk: drop every reference to earlier losses / outputs (anything with a '
                                   'grad_fn) before capturing: their autograd graph pins the parameters\' gradient '
                                   'accumulators to the training stream')
            torch.cuda.current_stream(dev).wait_stream(self.stream)
            torch.cuda.synchronize(dev)
            dnn._STATS_ARENA = self.arena
            self.arena.take(64, dev)              # (allocates the arena's buffer outside the capture)
            dnn.bump_param_epoch()               # every packed-weight cache misses: the packing is part of the graph
            self.fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.fwd, stream=self.stream):
                self.arena.reset()
                self.outs = model.extract_feat(self.x)
            self.gouts = [torch.zeros_like(o) for o in self.outs]
            self.bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.bwd, pool=self.fwd.pool(), stream=self.stream):
                self.arena.reset()
                self._backward(self.outs, self.gouts, params)
            ok = True
            self._epoch_dtypes = [dt for dt, ep in optimizer._packed_epoch.items() if ep == dnn.PARAM_EPOCH[0]]
        finally:
            ag.WGRAD_SIDE_STREAM = side_was
            dnn._STATS_ARENA = arena_was
            optimizer.overlap = overlap_was
            if not ok:      # a refused capture may leave the capture stream invalidated: drain the device, drop the stream
                try:
                    torch.cuda.synchronize(dev)
                except RuntimeError:
                    pass
                self.stream = None
            # the warm-up iterations (and a capture that raised half-way) must leave no trace in the model
            with torch.no_grad():
                for b, k in zip(buffers, keep_buf):
                    b.copy_(k)
                optimizer.flat_g.copy_(keep_g)
            dnn.bump_param_epoch()               # caches filled so far describe warm-up / graph memory: not for eager use
        for p in params:     # (the capture ran no kernel: nothing was accumulated, but autograd may have created .grad)
            assert p.grad is not None
        torch.cuda.synchronize(dev)

    @staticmethod
    def _backward(outs, gouts, params):
        """Backward of the trunk WITHOUT autograd's AccumulateGrad nodes: those are per-parameter objects bound to the
        stream they were first used on (the training stream) and stay alive as long as any earlier autograd graph does
        (a kept loss tensor is enough); run from the capture stream they synchronise the two streams, which is illegal
        inside a capture. torch.autograd.grad returns the gradients instead; the few that autograd delivers (most are
        added into the flat gradient by the kernels themselves and come back as None) are added by hand."""
        grads = torch.autograd.grad(outs, params, gouts, allow_unused=True)
        ag.finish_backward()      # (queued weight gradients launched, their side streams joined: inside the capture)
        with torch.no_grad():
            for p, g in zip(params, grads):
                if g is not None:
                    p.grad.add_(g)

    def mark_packed(self):
        """The forward graph has just repacked every conv weight of the flat buffer for the current parameters: the
        eager head must not do it again."""
        for dt in self._epoch_dtypes:
            self.opt._packed_epoch[dt] = dnn.PARAM_EPOCH[0]

    def matches(self, img):
        return tuple(img.shape) == self.shape and img.dtype == self.dtype and img.is_cuda

    def __call__(self, img):
        return _Replay.apply(img, self.anchor, self)


def enable_trunk_graphs(model, optimizer, example_img):
    """Capture the trunk for batches shaped like `example_img`; the detector's training forward uses it from now on.
    Returns the GraphedTrunk. With several ranks every rank must capture (same model, same shapes); raises when the
    trunk holds SyncBN layers."""
    trunk = GraphedTrunk(model, optimizer, example_img)
    model._graphed_trunk = trunk
    return trunk


class GraphedInference:
    """The eval-mode forward (backbone + neck + head: `bbox_head(extract_feat(img))`) as ONE hipGraph for inputs shaped like
    the example — configs[1] of BASELINE.json (1-stage, B = 8) is ~330 launches for 4 ms of GPU work, and a fifth of its
    step was launch gaps and ATen glue between them. The decode (one kernel + one device-to-host copy per batch, its
    arguments depend on `img_metas`) stays eager behind the replay.

    Legal because the eval forward is static once warm: packed weights, folded BatchNorm constants and the head's Scale
    values sit in the modules' caches (keyed on the parameter epoch and the tensors' version counters — a graph is only
    replayed while both are what they were at capture: load_state_dict / an optimizer step / an in-place edit drop the
    model back to the eager path),
    workspaces and kernel attributes exist after the two warm-up passes, the GroupNorm statistics come from a private
    zero arena whose fill is the graph's first node. The outputs are the graph's own buffers, overwritten by the next
    replay: `get_poses` reads them (stream order) into fresh result tensors."""

    def __init__(self, model, img, warmup=2):
        assert not model.training and img.is_cuda
        self.model = model
        self.shape, self.dtype = tuple(img.shape), img.dtype
        dev = img.device
        self.x = img.detach().clone()
        self.stream = torch.cuda.Stream(device=dev)
        arena_was = dnn._STATS_ARENA
        self.arena = dnn._ZeroArena()
        ok = False
        try:
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self.stream), torch.no_grad():
                for _ in range(warmup):
                    outs = model.bbox_head(model.extract_feat(self.x))
                    del outs
            torch.cuda.current_stream(dev).wait_stream(self.stream)
            torch.cuda.synchronize(dev)
            dnn._STATS_ARENA = self.arena
            self.arena.take(64, dev)              # (allocates the arena's buffers outside the capture)
            self.epoch = dnn.PARAM_EPOCH[0]
            self._state = [t for t in list(model.parameters()) + list(model.buffers())]
            self.sig = self._signature()
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph, stream=self.stream):
                self.arena.reset()
                self.outs = model.bbox_head(model.extract_feat(self.x))
            ok = True
        finally:
            dnn._STATS_ARENA = arena_was
            if not ok:
                try:
                    torch.cuda.synchronize(dev)
                except RuntimeError:
                    pass
                self.stream = None
        torch.cuda.synchronize(dev)

    def _signature(self):
        """Changes whenever a parameter or buffer is written in place through torch (load_state_dict, init, EMA ...): the
        modules' caches the graph's kernels read from are keyed on the same version counters."""
        return sum(t._version for t in self._state)

    def matches(self, img):
        return (tuple(img.shape) == self.shape and img.dtype == self.dtype and img.is_cuda and not self.model.training
                and dnn.PARAM_EPOCH[0] == self.epoch and self._signature() == self.sig)

    def __call__(self, img):
        self.x.copy_(img)
        self.graph.replay()
        return self.outs


def enable_inference_graph(model, example_img):
    """Capture the eval-mode forward for batches shaped like `example_img`; `model.simple_test` replays it for matching
    inputs from now on (anything else — another shape, training mode, changed parameters — runs launch by launch).
    Returns the GraphedInference."""
    model._graphed_infer = None
    g = GraphedInference(model, example_img)
    model._graphed_infer = g
    return g
