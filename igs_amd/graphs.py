"""`GraphedLoop`: a caller's loop body replayed from hipGraphs, one graph per key (camera view).

The reference's refine iteration (infer_batch.py:279-324) issues ~40 kernel launches, 33 of them the caller's own small PyTorch
kernels (activations and their backward, the PSNR line, fills): on this GPU the iteration is bound by the HOST cost of enqueueing them
(DESIGN.md 5), not by any kernel.  Nothing in the iteration depends on values read back from the device -- this package's forward needs
no host round trip on a capturing stream, its backward none, `igs_amd.optim.Adam(capturable=True)` keeps the step counts on the GPU --
so the whole body can be captured once per view with `torch.cuda.graph` and replayed with a single launch:

    loop = GraphedLoop(lambda view: refine_iteration(gs, cams[view], gts[view], bg, ...))
    for it in range(n):
        pkg, loss = loop(view_of(it))          # first visit of a view: eager; second: capture + replay; afterwards: replay

What a caller must know:
  * everything the body reads through Python (camera matrices, tan(fov), learning rates, the loss weights) is frozen into the graph of
    that key; tensors are read afresh at every replay (parameters, optimizer state, an image updated in place);
  * the values the body returns are the graph's own output tensors, overwritten by the next replay of the same key;
  * host-side asserts cannot be captured: the reference's NaN asserts do not run in a replayed iteration (they do in the eager visit);
  * the optimizer must keep its step count on the device (`igs_amd.optim.Adam(..., capturable=True)` or
    `torch.optim.Adam(..., capturable=True)`), and the body must not change the number of Gaussians (densification steps stay eager:
    call `loop.reset()` after one).
  * a replayed forward cannot tell the host that a tile outgrew the instance slab baked into the graph: call `loop.check()` now and
    then (it synchronises) -- it raises if the last replay overflowed, after which `reset()` and going on is the remedy;
  * eager visits and captures run on a side stream the loop owns (replays on the caller's current stream): autograd remembers the
    stream a parameter was first used on for as long as ANY earlier loss / output of the model is alive, and a capture whose backward has
    to synchronise with the default stream is invalid (this ROCm's hipStreamEndCapture then crashes instead of reporting it).  Drop
    losses / render outputs of iterations run OUTSIDE the loop before its first capture.
The first visit of a key runs eagerly and the capture itself executes nothing, so the sequence of steps is exactly the eager one.
"""
import torch

from . import rasterizer


def _tensors(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, dict):
        for v in x.values():
            yield from _tensors(v)
    elif isinstance(x, (tuple, list)):
        for v in x:
            yield from _tensors(v)


class GraphedLoop:
    def __init__(self, body, share_pool=True):
        self.body = body
        self._seen = set()
        self._graphs = {}
        self._pool = None
        self._stream = None
        self._scratch = rasterizer.CaptureScratch()      # rasterizer scratch shared by the graphs of this loop (no per-replay zero-fill)
        self.share_pool = share_pool       # graphs of one loop never run concurrently (one stream), so they can share their memory pool

    def reset(self):
        """Forget every captured graph (after anything that changes shapes or addresses: densification, new parameters)."""
        self._graphs.clear()
        self._seen.clear()
        self._pool = None
        self._scratch = rasterizer.CaptureScratch()

    def check(self):
        """Synchronises, then raises RasterizerError if the last replayed forward of this thread overflowed its tile slabs (results of
        that replay are invalid; the slab hint has been raised: reset() and continue)."""
        if self._graphs:
            torch.cuda.synchronize()
            n, ov = rasterizer.capture_status(any_capture=True)
            if ov:
                raise rasterizer.RasterizerError("a replayed forward overflowed its tile slabs (%d instances in one tile): reset() the loop" % ov)

    def __call__(self, key, *args):
        hit = self._graphs.get(key)
        if hit is not None:
            hit[0].replay()
            return hit[1]
        if not torch.cuda.is_available():
            raise RuntimeError("igs_amd.graphs.GraphedLoop needs a GPU (no CPU fallback)")
        cur = torch.cuda.current_stream()
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        side = self._stream
        side.wait_stream(cur)
        if key not in self._seen:
            self._seen.add(key)
            with torch.cuda.stream(side), rasterizer.capture_scratch(self._scratch):
                out = self.body(key, *args)
            cur.wait_stream(side)
            for t in _tensors(out):
                if t.is_cuda:
                    t.record_stream(cur)                   # (allocated on the side stream, read by the caller on its own)
            return out
        g = torch.cuda.CUDAGraph()
        if self.share_pool and self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, pool=self._pool if self.share_pool else None, stream=side), rasterizer.capture_scratch(self._scratch):
            out = self.body(key, *args)
        cur.wait_stream(side)
        self._graphs[key] = (g, out)
        g.replay()
        return out
