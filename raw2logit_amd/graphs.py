"""One training step of a processor as ONE HIP graph.

A step of the parametrized pipeline is two C-ABI calls (r2l_isp_step_fwd / r2l_isp_step_bwd, include/r2l_isp.h) that
only enqueue kernels on the current stream, plus the autograd bookkeeping around them (model.py:98-142 is the caller).
Below ~8 Mpix per step (the reference's 256 x 256 tiles, dataset.py:92: 64 frames = BASELINE config 5's share per GPU)
the host needs as long for that bookkeeping as the GPU for the kernels, and the step runs at the slower of the two.
StepGraph captures the whole step -- forward, backward, gradient accumulation, BatchNorm's running statistics -- into a
torch.cuda.CUDAGraph (hipGraph) and replays it with one launch: 64 x 256 x 256 runs at the kernels' own 0.16 ms per
step whatever the host does (eager: 0.16-0.33 ms depending on the box), bit-identical output and gradients
(tests/test_gpu_parity.py: test_whole_step_as_one_hip_graph; torch.cuda.make_graphed_callables, which replays forward
and backward as two graphs around static-buffer copies, is SLOWER than eager on this stack: 0.26 ms).

    g = StepGraph(processor, raw, cotangent)      # raw / cotangent: static input buffers (copy new batches into them)
    g.replay()                                    # g.out, p.grad of every parameter, the running statistics: updated
    g = StepGraph(processor, raw, None, loss=lambda rgb: criterion(head(rgb), target), loss_modules=(head,))

The processor must not have run on another stream before (its AccumulateGrad nodes are created by the warm-up here, on
the capture's side stream).

Several ranks (`process_group`): the two small all-gathers that split the step's calls under train-mode BatchNorm and the
flat all-reduce of the parameter gradients are issued on the capture stream and become nodes of the graph -- RCCL
collectives (backend "nccl") are capturable; a gloo group moves host memory and is refused.  One replay is then the whole
data-parallel step of the PROCESSOR: statistics of the global batch, the processor's gradients summed (or averaged) over
the ranks.  The gradients of `loss_modules` (a classifier head) are NOT reduced here: they are DistributedDataParallel's
business -- wrap the head in DDP inside `loss`, or pass `reduce_loss_modules=True` to put them into the same flat all-reduce.
"""
import gc

import torch


class StepGraph:
    def __init__(self, model, raw, cotangent, loss=None, loss_modules=(), warmup=3, process_group=None,
                 average_grads=False, reduce_loss_modules=False):
        """loss: optional callable out -> scalar (then `cotangent` is ignored and loss(out).backward() is captured);
        loss_modules: the modules `loss` runs (a classifier head ...): their parameters' gradients belong to the graph
        too -- like the processor's they are written by every replay into tensors the capture allocated; replay()
        re-attaches those tensors to `p.grad`, so optimizer.zero_grad(set_to_none=True) between replays is harmless.
        process_group: data-parallel ranks (one process per GPU over RCCL; default: the group the model already carries):
        the model runs with it for the lifetime of this object's captures (`model.process_group` is restored when the
        constructor returns) and the gradient all-reduce of the processor's parameters is captured behind the backward
        (average_grads: divide by the ranks; reduce_loss_modules: the `loss_modules` parameters join that all-reduce --
        otherwise several ranks with trainable loss modules raise unless those modules are DistributedDataParallel)."""
        from . import functional as F_
        self.model, self.raw, self.cotangent = model, raw, cotangent
        self.params = [p for m in (model,) + tuple(loss_modules) for p in m.parameters() if p.requires_grad]
        self._isp_params = [p for p in model.parameters() if p.requires_grad]
        self._loss = loss
        group = process_group if process_group is not None else getattr(model, 'process_group', None)
        self._group, self._average = group, average_grads
        self._collectives = False
        self._model_group = getattr(model, 'process_group', None)
        if group is not None:
            if F_._host_staged(group, raw):
                raise RuntimeError('StepGraph: a gloo group moves its vectors through host memory, which a HIP graph cannot '
                                   'capture -- use the eager step with gloo, or an RCCL ("nccl") group')
            self._collectives = F_._group_size(group) > 1 or F_.split_single_rank(group)
        self._reduced = self._isp_params
        if self._collectives:
            extra = [p for m in loss_modules for p in m.parameters() if p.requires_grad]
            if extra and reduce_loss_modules:
                self._reduced = self.params
            elif extra and F_._group_size(group) > 1 and not all(
                    isinstance(m, torch.nn.parallel.DistributedDataParallel) for m in loss_modules):
                raise RuntimeError('StepGraph: several ranks and trainable loss_modules -- their gradients would not be '
                                   'reduced and the ranks would diverge: wrap them in DistributedDataParallel, or pass '
                                   'reduce_loss_modules=True')
        model.process_group = group                       # (for the warm-up and the capture only: restored below)
        try:
            side = torch.cuda.Stream(device=raw.device)
            side.wait_stream(torch.cuda.current_stream(raw.device))
            with torch.cuda.stream(side):                 # warm-up off the default stream (allocator, autograd nodes,
                for _ in range(warmup):                   # and the communicator's own first-call set-up)
                    self._step()
            torch.cuda.current_stream(raw.device).wait_stream(side)
            torch.cuda.synchronize(raw.device)
            for p in self.params:                         # the captured step allocates the gradients in the graph's pool
                p.grad = None
            self.graph = torch.cuda.CUDAGraph()
            # no cyclic garbage collection while the stream is capturing: collecting an object that owns device resources (an
            # older graph, an event, a large cached block) from inside a capture aborts the process on this stack -- it depends
            # on what the caller left uncollected, i.e. on nothing this class controls (seen in the GPU suite, round 5)
            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            # With collectives in the step the capture must not be of the GLOBAL kind: the process group's watchdog thread polls the
            # events of the warm-up's collectives (hipEventQuery, every ~100 ms, until it has seen each of them complete) -- under a
            # global-mode capture that query from ANOTHER thread is "not permitted when stream is capturing", the watchdog thread
            # throws and the process aborts (seen in 1 of ~6 runs of bench.py's graph trial, round 5).  Thread-local mode restricts
            # the capturing thread only; torch does not hand the collectives issued DURING a capture to the watchdog.
            mode = 'thread_local' if self._collectives else 'global'
            try:
                with torch.cuda.graph(self.graph, capture_error_mode=mode):
                    self.out = self._step()
            finally:
                if gc_was_on:
                    gc.enable()
        finally:
            model.process_group = self._model_group
        self._grads = [p.grad for p in self.params]       # tensors of the graph's pool: every replay writes them

    def _step(self):
        out = self.model(self.raw)
        if self._loss is not None:
            self._loss(out).backward()
        else:
            out.backward(self.cotangent)
        if self._collectives:
            # the data-parallel sum of the processor's 132-float gradient: one flat all-reduce on this stream
            from .functional import GradAllReduce
            GradAllReduce(self._reduced, self._group, average=self._average).wait()
        return out

    def replay(self):
        """one hipGraphLaunch: gradients are OVERWRITTEN (the captured step starts from grad = None), not accumulated.
        `p.grad` is pointed back at the captured tensors first: a zero_grad(set_to_none=True) since the last replay
        would otherwise leave the optimiser without gradients while the graph keeps writing the old ones."""
        for p, g in zip(self.params, self._grads):
            if p.grad is not g:
                p.grad = g
        self.graph.replay()
        return self.out
