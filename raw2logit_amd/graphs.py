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
the capture's side stream); single-GPU or eval-mode / static processors only: a train-mode BatchNorm exchange between
ranks splits the step's calls on the host (functional.py).
"""
import torch


class StepGraph:
    def __init__(self, model, raw, cotangent, loss=None, loss_modules=(), warmup=3):
        """loss: optional callable out -> scalar (then `cotangent` is ignored and loss(out).backward() is captured);
        loss_modules: the modules `loss` runs (a classifier head ...): their parameters' gradients belong to the graph
        too -- like the processor's they are written by every replay into tensors the capture allocated, so do not set
        them to None afterwards (zero_grad(set_to_none=False) or nothing at all: a replay overwrites them)"""
        self.model, self.raw, self.cotangent = model, raw, cotangent
        self.params = [p for m in (model,) + tuple(loss_modules) for p in m.parameters() if p.requires_grad]
        self._loss = loss
        side = torch.cuda.Stream(device=raw.device)
        side.wait_stream(torch.cuda.current_stream(raw.device))
        with torch.cuda.stream(side):                     # warm-up off the default stream (allocator, autograd nodes)
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream(raw.device).wait_stream(side)
        torch.cuda.synchronize(raw.device)
        for p in self.params:                             # the captured step allocates the gradients in the graph's pool
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._step()

    def _step(self):
        out = self.model(self.raw)
        if self._loss is not None:
            self._loss(out).backward()
        else:
            out.backward(self.cotangent)
        return out

    def replay(self):
        """one hipGraphLaunch: gradients are OVERWRITTEN (the captured step starts from grad = None), not accumulated"""
        self.graph.replay()
        return self.out
