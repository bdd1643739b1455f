"""Whole-step HIP graph capture for the launch-bound train step.

At the reference's batch sizes (B = 128, latent S^2) a VAE step is ~100 tiny launches; the GPU is idle between them.
Every entry point of libvmf_hip.so is asynchronous, allocation-free and never synchronises, so forward + loss +
backward + optimizer step can be captured once and replayed as one graph launch.  Noise stays fresh across replays:
torch's generator is capture-aware (VMF_RNG=torch), and the in-kernel Philox streams read their call counter from
device memory (rng.PhiloxGenerator.enable_device_counter, which GraphedTrainStep always turns on).
"""
import os
import tempfile

import torch

from . import rng


class GraphedTrainStep:
    """step = GraphedTrainStep(model, optimizer, loss_fn, example_inputs); loss = step(*inputs)

    loss_fn(model, *inputs) -> scalar loss.  The optimizer must keep its state on the device
    (e.g. torch.optim.Adam(..., capturable=True, fused=True): the fused update is one launch for all parameters where
    the default for-each update is a dozen per step -- the S^9 step at batch 128 replays in 0.70 ms with it against
    0.84 ms without).  Inputs are copied into static buffers, so their shapes and
    dtypes are fixed at construction.  The returned loss is a static tensor that the next call overwrites.

    tune_gemms=True lets PyTorch's TunableOp choose the GEMM kernel of every nn.Linear shape during the warm-up steps
    (a few seconds; its log of choices goes to the temp directory unless PYTORCH_TUNABLEOP_FILENAME names a file).  The MLP bodies of the VAEs are stock nn.Linear; at batch 128 the
    default choice runs a 128 x 128 x 128 product in ONE workgroup (11-16 us each, a third of the replay), the tuned
    ones take 6 us: the S^9 step replays in 0.48 ms instead of 0.70."""

    def __init__(self, model, optimizer, loss_fn, example_inputs, warmup=3, generator=None, tune_gemms=False):
        for group in optimizer.param_groups:
            if "capturable" in group and not group["capturable"]:
                raise ValueError("GraphedTrainStep needs an optimizer with capturable=True (its step counter must "
                                 "live on the device)")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.static_inputs = [t.clone() for t in example_inputs]
        dev = self.static_inputs[0].device
        if dev.type != "cuda":
            raise RuntimeError("GraphedTrainStep runs on a HIP device only")
        # unconditionally: the general-m sampler and HypersphericalUniform draw from the in-kernel streams whatever
        # VMF_RNG says, and a host counter cannot be captured (harmless when nothing uses it)
        (generator or rng.default_generator).enable_device_counter(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        tuner = torch.cuda.tunable if tune_gemms else None
        if tuner is not None:
            was = (tuner.is_enabled(), tuner.tuning_is_enabled())
            tuner.enable(True)
            tuner.tuning_enable(True)
            if "PYTORCH_TUNABLEOP_FILENAME" not in os.environ:  # TunableOp logs its choices as it goes: not into the cwd
                tuner.set_filename(os.path.join(tempfile.gettempdir(), f"vmf_tunableop_{os.getpid()}.csv"))
        with torch.cuda.stream(side):  # warm-up off the capture stream: lazy initialisations, optimizer state
            for _ in range(warmup):
                self._eager_step()
        if tuner is not None:
            torch.cuda.synchronize(dev)
            tuner.tuning_enable(was[1])  # the choices made stay in use (also inside the capture); no further tuning
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.static_loss = self._eager_step(zero=False)
        if tuner is not None:
            # the captured launches keep the kernels chosen during the warm-up; the process-wide switch goes back to what
            # the caller had, so that code outside this step is not silently routed through TunableOp
            tuner.enable(was[0])

    def _eager_step(self, zero=True):
        if zero:
            self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model, *self.static_inputs)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_loss
