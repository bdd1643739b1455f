"""Counter-based noise state for the in-kernel Philox4x32-10 streams.

The samplers are stateless on the device: a draw is a pure function of (seed, offset, global row index,
slot).  The only state is this (seed, offset) pair; every sampling call takes the next offset.  Saving and
restoring `get_state()` therefore makes a run resumable, and sharding a batch over ranks (each rank passing its
global row offset) reproduces the unsharded draw bit for bit.
"""
import os
import threading

import torch


class PhiloxGenerator:
    def __init__(self, seed=0x5EED, offset=0):
        self._lock = threading.Lock()
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = int(offset) & 0xFFFFFFFFFFFFFFFF
        self._dev_counter = None

    def manual_seed(self, seed):
        with self._lock:
            self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
            self.offset = 0
            if self._dev_counter is not None:
                self._dev_counter.zero_()
        return self

    def next_offset(self):
        """Offset for one sampling call (all ranks of a sharded job must make the same calls in the same order)."""
        if self._dev_counter is not None:
            raise RuntimeError("this generator keeps its counter on the device: use ticket(device)")
        seed, o, _ = self.ticket(None)
        return seed, o

    # ---- device-resident counter (HIP graphs) ----------------------------------------------------------------
    def enable_device_counter(self, device):
        """Move the call counter into device memory.  The host arguments of a launch captured into a HIP graph are
        frozen, so a host counter would replay the same noise forever; with the counter on the device each call
        reads it (a one-element snapshot kept for the backward pass) and advances it with stream-ordered
        operations that are captured too, so every replay draws fresh noise.  The draws are the same as with the
        host counter.  Call this BEFORE capture (memory allocated during capture belongs to the graph)."""
        device = torch.device(device)
        with self._lock:
            if self._dev_counter is None:
                self._dev_counter = torch.tensor([self._as_i64(self.offset)], dtype=torch.int64, device=device)
            elif self._dev_counter.device != device:
                raise RuntimeError(f"the counter already lives on {self._dev_counter.device}")
        return self

    @staticmethod
    def _as_i64(u):
        return u - (1 << 64) if u >= (1 << 63) else u

    def ticket(self, device):
        """(seed, host_offset, device_offset_tensor_or_None) for one sampling call; the kernels use
        host_offset + device_offset[0]."""
        with self._lock:
            if self._dev_counter is None:
                if device is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                    raise RuntimeError(
                        "in-kernel noise under HIP graph capture needs a device-resident counter (a captured host "
                        "offset would replay identical noise): call generator.enable_device_counter(device) before "
                        "capturing (graph.GraphedTrainStep does).  VMF_RNG=torch moves only VonMisesFisher3D / "
                        "SphericalUniform / S1 noise to torch; the general-m sampler always uses this counter")
                o = self.offset
                self.offset = (self.offset + 1) & 0xFFFFFFFFFFFFFFFF
                return self.seed, o, None
            # snapshot = counter; counter += 1 as ONE launch of the library (a clone and an in-place add were two)
            from . import _lib
            snap = torch.empty_like(self._dev_counter)
            _lib.check(_lib.lib().vmf_philox_ticket(_lib.ptr(self._dev_counter), _lib.ptr(snap),
                                                    _lib.stream_ptr(self._dev_counter.device)), "vmf_philox_ticket")
            return self.seed, 0, snap

    def get_state(self):
        with self._lock:
            if self._dev_counter is not None:  # synchronises
                self.offset = int(self._dev_counter.item()) & 0xFFFFFFFFFFFFFFFF
            return {"seed": self.seed, "offset": self.offset}

    def set_state(self, state):
        with self._lock:
            self.seed = int(state["seed"]) & 0xFFFFFFFFFFFFFFFF
            self.offset = int(state["offset"]) & 0xFFFFFFFFFFFFFFFF
            if self._dev_counter is not None:
                self._dev_counter.fill_(self._as_i64(self.offset))


default_generator = PhiloxGenerator()


def manual_seed(seed):
    return default_generator.manual_seed(seed)


def rng_mode():
    """'torch' (default): VonMisesFisher3D / SphericalUniform / S1 draw their noise with torch.rand / torch.randn
    in the reference's order (sampler/von_mises_fisher.py:56,75), so a torch seed reproduces the reference's
    noise on the same device.  'philox': noise is generated inside the kernels (no HBM traffic for it)."""
    m = os.environ.get("VMF_RNG", "torch").lower()
    if m not in ("torch", "philox"):
        raise ValueError(f"VMF_RNG must be 'torch' or 'philox', got {m!r}")
    return m
