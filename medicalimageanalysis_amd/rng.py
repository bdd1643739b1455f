"""Counter-based noise state for the in-kernel Philox4x32-10 streams.

The samplers are stateless on the device: a draw is a pure function of (seed, offset, global row index,
slot).  The only state is this (seed, offset) pair; every sampling call takes the next offset.  Saving and
restoring `get_state()` therefore makes a run resumable, and sharding a batch over ranks (each rank passing its
global row offset) reproduces the unsharded draw bit for bit.
"""
import os
import threading


class PhiloxGenerator:
    def __init__(self, seed=0x5EED, offset=0):
        self._lock = threading.Lock()
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = int(offset) & 0xFFFFFFFFFFFFFFFF

    def manual_seed(self, seed):
        with self._lock:
            self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
            self.offset = 0
        return self

    def next_offset(self):
        """Offset for one sampling call (all ranks of a sharded job must make the same calls in the same order)."""
        with self._lock:
            o = self.offset
            self.offset = (self.offset + 1) & 0xFFFFFFFFFFFFFFFF
        return self.seed, o

    def get_state(self):
        with self._lock:
            return {"seed": self.seed, "offset": self.offset}

    def set_state(self, state):
        with self._lock:
            self.seed = int(state["seed"]) & 0xFFFFFFFFFFFFFFFF
            self.offset = int(state["offset"]) & 0xFFFFFFFFFFFFFFFF


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
