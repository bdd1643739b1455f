#!/usr/bin/env python3
"""Every float32 in the intervals where the reference's KL forms overflow (e^k, sinh k leave the float range), through
the per-element KL kernels and the CPU checker: NaN / inf positions must agree float for float.
    python tools/scan_kl_thresholds.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicalimageanalysis_amd.kl import VARIANTS, kl_per_sample  # noqa: E402
from oracle import vmf_oracle as orc  # noqa: E402

DEV = torch.device("cuda:0")


def floats_between(lo, hi):
    a, b = np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32)
    return np.arange(int(a), int(b) + 1, dtype=np.uint32).view(np.float32)


def scan():
    bad = 0
    for lo, hi in ((88.5, 89.6), (43.9, 44.9), (709.0, 711.0), (999.0, 1001.0)):
        k = floats_between(lo, hi)
        for name in VARIANTS:
            with np.errstate(all="ignore"):
                ref, _, _ = orc.vmf_kl(k, VARIANTS[name])
            for shift in (0, 1, 2, 3):  # every position of an element inside a quad
                kk = k[shift:]
                got = kl_per_sample(torch.from_numpy(kk.copy()).to(DEV), name).cpu().numpy()
                r = ref[shift:]
                mism = np.nonzero((np.isnan(got) != np.isnan(r)) | (np.isinf(got) != np.isinf(r)))[0]
                if mism.size:
                    bad += mism.size
                    i = mism[0]
                    print(f"{name} [{lo},{hi}] shift {shift}: {mism.size} mismatches, first k={kk[i]!r} ({kk[i].view(np.uint32):#x}) "
                          f"got {got[i]} ref {r[i]}")
    # double precision: 4000 doubles on either side of the points where e^k and sinh k leave the double range
    for centre in (709.782712893384, 710.4758600739439):
        c = np.float64(centre).view(np.uint64)
        k = np.arange(int(c) - 4000, int(c) + 4001, dtype=np.uint64).view(np.float64)
        for name in VARIANTS:
            with np.errstate(all="ignore"):
                ref, _, _ = orc.vmf_kl(k, VARIANTS[name])
            got = kl_per_sample(torch.from_numpy(k.copy()).to(DEV), name).cpu().numpy()
            mism = np.nonzero((np.isnan(got) != np.isnan(ref)) | (np.isinf(got) != np.isinf(ref)))[0]
            if mism.size:
                bad += mism.size
                i = mism[0]
                print(f"{name} fp64 around {centre}: {mism.size} mismatches, first k={k[i]!r} got {got[i]} ref {ref[i]}")
    return bad


if __name__ == "__main__":
    n_bad = scan()
    print("mismatches:", n_bad)
    sys.exit(1 if n_bad else 0)
