#!/usr/bin/env python3
"""Latency of the per-sample drop-in wrappers (host structured array in, host numpy array out): the
boundary the reference's adapters call once per sample (gen1_2yolo.py:296-304).  PCIe-inclusive by
construction -- H2D of the events, bin + build, D2H of the (H, W, 12) result.

    python tools/per_sample_latency.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_representation_study_amd.representations import gen1_transforms  # noqa: E402
from event_representation_study_amd.representations.event_stack import EventStack  # noqa: E402
from event_representation_study_amd.representations.time_surface import ToTimesurface  # noqa: E402
from event_representation_study_amd.representations.tore import events2ToreFeature  # noqa: E402
from event_representation_study_amd.representations.representation_search.mixed_density_event_stack import \
    MixedDensityEventStack  # noqa: E402
from event_representation_study_amd.representations.tonic_compat import ToVoxelGrid, ToImage  # noqa: E402
from event_representation_study_amd.synthetic import make_events, to_structured  # noqa: E402


def main():
    H, W, N = 480, 640, 50000
    table = {"VoxelGrid": ToVoxelGrid, "OptimizedRepresentation": MixedDensityEventStack, "EventStack": EventStack,
             "EventHistogram": ToImage, "TORE": events2ToreFeature, "TimeSurface": ToTimesurface}
    wins = [to_structured(make_events(N, W, H, seed=40 + i, polarity="01")) for i in range(8)]
    for rnd in range(2):                        # round 0 warms the host allocator and the HIP context
        for name, tr in table.items():
            ts = []
            for i in range(24):
                w = wins[i % len(wins)].copy()
                t0 = time.perf_counter()
                out = gen1_transforms.get_item_transform(w, str(tr), tr, H, W, N, 50000)
                ts.append(time.perf_counter() - t0)
            el = float(np.median(ts))
            # the same call with the result left on the GPU (get_item_transform_cuda, r04): no read-back
            td = []
            for i in range(24):
                w = wins[i % len(wins)].copy()
                t0 = time.perf_counter()
                dev = gen1_transforms.get_item_transform_cuda(w, str(tr), tr, H, W, N, 50000)
                td.append(time.perf_counter() - t0)
            eld = float(np.median(td))
            if rnd:
                print(json.dumps({"representation": name, "median_ms_per_sample": round(el * 1e3, 3),
                                  "median_ms_per_sample_device_out": round(eld * 1e3, 3),
                                  "events_per_s": round(N / el), "events_per_s_device_out": round(N / eld), "out_shape": list(out.shape),
                                  "out_dtype": str(out.dtype), "out_MB": round(out.nbytes / 1e6, 1)}))


if __name__ == "__main__":
    main()
