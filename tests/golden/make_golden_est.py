#!/usr/bin/env python3
"""Golden vectors for the EST quantisation layer (row F4): the reference's own QuantizationLayer.forward,
imported from ev-YOLOv6/yolov6/models/learned_repr.py and run on the CPU.

    python tests/golden/make_golden_est.py

Stand-ins (this process only): the object is assembled without the constructor's `.to("cuda")` (:91) and
`Tensor.cuda()` is the identity (:179).  The value MLP is the reference's own ValueLayer, initialised by its
own init_kernel (1000 Adam steps towards the trilinear kernel, :45-66) under torch.manual_seed(0); its weights
travel in the fixture.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from event_representation_study_amd.synthetic import make_events  # noqa: E402


def main():
    spec = importlib.util.spec_from_file_location(
        "ref_learned_repr", os.path.join(REF, "ev-YOLOv6", "yolov6", "models", "learned_repr.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    torch.set_num_threads(1)
    torch.manual_seed(0)
    C, H, W, S = 6, 48, 64, 96
    vl = m.ValueLayer([1, 100, 100, 1], activation=torch.nn.LeakyReLU(negative_slope=0.1), num_channels=C)
    q = m.QuantizationLayer.__new__(m.QuantizationLayer)
    torch.nn.Module.__init__(q)
    q.value_layer, q.dim, q.image_size = vl, (C, H, W), S
    torch.Tensor.cuda = lambda self, *a, **k: self
    g = {"dim": np.array([C, H, W]), "image_size": S}
    for k, v in vl.state_dict().items():
        g["w_" + k] = v.detach().numpy()
    rows = []
    for bi, n in enumerate([4000, 1500, 2500]):
        e = make_events(n, W, H, seed=800 + bi, polarity="01").astype(np.float32)
        rows.append(np.concatenate([e, np.full((n, 1), bi, np.float32)], axis=1))
    ev = np.concatenate(rows)
    g["events"] = ev
    with torch.no_grad():
        # the voxel grid before the letterbox: same statements as forward() up to :176
        q.image_size = None
        orig = q.crop_and_resize_to_resolution
        q.crop_and_resize_to_resolution = lambda x: x
        g["voxel"] = q.forward(torch.from_numpy(ev.copy())).numpy()
        q.crop_and_resize_to_resolution = orig
        q.image_size = S
        g["output"] = q.forward(torch.from_numpy(ev.copy())).numpy()
        u = torch.linspace(-1, 1, 4001)
        g["mlp_u"], g["mlp_f"] = u.numpy(), vl.forward(u).numpy()
    np.savez_compressed(os.path.join(HERE, "est.npz"), **g)
    print("wrote est.npz", {k: getattr(v, "shape", v) for k, v in g.items() if k in ("voxel", "output", "events")})


if __name__ == "__main__":
    main()
