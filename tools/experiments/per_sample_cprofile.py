import cProfile, pstats, sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from event_representation_study_amd.representations import gen1_transforms
from event_representation_study_amd.representations.representation_search.mixed_density_event_stack import MixedDensityEventStack
from event_representation_study_amd.representations.event_stack import EventStack
from event_representation_study_amd.representations.time_surface import ToTimesurface
from event_representation_study_amd.synthetic import make_events, to_structured
H, W, N = 480, 640, 50000
wins = [to_structured(make_events(N, W, H, seed=40 + i, polarity="01")) for i in range(8)]
tr = {"es": EventStack, "ts": ToTimesurface}.get(os.environ.get("REP", ""), MixedDensityEventStack)   # REP=es | ts
def run(k):
    for i in range(k):
        gen1_transforms.get_item_transform_cuda(wins[i % 8], str(tr), tr, H, W, N, 50000)
run(100)
pr = cProfile.Profile(); pr.enable(); run(2000); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "28")))
