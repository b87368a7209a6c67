"""get_optimized_representation -- mirrors representations/optimized_representation.py:86-134."""
from ._common import finish, sample_batch

N_CHANNELS = 12


def get_optimized_representation(reshaped_return_data, num_events, height, width, device_out=False):
    """ERGO-12: the 12 (window, function, aggregation) triples of the reference's second search,
    as an (H, W, 12) float64 array (``device_out=True``: a fresh CUDA tensor, no read-back)."""
    # x, y, p -> int32, t -> int64 and t - t.min(), as MixedDensityEventStack.stack does (:26-33); n_imagenet hands
    # all-float64 fields (imagenet.py:1002-1006)
    sb = sample_batch(reshaped_return_data, height, width, truncate=True, rebase_t=True, device_out=device_out)
    return finish(sb, sb.optimized(scale=1.0), allow_oob=True, what="get_optimized_representation", allow_unsorted=True)
