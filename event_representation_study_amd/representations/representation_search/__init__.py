"""Host mirrors of representations/representation_search/* (MixedDensityEventStack, compute_otmi, compute_repr) on the HIP path."""
