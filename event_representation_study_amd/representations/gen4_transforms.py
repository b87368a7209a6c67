"""get_item_transform -- mirrors representations/gen4_transforms.py:12-83 (the gen1 dispatcher
without the unused ``time_window`` argument)."""
from . import gen1_transforms


def get_item_transform(reshaped_return_data, representation_name, transform, height, width, num_events):
    return gen1_transforms.get_item_transform(reshaped_return_data, representation_name, transform, height, width,
                                              num_events, None)


def get_item_transform_cuda(reshaped_return_data, representation_name, transform, height, width, num_events):
    return gen1_transforms.get_item_transform_cuda(reshaped_return_data, representation_name, transform, height, width,
                                                   num_events, None)
