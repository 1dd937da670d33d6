"""Layer discovery (mirrors the reference's modelutils.py:5-16)."""
import torch
import torch.nn as nn

DEV = torch.device('cuda:0')


def find_layers(module, layers=[nn.Conv2d, nn.Linear], name=''):  # noqa: B006  (the reference's own default, a list)
    """Map dotted name -> module for every leaf whose type is in `layers`."""
    if type(module) in layers:
        return {name: module}
    found = {}
    for child_name, child in module.named_children():
        found.update(find_layers(child, layers=layers, name=f"{name}.{child_name}" if name else child_name))
    return found
