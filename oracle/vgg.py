"""CPU oracle of the CAD classifier (VGG-19, 10-way head; run_test.py:47-58).  TEST INFRASTRUCTURE ONLY.

Functional restatement of torchvision's ``vgg19`` forward - ``features`` (configuration "E": 3x3 conv + ReLU stacks of
64, 128, 256, 512, 512 channels, each followed by a 2x2 max-pool), ``AdaptiveAvgPool2d((7, 7))``, ``flatten``,
``Linear -> ReLU -> Dropout -> Linear -> ReLU -> Dropout -> Linear`` (dropout is the identity in eval mode) - driven
by a flat state_dict with torchvision's key names.  PARITY UNPINNED: torchvision is an un-vendored, unpinned
dependency of the reference (requirements.txt) that is absent from the build container, and the reference holds no
fixture for the classifier; the layer list is the published one.
"""
import torch
import torch.nn.functional as F

CFG_E = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


def vgg19_forward(sd, x: torch.Tensor) -> torch.Tensor:
    idx = 0
    for v in CFG_E:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            idx += 1
        else:
            x = F.relu(F.conv2d(x, sd[f"features.{idx}.weight"], sd[f"features.{idx}.bias"], padding=1))
            idx += 2
    x = torch.flatten(F.adaptive_avg_pool2d(x, (7, 7)), 1)
    x = F.relu(F.linear(x, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    x = F.relu(F.linear(x, sd["classifier.3.weight"], sd["classifier.3.bias"]))
    return F.linear(x, sd["classifier.6.weight"], sd["classifier.6.bias"])
