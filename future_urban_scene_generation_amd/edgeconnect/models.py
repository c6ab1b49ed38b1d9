"""Drop-in for the inference half of the reference's ``edgeconnect.models`` on MI355X.

``EdgeModel(config)`` / ``InpaintingModel(config)`` keep the reference's construction, ``load()``
(checkpoint dict ``{'iteration', 'generator'}`` at ``<PATH>/<name>_gen.pth``, models.py:17-30) and
``forward(images, edges, masks)`` (models.py:130-135, 236-240).  The training half (discriminators,
losses, optimisers, ``process``/``backward``) is out of scope - which also
removes the reference's import-time dependency on torchvision's pretrained VGG19.
The mask compositing + channel concat in front of the generator is one libfusg kernel.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import ops
from .networks import EdgeGenerator, InpaintGenerator


class BaseModel(nn.Module):
    def __init__(self, name, config):
        super().__init__()
        self.name = name
        self.config = config
        self.iteration = 0
        path = getattr(config, "PATH", ".") if config is not None else "."
        self.gen_weights_path = os.path.join(path, name + "_gen.pth")
        self.dis_weights_path = os.path.join(path, name + "_dis.pth")

    def load(self):
        if os.path.exists(self.gen_weights_path):
            print("Loading %s generator..." % self.name)
            data = torch.load(self.gen_weights_path, map_location=lambda storage, loc: storage)
            self.generator.load_state_dict(data["generator"])
            self.iteration = data["iteration"]

    def save(self):
        print("\nsaving %s...\n" % self.name)
        torch.save({"iteration": self.iteration, "generator": self.generator.state_dict()}, self.gen_weights_path)

    def process(self, *a, **k):
        raise NotImplementedError("training is out of scope of the MI355X inference path")

    backward = process


class EdgeModel(BaseModel):
    def __init__(self, config):
        super().__init__("EdgeModel", config)
        self.add_module("generator", EdgeGenerator(use_spectral_norm=True))

    def forward(self, images, edges, masks):
        g = self.generator
        g._ensure(images)
        return g._run(ops.ec_inputs(images, edges, masks, 0))     # cat(img*(1-m)+m, edge*(1-m), m)


class InpaintingModel(BaseModel):
    def __init__(self, config):
        super().__init__("InpaintingModel", config)
        self.add_module("generator", InpaintGenerator())

    def forward(self, images, edges, masks):
        g = self.generator
        g._ensure(images)
        return g._run(ops.ec_inputs(images, edges, masks, 1))     # cat(img*(1-m)+m, edge)
