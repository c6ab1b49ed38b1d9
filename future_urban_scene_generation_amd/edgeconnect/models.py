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
        """Reads `<PATH>/<name>_gen.pth` = {'iteration': int, 'generator': state_dict} when it exists (the
        reference's checkpoint contract, edgeconnect/models.py:25-30); a missing file leaves the initial weights."""
        if not os.path.exists(self.gen_weights_path):
            return
        ckpt = torch.load(self.gen_weights_path, map_location="cpu", weights_only=True)
        self.generator.load_state_dict(ckpt["generator"])
        self.iteration = ckpt["iteration"]

    def save(self):
        """Writes the same file `load` reads (edgeconnect/models.py:38-48, generator half)."""
        torch.save({"iteration": self.iteration, "generator": self.generator.state_dict()}, self.gen_weights_path)

    def process(self, *a, **k):
        raise NotImplementedError("training is out of scope of the MI355X inference path")

    backward = process


class EdgeModel(BaseModel):
    def __init__(self, config):
        super().__init__("EdgeModel", config)
        self.add_module("generator", EdgeGenerator(use_spectral_norm=True))

    def forward(self, images, edges, masks):
        return self.generator.run_model(images, edges, masks, 0)   # cat(img*(1-m)+m, edge*(1-m), m) -> generator


class InpaintingModel(BaseModel):
    def __init__(self, config):
        super().__init__("InpaintingModel", config)
        self.add_module("generator", InpaintGenerator())

    def forward(self, images, edges, masks):
        return self.generator.run_model(images, edges, masks, 1)   # cat(img*(1-m)+m, edge) -> generator
