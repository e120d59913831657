"""mlp.lua on MI355X: the caller of the hot path, with the reference's protocol
(`buildModel / resetGradients / sample / run / test / calc_lc`, mlp.lua:7-115).

This is the module-level path: nn.Sequential of Reshape -> [VBLinear -> ReLU]* -> Linear ->
LogSoftMax + ClassNLLCriterion, every module a call into libvbnn_hip.so, activations resident
in HBM between modules. The fused whole-step engine (vbnn_amd/engine.py) runs the same
arithmetic with the elementwise modules folded into the GEMM epilogues.
"""
import math

import torch

from . import _lib as L
from . import nn


class MLP:
    def buildModel(self, opt, device=None):                              # mlp.lua:7-60
        self.opt = opt
        self.vb_indices = []
        self.model = nn.Sequential()
        self.model.add(nn.Reshape(opt["input_size"]))                    # :12
        hidden = list(opt["hidden"])
        sizes = [opt["input_size"]] + hidden
        n_classes = opt["n_classes"] if "n_classes" in opt else len(opt["classes"])
        for i in range(1, len(hidden) + 1):
            if opt.get("type", "vb") == "vb":
                m = nn.VBLinear(sizes[i - 1], sizes[i], opt, device)     # :14,22
                self.vb_indices.append(2 * i)                            # :15,23
            else:
                m = nn.Linear(sizes[i - 1], sizes[i], opt, device)       # :17,25
            self.model.add(m)
            self.model.add(nn.ReLU())                                    # :19,27
        self.model.add(nn.Linear(sizes[-1], n_classes, opt, device))     # :29
        self.model.add(nn.LogSoftMax())                                  # :30
        self.criterion = nn.ClassNLLCriterion()                          # :32
        # mlp.lua:47-55: bias zero, weight ~ N(0, sqrt(2 / fan_in)) for every (VB)Linear.
        # (randomkit.normal there; the Philox HEINIT stream here, layer index = position.)
        for i in range(1, len(hidden) + 2):
            m = self.model.get(i * 2)
            m.layer_id = i - 1
            if isinstance(m, nn.VBLinear):
                m.set_layer_id(i - 1)
            m.bias.zero_()
            m.reset(math.sqrt(2.0 / m.weight.shape[1]))
        params, grads = self.model.parameters()                          # :37 getParameters
        self.parameters, self.gradParameters = params, grads
        self.W = sum(p.numel() for p in params)                          # :40
        return self

    def resetGradients(self):                                            # mlp.lua:62-67
        for g in self.gradParameters:
            g.zero_()
        for i in self.vb_indices:
            self.model.get(i).resetAcc(self.opt)

    def sample(self):                                                    # mlp.lua:69-74
        for i in self.vb_indices:
            self.model.get(i).sample(self.opt)

    def run(self, inputs, targets):                                      # mlp.lua:76-84
        outputs = self.model.forward(inputs)
        df_do = self.criterion.backward(outputs, targets)
        self.model.backward(inputs, df_do)
        error = self.criterion.forward(outputs, targets)
        accuracy = 100.0 * self.criterion.correct / outputs.shape[0]     # utils.get_accuracy (utils.lua:11-27)
        return error, accuracy

    def test(self, input, target):                                       # mlp.lua:86-107
        if self.opt.get("type", "vb") == "vb":
            if self.opt.get("quicktest"):
                for i in self.vb_indices:
                    self.model.get(i).clamp_to_map()
                return self.run(input, target)
            error = accuracy = 0.0
            for _ in range(self.opt["testSamples"]):
                self.sample()
                err, acc = self.run(input, target)
                error += err
                accuracy += acc
            return error / self.opt["testSamples"], accuracy / self.opt["testSamples"]
        return self.run(input, target)

    def calc_lc(self, opt=None):                                         # mlp.lua:109-115
        opt = opt or self.opt
        lc = 0.0
        for i in self.vb_indices:
            lc = lc + float(self.model.get(i).calc_lc(opt).sum().item())
        return lc


    def update(self, opt=None):                                          # mlp.lua:117-142
        """optim.sgd on the final Linear (`opt.state.learningRate`), then VBLinear:update per VB layer.
        Documented fix: the reference narrows the flat parameter vector to its LAST 110 entries
        (mlp.lua:41-42: 10*10 + 10, right only when the last hidden layer has 10 units); here the final
        Linear's whole weight and bias are updated, which is what that slice means for hidden = {10}."""
        opt = opt or self.opt
        lr = float(opt["state"]["learningRate"])
        last = self.model.get(2 * (len(opt["hidden"]) + 1))
        lib, h = L.lib(), last.ctx.h
        if opt.get("type", "vb") == "vb":
            L.check(lib.vbnn_sgd_step(h, nn._p(last.weight), nn._p(last.gradWeight), last.weight.numel(), lr))
            L.check(lib.vbnn_sgd_step(h, nn._p(last.bias), nn._p(last.gradBias), last.bias.numel(), lr))
            return [self.model.get(i).update(opt) for i in self.vb_indices]  # :137-141
        for p, g in zip(self.parameters, self.gradParameters):              # :125-128
            L.check(lib.vbnn_sgd_step(h, nn._p(p), nn._p(g), p.numel(), lr))
        return []


def buildModel(opt, device=None):
    return MLP().buildModel(opt, device)
