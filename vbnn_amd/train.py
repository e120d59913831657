"""main.lua's training driver over the fused engine: `train` (main.lua:12-50), `test` (:52-69) and the epoch loop of
`run` (:138-184) -- the immediate caller of the hot path, S-draw loop included (main.lua:28-40).

    opt = default_opt(hidden=[400, 400]); train_set, test_set = data.getMnist(root)
    Main(opt).run(train_set, test_set, epochs=10)

Per minibatch: resetGradients; prepare (compute_prior + operand packing); S x { sample; run }; update -- every
launch on the device, ONE host synchronisation per minibatch (the loss / accuracy read-back the reference's
`net:run` return values need). Per epoch: the five series of main.lua:169-177 through vbnn_amd.logger, and the run
directory's `parameters/means`, `parameters/vars`, `opt` (the files mainviz.lua:11-15 loads) through
vbnn_amd.t7file with utils.safe_save's keep-the-old-file rule (main.lua:181 saves the whole Lua `net` table, methods
included; Lua byte code cannot be produced here, so the data is what is saved).

Differences from main.lua, all deliberate: the loop ends after `epochs` (the reference loops forever, :164); a last
short minibatch is skipped rather than padded with uninitialised rows (data.lua:9-20); targets are 0-based.
"""
import os

import numpy as np
import torch

from . import utils as u
from .engine import FusedMLP
from .logger import Logger

# the series VBLinear:update logs per layer and minibatch, in the order of its Log:add calls (VBLinear.lua:149-164)
UPDATE_SERIES = ("vlc grad", "vle grad", "mlc grad", "mle grad", "min variance", "max variance", "mean variance", "var hat",
                 "mean means", "std means", "min. means", "max. means", "mu normratio", "var normratio")


def default_opt(**over):
    """config.lua's values under the engine's key names (config.lua:5-68)."""
    opt = {
        "network_name": "exp", "type": "vb", "dataset": "mnist",
        "batchSize": 100, "testBatchSize": 100, "trainSize": 100, "testSize": 1000,     # config.lua:11-16 (batchSize 1 there)
        "geometry": (28, 28), "input_size": 784, "n_classes": 10, "hidden": [10],        # :17-18, :31
        "B": 1e6, "S": 30, "testSamples": 30, "log": True,                                # :30-35
        "seed": 3, "mu_init": 0, "var_init": 1e-3,                                        # :40-44
        "state": {"learningRate": 1e-3}, "varState": {"learningRate": 5e-2}, "meanState": {"learningRate": 1e-4},  # :51-64
        "mode": "lrt", "dtype": "f32", "fuse_kl": True,
    }
    opt.update(over)
    return opt


class Main:
    def __init__(self, opt, device=None, net=None):
        # (the KL gradient: FusedMLP's own default -- exact, from the fp32 parameters in the update sweep wherever the fused
        # epilogue would read bf16 shadows; the saved opt records the form that actually ran)
        self.net = net or FusedMLP(opt, device=device)
        self.opt = dict(opt, kl_in_update=bool(getattr(self.net, "kl_in_update", False)))
        self.device = self.net.device
        self.rng = np.random.RandomState(int(opt.get("seed", 3)))                         # torch.manualSeed(3), config.lua:40
        self.indices = None
        self.log = Logger(opt["network_name"], append=bool(opt.get("network_to_load"))) if opt.get("log") else None

    def _to_device(self, inputs, targets):
        x = torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.float32)).to(self.device)
        t = torch.from_numpy(np.ascontiguousarray(targets)).to(self.device, dtype=torch.int32)
        return x, t

    def train(self, dataset):                                                             # main.lua:12-50
        opt, net = self.opt, self.net
        bs, n = int(opt["batchSize"]), int(opt["trainSize"])
        S = int(opt["S"]) if opt.get("type", "vb") == "vb" else 1
        if self.indices is None:
            self.indices = list(range(0, n - bs + 1, bs))                                 # torch.range(1, trainSize, batchSize)
        accuracy = error = 0.0
        net.prepare()                     # once: afterwards update() maintains the operand shadows and prior statistics
        stacked = net.mode == "lrt" and S > 1 and bool(opt.get("stack_draws", True))
        for batch_index in u.shuffle(self.indices, self.rng):
            inputs, targets = dataset.create_minibatch(batch_index, bs, n, opt.get("geometry"))
            x, t = self._to_device(inputs, targets)
            net.resetGradients()
            if stacked:                                                                   # main.lua:32-37 as rows of one pass
                net.run_draws(x, t, S)
            else:
                for _ in range(S):
                    net.sample()
                    net.run(x, t)
            loss, correct = net.loss_and_accuracy()           # sums over the S draws (the criterion accumulates)
            if not np.isfinite(loss):
                # Divergence must be seen HERE: the bf16 gradInput epilogue carries the ReLU mask through its second pass as a
                # NaN and zeroes every NaN it reads back (csrc/epilogues.h, EpiDx::apply_folded), so a genuine NaN above a layer
                # does not reach the layers below it the way model:backward propagates it in the reference (mlp.lua:79) -- the
                # loss and the last layer's gradients still show it (tests: test_nan_in_the_backward_...). No update on such a step.
                raise FloatingPointError(f"minibatch {batch_index}: the loss is {loss} -- the run has diverged; no parameter was updated")
            error += loss * net.world / S
            accuracy += 100.0 * correct / (bs * S)
            log_update = bool(self.log and opt.get("log_update") and net.mode == "lrt")
            net.update(opt, log=log_update)
            if log_update:                                                                # VBLinear.lua:149-164, per layer
                for row in net.update_log.cpu().tolist():
                    for name, value in zip(UPDATE_SERIES, row):
                        self.log.add(name, value)
        B = len(self.indices)
        return accuracy / B, error / B

    def test(self, dataset):                                                              # main.lua:52-69
        opt, net = self.opt, self.net
        bs, n = int(opt["testBatchSize"]), int(opt["testSize"])
        starts = list(range(0, n - bs + 1, bs))
        accuracy = error = 0.0
        for t0 in starts:
            inputs, targets = dataset.create_minibatch(t0, bs, n, opt.get("geometry"))
            x, t = self._to_device(inputs, targets)
            err, acc = net.test(x, t)
            accuracy += acc
            error += err
        return accuracy / len(starts), error / len(starts)

    def save(self):
        """The run directory's data files (mainviz.lua:11-15): every VB layer's means / vars flattened and
        concatenated in layer order -- the order of the reference's flat `parameters` vector (mlp.lua:37)."""
        net, d = self.net, self.opt["network_name"]
        means = torch.cat([v.means.reshape(-1) for v in net.vb]).cpu().numpy()
        vars_ = torch.cat([v.lvars.reshape(-1) for v in net.vb]).exp().cpu().numpy()
        u.safe_save(means, os.path.join(d, "parameters"), "means")
        u.safe_save(vars_, os.path.join(d, "parameters"), "vars")
        u.safe_save({k: (list(v) if isinstance(v, tuple) else v) for k, v in self.opt.items()}, d, "opt")

    def run(self, trainSet, testSet, epochs=1):                                           # main.lua:138-184
        history = []
        for _ in range(epochs):
            trainAccuracy, trainError = self.train(trainSet)
            testAccuracy, testError = self.test(testSet)
            rec = {"devacc": testAccuracy, "trainacc": trainAccuracy, "deverr": testError, "trainerr": trainError}
            if getattr(self.net, "sharded", False):
                self.net.gather_parameters()              # collective (every rank runs this loop): calc_lc and save read fp32 rows
            if self.opt.get("type", "vb") == "vb":
                rec["lc"] = self.net.calc_lc(self.opt)
            if self.log:
                for k in ("devacc", "trainacc", "deverr", "trainerr", "lc"):              # main.lua:169-177
                    if k in rec:
                        self.log.add(k, rec[k])
                self.log.flush()
                self.save()                                                                # main.lua:181
            history.append(rec)
        return history
