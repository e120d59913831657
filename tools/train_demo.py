#!/usr/bin/env python3
"""Train the same network with both estimators on data.synthetic_digits (or MNIST IDX files: --mnist DIR) and print
the per-epoch series of main.lua:169-177. Run on the GPU box:  python3 tools/train_demo.py [--dtype bf16] [--epochs 5]"""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbnn_amd import data, train

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32")
ap.add_argument("--epochs", type=int, default=5)
ap.add_argument("--hidden", default="400,400")
ap.add_argument("--batch", type=int, default=100)
ap.add_argument("--S", type=int, default=2)
ap.add_argument("--mnist", default="")
a = ap.parse_args()
if a.mnist:
    trainSet, testSet = data.getMnist(a.mnist)
else:
    trainSet, testSet = data.synthetic_digits(10000, 2000, seed=3, noise=4.0)
nt, ne = trainSet["inputs"].shape[0], testSet["inputs"].shape[0]
for mode in ("lrt", "wn"):
    opt = train.default_opt(network_name=os.path.join(tempfile.mkdtemp(), mode), hidden=[int(h) for h in a.hidden.split(",")],
                            batchSize=a.batch, testBatchSize=500, trainSize=nt, testSize=ne, S=a.S, testSamples=5, mode=mode,
                            dtype=a.dtype, state={"learningRate": 5e-2}, meanState={"learningRate": 1e-3},
                            varState={"learningRate": 5e-2})
    m = train.Main(opt)
    t0 = time.time()
    for e, rec in enumerate(m.run(trainSet, testSet, epochs=a.epochs)):
        print(json.dumps({"mode": mode, "dtype": a.dtype, "epoch": e + 1, **{k: round(v, 5) for k, v in rec.items()}}), flush=True)
    print(json.dumps({"mode": mode, "seconds": round(time.time() - t0, 2), "minibatches": a.epochs * (nt // a.batch), "S": a.S}), flush=True)
