#!/usr/bin/env python3
"""tests/golden/mt_tiny.npz: one round-robin pass (train_multi_task.py:249-265) of the IMPORTED
REFERENCE multi-task model (models/erfnet_multi_task.py, heads [20, 27]) on CPU, tiny shapes,
recorded dropout masks.  Arrays only."""
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import fixtures as fx          # noqa: E402
from oracle import rap_oracle as O         # noqa: E402
from tools.gen_golden_step3 import Replay, WEIGHT_IDD   # noqa: E402

torch.set_num_threads(8)


def main():
    sys.path.insert(0, "/root/reference")
    ref = importlib.import_module("models.erfnet_multi_task")
    G = {}
    torch.manual_seed(0)
    model = ref.Net([20, 27], 2, 0)
    sd = model.state_dict()
    fx.perturb_bn(sd, seed=31)
    G["state_keys"] = np.array(list(sd.keys()))
    named = [("module." + n, p) for n, p in model.named_parameters()]
    G["param_names"] = np.array([n for n, _ in named])
    opt = torch.optim.Adam([{"params": [p for n, p in named if "encoder" in n], "lr": 5e-4 / 2},
                            {"params": [p for n, p in named if "decoder" in n]}],
                           5e-4, (0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    state = {"masks": None, "k": 0}
    for blk in model.encoder.layers:
        if hasattr(blk, "dropout"):
            blk.dropout = Replay(blk.dropout.p, state)
    N, H, W = 2, 32, 64
    weights = [torch.tensor(fx.WEIGHT_BDD), torch.tensor(WEIGHT_IDD)]
    gen = torch.Generator().manual_seed(78)
    snap = lambda: [p.detach().clone() for _, p in named]
    prev = snap()
    model.train()
    losses = []
    for ind, nc in enumerate((20, 27)):
        images, labels = fx.make_batch(N, H, W, nc, seed=400 + ind)
        G[f"images{ind}"], G[f"labels{ind}"] = images.numpy(), labels.numpy()
        ms = O.draw_dropout_masks(N, gen)
        G[f"mask{ind}"] = np.stack([np.pad(m.reshape(N, -1).numpy(), ((0, 0), (0, 128 - m.shape[1]))) for m in ms])
        state["masks"], state["k"] = ms, 0
        out = model(images, ind)
        opt.zero_grad()
        loss = torch.nn.NLLLoss(weights[ind])(torch.log_softmax(out, 1), labels[:, 0])
        loss.backward()
        G[f"grad_is_none{ind}"] = np.array([p.grad is None for _, p in named])
        G[f"grad_digest{ind}"] = np.stack([fx.tensor_digest(p.grad)[:3].numpy() if p.grad is not None
                                           else np.full(3, np.nan) for _, p in named])
        opt.step()
        losses.append(loss.item())
        G[f"logits{ind}"] = out.detach().numpy()
        cur = snap()
        G[f"delta{ind}"] = np.stack([fx.tensor_digest(a - b)[:3].numpy() for a, b in zip(cur, prev)])
        prev = cur
    G["losses"] = np.array(losses, dtype=np.float64)
    G["digest_final"] = np.stack([fx.tensor_digest(p)[:3].numpy() for _, p in named])
    G["adam_steps"] = np.array([int(opt.state[p]["step"]) for _, p in named])
    for k, v in model.state_dict().items():
        if O.is_buffer(k):
            G["buf_" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "mt_tiny.npz"), **G)
    print("mt_tiny.npz:", len(G), "arrays; losses", G["losses"], "steps", np.unique(G["adam_steps"]))


if __name__ == "__main__":
    import warnings
    warnings.simplefilter("ignore")
    main()
