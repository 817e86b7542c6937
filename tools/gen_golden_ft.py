#!/usr/bin/env python3
"""tests/golden/ft_tiny.npz: the reference's fine-tuning baseline model (models/erfnet_ftp2.py,
heads 20/20/27, imported) on CPU, tiny shapes: state-dict layout, eval-mode logits of all three
heads, and one fine-tuning iteration of main_FT2_flexible_new.py:262-283 (train-mode forward
through decoder_new with recorded masks, CE, backward, Adam over encoder + decoder_new)."""
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
    ref = importlib.import_module("models.erfnet_ftp2")
    G = {}
    torch.manual_seed(0)
    model = ref.Net(20, 20, 27)
    sd = model.state_dict()
    fx.perturb_bn(sd, seed=41)
    G["state_keys"] = np.array(list(sd.keys()))
    images, labels = fx.make_batch(2, 32, 64, 27, seed=500)
    G["images"], G["labels"] = images.numpy(), labels.numpy()
    model.eval()
    with torch.no_grad():
        G["eval_old1"] = model(images, True, False, False).numpy()
        G["eval_old2"] = model(images, False, True, False).numpy()
        G["eval_new"] = model(images, False, False, True).numpy()
    for name, m in model.named_parameters():                       # :220-222
        if "decoder_old1" in name or "decoder_old2" in name:
            m.requires_grad = False
    named = list(model.named_parameters())
    G["param_names"] = np.array([n for n, _ in named])
    params = list(model.encoder.parameters()) + list(model.decoder_new.parameters())     # :224-229
    opt = torch.optim.Adam(params, 5e-4, (0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    state = {"masks": None, "k": 0}
    for blk in model.encoder.layers:
        if hasattr(blk, "dropout"):
            blk.dropout = Replay(blk.dropout.p, state)
    ms = O.draw_dropout_masks(2, torch.Generator().manual_seed(79))
    G["mask"] = np.stack([np.pad(m.reshape(2, -1).numpy(), ((0, 0), (0, 128 - m.shape[1]))) for m in ms])
    state["masks"] = ms
    before = [p.detach().clone() for _, p in named]
    model.train()
    out = model(images, decoder_old1=False, decoder_old2=False, decoder_new=True)
    opt.zero_grad()
    w = torch.tensor(WEIGHT_IDD)
    loss = torch.nn.NLLLoss(w)(torch.log_softmax(out, 1), labels[:, 0])
    loss.backward()
    opt.step()
    G["train_logits"], G["loss"] = out.detach().numpy(), np.float64(loss.item())
    G["grad_digest"] = np.stack([fx.tensor_digest(p.grad)[:3].numpy() if p.grad is not None
                                 else np.full(3, np.nan) for _, p in named])
    G["delta"] = np.stack([fx.tensor_digest(p.detach() - b)[:3].numpy() for (_, p), b in zip(named, before)])
    for k, v in model.state_dict().items():
        if O.is_buffer(k):
            G["buf_" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "ft_tiny.npz"), **G)
    print("ft_tiny.npz", len(G), "arrays; loss", G["loss"])


if __name__ == "__main__":
    import warnings
    warnings.simplefilter("ignore")
    main()
