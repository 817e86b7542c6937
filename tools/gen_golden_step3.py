#!/usr/bin/env python3
"""tests/golden/step3_tiny.npz: one step-3 iteration (train_new_task_step3.py:303-356) of the
IMPORTED REFERENCE model (CS|BDD -> IDD: classes [20,20,27]) on CPU, tiny shapes, recorded masks.
Arrays only; no reference source enters the repository."""
import importlib
import os
import re
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import fixtures as fx          # noqa: E402
from oracle import rap_oracle as O         # noqa: E402

torch.set_num_threads(8)
WEIGHT_IDD = [3.235635601598852, 6.76221624390441, 9.458242359884549, 9.446818215454014,
              9.947040673126763, 9.789672819856547, 9.476665808564432, 10.465565126694731,
              9.59189547383129, 7.637805282159825, 8.990899026692638, 9.26222234098628,
              10.265657138809514, 9.386517631614392, 8.357391489170013, 9.910382864314824,
              10.389977663948363, 8.997422571963602, 10.418070541191673, 10.483262606962834,
              9.511436923349441, 7.597725385711079, 6.1734896019878205, 9.787631041755187,
              3.9178330193378708, 4.417448652936843, 0.0]


class Replay(torch.nn.Module):
    def __init__(self, p, state):
        super().__init__()
        self.p, self.state = p, state

    def forward(self, x):
        if not self.training:
            return x
        m = self.state["masks"][self.state["k"]]
        self.state["k"] += 1
        return x * m


def main():
    sys.path.insert(0, "/root/reference")
    ref = importlib.import_module("models.erfnet_RA_parallel")
    G = {}
    torch.manual_seed(1)
    teacher = ref.Net([20, 20], 2, 1)
    torch.manual_seed(0)
    student = ref.Net([20, 20, 27], 3, 2)
    t_sd = teacher.state_dict()
    fx.perturb_bn(t_sd, seed=21)
    ckpt = {"module." + k: v.clone() for k, v in t_sd.items()}
    t = 2
    s_keys = {"module." + k for k in student.state_dict()}
    new = {k: v for k, v in ckpt.items() if k in s_keys}           # :566-592
    for k, v in ckpt.items():
        if "encoder" in k:
            if "parallel_conv" in k or "bn" in k:
                if ".{}.weight".format(t - 1) in k:
                    new[re.sub(".{}.weight".format(t - 1), ".{}.weight".format(t), k)] = v
                elif ".{}.bias".format(t - 1) in k:
                    new[re.sub(".{}.bias".format(t - 1), ".{}.bias".format(t), k)] = v
        elif "decoder" in k and "output_conv" not in k:
            new[re.sub("decoder.{}".format(t - 1), "decoder.{}".format(t), k)] = v
    G["init_loaded_keys"] = np.array(sorted(new.keys()))
    student.load_state_dict({k[7:]: v for k, v in new.items()}, strict=False)
    G["student_keys"] = np.array(list(student.state_dict().keys()))
    for p in teacher.parameters():
        p.requires_grad = False
    for name, m in student.named_parameters():                       # :229-241
        if "decoder" in name:
            if "decoder.{}".format(t) not in name:
                m.requires_grad = False
        elif "encoder" in name and ("bn" in name or "parallel_conv" in name):
            if not (".{}.weight".format(t) in name or ".{}.bias".format(t) in name):
                m.requires_grad = False
    named = [("module." + n, p) for n, p in student.named_parameters()]
    G["param_names"] = np.array([n for n, _ in named])
    G["requires_grad"] = np.array([p.requires_grad for _, p in named])
    opt = torch.optim.Adam([{"params": [p for n, p in named if O.is_shared(n)], "lr": 5e-6},
                            {"params": [p for n, p in named if O.is_ds_curr(n, t)]}],
                           5e-4, (0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    state = {"masks": None, "k": 0}
    for net in (student, teacher):
        for blk in net.encoder.layers:
            if hasattr(blk, "dropout"):
                blk.dropout = Replay(blk.dropout.p, state)
    N, H, W = 2, 32, 64
    images, labels = fx.make_batch(N, H, W, 27, seed=300)
    G["images"], G["labels"] = images.numpy(), labels.numpy()
    gen = torch.Generator().manual_seed(77)
    masks = {k: O.draw_dropout_masks(N, gen) for k in ("new", "prev1", "prev0", "teach1", "teach0")}
    for k, ms in masks.items():
        G["mask_" + k] = np.stack([np.pad(m.reshape(N, -1).numpy(), ((0, 0), (0, 128 - m.shape[1]))) for m in ms])
    weight = torch.tensor(WEIGHT_IDD)
    crit = torch.nn.NLLLoss(weight)
    kl = torch.nn.KLDivLoss()

    def fwd(net, task, key):
        state["masks"], state["k"] = masks[key], 0
        return net(images, task)

    snap = lambda: [p.detach().clone() for _, p in named]
    p_init = snap()
    student.train()                      # the teacher is left in its default (train) mode: quirk
    out = fwd(student, t, "new")
    ce = crit(torch.log_softmax(out, 1), labels[:, 0])
    opt.zero_grad()
    ce.backward()
    opt.step()
    G["logits_new"] = out.detach().numpy()
    p_a = snap()
    G["delta_ce_step"] = np.stack([fx.tensor_digest(a - b)[:3].numpy() for a, b in zip(p_a, p_init)])
    p1, p0 = fwd(student, t - 1, "prev1"), fwd(student, t - 2, "prev0")
    t1, t0 = fwd(teacher, t - 1, "teach1").detach(), fwd(teacher, t - 2, "teach0").detach()
    k1 = kl(torch.softmax(p1, 1), torch.softmax(t1, 1))
    k0 = kl(torch.softmax(p0, 1), torch.softmax(t0, 1))
    kd = 0.1 * (k1 + k0)
    opt.zero_grad()
    kd.backward()
    G["kd_grad_is_none"] = np.array([p.grad is None for _, p in named])
    opt.step()
    G["losses"] = np.array([ce.item(), k1.item(), k0.item()], dtype=np.float64)
    G["delta_kd_step"] = np.stack([fx.tensor_digest(a - b)[:3].numpy() for a, b in zip(snap(), p_a)])
    G["digest_final"] = np.stack([fx.tensor_digest(p)[:3].numpy() for _, p in named])
    for k, v in student.state_dict().items():
        if O.is_buffer(k):
            G["sbuf_" + k] = v.numpy().copy()
    for k, v in teacher.state_dict().items():
        if O.is_buffer(k):
            G["tbuf_" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "step3_tiny.npz"), **G)
    print("step3_tiny.npz:", len(G), "arrays; losses", G["losses"])


if __name__ == "__main__":
    import warnings
    warnings.simplefilter("ignore")
    main()
