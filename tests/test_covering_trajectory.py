"""GPU: multi-step FREE-GATE parity at the covering size (N = 2, 256x512: every dilation on the
F(4,3) / F(2,3) Winograd kernels the full-size step ships), stand-alone: tests/covering_trajectory.py
from a seeded pseudo-trained state.  The same check runs from the three trained states of the mIoU
protocol inside tests/test_miou_parity.py; this one needs no 12 k-iteration run in front of it.

State: the golden scenario's teacher / student (perturbed BatchNorm statistics and affines,
tests/helpers.golden_scenario) after 12 warm-up iterations of the shipped engine on covering-size
batches -- so the Adam moments, step counts and BN buffers are those of a run in progress, not the
zero moments of a first step (whose sign-like update lr * g / |g| amplifies noise-level gradient
elements to full-size steps in ANY two fp32 implementations).  Reference: the hot loop and its
scoring, train_new_task_step2.py:273-313,340-347.
"""
import pytest
import torch

from oracle import fixtures as fx
from tests import covering_trajectory as CT
from tests import helpers as Hh
from tests import miou_protocol as MP

pytestmark = pytest.mark.gpu

WARMUP = 12


def test_free_gate_trajectory_at_the_covering_size(golden):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_new_task_step2 as T
    from mdil_ss_amd.engine import Step2Engine
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    dev = torch.device("cuda:0")
    torch.set_num_threads(Hh.host_threads(16))          # the oracle leg runs on the host cores
    t_sd, s_sd = Hh.golden_scenario(golden)
    ops.invalidate_packs()
    weight = torch.tensor(fx.WEIGHT_BDD).to(dev)
    model, frozen = Net([20, 20], 2, 1), Net([20], 1, 0)
    model.load_state_dict(s_sd)
    frozen.load_state_dict(t_sd)
    model.to(dev)
    frozen.to(dev)
    ops.invalidate_packs()
    T.current_task = 1
    T.apply_step2_freeze(model, frozen, 1)
    eng = Step2Engine(model, frozen, weight, current_task=1, lambdac=MP.CONFIG["lambdac"],
                      is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
    for j in range(WARMUP):
        images, labels = MP.covering_batch(300000 + j)
        m_new, m_old = MP.masks_for(930000 + j, images.shape[0])
        q = [m_new, m_old]
        model.mask_provider = lambda n: q.pop(0)
        eng.iteration(images.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    model.mask_provider = None
    pre = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    adam = CT.adam_snapshot(eng.optimizer)
    assert all(step == WARMUP for step, _, _ in adam[2])
    del eng, model, frozen
    # (an untrained head -- mIoU 1.4 % on the held-out batches -- decides most pixels by near-ties: the argmax bound is
    # the loose one here, tests/covering_trajectory.py; the trained states of the mIoU protocol get the tight one)
    res = CT.covering_trajectory(dev, "hip", f"a pseudo-trained state ({WARMUP} warm-up steps)", pre, t_sd, adam, seed=5,
                                 min_agreement=0.99, max_rel_l2=2e-2)
    assert sorted(res) == list(CT.TRAJ_K) and all(set(v) == {"new", "old"} for v in res.values())
