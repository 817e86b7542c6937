"""GPU: multi-task joint model + round-robin engine (models/erfnet_multi_task.py,
train_multi_task.py:249-265) against the golden generated from the reference, and the trainer
mirror end to end."""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import rap_oracle as O
from tests import helpers as Hh
from tests.test_hip_parity import close

pytestmark = pytest.mark.gpu


def test_multi_task_round_against_reference_golden(golden_mt):
    gm = golden_mt
    dev = torch.device("cuda:0")
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd.engine import MultiTaskEngine
    from mdil_ss_amd.models.erfnet_multi_task import Net
    ops.invalidate_packs()
    model = Net([20, 27], 2, 0)
    model.load_state_dict(Hh.mt_scenario())
    model.to(dev)
    names = [n for n, _ in model.named_parameters()]
    assert ["module." + n for n in names] == list(gm["param_names"])
    weights = [torch.tensor(fx.WEIGHT_BDD, device=dev), torch.tensor(Hh.WEIGHT_IDD, device=dev)]
    eng = MultiTaskEngine(model, weights)
    params = dict(model.named_parameters())
    q = [Hh.mt_masks(gm, 0), Hh.mt_masks(gm, 1)]
    model.mask_provider = lambda n: q.pop(0)
    snap = lambda: [params[n].detach().cpu().clone() for n in names]
    prev = snap()
    enc = np.array([n.startswith("encoder") for n in names])
    for ind in (0, 1):
        images = torch.from_numpy(gm[f"images{ind}"]).to(dev)
        labels = torch.from_numpy(gm[f"labels{ind}"]).to(dev)
        if ind == 0:
            model.train()
            with torch.no_grad():          # logits of the untouched model (consumes no mask: eval off)
                pass
        ce = eng.sub_step(ind, images, labels)
        np.testing.assert_allclose(ce.item(), gm["losses"][ind], rtol=2e-5 if ind == 0 else 2e-3)
        cur = snap()
        got = np.stack([fx.tensor_digest(a - b)[:3].numpy() for a, b in zip(cur, prev)])
        ref = gm[f"delta{ind}"]
        head = np.array([n.startswith(f"decoder.{ind}.") for n in names])
        assert np.all(got[~(enc | head)] == 0), "the other head must not move"
        assert np.all(ref[~(enc | head)] == 0)
        rel = np.abs(got[:, 1] - ref[:, 1]) / (ref[:, 1] + 1e-12)
        assert np.median(rel[enc | head]) < 3e-2, np.median(rel[enc | head])
        if ind == 0:
            assert rel[enc | head].max() < 8e-2, rel.max()
        prev = cur
    steps = [g["step"] for g in eng.optimizer.param_groups]
    assert steps == [2, 1, 1]
    want = {True: 2, False: 1}
    assert [want[n.startswith("encoder")] for n in names] == list(gm["adam_steps"])
    sd = model.state_dict()
    for k, v in sd.items():
        if O.is_buffer(k):
            # second sub-step statistics are taken on post-Adam weights (see the drift note in
            # tests/test_oracle_golden.py::test_oracle_multi_task_round_matches_reference)
            close(v.float(), torch.from_numpy(gm["buf_" + k]).float(), rtol=5e-3, atol=3e-3, what=k)


def test_multi_task_forward_logits(golden_mt):
    """Train-mode logits of head 0 on the untouched model == the reference's (first sub-step)."""
    gm = golden_mt
    dev = torch.device("cuda:0")
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd.models.erfnet_multi_task import Net
    ops.invalidate_packs()
    model = Net([20, 27], 2, 0)
    model.load_state_dict(Hh.mt_scenario())
    model.to(dev).train()
    model.mask_provider = lambda n: Hh.mt_masks(gm, 0)
    with torch.no_grad():
        y = model(torch.from_numpy(gm["images0"]).to(dev), 0)
    # 13 train-mode BN layers over only 64..1024 pixels each: fp32 forward agrees to ~1e-4 of scale
    close(y, torch.from_numpy(gm["logits0"]), rtol=5e-4, atol=2e-4, what="head-0 logits")


def _scalars(work, run):
    """{tag: [(epoch, value)]} of the one event file under ``run`` (the reference's ``writer.add_scalar`` rows)."""
    import glob
    from mdil_ss_amd.scalar_log import read_scalars
    ev = glob.glob(str(work / run / "events.out.tfevents.*"))
    assert len(ev) == 1, (run, ev)
    out = {}
    for step, tag, value in read_scalars(ev[0]):
        out.setdefault(tag, []).append((step, value))
    return out


def test_multi_task_trainer_end_to_end(tmp_path, monkeypatch):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_multi_task as T
    ops.invalidate_packs()
    work = tmp_path / "run"
    work.mkdir()
    monkeypatch.chdir(work)
    args = T.build_parser().parse_args([
        "--savedir", "mt/CSBDDIDD", "--num-epochs", "1", "--batch-size", "2", "--dataset", "CSBDDIDD",
        "--datasets", "CS", "BDD", "IDD", "--num-classes", "20", "20", "27", "--nb_tasks", "3",
        "--height", "32", "--width", "64", "--synthetic", "8", "--num-workers", "0"])
    T.main(args)
    save = tmp_path / "save" / "mt" / "CSBDDIDD"
    name = "CSBDDIDD_erfnet_multi_task_1_2RAP_FT_step3.pth.tar"
    for f in ("opts.txt", "model.txt", "automated_log.txt", "checkpoint_" + name, "model_best_" + name):
        assert (save / f).exists(), f
    ck = torch.load(save / ("checkpoint_" + name), map_location="cpu", weights_only=False)
    assert len(ck["state_dict"]) == 232 + 3 * 199 or all(k.startswith("module.") for k in ck["state_dict"])
    steps = sorted({int(v["step"]) for v in ck["optimizer"]["state"].values()})
    assert steps == [4, 12], steps           # 4 iterations x 3 datasets: encoder 12 steps, heads 4
    # epoch-wise TensorBoard scalars (train_multi_task.py:122-124,290-301)
    sc = _scalars(work, "Adaptations/runs_CSBDDIDD_erfnet_multi_task_1_2RAP_FT_step3")
    assert sorted(sc) == sorted(f"{k}_{d}" for k in ("val_acc", "val_loss", "train_loss") for d in ("CS", "BDD", "IDD")), sc
    assert all(v[0][1] > 0 for t, v in sc.items() if t.startswith("train_loss"))
