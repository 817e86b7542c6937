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


def test_multi_task_free_gate_trajectory_at_the_covering_size():
    """Multi-step parity of the joint multi-task loop on the kernels the full-size network launches (round 6; the step-2
    counterpart is tests/covering_trajectory.py): from the multi-task scenario after 6 warm-up rounds of the shipped
    MultiTaskEngine the HIP path and the oracle both make 8 FREE-GATE round-robin passes (two sub-steps each: head 0 on a
    20-class batch, head 1 on a 27-class batch; train_multi_task.py:249-265) on N = 2, 256x512 batches with identical
    dropout masks, each side's own ReLU decisions and the oracle's own Adam restatement (one step count per head group,
    two per round for the encoder); after 1 / 2 / 4 / 8 rounds both states are scored in eval mode on 4 held-out batches per head."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd.engine import MultiTaskEngine
    from mdil_ss_amd.iouEval import iouEval
    from mdil_ss_amd.models.erfnet_multi_task import Net
    from tests import covering_trajectory as CT
    from tests import miou_protocol as MP
    dev = torch.device("cuda:0")
    torch.set_num_threads(Hh.host_threads(16))
    WARM, K, HELD = 6, 8, 4
    ops.invalidate_packs()
    model = Net([20, 27], 2, 0)
    model.load_state_dict(Hh.mt_scenario())
    model.to(dev)
    w_cpu = [torch.tensor(fx.WEIGHT_BDD), torch.tensor(Hh.WEIGHT_IDD)]
    eng = MultiTaskEngine(model, [w.to(dev) for w in w_cpu])

    def masks_of(it):
        g = torch.Generator().manual_seed(88000 + it)
        return [O.draw_dropout_masks(2, g), O.draw_dropout_masks(2, g)]

    def batches(it):
        return [MP.covering_batch(830000 + 2 * it), MP.covering_batch(830001 + 2 * it, old_domain=True)]

    def hip_round(it):
        q = masks_of(it)
        model.mask_provider = lambda n: q.pop(0)
        for ind, (images, labels) in enumerate(batches(it)):
            eng.sub_step(ind, images.to(dev), labels.to(dev))
        assert not q

    for it in range(WARM):
        hip_round(it)
    torch.cuda.synchronize()
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    S0 = cpu(model.state_dict())
    m_all, v_all, groups = CT.adam_snapshot(eng.optimizer)
    by_id = {id(p): n for n, p in model.named_parameters()}
    group_names = [[by_id[id(p)] for p in g["params"]] for g in eng.optimizer.param_groups]
    assert [st for st, _, _ in groups] == [2 * WARM, WARM, WARM], groups
    KS = (1, 2, 4, K)
    S_hip = {}
    for it in range(WARM, WARM + K):
        hip_round(it)
        if it - WARM + 1 in KS:
            torch.cuda.synchronize()
            S_hip[it - WARM + 1] = cpu(model.state_dict())
    model.mask_provider = None
    # ---- the oracle from the same state
    S = {k: v.clone() for k, v in S0.items()}
    for n in S:
        if S[n].is_floating_point() and not O.is_buffer(n):
            S[n].requires_grad_(True)
    mom, steps, lrs = {}, {}, {}
    for gi, names in enumerate(group_names):
        step, lr, off = groups[gi]
        steps[gi], lrs[gi] = step, lr
        for n in names:
            k = S[n].numel()
            mom[n] = (m_all[off:off + k].view(S[n].shape).clone(), v_all[off:off + k].view(S[n].shape).clone(), gi)
            off += k

    def opt_step(ind):
        touched = sorted({gi for n, (_, _, gi) in mom.items() if S[n].grad is not None})
        assert touched == [0, 1 + ind], (ind, touched)               # the encoder and the visited head only
        with torch.no_grad():
            for gi in touched:
                steps[gi] += 1
            for n, (m, v, gi) in mom.items():
                if S[n].grad is not None:
                    O.adam_l2_step(S[n], S[n].grad, m, v, steps[gi], lrs[gi])

    S_orc = {}
    for it in range(WARM, WARM + K):
        O.mt_round(S, batches(it), w_cpu, masks_of(it), opt_step)
        if it - WARM + 1 in KS:
            S_orc[it - WARM + 1] = {n: t.detach().clone() for n, t in S.items()}
    assert steps == {0: 2 * (WARM + K), 1: WARM + K, 2: WARM + K}
    # ---- both states, eval mode, both heads, held-out covering batches, through the same (HIP) eval path
    def score(sd):
        ops.invalidate_packs()
        mdl = Net([20, 27], 2, 0)
        mdl.load_state_dict(sd)
        mdl.to(dev).eval()
        out = {}
        with torch.no_grad():
            for task, nc in ((0, 20), (1, 27)):
                ev, outs = iouEval(nc, nc - 1), []
                for b in range(HELD):
                    images, labels = MP.covering_batch(840000 + b, old_domain=(task == 1))
                    y = mdl(images.to(dev), task)
                    ev.addBatch(y, labels.to(dev))
                    outs.append(y.contiguous().clone())
                out[task] = (outs, float(ev.getIoU()[0]))
        return out

    for k in KS:
        rh, ro = score(S_hip[k]), score(S_orc[k])
        for task in (0, 1):
            (a, ma), (b, mb) = rh[task], ro[task]
            num = sum(float((x.double() - y.double()).pow(2).sum()) for x, y in zip(a, b))
            den = sum(float(y.double().pow(2).sum()) for y in b)
            same = sum(int((x.argmax(1) == y.argmax(1)).sum()) for x, y in zip(a, b))
            tot = sum(x.shape[0] * x.shape[2] * x.shape[3] for x in a)
            rel, agree = (num / den) ** 0.5, same / tot
            print(f"multi-task covering-size trajectory, {k} free-gate round(s) at N=2 256x512: head {task} logits rel-L2 {rel:.2e}, "
                  f"argmax agreement {agree * 100:.4f} %, mIoU HIP {ma * 100:.4f} oracle {mb * 100:.4f} "
                  f"(d = {(ma - mb) * 100:+.4f} point)", flush=True)
            assert abs(ma - mb) * 100.0 <= 0.1, (k, task, ma, mb)
            # Here EVERYTHING trains at 2.5e-4 .. 5e-4 from a random init (the scenario's mIoU is below 1 %): two fp32
            # implementations part exponentially -- measured on the MI355X: rel-L2 5e-5 / 1.4e-3 / 9e-3 / 3.4e-2 after
            # 1 / 2 / 4 / 8 rounds, a factor ~5 per doubling -- where the step-2 / step-3 trajectories (shared encoder
            # at 5e-6, trained or BN-perturbed states) stay at 1e-5 .. 2e-3.  The first round pins the kernels (two
            # optimizer steps apart from the one-step parity of the golden test), the later ones only bound the growth.
            assert rel <= (2e-3 if k == 1 else 0.15) and agree >= (0.999 if k == 1 else 0.9), (k, task, rel, agree)
