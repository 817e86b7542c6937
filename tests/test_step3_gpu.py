"""GPU: the two-old-domain step (train_new_task_step3.py:303-356) through engine.Step3Engine
against the golden generated from the reference (tools/gen_golden_step3.py), the four-stream
schedule against the single-stream one, and the step-3 trainer mirror end to end."""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import rap_oracle as O
from tests import helpers as Hh
from tests.test_hip_parity import close

pytestmark = pytest.mark.gpu


def _build(dev):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    teacher_sd, student_sd = Hh.step3_scenario()
    student = Net([20, 20, 27], 3, 2)
    student.load_state_dict(student_sd)
    teacher = Net([20, 20], 2, 1)
    teacher.load_state_dict(teacher_sd)
    student.to(dev)
    teacher.to(dev)
    for p in teacher.parameters():
        p.requires_grad = False
    for n, p in student.named_parameters():
        p.requires_grad = O.step2_trainable("module." + n, 2)
    return student, teacher


def _engine(dev, g3, streams, repeat=1):
    from mdil_ss_amd import train_new_task_step3 as T
    from mdil_ss_amd.engine import Step3Engine
    T.current_task = 2
    student, teacher = _build(dev)
    eng = Step3Engine(student, teacher, torch.tensor(Hh.WEIGHT_IDD, device=dev), current_task=2,
                      lambdac=0.1, is_shared=T.is_shared, is_ds_curr=T.is_DS_curr, streams=streams)
    m = Hh.step3_masks(g3)
    qs = [m["new"], m["prev1"], m["prev0"]] * repeat
    qt = [m["teach1"], m["teach0"]] * repeat
    student.mask_provider = lambda n: qs.pop(0)
    teacher.mask_provider = lambda n: qt.pop(0)
    return eng, student, teacher


def test_step3_iteration_against_reference_golden(golden_step3):
    g3 = golden_step3
    dev = torch.device("cuda:0")
    eng, student, teacher = _engine(dev, g3, streams=False)
    names = [n for n, _ in student.named_parameters()]
    assert ["module." + n for n in names] == list(g3["param_names"])
    params = dict(student.named_parameters())
    assert [params[n].requires_grad for n in names] == list(g3["requires_grad"])
    images, labels = torch.from_numpy(g3["images"]).to(dev), torch.from_numpy(g3["labels"]).to(dev)
    snap = lambda: [params[n].detach().cpu().clone() for n in names]
    p0 = snap()
    inner = eng.optimizer.step
    snaps = []

    def spy(*a, **k):
        inner(*a, **k)
        snaps.append(snap())
    eng.optimizer.step = spy
    ce, k1, k0 = eng.iteration(images, labels)
    np.testing.assert_allclose([ce.item(), k1.item()], g3["losses"][:2], rtol=2e-5)
    # kld(t-2) is measured after the CE step moved the shared encoder by ~lr*sign(g)
    np.testing.assert_allclose(k0.item(), g3["losses"][2], rtol=2e-4)
    assert teacher.training, "the reference never puts the previous model in eval mode in step 3"
    g0, g1 = eng.optimizer.param_groups
    assert (g0["step"], g1["step"]) == (2, 1)
    shared = np.array([O.is_shared("module." + n) for n in names])
    ds = np.array([O.is_ds_curr("module." + n, 2) for n in names])
    for a, b, key in ((snaps[0], p0, "delta_ce_step"), (snaps[1], snaps[0], "delta_kd_step")):
        got = np.stack([fx.tensor_digest(x - y)[:3].numpy() for x, y in zip(a, b)])
        ref = g3[key]
        if key == "delta_kd_step":
            assert np.all(got[~shared] == 0), "the DS group must not step after the KD backward"
        assert np.all(got[~(shared | ds)] == 0), "frozen parameters moved"
        # Adam's first update is ~lr*sign(g): the summed |update| is insensitive to relu-gate
        # flips / rounding-noise gradients (see test_model_golden.py for the fp32 rationale)
        np.testing.assert_allclose(got[:, 1], ref[:, 1], rtol=5e-2, atol=1e-9)
        np.testing.assert_allclose(got[:, 2], ref[:, 2], rtol=5e-2, atol=1e-9)
    final = np.stack([fx.tensor_digest(x)[:3].numpy() for x in snaps[1]])
    # ||p + d|| moves by at most ||d_got - d_ref||: elements whose gradient is rounding noise take
    # +-lr with a noise-determined sign, so the bound is a fraction of the update norm itself
    slack = 0.5 * (g3["delta_ce_step"][:, 2] + g3["delta_kd_step"][:, 2])
    assert np.all(np.abs(final[:, 2] - g3["digest_final"][:, 2]) <= 2e-5 * g3["digest_final"][:, 2] + slack)
    for k, v in student.state_dict().items():
        if O.is_buffer(k):
            close(v.float(), torch.from_numpy(g3["sbuf_" + k]).float(), rtol=1e-3, atol=2e-4,
                  what=f"student buffer {k}")
    for k, v in teacher.state_dict().items():
        if O.is_buffer(k):
            close(v.float(), torch.from_numpy(g3["tbuf_" + k]).float(), rtol=1e-3, atol=2e-4,
                  what=f"previous-model buffer {k}")


def test_step3_four_stream_schedule_matches_single_stream(golden_step3):
    """Iteration 2 (the first one that runs on four streams) against the same iteration on one
    stream: same inputs, same masks -> same losses, same parameters after both optimizer steps."""
    g3 = golden_step3
    dev = torch.device("cuda:0")
    images, labels = torch.from_numpy(g3["images"]).to(dev), torch.from_numpy(g3["labels"]).to(dev)
    res = []
    for streams in (False, True):
        eng, student, teacher = _engine(dev, g3, streams=streams, repeat=2)
        eng.iteration(images, labels)
        out = eng.iteration(images, labels)
        torch.cuda.synchronize()
        assert eng.multi_stream == streams
        res.append(([float(v) for v in out], eng.optimizer.flat_param.clone(),
                    {k: v.clone() for k, v in teacher.state_dict().items() if O.is_buffer(k)}))
    (l_a, p_a, b_a), (l_b, p_b, b_b) = res
    np.testing.assert_allclose(l_a, l_b, rtol=1e-5)
    # sink order differs (two buffers summed once instead of one buffer in sequence): fp32 sums
    # agree to rounding; after Adam that is at most a few ulp of the update
    assert float((p_a - p_b).abs().max()) < 2e-6
    for k in b_a:
        close(b_a[k].float(), b_b[k].float(), rtol=1e-5, atol=1e-6, what=k)


def test_step3_staggered_phase_b_is_bit_identical_to_lock_step(golden_step3):
    """Phase B of engine.Step3Engine (round 4): the second old-domain graph staggered behind the first,
    one backward per graph on its own stream -- same launches, same per-stream order, the two graphs'
    shared-encoder gradients still in separate flat buffers summed once: losses and parameters after
    three iterations (six optimizer steps) must equal the lock-step schedule's bit for bit."""
    g3 = golden_step3
    dev = torch.device("cuda:0")
    images, labels = torch.from_numpy(g3["images"]).to(dev), torch.from_numpy(g3["labels"]).to(dev)
    res = []
    for stagger in (None, 0, 8, 99):
        eng, student, teacher = _engine(dev, g3, streams=True, repeat=3)
        eng.stagger = stagger
        outs = [[float(v) for v in eng.iteration(images, labels)] for _ in range(3)]
        torch.cuda.synchronize()
        assert eng.multi_stream
        res.append((outs, eng.optimizer.flat_param.clone()))
    for outs, p in res[1:]:
        assert outs == res[0][0], (outs, res[0][0])
        assert torch.equal(p, res[0][1]), float((p - res[0][1]).abs().max())


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


def test_step3_trainer_from_step2_checkpoint(tmp_path, monkeypatch):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_new_task_step3 as T
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    work = tmp_path / "run"
    work.mkdir()
    monkeypatch.chdir(work)
    torch.manual_seed(3)
    step2 = Net([20, 20], 2, 1)
    ckpt = tmp_path / "step2.pth.tar"
    torch.save({"state_dict": {"module." + k: v for k, v in step2.state_dict().items()}}, ckpt)
    args = T.build_parser().parse_args([
        "--savedir", "t/CS1_BDD2_IDD3", "--num-epochs", "1", "--batch-size", "2", "--state", str(ckpt),
        "--dataset-new", "IDD", "--datasets", "cityscapes", "BDD", "IDD", "--num-classes", "20", "20",
        "27", "--num-classes-old", "20", "20", "--nb_tasks", "3", "--current_task", "2", "--height",
        "32", "--width", "64", "--synthetic", "8", "--num-workers", "0", "--steps-loss", "2"])
    T.main(args)
    save = tmp_path / "save" / "t" / "CS1_BDD2_IDD3"
    name = "IDD_erfnet_RA_parallel_1_2RAPFT_KLD_step3.pth.tar"
    for f in ("opts.txt", "model.txt", "automated_log.txt", "checkpoint_" + name, "model_best_" + name):
        assert (save / f).exists(), f
    ck = torch.load(save / ("checkpoint_" + name), map_location="cpu", weights_only=False)
    new = {k[7:]: v for k, v in ck["state_dict"].items()}
    old = step2.state_dict()
    assert tuple(new["decoder.2.output_conv.weight"].shape) == (16, 27, 2, 2)
    for k in ("decoder.0.output_conv.weight", "decoder.1.output_conv.weight",
              "encoder.layers.1.parallel_conv_1.1.weight", "encoder.layers.1.bns_1.0.weight"):
        assert torch.equal(new[k], old[k]), k
    assert not torch.equal(new["encoder.layers.1.conv3x1_1.weight"], old["encoder.layers.1.conv3x1_1.weight"])
    assert not torch.equal(new["encoder.layers.1.parallel_conv_1.2.weight"],
                           old["encoder.layers.1.parallel_conv_1.1.weight"])
    st = ck["optimizer"]["state"]
    steps = sorted({int(v["step"]) for v in st.values()})
    assert steps == [4, 8], steps            # 4 iterations: DS group 1 step each, shared group 2
    # epoch-wise TensorBoard scalars (train_new_task_step3.py:116-118,417-426): val_acc / val_loss per dataset
    sc = _scalars(work, "Adaptations/runs_IDD_erfnet_RA_parallel_1_2RAPFT_KLD_step3")
    assert sorted(sc) == sorted(f"val_{k}_{d}" for k in ("acc", "loss") for d in ("cityscapes", "BDD", "IDD")), sc
    assert all([e for e, _ in v] == [1] for v in sc.values())


def test_step3_free_gate_trajectory_at_the_covering_size():
    """Multi-step parity of the two-old-domain step on the kernels the full-size network launches (round 6; the step-2
    counterpart is tests/covering_trajectory.py): from the step-3 scenario after 6 warm-up iterations of the shipped
    Step3Engine (four streams) the HIP path and the oracle both make 8 FREE-GATE iterations -- two optimizer steps each,
    the previous model in train mode (its running statistics move, train_new_task_step3.py:301-356) -- on N = 2, 256x512
    batches with identical dropout masks, each side's own ReLU decisions and the oracle's own Adam restatement (per-group
    step counts: shared encoder 2 per iteration, new-domain group 1); then both students are scored in eval mode on 4
    held-out batches, all three heads: logits rel-L2, argmax agreement, confusion-matrix mIoU |d| <= 0.1 point."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_new_task_step3 as T
    from mdil_ss_amd.engine import Step3Engine
    from mdil_ss_amd.iouEval import iouEval
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    from tests import covering_trajectory as CT
    from tests import miou_protocol as MP
    dev = torch.device("cuda:0")
    torch.set_num_threads(Hh.host_threads(16))
    WARM, K, HELD = 6, 8, 4
    T.current_task = 2
    student, teacher = _build(dev)
    weight_cpu = torch.tensor(Hh.WEIGHT_IDD)
    eng = Step3Engine(student, teacher, weight_cpu.to(dev), current_task=2, lambdac=0.1, is_shared=T.is_shared,
                      is_ds_curr=T.is_DS_curr)

    def masks_of(it):
        g = torch.Generator().manual_seed(77000 + it)
        return {k: O.draw_dropout_masks(2, g) for k in ("teach1", "teach0", "new", "prev1", "prev0")}

    def batch(it):
        return MP.covering_batch(810000 + it)

    def hip_iteration(it):
        m = masks_of(it)
        qs, qt = [m["new"], m["prev1"], m["prev0"]], [m["teach1"], m["teach0"]]
        student.mask_provider = lambda n: qs.pop(0)
        teacher.mask_provider = lambda n: qt.pop(0)
        images, labels = batch(it)
        eng.iteration(images.to(dev), labels.to(dev))
        assert not qs and not qt

    for it in range(WARM):
        hip_iteration(it)
    torch.cuda.synchronize()
    assert getattr(eng, "multi_stream", False)
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    S0, T0 = cpu(student.state_dict()), cpu(teacher.state_dict())
    m_all, v_all, groups = CT.adam_snapshot(eng.optimizer)
    by_id = {id(p): n for n, p in student.named_parameters()}
    group_names = [[by_id[id(p)] for p in g["params"]] for g in eng.optimizer.param_groups]
    assert [st for st, _, _ in groups] == [2 * WARM, WARM], groups
    for it in range(WARM, WARM + K):
        hip_iteration(it)
    torch.cuda.synchronize()
    S_hip = cpu(student.state_dict())
    # ---- the oracle from the same state
    S = {k: v.clone() for k, v in S0.items()}
    Tt = {k: v.clone() for k, v in T0.items()}
    for n in S:
        if S[n].is_floating_point() and not O.is_buffer(n):
            S[n].requires_grad_(O.step2_trainable("module." + n, 2))
    mom, steps, lrs = {}, {}, {}
    for gi, names in enumerate(group_names):
        step, lr, off = groups[gi]
        steps[gi], lrs[gi] = step, lr
        for n in names:
            k = S[n].numel()
            mom[n] = (m_all[off:off + k].view(S[n].shape).clone(), v_all[off:off + k].view(S[n].shape).clone(), gi)
            off += k

    def opt_step(tag):
        touched = sorted({gi for n, (_, _, gi) in mom.items() if S[n].grad is not None})
        assert touched == ([0, 1] if tag == "ce" else [0]), (tag, touched)      # torch >= 2: zero_grad -> None
        with torch.no_grad():
            for gi in touched:
                steps[gi] += 1
            for n, (m, v, gi) in mom.items():
                if S[n].grad is not None:
                    O.adam_l2_step(S[n], S[n].grad, m, v, steps[gi], lrs[gi])

    for it in range(WARM, WARM + K):
        images, labels = batch(it)
        O.step3_iteration(S, Tt, images, labels, weight_cpu, 2, 0.1, masks_of(it), opt_step)
    assert steps == {0: 2 * (WARM + K), 1: WARM + K}
    # the previous model's running statistics moved identically on both sides (it is never put in eval mode)
    for k, v in cpu(teacher.state_dict()).items():
        if O.is_buffer(k) and v.is_floating_point():
            close(v, Tt[k], rtol=1e-3, atol=2e-4, what=f"previous-model buffer {k} after {WARM + K} iterations")
    # ---- both students, eval mode, three heads, held-out covering batches, through the same (HIP) eval path
    res = {}
    for who, sd in (("hip", S_hip), ("oracle", {n: t.detach().clone() for n, t in S.items()})):
        ops.invalidate_packs()
        model = Net([20, 20, 27], 3, 2)
        model.load_state_dict(sd)
        model.to(dev).eval()
        with torch.no_grad():
            for task, nc in ((2, 27), (1, 20), (0, 20)):
                ev, outs = iouEval(nc, nc - 1), []
                for b in range(HELD):
                    images, labels = MP.covering_batch(820000 + b, old_domain=(task == 0))
                    y = model(images.to(dev), task)
                    ev.addBatch(y, labels.to(dev))
                    outs.append(y.contiguous().clone())
                res[(who, task)] = (outs, float(ev.getIoU()[0]))
    for task in (2, 1, 0):
        (a, ma), (b, mb) = res[("hip", task)], res[("oracle", task)]
        num = sum(float((x.double() - y.double()).pow(2).sum()) for x, y in zip(a, b))
        den = sum(float(y.double().pow(2).sum()) for y in b)
        same = sum(int((x.argmax(1) == y.argmax(1)).sum()) for x, y in zip(a, b))
        tot = sum(x.shape[0] * x.shape[2] * x.shape[3] for x in a)
        rel, agree = (num / den) ** 0.5, same / tot
        print(f"step-3 covering-size trajectory, {K} free-gate iterations (16 optimizer steps) at N=2 256x512: head {task} "
              f"logits rel-L2 {rel:.2e}, argmax agreement {agree * 100:.4f} %, mIoU HIP {ma * 100:.4f} oracle {mb * 100:.4f} "
              f"(d = {(ma - mb) * 100:+.4f} point)", flush=True)
        assert abs(ma - mb) * 100.0 <= 0.1, (task, ma, mb)
        # (measured on the MI355X: see the printed line; the new 27-class head is 14 iterations from its random init:
        # near-ties everywhere, like the stand-alone step-2 test)
        assert rel <= (2e-2 if task == 2 else 2e-3) and agree >= (0.99 if task == 2 else 0.999), (task, rel, agree)
    student.mask_provider = teacher.mask_provider = None
