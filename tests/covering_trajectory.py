"""Multi-step parity ON THE KERNEL MIX THE BENCH SHIPS (VERDICT r5 #3 / missing #2).

The mIoU protocol and its paired trajectories run 32x64 images: the deepest map is 4x8, so for
d = 4 / 8 / 16 (and d = 2 along H) no Winograd quad is complete and the direct `sconv` kernel runs --
the F(4,3) `w4conv` forms that carry a quarter of the bench's kernel time contribute almost nothing
there, and the covering-size check of round 4 was ONE gate-forced step per state.  Here, from a
given trained state (weights, BN buffers, Adam moments / step counts / learning rates), the HIP path
and the oracle both make K = 16 FREE-GATE training steps (each implementation's own ReLU decisions,
the oracle's own Adam restatement) on N = 2, 256x512 batches (`tests/miou_protocol.covering_batch`:
deepest map 32x64, every dilation 2..16 forms complete quads on both axes), identical batches and
dropout masks; after 8 and 16 steps both states are scored on 8 held-out covering batches through
the same (HIP) eval path:

  * logits of both heads: relative L2 distance,
  * argmax agreement (fraction of pixels),
  * confusion-matrix mIoU of both heads, pairwise |HIP - oracle| <= 0.1 point (the north_star's
    tolerance; the reference scores after training, not after one step:
    train_new_task_step2.py:340-347),

and the in-library launch profile of EVERY training step must show `w4conv` carrying more than half
of the 3-tap C -> C conv launches and no launch on the direct `sconv` path.

Used by tests/test_miou_parity.py (the three trained states of the protocol) and by
tests/test_covering_trajectory.py (a seeded pseudo-trained state: a quick stand-alone run).
Test infrastructure: imports the oracle.
"""
import torch

from oracle import fixtures as fx
from oracle import rap_oracle as O
from tests import miou_protocol as MP

TRAJ_K = (8, 16)
HELD_OUT = 8
# Bounds on the two auxiliary statistics (the statement proper is |d mIoU| <= 0.1 point).  Measured on the MI355X
# (profiles/r06_covering_trajectories.txt): from TRAINED states (new-domain head at 80+ % mIoU) the argmax of the two
# implementations agrees on >= 99.998 % of the held-out pixels after 16 steps; from the stand-alone test's
# pseudo-trained state (a head 12 steps away from its random init, mIoU 1.4 %: nearly every pixel is a near-tie
# between classes) 99.96 % after 8 and 99.5 % after 16 steps -- callers pass the bound that fits their state.
MIN_ARGMAX_AGREEMENT = 0.9999     # measured from the protocol's three trained states: 99.9981 .. 100.0000 %
MAX_LOGIT_REL_L2 = 2e-3           # measured: 3.7e-07 .. 2.1e-04


def _cpu(sd):
    return {k: v.detach().cpu().clone() for k, v in sd.items()}


def adam_snapshot(opt):
    m, v = opt.exp_avg.detach().cpu().clone(), opt.exp_avg_sq.detach().cpu().clone()
    return m, v, [(g["step"], g["lr"], g["offset"]) for g in opt.param_groups]


def _train_batch(seed, j, step1):
    images, labels = MP.covering_batch(100000 + 1000 * seed + j, old_domain=step1)
    m_new, m_old = MP.masks_for(910000 + 100 * seed + j, images.shape[0])
    return images, labels, m_new, m_old


def _held_out(seed, b, task):
    return MP.covering_batch(700000 + 100 * seed + b, old_domain=(task == 0))


def _profile_paths(prof, ops):
    by = {}
    for k, cin, cout, nt, _, _ in prof:
        if k in ops._PROF_CONV and cin == cout and cin in (64, 128) and nt in (3, 4):
            by[k] = by.get(k, 0) + 1
    return by


def hip_trajectory(dev, pre_sd, teacher_sd, adam, seed, ks=TRAJ_K):
    """max(ks) free-gate training iterations of the SHIPPED engine (Step2Engine on three streams, or
    Step1Engine when ``teacher_sd`` is None) from the given state on covering-size batches.
    -> ({K: state dict}, [names per optimizer group], per-step conv launch counts by kernel family)"""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_new_task_step2 as T
    from mdil_ss_amd.engine import Step1Engine, Step2Engine
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    weight = torch.tensor(fx.WEIGHT_BDD).to(dev)
    step1 = teacher_sd is None
    if step1:
        model = Net([20], 1, 0)
        model.load_state_dict(pre_sd)
        model.to(dev)
        eng = Step1Engine(model, weight, current_task=0)
        eng.optimizer.param_groups[0]["names"] = [n for n, _ in model.named_parameters()]
    else:
        model = Net([20, 20], 2, 1)
        model.load_state_dict(pre_sd)
        model.to(dev)
        frozen = Net([20], 1, 0)
        frozen.load_state_dict(teacher_sd)
        frozen.to(dev)
        ops.invalidate_packs()
        T.current_task = 1
        T.apply_step2_freeze(model, frozen, 1)
        eng = Step2Engine(model, frozen, weight, current_task=1, lambdac=MP.CONFIG["lambdac"],
                          is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
        by_id = {id(p): n for n, p in model.named_parameters()}
        for g in eng.optimizer.param_groups:
            g["names"] = [by_id[id(p)] for p in g["params"]]
    m_all, v_all, groups = adam

    def restore():
        model.load_state_dict(pre_sd)                   # in place: weights and BN buffers
        ops._stale_refresh()                            # packed weight images follow, on this stream
        eng.optimizer.exp_avg.copy_(m_all.to(dev))
        eng.optimizer.exp_avg_sq.copy_(v_all.to(dev))
        for g, (step, lr, off) in zip(eng.optimizer.param_groups, groups):
            assert g["offset"] == off
            g["step"], g["lr"] = step, lr
        torch.cuda.synchronize()

    restore()
    if not step1:
        # the engine's first iteration runs on one stream (it creates the packed images and scratch
        # every stream reads afterwards): spend it, put the state back, then run the shipped schedule
        images, labels, m_new, m_old = _train_batch(seed, 0, step1)
        q = [m_new, m_old]
        model.mask_provider = lambda n: q.pop(0)
        eng.iteration(images.to(dev), labels.to(dev))
        restore()
    states, paths = {}, []
    for j in range(max(ks)):
        images, labels, m_new, m_old = _train_batch(seed, j, step1)
        q = [m_new] if step1 else [m_new, m_old]
        model.mask_provider = lambda n: q.pop(0)
        ops.profile_begin()
        eng.iteration(images.to(dev), labels.to(dev))
        paths.append(_profile_paths(ops.profile_end(), ops))
        if not step1:
            assert getattr(eng, "multi_stream", False)
        if j + 1 in ks:
            torch.cuda.synchronize()
            states[j + 1] = _cpu(model.state_dict())
    model.mask_provider = None
    return states, [list(g["names"]) for g in eng.optimizer.param_groups], paths


def oracle_trajectory(pre_sd, teacher_sd, adam, group_names, seed, ks=TRAJ_K):
    """The same steps by the oracle (its own gates, its own Adam restatement) -> {K: state dict}."""
    weight = torch.tensor(fx.WEIGHT_BDD)
    step1 = teacher_sd is None
    trainable = (lambda n: True) if step1 else (lambda n: O.step2_trainable("module." + n, 1))
    S = {k: v.clone() for k, v in pre_sd.items()}
    for n in S:
        if S[n].is_floating_point() and not O.is_buffer(n):
            S[n].requires_grad_(trainable(n))
    T_sd = None if step1 else {k: v.clone() for k, v in teacher_sd.items()}
    m_all, v_all, groups = adam
    mom = {}
    for gi, names in enumerate(group_names):
        step, lr, off = groups[gi]
        for n in names:
            k = S[n].numel()
            mom[n] = (m_all[off:off + k].view(S[n].shape).clone(), v_all[off:off + k].view(S[n].shape).clone(), gi)
            off += k
    states = {}
    for j in range(max(ks)):
        images, labels, m_new, m_old = _train_batch(seed, j, step1)
        for n in mom:
            S[n].grad = None
        if step1:
            O.ce2d(O.net_forward(S, images, 0, True, m_new), labels[:, 0], weight).backward()
        else:
            O.step2_iteration(S, T_sd, images, labels, weight, 1, MP.CONFIG["lambdac"], m_new, m_old)
        with torch.no_grad():
            for n, (m, v, gi) in mom.items():
                if S[n].grad is not None:
                    step, lr, _ = groups[gi]
                    O.adam_l2_step(S[n], S[n].grad, m, v, step + j + 1, lr)
        if j + 1 in ks:
            states[j + 1] = {n: t.detach().clone() for n, t in S.items()}
    return states


def score_pair(dev, sd_hip, sd_oracle, step1, seed, held_out=HELD_OUT):
    """Both states through the HIP eval path on the held-out covering batches.
    -> {head: (logit rel-L2, argmax agreement, mIoU hip, mIoU oracle)}"""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd.iouEval import iouEval
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    heads = ((0, "old"),) if step1 else ((1, "new"), (0, "old"))
    logits, miou = {}, {}
    for who, sd in (("hip", sd_hip), ("oracle", sd_oracle)):
        ops.invalidate_packs()
        model = Net([20], 1, 0) if step1 else Net([20, 20], 2, 1)
        model.load_state_dict(sd)
        model.to(dev).eval()
        with torch.no_grad():
            for task, name in heads:
                ev = iouEval(20, 19)
                outs = []
                for b in range(held_out):
                    images, labels = _held_out(seed, b, task)
                    y = model(images.to(dev), task)
                    ev.addBatch(y, labels.to(dev))
                    outs.append(y.contiguous().clone())
                logits[(who, name)] = outs
                miou[(who, name)] = float(ev.getIoU()[0])
    res = {}
    for _, name in heads:
        num = den = 0.0
        same = tot = 0
        for a, b in zip(logits[("hip", name)], logits[("oracle", name)]):
            num += float((a.double() - b.double()).pow(2).sum())
            den += float(b.double().pow(2).sum())
            # (class 19 = ignore takes part in the argmax like in iouEval's fused kernel)
            same += int((a.argmax(1) == b.argmax(1)).sum())
            tot += a.shape[0] * a.shape[2] * a.shape[3]
        res[name] = ((num / den) ** 0.5, same / tot, miou[("hip", name)], miou[("oracle", name)])
    return res


def covering_trajectory(dev, tag, where, pre_sd, teacher_sd, adam, seed, ks=TRAJ_K,
                        min_agreement=MIN_ARGMAX_AGREEMENT, max_rel_l2=MAX_LOGIT_REL_L2):
    """Run both trajectories from one trained state and assert the statements of the module
    docstring.  -> {K: {head: (rel-L2, agreement, mIoU hip, mIoU oracle)}}"""
    step1 = teacher_sd is None
    hip, group_names, paths = hip_trajectory(dev, pre_sd, teacher_sd, adam, seed, ks)
    n_graphs = (1 + 1) if step1 else (3 + 2)          # forwards + backward graphs
    for j, by in enumerate(paths):
        wino = by.get("wconv", 0) + by.get("w4conv", 0)
        assert by.get("sconv", 0) == 0, (where, j, by)
        assert wino == 15 * 4 * n_graphs, (where, j, by)
        assert 2 * by.get("w4conv", 0) > wino + by.get("tapconv", 0), (where, j, by)
    orc = oracle_trajectory(pre_sd, teacher_sd, adam, group_names, seed, ks)
    out = {}
    for k in ks:
        out[k] = score_pair(dev, hip[k], orc[k], step1, seed)
        for name, (rel, agree, mh, mo) in out[k].items():
            print(f"[{tag}] covering-size trajectory from {where}, K={k:2d} free-gate steps at N={MP.COVER['batch']} "
                  f"{MP.COVER['height']}x{MP.COVER['width']} (w4conv {paths[0].get('w4conv', 0)} / wconv "
                  f"{paths[0].get('wconv', 0)} conv launches per step): {name} head logits rel-L2 {rel:.2e}, argmax "
                  f"agreement {agree * 100:.4f} %, mIoU HIP {mh * 100:.4f} oracle {mo * 100:.4f} "
                  f"(d = {(mh - mo) * 100:+.4f} point)", flush=True)
    for k in ks:
        for name, (rel, agree, mh, mo) in out[k].items():
            assert abs(mh - mo) * 100.0 <= 0.1, (where, k, name, mh, mo)
            assert agree >= min_agreement, (where, k, name, agree)
            assert rel <= max_rel_l2, (where, k, name, rel)
    return out
