"""CPU: the C-ABI library loads without a GPU and exports every symbol include/mdil_hip.h
declares; host-side logic that needs no device (predicates, init rule, LR rule, packing keys)."""
import os
import re

import pytest
import torch


def _declared():
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "mdil_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mdil_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mdil_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names
    assert lib.mdil_version() >= 100


def test_product_refuses_cpu_tensors():
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    net = Net([20], 1, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 32, 64), 0)


def test_trainer_host_logic_matches_oracle(golden):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import train_new_task_step2 as T
    from mdil_ss_amd.engine import poly_factor
    from oracle import rap_oracle as O
    T.current_task = 1
    names = list(golden["param_names"])
    assert [T.is_shared(n) for n in names] == list(golden["is_shared"])
    assert [bool(T.is_DS_curr(n)) for n in names] == list(golden["is_ds_curr"])
    for e, (lr0, lr1) in zip(golden["lr_epochs"], golden["lr_values"]):
        assert 5e-6 * poly_factor(int(e), 150) == pytest.approx(lr0, rel=1e-12)
        assert 5e-4 * poly_factor(int(e), 150) == pytest.approx(lr1, rel=1e-12)
    # init rule on a fake step-1 checkpoint
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    t_sd = {"module." + k: v for k, v in Net([20], 1, 0).state_dict().items()}
    s_keys = {"module." + k for k in Net([20, 20], 2, 1).state_dict()}
    new = T.student_init_dict(t_sd, s_keys, 1)
    assert sorted(new.keys()) == list(golden["init_loaded_keys"])
    ref = O.student_init_from_teacher(t_sd, {k: None for k in s_keys}, 1)
    assert sorted(ref.keys()) == sorted(new.keys())
    # freeze rule
    student, teacher = Net([20, 20], 2, 1), Net([20], 1, 0)
    T.apply_step2_freeze(student, teacher, 1)
    assert [p.requires_grad for _, p in student.named_parameters()] == list(golden["requires_grad"])
    assert not any(p.requires_grad for p in teacher.parameters())
    w = T.class_weights("BDD")
    assert w[19] == 0 and w.numel() == 20


def test_event_file_writer_roundtrip(tmp_path):
    """mdil_ss_amd/scalar_log.py: TensorBoard event files without tensorboard (CRC-32C known answer,
    TFRecord framing and the Event / Summary encoding read back)."""
    import glob
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "_scalar_log", os.path.join(os.path.dirname(os.path.dirname(__file__)), "mdil_ss_amd", "scalar_log.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m._crc32c(b"123456789") == 0xE3069283                      # the CRC-32C check value
    w = m.EventFileWriter(str(tmp_path))
    for epoch in (1, 2, 300):
        w.add_scalar("val_acc_BDD", 0.25 * epoch, epoch)
        w.add_scalar("KLD_loss_train", -0.5 / epoch, epoch)
    w.close()
    got = m.read_scalars(glob.glob(str(tmp_path / "events.out.tfevents.*"))[0])
    assert [(s, t) for s, t, _ in got] == [(e, t) for e in (1, 2, 300) for t in ("val_acc_BDD", "KLD_loss_train")]
    assert abs(got[4][2] - 75.0) < 1e-6 and abs(got[1][2] + 0.5) < 1e-7
    head = open(w.path, "rb").read(64)
    assert b"brain.Event:2" in head
