"""GPU: fine-tuning / feature-extraction baselines (models/erfnet.py, erfnet_ftp1.py,
erfnet_ftp2.py; main_ftp1_enc_newbn.py, main_FT2_flexible_new.py) against the golden generated
from the reference, and the two trainer mirrors chained FT1 -> FT2."""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import rap_oracle as O
from tests import helpers as Hh
from tests.test_hip_parity import close

pytestmark = pytest.mark.gpu


def _model(dev):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd.models.erfnet import NetFT2 as Net
    ops.invalidate_packs()
    m = Net(20, 20, 27)
    m.load_state_dict(Hh.ft_scenario())
    return m.to(dev)


def test_ft_model_eval_heads(golden_ft):
    gf, dev = golden_ft, torch.device("cuda:0")
    m = _model(dev).eval()
    x = torch.from_numpy(gf["images"]).to(dev)
    with torch.no_grad():
        close(m(x, True, False, False), torch.from_numpy(gf["eval_old1"]), rtol=5e-4, atol=1e-4, what="old1")
        close(m(x, False, True, False), torch.from_numpy(gf["eval_old2"]), rtol=5e-4, atol=1e-4, what="old2")
        close(m(x, False, False, True), torch.from_numpy(gf["eval_new"]), rtol=5e-4, atol=1e-4, what="new")


@pytest.mark.parametrize("finetune", [True, False])
def test_finetune_engine_iteration(golden_ft, finetune):
    gf, dev = golden_ft, torch.device("cuda:0")
    from mdil_ss_amd.engine import FineTuneEngine
    m = _model(dev)
    names = [n for n, _ in m.named_parameters()]
    assert names == list(gf["param_names"])
    m.mask_provider = lambda n: Hh.ft_masks(gf)
    eng = FineTuneEngine(m, torch.tensor(Hh.WEIGHT_IDD, device=dev), finetune,
                         lambda x: m(x, decoder_old1=False, decoder_old2=False, decoder_new=True))
    params = dict(m.named_parameters())
    before = [params[n].detach().cpu().clone() for n in names]
    ce = eng.iteration(torch.from_numpy(gf["images"]).to(dev), torch.from_numpy(gf["labels"]).to(dev))
    assert float(ce) == pytest.approx(float(gf["loss"]), rel=2e-5)
    close(eng.last_outputs, torch.from_numpy(gf["train_logits"]), rtol=5e-4, atol=2e-4, what="train logits")
    delta = np.stack([fx.tensor_digest(params[n].detach().cpu() - b)[:3].numpy() for n, b in zip(names, before)])
    ref = gf["delta"]
    dec_new = np.array([n.startswith("decoder_new") for n in names])
    enc = np.array([n.startswith("encoder") for n in names])
    assert np.all(delta[~(dec_new | enc)] == 0), "old decoders must not move"
    moved = dec_new | (enc if finetune else np.zeros_like(enc))
    if not finetune:
        assert np.all(delta[enc] == 0), "feature extraction: the encoder is not stepped"
    rel = np.abs(delta[moved, 1] - ref[moved, 1]) / (ref[moved, 1] + 1e-12)
    assert np.median(rel) < 3e-2 and rel.max() < 0.15, (np.median(rel), rel.max())
    for k, v in m.state_dict().items():
        if O.is_buffer(k):        # encoder + decoder_new statistics move in both modes
            close(v.float(), torch.from_numpy(gf["buf_" + k]).float(), rtol=1e-3, atol=3e-4, what=k)


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


def test_ft_trainers_chain(tmp_path, monkeypatch):
    """single-task ERFNet checkpoint -> main_ftp1_enc_newbn (--finetune) -> main_FT2_flexible_new (FE)."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import main_FT2_flexible_new as F2
    from mdil_ss_amd import main_ftp1_enc_newbn as F1
    from mdil_ss_amd import ops
    from mdil_ss_amd.models.erfnet import Net as ERFNet
    ops.invalidate_packs()
    work = tmp_path / "run"
    work.mkdir()
    monkeypatch.chdir(work)
    torch.manual_seed(2)
    base = ERFNet(20)
    ck0 = tmp_path / "erfnet_cs.pth.tar"
    torch.save({"state_dict": {"module." + k: v for k, v in base.state_dict().items()}}, ck0)
    common = ["--batch-size", "2", "--height", "32", "--width", "64", "--synthetic", "8",
              "--num-workers", "0", "--steps-loss", "0", "--num-epochs", "1"]
    F1.main(F1.build_parser().parse_args(["--savedir", "ft1", "--state", str(ck0), "--finetune",
                                         "--dataset-old", "cityscapes", "--dataset-new", "BDD",
                                         "--model-name-suffix", "FT-CStoBDD"] + common))
    ck1 = tmp_path / "save" / "ft1" / "checkpoint_erfnet_ftp1_1_2_FT-CStoBDD.pth.tar"
    assert ck1.exists()
    sd1 = torch.load(ck1, map_location="cpu", weights_only=False)["state_dict"]
    assert torch.equal(sd1["module.decoder_old.output_conv.weight"], base.state_dict()["decoder.output_conv.weight"])
    assert not torch.equal(sd1["module.encoder.layers.1.conv3x1_1.weight"],
                           base.state_dict()["encoder.layers.1.conv3x1_1.weight"])
    log = (tmp_path / "save" / "ft1" / "automated_log.txt").read_text().splitlines()
    assert len(log) == 2 and len(log[1].split("\t\t")) == 8
    # TensorBoard scalars of the fine-tuning baselines (main_ftp1_enc_newbn.py:109-111,327-332)
    sc = _scalars(work, "Finetuning_Baselines/runs_erfnet_ftp1_1_2FT-CStoBDD")
    assert sorted(sc) == sorted(["train_loss", "val_loss_BDD", "val_accuracy_BDD", "val_loss_cityscapes",
                                 "val_accuracy_cityscapes"]), sc
    ops.invalidate_packs()
    F2.main(F2.build_parser().parse_args(["--savedir", "ft2", "--state", str(ck1), "--dataset-new", "IDD",
                                         "--datasets", "cityscapes", "BDD", "IDD", "--num-classes", "20",
                                         "20", "27", "--model-name-suffix", "FE-CSBDDtoIDD"] + common))
    ck2 = tmp_path / "save" / "ft2" / "checkpoint_erfnet_ftp2_1_2_FE-CSBDDtoIDD.pth.tar"
    sd2 = torch.load(ck2, map_location="cpu", weights_only=False)["state_dict"]
    assert torch.equal(sd2["module.decoder_old1.output_conv.weight"], sd1["module.decoder_old.output_conv.weight"])
    assert torch.equal(sd2["module.decoder_old2.output_conv.weight"], sd1["module.decoder_new.output_conv.weight"])
    assert torch.equal(sd2["module.encoder.layers.1.conv3x1_1.weight"], sd1["module.encoder.layers.1.conv3x1_1.weight"])
    assert tuple(sd2["module.decoder_new.output_conv.weight"].shape) == (16, 27, 2, 2)
    sc = _scalars(work, "Finetuning_Baselines/runs_erfnet_ftp2_1_2FE-CSBDDtoIDD")          # main_FT2_flexible_new.py:108-110,313-322
    assert sorted(sc) == sorted(f"val_{k}_{d}" for k in ("acc", "loss") for d in ("cityscapes", "BDD", "IDD")), sc
