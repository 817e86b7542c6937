"""One run of the mIoU protocol (tests/miou_protocol.py) on the HIP path, printing the final mIoU
of both heads.  The kernel variant is chosen by the environment (MDIL_NO_WCONV, MDIL_NO_WGRADW,
MDIL_NO_WGRAD2, MDIL_NO_BNFUSE, MDIL_NO_SCONV, MDIL_NO_C16CONV ...): every variant sums in another
order, i.e. is an independent sample of the run-to-run noise of the protocol."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_miou_parity import _run_protocol  # noqa: E402

if __name__ == "__main__":
    tag = " ".join(k for k in sorted(os.environ) if k.startswith("MDIL_NO_")) or "shipped build"
    r = _run_protocol(torch.device("cuda:0"), tag)
    print(f"SAMPLE [{tag}] new {r['miou_new'] * 100:.3f} old {r['miou_old'] * 100:.3f}")
