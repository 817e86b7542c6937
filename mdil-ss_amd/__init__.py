"""mdil-ss_amd: MI355X-native ERFNet + parallel-residual-adapter (RAP) step-2 training path.

Host side mirrors the reference's Python surface (``models.erfnet_RA_parallel.Net``,
``train_new_task_step2`` entry points, ``iouEval``); all tensor math runs in hand-written HIP
kernels for gfx950 behind the C ABI of ``include/mdil_hip.h`` (``libmdil_hip.so``).
Import as ``mdil_ss_amd`` (alias module at the repo root; the directory name carries a hyphen).
"""
__version__ = "0.1.0"
