"""mdil_ss_amd: MI355X-native ERFNet + parallel-residual-adapter (RAP) step-2 training path.

Host side mirrors the reference's Python surface (``models.erfnet_RA_parallel.Net``,
``train_new_task_step2`` entry points, ``iouEval``); all tensor math runs in hand-written HIP
kernels for gfx950 behind the C ABI of ``include/mdil_hip.h`` (``libmdil_hip.so``).
The package directory is ``mdil_ss_amd`` (underscore: importable as it is; the project name
``mdil-ss_amd`` of the task statement is not a valid Python identifier).
"""
__version__ = "0.1.0"

import os as _os

# Kernel arguments in device memory: ~1,750 launches per step sit on dependent chains, and the
# launch-to-start latency is 3 % of the step (171.5 vs 177 img/s with the variable forced to 0).
# Recent PyTorch-ROCm builds already default to it; make it explicit (must precede HIP init).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# ROCm multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default).  The step
# uses the null stream + three engine streams; a fifth one (the communication stream of a multi-rank run, a
# library's internal stream) would share a queue with one of them and run BEHIND it: measured 22.6 -> 25.3 ms
# per step when six other streams had been used first, 22.6 ms again with 8 queues (profiles/r04_experiments.txt
# #17).  Must precede HIP initialisation; a value the caller set is left alone.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
