"""Library-GEMM solution selection for MI355X (PyTorch TunableOp over rocBLAS / hipBLASLt), recorded offline.

The Linear and attention GEMMs of the models are plain library GEMMs (SURVEY section 8 keeps them on rocBLAS /
hipBLASLt through torch).  The library's default heuristic is poor for the shapes a T=5 clip produces -- e.g. the weight
gradients of the encoder's FFN, [1 024 x 51 000] x [51 000 x 256], run as 32 output tiles on a 256-CU part: 573 us where
the tuned solution takes 197 -- and the training step of SeqFormer-R50 (two clips per GPU, fp32) goes from 80.5 to
69.3 ms on MI355X with the recorded choices (tools/tune_gemms.py, round 4).  `tunableop_mi355x.csv` holds them;
`enable()` loads the file with TUNING OFF: a GEMM with an entry takes the recorded solution, every other shape takes the
library default, nothing is ever timed at run time.  TunableOp itself refuses the file when the PyTorch / ROCm / hipBLASLt
/ rocBLAS versions or the GPU architecture differ from the ones it was recorded with (the `Validator` rows).

Only fp32 entries are recorded: the bf16 tuning pass took a GPU memory-access fault inside a library candidate on this
stack, with and without the hipBLASLt candidates (two attempts, round 4).
"""
from __future__ import annotations

import os
import tempfile

TUNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_mi355x.csv")
_state = {"enabled": False, "entries": 0, "why": "not requested"}
_prev = {}        # TunableOp's results file name before enable() moved it (restored by disable())


def enable(path: str | None = None) -> dict:
    """Load the recorded GEMM solutions (idempotent).  -> {"enabled", "entries", "why"} for the bench line."""
    import torch
    if _state["enabled"]:
        return dict(_state)
    path = path or os.environ.get("VNX_TUNED_GEMMS") or TUNED_FILE
    if os.environ.get("VNX_TUNED_GEMMS", "") == "0":
        _state.update(why="disabled by VNX_TUNED_GEMMS=0")
        return dict(_state)
    if not torch.cuda.is_available() or getattr(torch.version, "hip", None) is None:
        _state.update(why="no ROCm device")
        return dict(_state)
    if not os.path.exists(path):
        _state.update(why=f"{path} is missing")
        return dict(_state)
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(False)          # never time anything at run time
        # PyTorch builds that write TunableOp's results back at exit have `write_file_on_exit`: switch it off -- tuning is off,
        # there is nothing new to persist, and every process that enables the recorded GEMMs (each DDP rank, each bench leg)
        # would leave a CSV behind.  PyTorch 2.10 has no such switch (it writes a result when tuning produces
        # one, i.e. never here); the file name is moved away from the working directory and the repo either way, and
        # disable() puts it back.
        _prev["filename"] = tunable.get_filename()
        tunable.set_filename(os.path.join(tempfile.gettempdir(), "vnx_tunableop_%d.csv" % os.getpid()))
        if hasattr(tunable, "write_file_on_exit"):
            tunable.write_file_on_exit(False)
        ok = bool(tunable.read_file(path))
        n = len(tunable.get_results())
        if not ok or n == 0:
            tunable.enable(False)
            _state.update(why="TunableOp rejected the file (validators: PyTorch / ROCm / hipBLASLt / rocBLAS version or "
                              "GPU architecture differ from the recording)")
            return dict(_state)
        _state.update(enabled=True, entries=n, why=os.path.relpath(path, os.path.dirname(os.path.dirname(TUNED_FILE))))
    except Exception as e:       # a TunableOp problem must never cost the run
        _state.update(why=f"TunableOp unavailable: {type(e).__name__}: {e}")
    return dict(_state)


def disable() -> None:
    import torch
    if _state["enabled"]:
        import torch.cuda.tunable as tunable
        tunable.enable(False)
        if _prev.get("filename"):
            tunable.set_filename(_prev.pop("filename"))
        _state.update(enabled=False, entries=0, why="disabled")


def status() -> dict:
    return dict(_state)
