"""Library-GEMM solution selection for MI355X (PyTorch TunableOp over rocBLAS / hipBLASLt), recorded offline.

The Linear and attention GEMMs of the models are plain library GEMMs (SURVEY section 8 keeps them on rocBLAS /
hipBLASLt through torch).  The library's default heuristic is poor for the shapes a T=5 clip produces -- e.g. the weight
gradients of the encoder's FFN, [1 024 x 51 000] x [51 000 x 256], run as 32 output tiles on a 256-CU part: 573 us where
the tuned solution takes 197 -- and the training step of SeqFormer-R50 (two clips per GPU, fp32) goes from 80.5 to
69.3 ms on MI355X with the recorded choices (tools/tune_gemms.py, round 4).  `tunableop_mi355x.csv` holds them;
`enable()` loads the file with TUNING OFF: a GEMM with an entry takes the recorded solution, every other shape takes the
library default, nothing is ever timed at run time.  TunableOp itself refuses the file when the PyTorch / ROCm / hipBLASLt
/ rocBLAS versions or the GPU architecture differ from the ones it was recorded with (the `Validator` rows).

fp32 entries since round 4; bf16 entries (the autocast legs: BASELINE config 3) since round 6.  Tuned ONLINE, inside the autocast
step, the bf16 pass took a GPU memory-access fault inside a library candidate on this stack (two attempts, round 4);
tools/tune_gemms_bf16.py records the step's GEMM signatures with tuning off and tunes each offline, on random operands, in a worker
process -- a fault would cost one shape; there was none (155 shapes, 73 s).  `bf16_step_shapes.csv` is that recorded list.

`enable_conv_search()` (below) does the same for MIOpen's convolution algorithms: a find-db recorded offline, no search at run time.
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


# ---- MIOpen: convolution algorithms found by measurement, recorded offline (round 6) -------------------------------------------
CONV_DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_userdb")
_conv_state = {"enabled": False, "why": "not requested"}


def enable_conv_search(db_dir: str | None = None) -> dict:
    """MIOpen picks the trunk's convolution algorithms by MEASUREMENT instead of by its heuristic
    (`torch.backends.cudnn.benchmark = True`), answering from the find-db recorded on MI355X for the models' problems
    (tools/record_miopen_db.py -> tuning/miopen_userdb/): SeqFormer-R50 training step 59.0 -> 55.4 ms fp32, 54.0 -> 46.0 ms under
    bf16 autocast (two T = 5 360p clips, one GPU, round 6).  A recorded problem costs nothing at run time; a problem that is
    not in the file (another frame size or batch) is searched once per process -- tens of seconds for a new resolution, which
    is why this is an explicit call and not an import side effect.

    MIOpen reads MIOPEN_USER_DB_PATH when the process runs its first convolution: call this BEFORE it.  The library also
    WRITES there, so it is handed a private copy of the recorded directory (per process: DDP ranks do not share a file).
    An explicit MIOPEN_USER_DB_PATH in the environment is respected (only the search is switched on); VNX_CONV_SEARCH=0
    opts out.  -> {"enabled", "why", "db"} for the bench line."""
    import shutil
    import torch
    if _conv_state["enabled"]:
        return dict(_conv_state)
    if os.environ.get("VNX_CONV_SEARCH", "1") == "0":
        _conv_state.update(why="VNX_CONV_SEARCH=0")
        return dict(_conv_state)
    if not torch.cuda.is_available() or getattr(torch.version, "hip", None) is None:
        _conv_state.update(why="no ROCm device")
        return dict(_conv_state)
    db = "MIOPEN_USER_DB_PATH from the environment"
    if "MIOPEN_USER_DB_PATH" not in os.environ:
        src = db_dir or CONV_DB_DIR
        dst = os.path.join(tempfile.gettempdir(), "vnx_miopen_userdb_%d" % os.getpid())
        try:
            if os.path.isdir(dst):
                shutil.rmtree(dst)
            if os.path.isdir(src) and any(f.endswith(".txt") for f in os.listdir(src)):
                shutil.copytree(src, dst)
                db = "a private copy of " + os.path.relpath(src, os.path.dirname(os.path.dirname(CONV_DB_DIR)))
            else:
                os.makedirs(dst, exist_ok=True)
                db = "empty (no recorded find-db at %s): every problem is searched once" % src
            os.environ["MIOPEN_USER_DB_PATH"] = dst
            import atexit
            atexit.register(shutil.rmtree, dst, True)      # (the private copy goes with the process)
        except OSError as e:       # never cost the run: MIOpen then keeps its own default location
            db = "MIOpen's default location (%s: %s)" % (type(e).__name__, e)
    torch.backends.cudnn.benchmark = True
    _conv_state.update(enabled=True, why="torch.backends.cudnn.benchmark = True: MIOpen answers from its find-db, or times its solvers "
                                         "the first time it sees a problem", db=db)
    return dict(_conv_state)
