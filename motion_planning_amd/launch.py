"""One process per GPU from a plain command line.

``python bench.py --gpus 8`` (no torchrun) must still run eight ranks: `self_launch` starts
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <script> <args>``
as a CHILD process, relays its output and returns its exit code.  It must be called before anything in the parent
touches the GPU (a process that initialised HIP never replaces itself with another program, and the parent of the ranks
has no business holding a GPU context): the only torch call here is `torch.cuda.device_count()`, which does not
initialise the device on this image.

The reference is a single process; this is plumbing of the build's own multi-GPU form (§8e).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence


def under_launcher() -> bool:
    """True inside a rank started by torch.distributed.run (or any launcher that exports the rendezvous variables)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def visible_gpus() -> int:
    import torch

    return int(torch.cuda.device_count())


def launcher_command(script: str, argv: Sequence[str], n: int, port: Optional[int] = None) -> List[str]:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def self_launch(script: str, argv: Sequence[str], n: int, one_device: bool = False) -> int:
    """Run `script argv` as `n` ranks under torch.distributed.run and return the launcher's exit code.  Fewer than `n`
    visible GPUs is an error (exit code 2, message on stderr) unless `one_device` (rehearsal: every rank on cuda:0)."""
    if n < 2:
        raise ValueError("self_launch is for n >= 2 ranks")
    if not one_device:
        have = visible_gpus()
        if have < n:
            print(f"{os.path.basename(script)}: --gpus {n} asked for, {have} HIP device(s) visible: not starting a run that "
                  f"would report fewer GPUs than requested", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    # a one-GPU job's CPU share is 16 threads; N ranks must not each start 16 OpenMP threads
    env.setdefault("OMP_NUM_THREADS", "2")
    proc = subprocess.run(launcher_command(script, argv, n), env=env)
    return int(proc.returncode)
