"""One process per GPU: the parent side of `bench.py --gpus N` / `python -m amuse_amd.train_gesture --gpus N` when nothing else
launched the ranks.  The parent never touches the GPU (no torch.cuda call, no HIP library): it starts the ranks as CHILD
processes through torch.distributed.run and hands back their exit code - never os.exec* (a process image replaced after the
GPU runtime was initialised takes the node down on this pool, and a child is the portable form anyway)."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Optional, Sequence


def launched_by_torchrun() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def torchrun_command(script: str, argv: Sequence[str], n: int, port: Optional[int] = None, module: bool = False) -> list:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), *(["-m"] if module else []), script, *argv]


def run_ranks(script: str, argv: Sequence[str], n: int, timeout: Optional[float] = None, module: bool = False) -> int:
    """Start `n` ranks of `script argv...` (module = True: `-m script`) on this node and wait for them; their stdout / stderr
    pass straight through (rank 0 prints the JSON line).  Returns the launcher's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = repo + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    return subprocess.run(torchrun_command(script, argv, n, module=module), env=env, timeout=timeout).returncode
