"""Single-node launcher: one child process per GPU, rendezvous on 127.0.0.1.

Kept free of package-relative imports and of anything that loads the HIP library, so that a parent
process (bench.py --gpus N) can load this file by path and start its ranks BEFORE anything in it has
touched the GPU; the parent is never replaced (children are started with subprocess).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_commands(script: str, argv: Sequence[str], n_ranks: int, port: Optional[int] = None):
    """[(argv, env-additions)] for the n_ranks children of a single-node job (rendezvous on 127.0.0.1)."""
    port = port or free_port()
    cmds = []
    for r in range(n_ranks):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
        cmds.append(([sys.executable, script] + list(argv), env))
    return cmds


def launch_ranks(script: str, argv: Sequence[str], n_ranks: int, timeout: Optional[float] = None) -> int:
    """Starts n_ranks children (one per GPU) and waits for them.  Must be called from a process that has NOT
    touched the GPU (children are started with subprocess, the caller is never replaced).  Rank 0 inherits
    stdout, so its JSON line is the caller's; every rank inherits stderr.  Returns the worst exit code."""
    procs = []
    for cmd, env_add in rank_commands(script, argv, n_ranks):
        env = dict(os.environ)
        env.update(env_add)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if env_add["RANK"] == "0" else subprocess.DEVNULL
        procs.append(subprocess.Popen(cmd, env=env, stdout=out))
    rc, t0 = 0, time.monotonic()
    try:
        while any(p.poll() is None for p in procs):
            failed = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
            if failed and rc == 0:              # one rank down: the others would wait in a collective forever
                rc = failed[0] if failed[0] > 0 else 1
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
            if timeout is not None and time.monotonic() - t0 > timeout:
                rc = rc or 124
                break
            time.sleep(0.05)
        for p in procs:
            if p.poll() is not None and p.returncode != 0 and rc == 0:
                rc = p.returncode if p.returncode > 0 else 1
    finally:
        for q in procs:                         # only the exact children started above
            if q.poll() is None:
                q.kill()
                q.wait()
    return rc
