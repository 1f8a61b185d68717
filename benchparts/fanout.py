"""`python bench.py --gpus N` as a plain command: the parent that starts the N ranks.

Part of bench.py (the repo-root benchmark driver), split out in round 6: bench.py keeps the command line, the timed regions
of the headline metric and the assembly of the ONE JSON line; this module holds `fan_out`."""
from __future__ import annotations

import os
import sys
import time

BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def fan_out(n: int) -> int:
    """`python bench.py --gpus N` started as a PLAIN process (no torchrun: WORLD_SIZE unset): this parent starts the N
    ranks itself -- one child per GPU running this same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    exactly what `python -m torch.distributed.run --nproc-per-node N` would give them -- touches no GPU, lets rank 0's
    JSON line through on the inherited stdout, and returns non-zero if any child fails (the others are then stopped by
    their exact PIDs: SIGTERM, then SIGKILL after a bounded grace period).  Under torchrun this function is never reached."""
    import signal
    import subprocess
    # (the parent asks the runtime NOTHING about devices -- on ROCm builds without amdsmi even device_count() initialises
    # HIP/HSA in this process before it forks; each rank checks its own LOCAL_RANK against the device count in main())
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + sys.argv[1:], env=env))
    rc = 0
    pending = set(range(n))
    kill_at = None                                         # after a failure: when the survivors' grace period ends
    grace = float(os.environ.get("BENCH_FANOUT_GRACE_S", "20"))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].send_signal(signal.SIGTERM)
                kill_at = time.monotonic() + grace
        if kill_at is not None and pending and time.monotonic() > kill_at:
            # a rank stuck in a collective does not act on SIGTERM: end it by its exact PID
            for q in sorted(pending):
                print(f"[bench] rank {q} (pid {procs[q].pid}) still running {grace:.0f} s after SIGTERM: SIGKILL", file=sys.stderr)
                procs[q].kill()
            for q in sorted(pending):
                procs[q].wait()
            pending.clear()
        time.sleep(0.05)
    return rc
