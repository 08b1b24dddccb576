"""Self-launch of a one-node data-parallel job: one fresh child process per GPU, each a rank of an RCCL job -- what
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N` does, without the launcher.  Used by `train.py -gpu=0,1,...`
(the reference parses device lists, params/params_util.py:20-55, and trains on the first one) and by `bench.py --gpus N` when no
launcher set WORLD_SIZE.  Nothing here touches the GPU: the parent never initialises HIP, the children do."""
import os
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environments(devices, narrow_visible=True):
    """One environment per rank.  The listed indices select among the devices an outer HIP_VISIBLE_DEVICES already exposes
    (narrow_visible=False: single-device test runs that put every rank on device 0 leave the visibility alone)."""
    port = free_port()
    extra = {}
    if narrow_visible:
        outer = [v for v in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if v != ""]
        if outer:
            if max(devices) >= len(outer):
                raise SystemExit(f"devices {devices} do not fit HIP_VISIBLE_DEVICES={','.join(outer)}")
            visible = [outer[d] for d in devices]
        else:
            visible = [str(d) for d in devices]
        extra["HIP_VISIBLE_DEVICES"] = ",".join(visible)
    return [dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(len(devices)), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), **extra) for rank in range(len(devices))]


def spawn_ranks(script, argv, devices, narrow_visible=True):
    """Runs `script argv` once per device as rank 0 ... len(devices) - 1 and returns the job's exit code: 0 only if every rank
    exited 0.  When a rank dies the others are terminated (they would otherwise sit in their next collective until the
    process-group timeout) and the dead rank's code is returned."""
    procs = [subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env)
             for env in rank_environments(devices, narrow_visible)]
    code = 0
    while any(p.poll() is None for p in procs):
        failed = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if failed:
            code = failed[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=30)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    if code == 0:
        code = max((p.returncode for p in procs), key=abs)
    return code
