"""`--gpus N` of bench.py / bench_corpus.py / bench_fit.py: start N ranks.

The process the user (or the driver) starts with `python bench.py --gpus N` has
made no GPU call when it gets here.  When N > 1 and it is not already a rank
(no WORLD_SIZE in the environment, i.e. nobody started it under
torch.distributed.run), it becomes the LAUNCHER: N fresh child processes of
the same command line, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE /
LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, rank 0's standard output
forwarded (the JSON line), the others' kept out of it.  It waits for all of
them and exits with the first non-zero code; when a rank dies the others are
ended (by the exact PIDs started here) instead of waiting in a collective
forever.  Never an exec, never a retry, and the launcher itself never imports
torch or the HIP library.

The loop this shards is the reference's per-file loop,
/root/reference/kwiiyatta/convert_voice.py:17-32: ranks take
pairs / utterances round-robin, no collective on the data path.
"""
import os
import socket
import subprocess
import sys
import time

GRACE_S = 20.0        # after a rank has failed: how long its peers get before SIGTERM


def is_rank():
    """True in a process that somebody already started as a rank (torch.distributed.run or this launcher)."""
    return 'WORLD_SIZE' in os.environ and 'RANK' in os.environ


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    """The environment of rank `rank` of `world` on this node (what torch.distributed.run would set)."""
    env = dict(os.environ if base is None else base)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_WORLD_SIZE': str(world),
                'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'KWY_LAUNCHED_BY': 'bench_launch',
                'HSA_ENABLE_IPC_MODE_LEGACY': env.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')})
    return env


def launch_ranks(world, argv=None, python=None, grace_s=GRACE_S, poll_s=0.05):
    """Start `world` ranks of `argv` (default: this command line), forward rank 0's stdout, return the exit code:
    0 when every rank returned 0, otherwise the first non-zero code seen (a rank killed by signal n gives 128 + n)."""
    argv = list(sys.argv if argv is None else argv)
    python = python or sys.executable
    port = int(os.environ.get('MASTER_PORT', 0)) or free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([python] + argv, env=rank_env(r, world, port),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    code, failed_at = 0, None
    live = set(range(world))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and code == 0:
                code = 128 - rc if rc < 0 else rc
                failed_at = time.monotonic()
                sys.stderr.write(f'bench_launch: rank {r} of {world} exited with {rc}; ending the other ranks\n')
        if failed_at is not None and live and time.monotonic() - failed_at > grace_s:
            for r in sorted(live):
                procs[r].terminate()
            deadline = time.monotonic() + 10.0
            for r in sorted(live):
                try:
                    procs[r].wait(max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        if live:
            time.sleep(poll_s)
    return code


def maybe_launch(gpus):
    """Call right after the arguments are parsed.  Returns None in a rank (or when gpus == 1); in the launcher it
    does not return: the process exits with the ranks' code."""
    if gpus <= 1 or is_rank():
        if is_rank() and int(os.environ['WORLD_SIZE']) != gpus and int(os.environ.get('RANK', 0)) == 0:
            sys.stderr.write(f'bench: --gpus {gpus} but WORLD_SIZE={os.environ["WORLD_SIZE"]}: the environment decides\n')
        return None
    sys.stdout.flush()
    sys.exit(launch_ranks(gpus))
