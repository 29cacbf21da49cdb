"""
One node, one process per GPU, with automatic re-queue of what a failed attempt left undone.

    python -m setk_amd.launch --nproc 8 [--retries 2] scripts/sptk/apply_adaptive_beamformer.py ARGS...

The reference parallelises this path with `split_scp.pl` + `run.pl JOB=1:nj`
(scripts/run_adapt_beamformer.sh:69-92): a shard whose process dies is re-run by hand.  Here the
ranks are started with `torch.distributed.run` (rendezvous on 127.0.0.1, a free port); when the
attempt ends with a non-zero status -- a rank raised, was killed, lost its GPU: torchrun tears
the other ranks down with it -- the job is started again with `--skip-existing true --requeue
true`: finished wave files are complete by construction (written to `{key}.wav.part`, renamed
when closed), so the new attempt finds what is missing and deals ONLY that over all ranks.
The command lines that take these two flags: apply_adaptive_beamformer.py.
"""
import argparse
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def build_command(nproc, script_and_args, attempt, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), *extra, *script_and_args]
    if attempt > 0:
        cmd += ["--skip-existing", "true", "--requeue", "true"]
    return cmd


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--nproc", type=int, required=True, help="ranks = GPUs of this node")
    ap.add_argument("--retries", type=int, default=2,
                    help="re-launches after a failed attempt (each does only what is missing)")
    ap.add_argument("command", nargs=argparse.REMAINDER, help="script and its arguments")
    a = ap.parse_args(argv)
    cmd = [c for c in a.command if c != "--"] if a.command[:1] == ["--"] else a.command
    if not cmd:
        ap.error("no command given")
    rc = 1
    for attempt in range(a.retries + 1):
        env = dict(os.environ, SETK_LAUNCH_ATTEMPT=str(attempt))
        full = build_command(a.nproc, cmd, attempt)
        print(f"[setk_amd.launch] attempt {attempt + 1} of {a.retries + 1}: {' '.join(full)}",
              file=sys.stderr, flush=True)
        rc = subprocess.call(full, env=env)
        if rc == 0:
            return 0
        print(f"[setk_amd.launch] attempt {attempt + 1} ended with status {rc}"
              + ("; re-queueing what is missing" if attempt < a.retries else "; giving up"),
              file=sys.stderr, flush=True)
    return rc


if __name__ == "__main__":
    sys.exit(main())
