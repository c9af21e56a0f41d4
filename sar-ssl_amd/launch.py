"""One process per GPU, started by the entry point itself.

The reference's multi-GPU form is ONE command - ``python run_pretrain.py --gpu-id 0,1,2,3`` builds ``nn.DataParallel`` inside the
process (code/run_pretrain.py:204-205 -> code/learner.py:25-31).  Here multi-GPU is one process per GPU over RCCL, so the same
command line has to start those processes: when ``WORLD_SIZE`` is not in the environment and more than one GPU is asked for, the
entry point calls :func:`spawn_ranks` BEFORE anything touches the GPU.  The parent then only waits: it never initialises HIP (a
process that has must not be replaced or forked on this platform), the children are fresh interpreters with torchrun's environment
contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Rank 0 inherits the parent's stdout (so a benchmark's single
JSON line is the command's single line); the other ranks' stdout goes to stderr.  The exit code is the worst child's; when one rank
dies the others are terminated by PID (never by pattern).

Under ``torchrun`` (WORLD_SIZE already set) nothing here runs: the driver's own launch form keeps working unchanged.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def launched():
    """True inside a rank that a launcher (torchrun or spawn_ranks) started."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def parse_gpu_ids(spec):
    """'0,1,2,' -> ['0', '1', '2'] (the reference's --gpu-id syntax, trailing comma allowed: code/opt.py:16)."""
    return [g.strip() for g in str(spec).split(",") if g.strip() != ""]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_envs(nproc, gpu_ids=None, base=None, port=None):
    """The environment of each of `nproc` ranks.  `gpu_ids` (strings, one per rank; duplicates allowed - two ranks may share a GPU
    over gloo in functional tests) become HIP_VISIBLE_DEVICES = the distinct ids in order, LOCAL_RANK = the rank's index into it."""
    base = dict(os.environ if base is None else base)
    port = port or free_port()
    envs = []
    distinct = None
    if gpu_ids:
        assert len(gpu_ids) == nproc, "one GPU id per rank"
        distinct = list(dict.fromkeys(gpu_ids))
    for r in range(nproc):
        e = dict(base)
        e.update({"RANK": str(r), "WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                  "SARSSL_SELF_LAUNCHED": "1"})
        if distinct is not None:
            e["HIP_VISIBLE_DEVICES"] = ",".join(distinct)
            e["LOCAL_RANK"] = str(distinct.index(gpu_ids[r]))
        else:
            e["LOCAL_RANK"] = str(r)
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this platform (RCCL needs it across processes)
        envs.append(e)
    return envs


def spawn_ranks(script, argv, nproc, gpu_ids=None, poll_s=0.2, grace_s=10.0):
    """Start `nproc` fresh interpreters on `script argv` (one per rank) and wait for all of them.  Returns the exit code the caller
    should exit with: 0 when every rank returned 0, otherwise the first non-zero code seen (a signal -N is reported as 128 + N)."""
    assert nproc >= 1
    envs = rank_envs(nproc, gpu_ids)
    procs = []
    # a SIGTERM / SIGHUP to the parent (scheduler kill, timeout(1), closed terminal) must reach the ranks: they hold GPUs and may sit in a
    # collective for ever (advisor, round 5) - the handlers turn the signal into an exception the finally block below acts on
    class _Stop(Exception):
        pass

    def _on_signal(signum, frame):
        raise _Stop(signum)
    old_handlers = {}
    for sg in (signal.SIGTERM, getattr(signal, "SIGHUP", None)):
        if sg is not None:
            try:
                old_handlers[sg] = signal.signal(sg, _on_signal)
            except (ValueError, OSError):                        # (not the main thread: leave the default disposition)
                pass
    worst = 0
    try:
        for r, e in enumerate(envs):
            out = None if r == 0 else sys.stderr                 # rank 0 owns stdout
            procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=e, stdout=out, stdin=subprocess.DEVNULL))
        alive = set(range(nproc))
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0 and worst == 0:
                    worst = 128 - rc if rc < 0 else rc
                    print("launch: rank %d exited with code %d - stopping the other ranks" % (r, rc), file=sys.stderr, flush=True)
                    deadline = time.time() + grace_s
                    for o in sorted(alive):                      # a rank is gone: the others would wait in a collective for ever
                        procs[o].send_signal(signal.SIGTERM)
                    for o in sorted(alive):
                        try:
                            procs[o].wait(timeout=max(0.1, deadline - time.time()))
                        except subprocess.TimeoutExpired:
                            procs[o].kill()
                            procs[o].wait()
                    alive.clear()
                    break
            if alive:
                time.sleep(poll_s)
    except (KeyboardInterrupt, _Stop) as stop:
        sig = signal.SIGINT if isinstance(stop, KeyboardInterrupt) else signal.SIGTERM
        for p in procs:
            if p.poll() is None:
                p.send_signal(sig)
        worst = worst or (130 if sig == signal.SIGINT else 143)
    finally:
        # whatever ends the parent - normal return, a signal, an exception - no rank survives it
        deadline = time.time() + grace_s
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        for sg, h in old_handlers.items():
            try:
                signal.signal(sg, h)
            except (ValueError, OSError):
                pass
    return worst
