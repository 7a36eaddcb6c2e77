"""One process per GPU, started by the command itself (test/stark_aggregation.sh:70-73 starts its per-task provers the same
way: a loop of child processes on one host).

`spawn_ranks(argv, n)` is called by a command that was asked for n > 1 GPUs and finds no rank environment (no torchrun in front
of it): it starts n FRESH child processes of the same command with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
set, relays rank 0's standard output, and returns the first non-zero exit status (0 when every rank succeeded).

The parent must not have touched a GPU: no process that has initialised HIP is ever re-exec'ed or forked here -- the children
are new interpreters (`subprocess.Popen`), and this module imports nothing beyond the standard library, so a caller can run it
before `torch` or `libzkgpu.so` are loaded.
"""
import os
import socket
import subprocess
import sys
import threading
import time

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE")


def under_launcher(environ=None):
    """True when a launcher (torchrun, or spawn_ranks itself) has already given this process a rank"""
    e = os.environ if environ is None else environ
    return "RANK" in e and "WORLD_SIZE" in e


def free_port(addr="127.0.0.1"):
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind((addr, 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _relay(pipe, sink, prefix, json_only_to=None):
    """copy a child's pipe line by line; with `json_only_to`, lines that look like a JSON object go there and the rest (library
    chatter on stdout, e.g. gloo's connection notes) to `sink`"""
    for line in iter(pipe.readline, b""):
        text = line.decode("utf-8", "replace")
        if json_only_to is not None and text.lstrip().startswith("{"):
            json_only_to.write(text); json_only_to.flush()
            continue
        sink.write(prefix + text)
        sink.flush()
    pipe.close()


def spawn_ranks(argv, n, extra_env=None, timeout=None, python=None, json_stdout=True):
    """Start `python argv...` n times, rank r with RANK = LOCAL_RANK = r; -> exit status (0 = every rank exited 0).
    Rank 0's JSON lines go to this process's stdout (the ONE line of bench.py; json_stdout=False: all of rank 0's stdout, unchanged);
    everything else the ranks print goes to stderr with a `[rank r]` prefix.  If a rank fails the others are terminated (they would wait
    in a collective for ever)."""
    assert n >= 1
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", 0)) or free_port(addr)
    procs, threads = [], []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": addr, "MASTER_PORT": str(port), "ZK_SPAWNED_BY": str(os.getpid())})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                  # dmabuf IPC only on this driver (RCCL across processes)
        if extra_env:
            env.update(extra_env)
        p = subprocess.Popen([python or sys.executable] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        procs.append(p)
        threads.append(threading.Thread(target=_relay, args=(p.stdout, sys.stderr, "[rank %d] " % r, sys.stdout if r == 0 and json_stdout else None), daemon=True)
                       if json_stdout or r else threading.Thread(target=_relay, args=(p.stdout, sys.stdout, ""), daemon=True))
        threads.append(threading.Thread(target=_relay, args=(p.stderr, sys.stderr, "[rank %d] " % r if n > 1 else ""), daemon=True))
    for t in threads:
        t.start()
    status, deadline = 0, None if timeout is None else time.time() + timeout
    live = set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 128 - rc                        # killed by signal s: 128 + s
                print("launcher: rank %d exited with status %d; stopping the other ranks" % (r, rc), file=sys.stderr, flush=True)
                for q in live:
                    procs[q].terminate()
        if deadline is not None and time.time() > deadline and live:
            print("launcher: timeout after %.0f s; stopping ranks %s" % (timeout, sorted(live)), file=sys.stderr, flush=True)
            for q in live:
                procs[q].terminate()
            status, deadline = status or 124, None
        if live:
            time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    for t in threads:
        t.join(timeout=5)
    return status
