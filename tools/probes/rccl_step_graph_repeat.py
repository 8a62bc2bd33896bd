#!/usr/bin/env python3
"""Probe (not product code): tests/test_ddp_gpu.py::_run_graph(graph=True) in a child process, N times, with time-stamped markers on
stderr, to see WHERE a ProcessGroupNCCL watchdog abort (hipErrorCapturedEvent) falls relative to the capture.
usage (GPU box): python tools/probes/rccl_step_graph_repeat.py [N]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
t0 = time.time()
def mark(s): print(f"[{time.time() - t0:7.3f}] {s}", file=sys.stderr, flush=True)
import test_ddp_gpu as T
from ssecg import graph as G
orig_capture = G.StepGraph._capture
def cap(self, inputs):
    if os.environ.get("PROBE_DRAIN"):   # let the watchdog drop the eager steps' completed works before RCCL's stream is captured
        import torch
        torch.cuda.synchronize()
        time.sleep(float(os.environ["PROBE_DRAIN"]))
    mark("capture begins")
    r = orig_capture(self, inputs)
    mark("capture ended: " + str(r))
    return r
G.StepGraph._capture = cap
orig_call = G.StepGraph.__call__
def call(self, *a):
    mark(f"step {self.calls + 1} begins")
    r = orig_call(self, *a)
    mark(f"step {self.calls} issued")
    return r
G.StepGraph.__call__ = call
import torch.distributed as dist
orig_barrier = dist.barrier
def bar(*a, **k):
    mark("barrier")
    r = orig_barrier(*a, **k)
    mark("barrier returned")
    return r
dist.barrier = bar
if os.environ.get("PROBE_WIDEN"):   # hold the capture open for more than one watchdog period (100 ms), after the forward's collectives
    import torch
    import algorithms.fixmatch as AF
    orig_step = AF.fixmatch_step
    def slow_step(*a, **k):
        r = orig_step(*a, **k)
        if torch.cuda.is_current_stream_capturing():
            mark("inside the capture: sleeping")
            time.sleep(float(os.environ["PROBE_WIDEN"]))
        return r
    AF.fixmatch_step = slow_step
out = {}
T._run_graph(0, 1, PORT, out, True)
mark("done, replays " + str(out.get("replays")))
time.sleep(0.3)
mark("exit")
'''


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    bad = 0
    for i in range(n):
        code = f"ROOT = {ROOT!r}\nPORT = {29800 + i}\n" + CHILD
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280)
        marks = [ln for ln in p.stderr.splitlines() if ln.startswith("[") and "]" in ln[:12] and "rank" not in ln[:8]]
        err = [ln for ln in p.stderr.splitlines() if "HIP error" in ln][:1]
        print(f"run {i}: exit {p.returncode}", "|", " ; ".join(marks), "|", *err, flush=True)
        bad += p.returncode != 0
    print(f"{bad} of {n} runs aborted")


if __name__ == "__main__":
    main()
