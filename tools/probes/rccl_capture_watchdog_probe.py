#!/usr/bin/env python3
"""Hardware / runtime probe (not product code): does ProcessGroupNCCL's watchdog survive a HIP-graph capture that pulls RCCL's stream in
while works of EAGER collectives are still on its list?

Background (round 6, second session): tests/test_ddp_gpu.py::test_rccl_single_rank_step_graph_is_bit_identical aborted once in three
full-suite runs with `HIP error: operation not permitted on an event last recorded in a capturing stream` raised from
WorkNCCL::isCompleted() on the watchdog thread.  Reading: hipEventQuery reports hipErrorCapturedEvent when the stream an event was last
recorded on is capturing NOW - also for an event that was recorded eagerly, before the capture began (cudaEventQuery does not) - and this
torch build has no "wait for pending event queries" step in CUDAGraph::capture_begin.  The watchdog polls every 100 ms and only then drops
completed works, so a capture that starts within ~100 ms of the last eager collective races with its next poll.

usage (GPU box): python tools/probes/rccl_capture_watchdog_probe.py            -> runs both arms in child processes
                 python tools/probes/rccl_capture_watchdog_probe.py child DRAIN -> one arm (DRAIN = seconds slept before each capture)
"""
import os
import subprocess
import sys
import time


def child(drain: float, rounds: int = 12) -> None:
    import torch
    import torch.distributed as dist
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29731"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    x = torch.ones(1024, device=dev, dtype=torch.float64)
    y = torch.ones(1 << 20, device=dev)
    for r in range(rounds):
        for _ in range(40):          # the eager step's collectives: their works sit on the watchdog's list until its next poll
            dist.all_reduce(x)
        torch.cuda.synchronize()
        if drain > 0:
            time.sleep(drain)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="relaxed"):
            for i in range(40):      # host time inside the capture: the window the watchdog's poll must not fall into
                y.mul_(1.0001)
                dist.all_reduce(x)
                if i == 20:
                    time.sleep(0.15)  # (a whole-step capture takes this long on the host: a watchdog poll falls inside for sure)
        g.replay()
        torch.cuda.synchronize()
        time.sleep(0.013 * (r % 7))  # vary the phase against the watchdog's 100 ms period
        print(f"round {r} ok", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("child done", flush=True)


def main() -> None:
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(float(sys.argv[2]))
        return
    for drain in (0.0, 0.35):
        env = dict(os.environ, MASTER_PORT=str(29731 + int(drain * 100)))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(drain)], env=env, capture_output=True, text=True, timeout=280)
        ok = p.stdout.count(" ok")
        tail = [ln for ln in p.stderr.splitlines() if "HIP error" in ln or "terminate" in ln][:2]
        print(f"drain {drain:4.2f} s before each capture: exit code {p.returncode}, {ok} of 12 rounds completed", *tail, sep="\n    ", flush=True)


if __name__ == "__main__":
    main()
