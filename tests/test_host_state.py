"""Host-side state and launch plumbing that needs no GPU: the per-module execution context (weight plan, derived
weight buffers), bench.py's self-launch of N ranks, the CPU-baseline core count, the RCCL log digest."""
import json
import os
import sys
import threading
import types

import torch

from stabletriton_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_weight_plan_records_then_hints_the_next_launch():
    plan = ops.WeightPlan()
    w = [torch.zeros(4, 4) for _ in range(3)]
    assert plan.next_after(w[0]) is None                      # off until a step begins
    plan.begin()
    assert [plan.next_after(t) for t in w] == [None, None, None]          # first pass records
    plan.begin()
    assert plan.state == "replay"
    assert plan.next_after(w[0]) is w[1] and plan.next_after(w[1]) is w[2] and plan.next_after(w[2]) is w[0]
    plan.begin()
    assert plan.next_after(w[0]) is w[1]
    assert plan.next_after(w[2]) is None and plan.state == "off"          # sequence changed: stop hinting
    # entries are the tensors themselves: a hinted buffer cannot be freed while the plan (or a graph built from it) lives
    plan.begin(); plan.next_after(w[0]); plan.begin()
    assert plan.entries[0] is w[0]


def test_step_scope_hints_only_inside_steps_and_contexts_nest_per_thread():
    ctx = ops.ExecContext()
    cpu = torch.device("cpu")
    assert ops.current_context(cpu) is not ctx
    w = [torch.zeros(2, 2), torch.ones(2, 2)]
    with ctx:
        assert ops.current_context(cpu) is ctx
        assert ops._next_weights(w[0]) == (None, 0)           # a one-off pass (context / time table) stays out of the plan
        assert ctx.plan.state == "off"
    for _ in range(2):
        with ctx.step():
            got = [ops._next_weights(t) for t in w]
    assert got[0] == (w[1].data_ptr(), 16) and got[1] == (w[0].data_ptr(), 16)
    seen = []
    t = threading.Thread(target=lambda: seen.append(ops.current_context(cpu) is ctx))
    with ctx:
        t.start(); t.join()
    assert seen == [False]                                    # the current context is per thread
    assert ops.current_context(cpu) is ops.current_context(cpu)


def test_derived_weights_refresh_in_place():
    lin = [torch.nn.Linear(8, 4, bias=False), torch.nn.Linear(8, 6, bias=False)]
    ctx = ops.ExecContext()
    src = [l.weight for l in lin]
    compute = lambda: (torch.cat([l.weight.detach() for l in lin]).contiguous(),)
    d = ctx.derived_weights(("cat", 1), src, compute)
    buf = d.value[0]
    ptr = buf.data_ptr()
    assert ctx.refresh_derived(full=True) == 0
    with torch.no_grad():
        lin[1].weight.mul_(2.0)                               # in-place update (LoRA merge): version counter moves
    assert ctx.refresh_derived() == 1
    assert buf.data_ptr() == ptr and torch.equal(buf, compute()[0])       # same storage, new values
    with torch.no_grad():
        lin[0].weight.data = torch.randn(4, 8)                # storage swap: only the full check sees it
    assert ctx.refresh_derived() == 0 and ctx.refresh_derived(full=True) == 1
    assert buf.data_ptr() == ptr and torch.equal(buf, compute()[0])
    assert ctx.derived_weights(("cat", 1), src, compute) is d
    other = [torch.nn.Linear(8, 4, bias=False).weight]
    assert ctx.derived_weights(("cat", 1), other, lambda: (other[0].detach().clone(),)) is not d       # key reused by new modules


def _rank_script(fail_rank=-1):
    return ("import os, sys, json\n"
            "r = int(os.environ['RANK']); w = int(os.environ['WORLD_SIZE'])\n"
            "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['LOCAL_RANK']) == r\n"
            f"if r == {fail_rank}: sys.exit(3)\n"
            "print(json.dumps({'rank': r, 'world': w, 'port': os.environ['MASTER_PORT']}))\n")


def test_bench_self_launch_relays_rank0_and_fails_loudly(capsys):
    args = types.SimpleNamespace(gpus=3)
    assert bench.self_launch(args, [sys.executable, "-c", _rank_script()]) == 0
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0])["rank"] == 0 and json.loads(out[0])["world"] == 3
    assert bench.self_launch(args, [sys.executable, "-c", _rank_script(fail_rank=2)]) != 0
    assert "rank 2" in capsys.readouterr().err


def test_host_cores_is_bounded_by_this_process():
    n, desc = bench.host_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and n <= len(os.sched_getaffinity(0))
    assert "physical" in desc


def test_rccl_report_digest(tmp_path):
    log = tmp_path / "rccl.log"
    log.write_text("host:1:1 [0] NCCL INFO comm 0x1 rank 0 nranks 8 cudaDev 0 busId c000 - Init START\n"
                   "host:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC\n"
                   "host:1:1 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC\n")
    rep = bench.rccl_report(str(log))
    assert rep["nranks_logged"] == 8 and rep["transports"] == ["P2P/IPC"] and rep["channels"] == 2
    assert bench.rccl_report(str(tmp_path / "missing.log")) is None


def test_rccl_self_check_fails_a_partial_communicator(tmp_path):
    """backend=nccl: the first real multi-GPU run checks itself - a communicator that does not span the job is an error."""
    assert bench.rccl_self_check({"nranks_logged": 8, "transports": ["P2P/IPC"], "channels": 2}, 8) is None
    assert "8" in bench.rccl_self_check({"nranks_logged": 4, "transports": [], "channels": 0}, 8)
    assert bench.rccl_self_check(None, 2) is not None                      # no log and no all-reduce witness
    assert bench.rccl_self_check(None, 2, ranks_seen=2) is None              # the all-reduce saw every rank: a missing log does not kill the job
    assert bench.rccl_self_check({"nranks_logged": None, "transports": [], "channels": 0}, 8, ranks_seen=8) is None
    assert "4" in bench.rccl_self_check({"nranks_logged": 8, "transports": [], "channels": 0}, 8, ranks_seen=4)
    assert "4" in bench.rccl_self_check({"nranks_logged": 4, "transports": [], "channels": 0}, 8, ranks_seen=8)
