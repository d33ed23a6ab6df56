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


def test_hint_size_rule_and_strided_encoding():
    """Which matrices a launch is asked to touch (ops._next_weights): whole up to 10 MB from launches of at most 1,024 rows, up to
    4 MB from larger ones (8 MB for e4m3), the others strided - byte count | row length in 128-byte lines << 40 | log2(lines per
    row) << 60 (include/stabletriton_amd.h, next_weights_bytes) - and nothing where a row is not a whole number of lines."""
    small = torch.empty((1280, 1280), dtype=torch.bfloat16)           # 3.3 MB
    mid = torch.empty((3840, 1280), dtype=torch.bfloat16)             # 9.8 MB
    big = torch.empty((10240, 1280), dtype=torch.bfloat16)            # 26 MB: rows of 20 lines
    q8 = torch.empty((5120, 1280), dtype=torch.uint8)                 # 6.5 MB of e4m3
    odd = torch.empty((40000, 100), dtype=torch.bfloat16)             # 8 MB, 200-byte rows
    order = [small, mid, big, q8, odd]
    ctx = ops.ExecContext()

    def hints(rows):
        for _ in range(2):
            with ctx.step():
                got = [ops._next_weights(t, rows) for t in order]
        return got                                                    # got[i] = the hint of the matrix AFTER order[i]
    nb = lambda t: t.numel() * t.element_size()
    strided = lambda t: nb(t) | ((nb(t) // t.shape[0] // 128) << 40) | (ops.HINT_LEAD_SHIFT << 60)
    g = hints(1024)
    assert g[4] == (small.data_ptr(), nb(small)) and g[0] == (mid.data_ptr(), nb(mid)) and g[1] == (big.data_ptr(), strided(big))
    assert g[2] == (q8.data_ptr(), nb(q8)) and g[3] == (odd.data_ptr(), nb(odd))          # (8 MB: whole from a small launch)
    g = hints(4096)
    assert g[4] == (small.data_ptr(), nb(small)) and g[0] == (mid.data_ptr(), strided(mid)) and g[1] == (big.data_ptr(), strided(big))
    assert g[2] == (q8.data_ptr(), nb(q8)) and g[3] == (None, 0)
    assert (strided(big) >> 40) & 0xfffff == 20 and strided(big) >> 60 == ops.HINT_LEAD_SHIFT and strided(big) & ((1 << 40) - 1) == nb(big)


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


def test_split_image_notes_follow_the_output_memory_and_die_with_the_pass():
    """Host side of st_arm_split_output (strict mode): a producer's split image is found by the consumer through the OUTPUT's
    memory (address, rows, row length), column ranges of it (the K / V slices of a fused q|k|v output) by their offset, the
    list keeps only the last few outputs, and nothing outlives the execution context's pass.  CPU tensors: no launch involved."""
    import torch
    from stabletriton_amd import ops
    ops_ = ops
    with ops_.ExecContext() as ctx:
        outs = [torch.zeros(8, 96) for _ in range(ops_._RECENT_SPLITS + 2)]
        imgs = [torch.full((8, 96), float(i)) for i in range(len(outs))]
        for o, im in zip(outs, imgs):
            ops_._note_split(o, im, 8, 96)
        assert len(ctx.recent_splits) == ops_._RECENT_SPLITS                      # the oldest notes are gone ...
        assert ops_._image_columns(outs[0], 8, 96) is None
        last = outs[-1]
        got = ops_._image_columns(last, 8, 96)                                   # ... the newest is found by address
        assert got is not None and got[0] is imgs[-1] and got[1] == 0
        mid = last[:, 32:64]                                                     # a column range: same rows, same row stride
        got = ops_._image_columns(mid, 8, 96)
        assert got is not None and got[0] is imgs[-1] and got[1] == 32
        assert ops_._image_columns(last[:, 16:48], 8, 96) is None                # not on a 32-value segment
        assert ops_._image_columns(last[:4], 4, 96) is None                      # other row count: not this matrix
        assert ops_._image_columns(torch.zeros(8, 96), 8, 96) is None            # another tensor
        last.add_(1.0)                                                           # an in-place update: the image is stale, the note no longer honoured
        assert ops_._image_columns(last, 8, 96) is None and ops_._image_columns(last[:, 32:64], 8, 96) is None
        held = [e[0] for e in ctx.recent_splits]
        assert all(any(h is o for o in outs) for h in held)                      # the notes HOLD the outputs (their memory cannot be recycled)
    assert not hasattr(ctx, "recent_splits")                                     # the pass is over: nothing is kept
    before, ops_.EMIT_SPLIT = ops_.EMIT_SPLIT, False
    try:
        with ops_.ExecContext() as ctx:
            ops_._note_split(outs[0], imgs[0], 8, 96)
            assert ops_._image_columns(outs[0], 8, 96) is None                   # switched off: consumers split for themselves
    finally:
        ops_.EMIT_SPLIT = before
