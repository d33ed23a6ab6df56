"""Developer measurement: per-kernel floor inside a replayed hipGraph (a one-thread kernel back to back)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabletriton_amd import ops
from tools.op_bench import timeit
dev = torch.device("cuda:0")
step = torch.zeros(1, dtype=torch.int32, device=dev)
print(f"step_advance (1 thread): {timeit(lambda: ops.step_advance(step, 50), iters=200):.2f} us per launch in a graph")
x = torch.zeros(64, device=dev)
print(f"torch add_ (64 elements): {timeit(lambda: x.add_(1.0), iters=200):.2f} us per launch in a graph")
