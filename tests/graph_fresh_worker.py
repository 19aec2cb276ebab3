"""Worker of tests/test_graph_gpu.py::test_step_graphs_small_then_large_in_a_fresh_process (run as a child process, so that nothing has
sized the library's cached buffers before the graphs are built).

ADVICE round 3 (high): the weight-gradient slab buffers and the device table of their batched reduction are cached per launch index; a
hipGraph captured for a small input shape has their addresses baked in, and a LARGER shape arriving later used to replace (free) them.
Here StepGraphs sees the small shape first, then the large one, with NO eager step before either capture; the four replayed steps
A, B, A, B must equal the same four steps issued eagerly afterwards, bit for bit.  Prints one JSON line.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (REPO, os.path.join(REPO, "unsupervised-pseuso-lidar_amd"), HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    import test_graph_gpu as T
    from mcav import nn as N
    from mcav.graph import StepGraphs
    dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else None
    shapes = [(2, 64, 128), (12, 192, 640), (2, 64, 128), (12, 192, 640)]
    data = T.batches(shapes, seed0=90)
    d, p, opt = T.build(dtype)
    graphs = StepGraphs(T.make_fwd_bwd(d, p, opt), opt, capture_adam=True, buffers=list(d.buffers()) + list(p.buffers()))
    losses, sizes = [], []
    for b in data:
        losses.append([float(l) for l in graphs(*b)])
        sizes.append(sum(t.numel() for t in N.WGRAD_BATCH.pool.values()))
    got = T.state_of(d, p, opt)
    retired = len(N.WGRAD_BATCH.retired)
    del graphs
    d, p, opt = T.build(dtype)
    ref_losses, ref = T.run_eager(d, p, opt, data, capturable=True, serial=True)
    same = all(torch.equal(got[k], ref[k]) for k in ("flat", "m", "v")) and got["step"] == ref["step"]
    same_buf = all(torch.equal(x, y) for (_, x), (_, y) in zip(got["buffers"], ref["buffers"]))
    worst = max(float((got[k] - ref[k]).abs().max()) for k in ("flat", "m", "v"))
    print(json.dumps(dict(state_equal=bool(same), buffers_equal=bool(same_buf), losses=losses, ref_losses=ref_losses, worst_abs=worst,
                          slab_buffers_retired=retired, pool_bytes_after_each_step=sizes)))


if __name__ == "__main__":
    main()
