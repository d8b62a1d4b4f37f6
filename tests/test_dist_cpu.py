"""N > 1 host logic on CPU: world_size 2 over gloo (the product kernels need a GPU; here the oracle checks the
sharding rule and the reductions it implies)."""
import os
import subprocess
import sys

import numpy as np

from nllssolver_jl_amd.dist import partition_by_weight

HERE = os.path.dirname(os.path.abspath(__file__))


def test_partition_by_weight():
    w = np.array([5, 1, 1, 1, 4, 4, 2, 2], dtype=np.int64)
    b = partition_by_weight(w, 4)
    assert b[0] == 0 and b[-1] == len(w) and np.all(np.diff(b) >= 0)
    sums = [w[b[k]:b[k + 1]].sum() for k in range(4)]
    assert sum(sums) == w.sum() and max(sums) <= w.sum() / 4 + w.max()
    assert list(partition_by_weight(np.ones(10), 1)) == [0, 10]


def test_world_size_2_gloo():
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_cpu_worker.py"), str(r), "2", "29611"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0 and "ok" in out, f"rank {r}:\n{out[-3000:]}"
