"""Per-iteration callbacks, mirroring src/callbacks.jl (host-side observability)."""
import time

import numpy as np


def nullcallback(cost, *unused):
    """src/callbacks.jl:20"""
    return cost, 0


def printoutcallback(cost, problem, data, iteratedata=None):
    """src/callbacks.jl:39-60"""
    prev = data.bestcost
    tr = getattr(iteratedata, "printable", lambda: None)() if iteratedata is not None else None
    if data.iternum == 1:
        prev = data.startcost
        print("iter      cost      cost_change    |step|" + ("    tr_radius" if tr is not None else ""))
        print("% 4d % 8e  % 4.3e   % 3.2e" % (0, prev, 0, 0) + ("   % 2.1e" % tr if tr is not None else ""))
    print("% 4d % 8e  % 4.3e   % 3.2e" % (data.iternum, cost, prev - cost, data.linsystem.step_norm())
          + ("   % 2.1e" % tr if tr is not None else ""))
    return cost, 0


class CostTrajectory:
    """src/callbacks.jl:85-107"""

    def __init__(self):
        self.costs, self.times_ns, self.trajectory = [], [], []

    def empty(self):
        self.costs.clear(); self.times_ns.clear(); self.trajectory.clear()
        return self


def storecostscallback(store):
    """src/callbacks.jl:63-66,100-133: store is a list (costs only) or a CostTrajectory."""
    if isinstance(store, CostTrajectory):
        def cb(cost, problem, data, *unused):
            store.costs.append(cost)
            store.times_ns.append(time.perf_counter_ns() - data.starttime)
            store.trajectory.append(np.array(data.linsystem.x))
            return cost, 0
        return cb

    def cb(cost, *unused):
        store.append(cost)
        return cost, 0
    return cb
