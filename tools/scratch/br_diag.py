import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from oracle import csf_oracle as orc
from conftest import shadow_run
from cyclistsocialforce_amd import engine, parameters
from test_gpu_balancingrider import crowd

class NS: pass
amd = NS(); amd.Engine = engine.Engine; amd.pod = parameters.default_pod

def mixed():
    rng = np.random.default_rng(17)
    n = 600
    order = ["balancingrider", "twod", "invpend", "bicycle", "balancingrider"]
    pods = [amd.pod(m) for m in order[:4]] + [amd.pod("balancingrider", hfov=2.0, k_p_v=4.0)]
    s0, off, dq = crowd(n, 90.0, seed=21)
    cls = rng.integers(0, len(pods), n).astype(np.uint8)
    for k, m in enumerate(order):
        s0[cls == k, orc.N_STATES[orc.MODEL_IDS[m]]:] = 0.0
    e = amd.Engine(pods[0], n + 8)
    e.set_param_classes(pods)
    e.add_agents(s0, 4.5)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.set_agent_class(np.arange(n), cls)
    classes = [orc.Params.from_buffer_copy(bytes(p)) for p in pods]
    pop = orc.Population(classes[0], s0, 4.5, off, dq, ns=8)
    pop.set_classes(classes, cls)
    for t in range(3):
        e.step(1); pop.step(1)
        got, ref = e.state(), pop.state()
        d = np.abs(got - ref)
        for k in range(5):
            m = cls == k
            print("tick", t, "class", k, order[k], "max |ds| per column", d[m].max(axis=0))
        x, g, _ = e.integrator_state(); ox = pop.lti()[0]
        print("   x dev", np.abs(x - ox).max(axis=0))

def big():
    n, box = 5000, 260.0
    s0, off, dq = crowd(n, box, seed=n)
    vdes = np.random.default_rng(1).uniform(3.0, 5.5, n)
    p = amd.pod("balancingrider", hfov=3.0)
    e = amd.Engine(p, n); e.add_agents(s0, vdes); e.set_dest_queue(np.arange(n), off, dq, reset=True)
    pop = orc.Population(orc.default_params("balancingrider", hfov=3.0), s0, vdes, off, dq)
    worst, devs, got, ref = shadow_run(e, pop, 60, 10)
    dd = np.abs(got[:, 2:] - ref[:, 2:])
    bad = np.where((dd > np.array([2e-3, 5e-3, 5e-3, 5e-3, 5e-2, 5e-2])).any(axis=1))[0]
    for j in bad:
        print("rider", j, "got", got[j], "\n      ref", ref[j], "dpos", devs[j])
    # how wild is the population
    print("riders with |steer| > 0.5:", (np.abs(ref[:, 4]) > 0.5).sum(), " |roll| > 0.5:", (np.abs(ref[:, 5]) > 0.5).sum(), " max |steer rate|", np.abs(ref[:, 6]).max())

mixed(); big()
