#!/usr/bin/env python3
"""Differential run: the same random calls on two engines whose road users own several parameter sets of several vehicle
classes - one re-bins into the class-segmented order (CSF_SEGMENTS=1: a launch of the culling kernel per set), the other
keeps the plain kernel that looks every source's set up (CSF_SEGMENTS=0; the variable is read at every re-binning, so it
is switched between the two engines' calls).  usage: fuzz_segments.py FIRST_SEED N_SEEDS"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CSF_PAIR_VARIANT"] = "0"
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402


def run(seed):
    rng = np.random.default_rng(seed)
    n0, box = int(rng.integers(1500, 4000)), float(rng.uniform(250, 500))
    cap = n0 + 1500
    pool, _, pdq = synthetic_population(cap + 4000, box, seed=seed + 7)
    pool = np.c_[pool, np.zeros(pool.shape[0])]
    pdq = pdq.reshape(-1, 4, 3)
    pods = [parameters.default_pod("twod"), parameters.default_pod("bicycle", hfov=2.5, p_0=35.0),
            parameters.default_pod("invpend", hfov=1.0, f_0=10.0), parameters.default_pod("planarpoint", hfov=3.6, sigma_0=0.6),
            parameters.default_pod("twod", f_0=0.0)]
    K = int(rng.integers(2, 6))
    pods = pods[:K]
    cls_of = rng.integers(0, K, pool.shape[0])
    engines = []
    for seg in ("1", "0"):
        os.environ["CSF_SEGMENTS"] = seg
        e = Engine(pods[0], cap)
        e.set_param_classes(pods)
        e.add_agents(pool[:n0], 5.0)
        e.set_dest_queue(np.arange(n0), np.arange(n0 + 1) * 4, pdq[:n0].reshape(-1, 3), reset=True)
        e.set_agent_class(np.arange(n0), cls_of[:n0])
        e.step(2)
        engines.append((seg, e))
    names = [e.count_pairs()[1] for _, e in engines]
    fresh, n = n0, n0
    cls_now = list(cls_of[:n0])
    hist = []
    worst = worst_f = 0.0
    for it in range(60):
        op = str(rng.choice(["step", "step", "step", "remove", "add", "replace", "table", "vdes"]))
        args = None
        if op == "step":
            args = int(rng.integers(1, 12))
        elif op == "remove" and n > 800:
            args = np.sort(rng.choice(n, int(rng.integers(1, 60)), replace=False))
        elif op == "add" and n + 80 < cap:
            k = int(rng.integers(1, 80)); args = np.arange(fresh, fresh + k); fresh += k
        elif op == "replace":
            k = int(rng.integers(1, 30)); args = (np.sort(rng.choice(n, k, replace=False)), rng.integers(0, pdq.shape[0], k))
        elif op == "table":
            c = int(rng.integers(0, K)); args = c
            m = pods[c].model
            name = {0: "bicycle", 1: "twod", 2: "invpend", 3: "planarpoint"}[m]
            pods[c] = parameters.default_pod(name, hfov=float(rng.uniform(0.8, 4.0)))
        elif op == "vdes":
            args = (np.sort(rng.choice(n, 20, replace=False)), rng.uniform(3.5, 5.5, 20))
        if args is None:
            continue
        for seg, e in engines:
            os.environ["CSF_SEGMENTS"] = seg
            if op == "step":
                e.step(args)
            elif op == "remove":
                e.remove_agents(args)
            elif op == "add":
                k = args.size
                e.add_agents(pool[args], 5.0)
                e.set_dest_queue(np.arange(n, n + k), np.arange(k + 1) * 4, pdq[args].reshape(-1, 3), reset=True)
                e.set_agent_class(np.arange(n, n + k), cls_of[args])
            elif op == "replace":
                e.set_dest_queue(args[0], np.arange(args[0].size + 1) * 4, pdq[args[1]].reshape(-1, 3), reset=1)
            elif op == "table":
                e.set_param_classes(pods)
            elif op == "vdes":
                e.set_v_desired(*args)
        if op == "remove":
            n -= args.size
            gone = set(args.tolist())
            cls_now = [c for i, c in enumerate(cls_now) if i not in gone]
        elif op == "add":
            n += args.size
            cls_now.extend(cls_of[args].tolist())
        hist.append(op)
        A, B = engines[0][1].state(), engines[1][1].state()
        assert A.shape == B.shape and A.shape[0] == n, (A.shape, B.shape, n)
        dp = np.abs(A[:, :2] - B[:, :2]).max(axis=1)
        out = int((dp > 1e-4).sum())
        worst = max(worst, float(dp.max()))
        # (a crowd amplifies rounding differences: the second engine is put back on the first one's state after every call,
        # so that what is compared is the last call alone; a source crossing a field-of-view edge a tick apart - D6 - still
        # moves a handful of road users by millimetres within a few ticks)
        if (out > max(40, n // 50) or dp.max() > 0.5 or not np.isfinite(A).all()) and os.environ.get("FUZZ_DEBUG"):
            bad = np.where(dp > 1e-4)[0]
            cl = np.array(cls_now)
            print("  deviating road users by set:", np.bincount(cl[bad], minlength=K), "of", np.bincount(cl, minlength=K), "models", [p.model for p in pods])
            for r in bad[:6]:
                print("   ", r, "set", cl[r], "A", A[r], "B", B[r])
        if out > max(40, n // 50) or dp.max() > 0.5 or not np.isfinite(A).all():
            raise AssertionError(f"seed {seed} call {it} ({op}): {out} road users differ, max {dp.max():.2e} m; kernels {names}; calls {hist}")
        os.environ["CSF_SEGMENTS"] = "0"
        engines[1][1].push_state(np.arange(n), A)
        # ... and on that common state the two kernels must agree on every receiver's repulsive sum (the sharp check: a
        # wrong set, a missed run, a stale circle shows here at once; chaos does not)
        os.environ["CSF_SEGMENTS"] = "1"
        engines[0][1].calc_forces()
        _, _, ax, ay = engines[0][1].force_parts()
        os.environ["CSF_SEGMENTS"] = "0"
        engines[1][1].calc_forces()
        _, _, bx, by = engines[1][1].force_parts()
        scale = max(np.hypot(ax, ay).max(), 1.0)
        # (the repulsive sum is clamped to |F_dest|, intersection.py:841-845, and the second engine's destination force
        # comes from its own ring history: where the clamp is active the two sums are compared after bringing them to the
        # same length)
        ma, mb = np.hypot(ax, ay), np.hypot(bx, by)
        fa_, fb_ = engines[0][1].force_parts(), engines[1][1].force_parts()
        clamped = (ma > 0.98 * np.hypot(fa_[0], fa_[1])) | (mb > 0.98 * np.hypot(fb_[0], fb_[1]))
        k = np.where(clamped & (mb > 0), ma / np.maximum(mb, 1e-300), 1.0)
        df = np.maximum(np.abs(ax - k * bx), np.abs(ay - k * by)) / scale
        nbad = int((df > 1e-3).sum())
        worst_f = max(worst_f, float(np.percentile(df, 99.9)))
        if nbad > 2:
            w = int(df.argmax())
            if os.environ.get("FUZZ_DEBUG"):
                sys.path.insert(0, os.path.join(ROOT))
                from oracle import csf_oracle as orc
                tab = [orc.Params.from_buffer_copy(bytes(p)) for p in pods]
                cl = np.array([cls_now[i] for i in range(n)], dtype=np.uint8)
                bad = np.where(df > 1e-3)[0]
                ox, oy = orc.column_sums(tab, A[:, 0], A[:, 1], A[:, 2], A[:, 3], bad, cls=cl)
                fa = engines[0][1].force_parts(); fb = engines[1][1].force_parts()
                for j, r in enumerate(bad):
                    lim = np.hypot(fa[0][r], fa[1][r]); mag = np.hypot(ox[j], oy[j]); sc = lim / mag if mag > lim else 1.0
                    print("  receiver", r, "class", cl[r], "oracle (clamped)", ox[j] * sc, oy[j] * sc, "A", ax[r], ay[r], "B", bx[r], by[r],
                          "destA", fa[0][r], fa[1][r], "destB", fb[0][r], fb[1][r])
            raise AssertionError(f"seed {seed} call {it} ({op}): repulsive sums of {nbad} receivers differ on the same state, worst {df.max():.2e} at {w} "
                                 f"(A {ax[w]:.4f} {ay[w]:.4f}, B {bx[w]:.4f} {by[w]:.4f}); kernels {names}; calls {hist}")
    for _, e in engines:
        e.close()
    return n, K, names, worst, worst_f


first, count = int(sys.argv[1]), int(sys.argv[2])
ok = bad = 0
for seed in range(first, first + count):
    try:
        print("seed", seed, run(seed))
        ok += 1
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print("SEED", seed, "FAILED", type(ex).__name__, str(ex)[:800])
print("ok", ok, "bad", bad)
