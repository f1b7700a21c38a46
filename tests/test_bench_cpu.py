"""bench.py's host logic (no GPU): the untimed pre-heat runs the SAME number of steps on every rank (a step contains a collective),
the timed region is exactly K steps between two fences, and the fixed-stream mode derives its step count from the frame total."""
import importlib.util
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_timed_region_is_exactly_k_steps_after_warmup_and_preheat(bench):
    log = []
    bench.time.sleep(0)

    def step():
        log.append("s")

    def fence():
        log.append("|")

    elapsed, pre = bench.timed_steps(step, fence, steps=7, warmup=3, preheat_s=0.0, frames_per_step=256)
    assert "".join(log) == "sss|" + "s" * 7 + "|" and pre is None and elapsed >= 0
    log.clear()
    elapsed, pre = bench.timed_steps(step, fence, steps=4, warmup=1, preheat_s=0.01, frames_per_step=256)
    text = "".join(log)
    assert text.startswith("s|" + "s" * 10 + "|") and text.endswith("|ssss|") and pre > 0


def test_preheat_step_count_follows_the_all_reduced_time(bench):
    """Every rank must run the same number of pre-heat steps: the count is derived from the MAX-reduced time of the first ten."""
    counts = []
    for fake_dt in (0.001, 0.004):                     # what sync_max returns (the slowest rank's time for ten steps)
        n = [0]

        def step():
            n[0] += 1

        bench.timed_steps(step, lambda: None, steps=2, warmup=0, preheat_s=0.02, frames_per_step=1, sync_max=lambda v, d=fake_dt: d)
        counts.append(n[0] - 2)
    # 10 steps measured + ceil((0.02 - dt) / (dt / 10)) more: a function of the reduced time only, not of this rank's own clock
    assert counts == [10 + 190, 10 + 40]


def test_host_cores_is_positive_and_bounded(bench):
    assert 1 <= bench.host_cores() <= 64
