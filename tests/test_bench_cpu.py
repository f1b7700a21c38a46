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


def _write_counter_csv(path, counter, launches):
    """launches: (dispatch id, kernel name, grid, KiB)"""
    import csv
    cols = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
            "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
            "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(cols)
        for did, name, grid, kib in launches:
            w.writerow([did, did, "Agent 2", 1, 1, 1, grid, 9, name, 512, 0, 0, 8, 0, 32, counter, kib, 0, 1])


def test_pmc_summary_counts_full_batch_launches_only(tmp_path):
    """Round-2 bug: bench.py's batch-2 check forward was averaged into every class (stem 77.3 MB x 7 and one launch of 1.4 MB).  The
    summary keeps only (kernel, grid) groups that occur a whole number of times per full-batch forward."""
    sys.path.insert(0, str(ROOT / "scripts"))
    import io
    import pmc_bench_summary as P
    stem, conv = "void stem_fused3_kernel<0>(float const*)", "void igemm_ws_kernel<0, 128, 224, 4, 2, 4, 3>(ConvArgs)"
    fetch, write = [], []
    did = 0
    for f in range(7):                                   # 7 full-batch forwards: stem, conv, conv (two layers on one kernel, same grid)
        for name, grid, kib_f, kib_w in ((stem, 131072, 77300.0, 200000.0), (conv, 196608, 50000.0, 25000.0), (conv, 196608, 30000.0, 25000.0)):
            did += 1
            fetch.append((did, name, grid, kib_f)); write.append((did, name, grid, kib_w))
    for name, grid, kib_f, kib_w in ((stem, 1024, 1400.0, 1500.0), (conv, 3072, 400.0, 200.0), (conv, 3072, 300.0, 200.0)):   # the batch-2 check forward
        did += 1
        fetch.append((did, name, grid, kib_f)); write.append((did, name, grid, kib_w))
    ff, wf = tmp_path / "f_counter_collection.csv", tmp_path / "w_counter_collection.csv"
    _write_counter_csv(ff, "FETCH_SIZE", fetch)
    _write_counter_csv(wf, "WRITE_SIZE", write)
    log = io.StringIO()
    layers = [{"name": "conv1", "bytes_per_launch": 360e6}, {"name": "layer3.0.conv2", "bytes_per_launch": 120e6}, {"name": "layer3.1.conv2", "bytes_per_launch": 90e6}]
    res, table = P.summarize([str(ff)], [str(wf)], forwards=7, layers=layers, log=log)
    assert "dropped 1 launch" in log.getvalue() and "dropped 2 launch" in log.getvalue()
    assert res["conv1"]["launches_counted"] == 7 and res["igemm"]["launches_counted"] == 14
    assert res["conv1"]["fetch_bytes_per_launch_raw"] == pytest.approx(77300.0 * 1024)
    assert res["igemm"]["hbm_bytes_per_launch"] == pytest.approx((2 * 40000.0 + 25000.0) * 1024)
    assert [r["layer"] for r in table] == ["conv1", "layer3.0.conv2", "layer3.1.conv2"]
    assert table[1]["read_bytes_x2"] == pytest.approx(2 * 50000.0 * 1024) and table[2]["written_bytes"] == pytest.approx(25000.0 * 1024)
    assert table[1]["ratio"] == pytest.approx((2 * 50000.0 + 25000.0) * 1024 / 120e6)
    assert res["_step"]["launches"] == 3
    # without the guard the check launches pull the averages down (what round 2 committed)
    res0, table0 = P.summarize([str(ff)], [str(wf)], forwards=0, log=log)
    assert res0["conv1"]["launches_counted"] == 8 and res0["conv1"]["fetch_bytes_per_launch_raw"] < 0.9 * 77300.0 * 1024 and table0 is None


def test_lane_plan_falls_back_to_one_lane_when_the_lanes_cost_time():
    """BackboneLanes.tune's decision (`fp8_b512` reported 122.1 k frames/s on two lanes beside its own `single_lane` 125.7 k):
    a last ratio below 1 means submit() deals every batch to lane 0; between 1 and the tuning target both lanes stay."""
    from implementation_phd_lab_vision_amd.backbone import BackboneLanes
    assert BackboneLanes.lane_plan([], 2)[0] == 2
    assert BackboneLanes.lane_plan([1.07], 2) == (2, "2 lanes (1.070 x one lane)")
    n, mode = BackboneLanes.lane_plan([0.99, 0.971], 2)
    assert n == 1 and "fallback" in mode and "0.971" in mode
    n, mode = BackboneLanes.lane_plan([0.99, 1.005], 2)
    assert n == 2 and "below the tuning target" in mode
    assert BackboneLanes.lane_plan([0.5], 1) == (1, "one lane")
    assert BackboneLanes.lane_plan([0.97], 2, drop_below=0.0)[0] == 2


def test_roofline_object_describes_the_largest_class(bench):
    """The `roofline` object of the bench line is the kernel class with the largest ms_per_step, carries both
    fractions and its bound = the larger one; the igemm class rides beside it."""
    prof = {"igemm": {"launches": 38, "ms": 2.2, "flops": 1.97e12, "bytes": 4.0e9},
            "bneck_block2": {"launches": 12, "ms": 2.3, "flops": 1.4e12, "bytes": 8.7e9},
            "bneck_tail3": {"launches": 10, "ms": 0.7, "flops": 0.5e12, "bytes": 2.5e9},
            "conv1": {"launches": 2, "ms": 0.28, "flops": 0.13e12, "bytes": 0.72e9},
            "avgpool": {"launches": 2, "ms": 0.02, "flops": 0.0, "bytes": 0.1e9}}
    r = bench.roofline_object(prof, steps=2, precision="bf16", traffic={"bneck_block": {"hbm_bytes_per_launch": 8.0e8}}, traffic_source="static")
    assert r["class"] == "bneck_block2" and r["bound"] == "hbm" and r["unit"] == "GB/s"
    assert r["achieved"] == pytest.approx(8.7e9 / 2.3e-3 / 1e9) and r["frac"] == pytest.approx(r["achieved"] / 8000.0)
    assert r["frac_of_mfma_peak"] == pytest.approx(1.4e12 / 2.3e-3 / 1e12 / 2500.0) and r["traffic"] == 8.0e8
    assert r["launches_per_step"] == 6 and r["ms_per_step"] == pytest.approx(1.15)
    ig = r["classes"]["igemm"]
    assert ig["bound"] == "mfma" and ig["frac"] == pytest.approx(1.97e12 / 2.2e-3 / 1e12 / 2500.0) and ig["launches_per_step"] == 19
    assert set(r["classes"]) == {"igemm", "bneck_tail3", "conv1", "avgpool"}
