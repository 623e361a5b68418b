"""CPU: the measurement tool chain behind `roofline.traffic` — tools/pmc_traffic.py turns one `rocprofv3 --pmc FETCH_SIZE`
pass of the bench plus the calibration pass of tools/ubench/fetch_calib into profiles/rNN_traffic.json, and bench.py
reads the newest such file.  Run here on the committed round-3 inputs: the committed output must be what the tool
makes of them, the calibration factors must be what the counters say, and bench.py must pick that file up for the
workload it was taken on (and for no other)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def test_committed_traffic_file_is_what_the_tool_makes_of_the_committed_counters(tmp_path):
    bench_line = os.path.join(P, "r03_bench_final.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(P, "r03_pmc_fetch_size.txt"),
                          os.path.join(P, "r03_fetch_calib_pmc.txt"), os.path.join(P, "r03_fetch_calib_stdout.txt"), bench_line, "3"],
                         check=True, capture_output=True, text=True).stdout
    made = json.loads(out)
    with open(os.path.join(P, "r03_traffic.json")) as f:
        committed = json.load(f)
    assert made["workload"] == committed["workload"]
    assert set(made["groups"]) == set(committed["groups"]) >= {"blake2b_cid", "cid_index", "amt_walk", "event_scan", "event_verify"}
    for g in made["groups"]:
        a, b = made["groups"][g]["traffic_bytes_per_step"], committed["groups"][g]["traffic_bytes_per_step"]
        assert abs(a - b) <= 1e-6 * max(a, b), (g, a, b)
    # the calibration: 2 GiB touched exactly once reads as half of it for the streaming and the random patterns alike
    pat = made["calibration"]["patterns"]
    for name in ("stream", "rand16", "rand64", "rand128"):
        assert abs(pat[name]["factor"] - 2.0) < 0.01, (name, pat[name])
    assert pat["lane_seq"]["refetch_ratio"] > 3.0  # a lane walking its own record re-fetches lines at full occupancy
    # K1: line padding of ~341-byte blocks, nothing else
    with open(bench_line) as f:
        line = json.load(f)
    algo = line["roofline"]["algorithmic_bytes_per_launch"]
    ratio = made["groups"]["blake2b_cid"]["traffic_bytes_per_step"] / algo
    assert 1.2 < ratio < 1.4, ratio


def test_bench_reads_the_newest_traffic_file_of_the_same_workload():
    sys.path.insert(0, ROOT)
    import bench

    with open(os.path.join(P, "r03_traffic.json")) as f:
        committed = json.load(f)
    n_blocks = committed["workload"]["witness_blocks"]
    tr = bench.load_traffic(n_blocks)
    assert tr is not None and tr["file"].startswith("profiles/r") and tr["file"].endswith("_traffic.json")
    assert tr["workload"]["witness_blocks"] == n_blocks
    assert bench.load_traffic(n_blocks + 1) is None  # another workload: no figure rather than a wrong one


def test_workload_traffic_tool_on_a_small_pmc_table(tmp_path):
    """tools/pmc_traffic_workload.py: FETCH_SIZE KB (summed over dispatches) x 1024 x 2.00 / steps, over the library's kernels
    only; the steps are the dispatch count of the workload's once-per-step kernel."""
    pmc = tmp_path / "hamt_pmc1.txt"
    pmc.write_text("# kernel | counter | samples | sum | per-dispatch-sample mean\n"
                   "ipcfp::k_hamt_lv_start | FETCH_SIZE | 7 | 700 | 100\n"
                   "ipcfp::k_hamt_lv_advance | FETCH_SIZE | 42 | 84000 | 2000\n"
                   "void ipcfp::k_hamt_lv_parse_actor<6912u, 96u, false> | FETCH_SIZE | 42 | 420000 | 10000\n"
                   "__amd_rocclr_fillBufferAligned | FETCH_SIZE | 9 | 900 | 100\n"
                   "ipcfp::k_hamt_lv_start | SQ_WAVES | 7 | 7 | 1\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic_workload.py"), "5", "hamt=%s" % pmc],
                         check=True, capture_output=True, text=True).stdout
    made = json.loads(out)
    w = made["workloads"]["hamt"]
    assert made["round"] == 5 and w["steps_in_pass"] == 7 and len(w["kernels"]) == 3
    assert abs(w["traffic_bytes_per_step"] - (700 + 84000 + 420000) * 1024 * 2.0 / 7) < 1e-3
