"""bench.py end to end: the single-GPU line, and the multi-rank path rehearsed with two gloo ranks that
share the one GPU of the test box (the driver runs the real N = 2, 4, 8 over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--columns", "64", "--no-cpu-baseline", "--no-extras"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["unit"] == "columns/s" and d["value"] > 0 and d["dtype"] == "f64"
    assert d["roofline"]["bound"] in ("mfma", "hbm") and 0 < d["roofline"]["frac"] < 1
    assert {d["roofline"]["kernel"], d["roofline_other"]["kernel"]} == {"k_jn_gemm", "k_transport_ring + k_transport_scan"}
    assert "roofline_order_loop" not in d                  # (the order-loop launches are off by default)
    assert d["config"]["not_converged"] == 0 and d["config"]["columns_per_gpu"] == 64
    # the run checks itself: sampled columns against the oracle, outside the timed region
    assert d["check"]["ok"] and d["check"]["max_rel_err_vs_oracle"] <= 1e-10 and d["check"]["orders_match"]
    assert d["check"]["p0_max_rel_err_vs_oracle"] <= 1e-12
    # the default workload carries the aerosol BASELINE names
    assert d["config"]["aerosol"] == "eva" and "EVA log-normal Mie" in d["config"]["workload"]
    # every frac follows from the line with one division
    for r in (d["roofline"], d["roofline_other"]):
        assert abs(r["achieved"] - r["work_per_launch"] / (r["avg_launch_ms"] * 1e-3) / (1e9 if r["unit"] == "GB/s" else 1e12)) < 1e-6 * r["achieved"]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    tr = d["roofline"] if d["roofline"]["bound"] == "hbm" else d["roofline_other"]
    assert abs(tr["work_per_launch"] * tr["launches"] - 32.0 * 200 * 256 * d["config"]["orders_per_step"] * d["steps"]) < 1.0
    rs = d["roofline_step"]
    assert 0 < rs["frac_executed"] <= rs["frac_full_product"] < 1
    assert d["strong"]["status"].startswith("unmeasured")


def test_bench_accounts_for_order_loop_launches():
    """SOSRT_ORDER_LOOP=1 (opt-in): 216 columns -- the first orders are two launches each; once at most 128 columns are live the
    rest of the solve is ONE order-loop launch, priced on a line of its own; the (column, order) pairs of a step are in one or
    the other."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--columns", "216", "--no-cpu-baseline", "--no-extras",
                        "--pipelined", "0"], cwd=ROOT, env=dict(os.environ, SOSRT_ORDER_LOOP="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["check"]["ok"] and d["config"]["not_converged"] == 0
    ol = d["roofline_order_loop"]
    assert ol["kernel"].startswith("k_order_loop") and ol["launches"] == d["steps"] and ol["refused_launches"] == 0
    for r in (d["roofline"], d["roofline_other"], ol):
        assert abs(r["achieved"] - r["work_per_launch"] / (r["avg_launch_ms"] * 1e-3) / (1e9 if r["unit"] == "GB/s" else 1e12)) < 1e-6 * r["achieved"]
    two = [r for r in (d["roofline"], d["roofline_other"]) if not r["kernel"].startswith("k_order_loop")]
    assert two and all(r["column_orders_per_step"] + ol["column_orders_per_step"] == d["config"]["orders_per_step"] for r in two)


def test_bench_extras_c2_c3_c5_and_hg_stand_in():
    """The other single-GPU configurations of BASELINE.json beside the headline: one EVA column at N_mu = 128 and 256
    (single-column latency), the 4096-column wildfire sweep at L = 400, N = 256, each checked against the oracle."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--columns", "27", "--no-cpu-baseline",
                        "--aerosol", "hg", "--pipelined", "0"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["config"]["aerosol"] == "hg" and d["check"]["ok"]
    ex = d["extras"]
    assert "error" not in ex, ex
    for k, cols, N in (("c2", 1, 128), ("c3", 1, 256), ("c5", 4096, 256)):
        e = ex[k]
        assert e["check"]["ok"] and e["check"]["max_rel_err_vs_oracle"] <= 1e-10 and e["not_converged"] == 0, (k, e)
        assert ("%d column" % cols) in e["workload"] and ("N=%d" % N) in e["workload"] and e["columns_per_s"] > 0
    assert ex["c5"]["columns_per_s"] > 10 * ex["c3"]["columns_per_s"]          # a batch fills the GPU, a lone column cannot
    # beside the headline (one column group, every kernel alone): the library's own order loop for the batch, same bits
    assert "one column group" in d["config"]["order_loop"]
    assert d["two_groups"]["same_bits_as_headline"] and d["two_groups"]["value"] > 0


def test_bench_two_column_groups_at_the_headline_size():
    """512 columns at (200, 128): the library takes two column groups by itself (one field of the batch = 210 MB); `two_groups`
    measures that loop beside the one-group headline and must find the headline's bits."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--pipelined", "0",
                        "--check-columns", "1"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["check"]["ok"] and d["two_groups"]["same_bits_as_headline"]
    # (the ratio is reported, not asserted: 0.94 on a quiet box, but a wall-clock ratio flakes on a busy one -- ADVICE r3)
    print("two groups / one group: %.3f" % (d["two_groups"]["ms_per_step"] / d["ms_per_step"]))
    assert d["strong"]["status"].startswith("unmeasured")
    sh = d["extras"]["c4_shard"]
    assert sh["check"]["ok"] and sh["workload"].startswith("64 columns") and 0 < sh["implied_8gpu_strong_ceiling"]["efficiency"] < 1
    assert d["extras"]["c2"]["check"]["ok"] and d["extras"]["c3"]["check"]["ok"] and d["extras"]["c5"]["check"]["ok"]


def test_bench_two_ranks_share_the_gpu():
    env = dict(os.environ, SOSRT_BENCH_BACKEND="gloo", SOSRT_BENCH_SHARE_GPU="1")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--columns", "27",
           "--inflight", "2", "--aerosol", "hg"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["columns_per_gpu"] == [27, 27] and "cpu_baseline" not in d
    assert len(d["config"]["orders_per_step_per_rank"]) == 2 and d["check"]["ok"]
    # the same run also measures ONE sweep over the two ranks (BASELINE configs[3]) beside the weak-scaling headline
    st = d["strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["value"] > 0 and st["gather_places_columns"]
    assert sorted(st["columns_per_gpu"]) == [13, 14]


def test_bench_strong_scaling_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment starts the two ranks itself; --scaling strong deals ONE
    sweep to them by expected work and gathers the whole fields to rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SOSRT_BENCH_BACKEND="gloo", SOSRT_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--scaling", "strong", "--steps", "2", "--warmup", "1",
                        "--columns", "27"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert sorted(d["config"]["columns_per_gpu"]) == [13, 14]
    per = d["config"]["orders_per_step_per_rank"]
    assert max(per) / min(per) < 1.3                      # the deal balances the sum of orders
    assert d["check"]["ok"] and d["check"]["gather_places_columns"]
