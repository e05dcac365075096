"""`python bench.py --gpus N` without a torch.distributed.run parent must start its own N ranks as a child
process before anything touches the GPU (the driver's scaling run may invoke it that way)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                       timeout=300)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{") and "rank_check" in l]
    return p, lines


def test_bench_self_launches_its_ranks_before_any_gpu_call():
    p, lines = _run(None, "--gpus", "2", "--rank-check")
    assert p.returncode == 0, p.stderr[-2000:]
    assert sorted(l["rank"] for l in lines) == [0, 1]
    assert all(l["world"] == 2 and l["cuda_initialized"] is False for l in lines)


def test_bench_under_an_external_launcher_does_not_relaunch():
    # what the driver does: torch.distributed.run has already set WORLD_SIZE / RANK for this process
    p, lines = _run({"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, "--gpus", "2", "--rank-check")
    assert p.returncode == 0, p.stderr[-2000:]
    assert lines == [{"rank_check": True, "rank": 1, "world": 2, "local_rank": 1, "cuda_initialized": False}]


def test_bench_rejects_a_world_that_differs_from_gpus():
    p, _ = _run({"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--rank-check")
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)


# ---- the final stdout line (a 20.9 KB line was lost by the driver's parser) -------------------------------------
def _fat_record():
    """a result record as bench.py assembles it, with the bulk the lab harness adds (long prose, per-rank arrays, per-tensor tables)"""
    prose = "x" * 900
    per_rank = [1.1488677992019802 + i * 1e-3 for i in range(8)]
    vw = {"hubs_sage": {"what": prose, "world": 8, "t1_ms": 6.64, "per_rank_ms": per_rank, "per_rank_entries": [2623308] * 8,
                        "balance": 0.99, "compute_ceiling": 5.72, "bytes_per_collective": {"all_gather": {"calls": 2, "payload_bytes": 204800000}},
                        "emulated_wire": {"assumptions": {"what": prose}, "by_wire_GBps": {"800": {"rank0_ms": 1.24}}}},
          "hubs_gat": {"per_rank_ms": per_rank, "what": prose}, "rows_sage": {"per_rank_ms": per_rank, "what": prose},
          "hubs_sage_by_world": {"2": {"compute_ceiling": 1.79, "per_rank_ms": per_rank[:2]}, "4": {"compute_ceiling": 3.3}}, "note": prose}
    cfg = {k: {"workload": prose, "ms_per_step": 0.5, "ms_per_step_graph": 0.33, "parity_max_abs_err": 1e-6, "note": prose}
           for k in ("C1", "C2", "C3")}
    cfg["R_net1_step"] = {"workload": prose, "ms_per_step": 0.98, "ms_per_step_eager": 1.5}
    for k in ("gcn_c4", "gat_c4", "C4_bf16_storage"):
        cfg[k] = {"workload": prose, "ms_per_step": 8.4, "roofline": {"gat_fwd_aggregate": {"kernel": prose, "frac": 0.84}}}
    cfg["C5_1gpu"] = {"workload": prose, "ms_per_step": 125.3, "w8_virtual": {"parity": {"against": prose * 3, "by_tensor": {f"t{i}": 1e-6 for i in range(40)}}}}
    cfg["C4_w8_virtual"] = vw
    return {
        "metric": "edges/sec per GNN layer (fwd+bwd)", "value": 3011703939.1234, "unit": "edges/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 6.640756500564748, "ms_per_step_repeats": [6.6, 6.59], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C4 " + prose[:300], "parallelism": "single GPU", "hip_graph_replay": False, "setup_steps": 40,
                   "csr_build_s": 0.1, "csr_sorted_columns": True, "fallback": None, "autotune": None, "communicators": None},
        "parity_max_err": 3.3e-7, "parity": {"parity_max_err": 3.3e-7, "by_tensor": {"out": 1e-7, "dX": 2e-7, "dW": 3.3e-7, "db": 1e-7}, "against": prose},
        "roofline": {"bound": "hbm", "achieved": 8840.0, "achieved_traffic": 6480.0, "peak": 8000.0, "unit": "GB/s", "frac_algorithmic": 1.105,
                     "frac_traffic": 0.81, "frac": 0.81, "fwd_launch_ms": 2.36, "bwd_launch_ms": 2.80, "frac_basis": prose, "traffic": 16646937064.9, "traffic_source": prose, "kernel": prose,
                     "algorithmic_bytes_per_launch": 22612000000, "avg_launch_ms": 2.5588, "launches_timed": 40,
                     "control_uniform": {"workload": prose, "avg_launch_ms": 3.56, "frac_algorithmic": 0.79, "frac_traffic": 0.79, "traffic_source": prose}},
        "exchange": None, "aggregation_only": {"edges_per_s": 7.8e9, "ms_per_step": 5.1, "note": prose},
        "projection": {"bound": "mfma", "kernels": prose, "achieved_f32_equivalent": 180.0, "achieved": 1080.0, "peak": 2500.0, "frac": 0.43,
                       "frac_note": prose, "per_gemm_ms": {"fwd": 0.73, "bwd_data": 0.73, "bwd_weight": 2.2}, "note": prose},
        "configs": cfg,
        "cpu_baseline": {"value": 1.97e6, "unit": "edges/s", "cores": 32, "kind": "port", "sample": prose, "ran": "full C4 configuration",
                         "nan_field": float("nan")},
        "wall_s": 95.0,
    }


def test_the_final_line_is_compact_strict_json_with_the_contract_blocks():
    sys.path.insert(0, ROOT)
    import bench
    rec = _fat_record()
    assert len(json.dumps(rec)) > 20000                       # what round 4 printed
    line = bench.compact_line(rec)
    assert "\n" not in line and len(line) < 4096 < bench.LINE_CAP, len(line)
    assert "NaN" not in line and "Infinity" not in line
    got = json.loads(line)
    assert json.loads(json.dumps(got)) == got                 # round-trips
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity_max_err", "configs"):
        assert k in got, k
    assert got["config"]["workload"].startswith("C4") and got["config"]["parallelism"] == "single GPU"
    r = got["roofline"]
    assert r["bound"] == "hbm" and r["frac"] == 0.81 and r["frac_algorithmic"] == 1.105 and r["peak"] == 8000.0
    assert abs(r["achieved"] / r["peak"] - r["frac_algorithmic"]) < 1e-3 and r["control_uniform"]["frac"] == 0.79
    # the object agrees with itself: the headline fraction follows from a rate and the peak that are both in it
    assert abs(r["achieved_traffic"] / r["peak"] - r["frac"]) < 1e-6 and "upper bound" in r["frac_basis"]
    assert r["fwd_launch_ms"] == 2.36 and r["bwd_launch_ms"] == 2.80
    assert set(got["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and got["cpu_baseline"]["cores"] == 32
    c = got["configs"]
    assert c["C2"] == 0.33 and c["C2_eager"] == 0.5 and c["gat_c4"] == 8.4 and c["C5_1gpu"] == 125.3
    assert c["w8_hubs_sage_ceiling"] == 5.72 and c["w2_hubs_sage_ceiling"] == 1.79 and abs(c["w8_hubs_sage_rank_ms_max"] - 1.156) < 1e-3
    # an N > 1 record: exchange + autotune + fallback survive, still compact
    rec.update(n_gpus=8, exchange={"exposed_ms_per_step": 0.4, "by_collective_ms_per_step": {f"fwd_ag{i}": 0.05 for i in range(8)}, "note": "x"})
    rec["config"].update(autotune={"ms_per_step": {"default": 1.4, "one communicator": 1.5}, "chosen": "default"}, fallback="y" * 1000,
                         communicators=2)
    line = bench.compact_line(rec)
    got = json.loads(line)
    assert len(line) < 6000 and got["config"]["autotune"]["chosen"] == "default" and len(got["config"]["fallback"]) <= 200
    assert got["exchange"]["exposed_ms_per_step"] == 0.4


def test_roofline_rates_agree_with_the_fractions_the_line_prints():
    """bench.roofline_rates is what main() fills the roofline object from: achieved / peak = frac_algorithmic and
    achieved_traffic / peak = frac_traffic (= frac), with or without a PMC figure"""
    sys.path.insert(0, ROOT)
    import bench
    ach, ach_t, fa, ft = bench.roofline_rates(22_612_000_000, 16_650_000_000.0, 2.583)
    assert abs(ach / bench.HBM_PEAK_GBS - fa) < 1e-12 and abs(ach_t / bench.HBM_PEAK_GBS - ft) < 1e-12
    assert abs(fa - 1.0943) < 1e-3 and abs(ft - 0.8057) < 1e-3
    ach, ach_t, fa, ft = bench.roofline_rates(22_612_000_000, None, 2.583)
    assert ach_t is None and ft is None and fa > 1.0
    assert bench.roofline_rates(1, 1, 0.0) == (0.0, None, 0.0, None)


def test_a_failed_first_attempt_restarts_every_rank_as_a_fresh_worker(tmp_path):
    """N > 1: the launched ranks are supervisors that never touch the GPU.  Rank 1's worker fails its pre-flight while rank 0's
    sits in a collective for ever: both supervisors end their own worker and start a second, fresh one (conservative schedule);
    rank 0's line says why.  (Fake workers: no GPU here.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(WORLD_SIZE="2", NPI_BENCH_FAKE_WORKER="fail0", NPI_BENCH_RDV=str(tmp_path / "rdv"), MASTER_PORT="29999")
    ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=120) for p in ps]
    assert [p.returncode for p in ps] == [0, 0], outs
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["attempt"] == 1 and "injected pre-flight failure" in line["fallback"]
    assert "starting a fresh worker" in outs[0][1] and "starting a fresh worker" in outs[1][1]
    names = sorted(os.listdir(tmp_path / "rdv"))
    assert {"fail_0_0", "fail_0_1", "err_0_1", "ok_1_0", "ok_1_1"} <= set(names), names       # (ok_0_0 only if rank 0 got that far in time)
