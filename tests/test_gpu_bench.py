"""bench.py's multi-rank path with the REAL library: two self-launched ranks on the one GPU of the box
(ZJ_BENCH_SAME_GPU=1: both use cuda:0, collectives on gloo), each decoding its own shard with libzjhip; the gathered
per-frame checksums must equal the oracle's golden ones (tests/golden/checksums_seed1234.json)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_self_launched_ranks_decode_their_shards_to_the_golden_checksums():
    res = _bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--min-untimed", "1", "--shard-frames", "32",
                 "--no-cpu-baseline", "--no-live-traffic", "--no-single-frame", env_extra={"ZJ_BENCH_SAME_GPU": "1"})
    assert res["n_gpus"] == 2 and res["frames_checksummed"] == 64 and res["config"]["frames_total"] == 64
    assert res["checksums_match_golden"] is True
    assert res["collective_backend"] == "gloo" and res["rccl_ranks"] == 0     # same-GPU plumbing run: no RCCL
    assert len(res["per_rank_ms"]) == 2 and all(v > 0 for v in res["per_rank_ms"])
    assert max(res["per_rank_ms"]) <= res["ms_per_step"] * 1.001                # the line's time is the MAX over ranks
    rf = res["roofline"]
    assert len(rf["per_rank_kernel_ms"]) == 2 and rf["kernel_ms"] == max(rf["per_rank_kernel_ms"])  # the slowest rank's
    assert rf["live_counters_dropped"] and rf["scattered_batch"]["same_checksums_as_contiguous"] is True


@pytest.mark.gpu
def test_two_ranks_carry_the_cpu_baseline_in_the_same_run():
    """north_star: the AVX2 figure "in the same run" at every N -- rank 0 times it after the gather while the other ranks
    sleep on the store (the driver's command shape, shortened)."""
    res = _bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--min-untimed", "1", "--shard-frames", "16",
                 "--no-live-traffic", "--no-single-frame", "--no-dense-control", env_extra={"ZJ_BENCH_SAME_GPU": "1"})
    cb = res["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 100 and cb["cores"] >= 1 and "4 threads" in cb["sample"]
    # north_star "core count stated": the thread count, the cgroup's CPU quota and the host's size, cores = what the run could occupy
    assert cb["threads"] >= cb["cores"] == min(cb["threads"], cb["cpu_quota"]) and cb["logical_cpus"] >= cb["cpu_quota"] >= 1
    assert str(cb["threads"]) in cb["by_threads"] and "4" in cb["by_threads"]
    assert res["checksums_match_golden"] is True and res["n_gpus"] == 2
    numa = res["roofline"]["per_rank_numa"]                                     # every rank says where its host side ran
    assert len(numa) == 2 and all(set(n) == {"device_node", "thread_node", "bound"} for n in numa)
    assert all(n["thread_node"] == n["device_node"] and n["bound"] for n in numa if n["device_node"] >= 0)


@pytest.mark.gpu
@pytest.mark.parametrize("r", range(8))
def test_virtual_rank_decodes_its_shard_of_configs4_to_the_golden_checksums(r):
    """--as-rank R/8: rank R's shard of the 1024-frame batch (global frames [128 R, 128 R + 128)), its frame indices and
    golden offsets, on the box's one GPU.  R = 0..7 together: every frame of configs[4] decoded on hardware."""
    res = _bench("--as-rank", f"{r}/8", "--steps", "8", "--warmup", "1", "--min-untimed", "1", "--no-cpu-baseline",
                 "--no-live-traffic", "--no-single-frame", "--no-e2e", "--no-dense-control")
    assert res["as_rank"] == {"rank": r, "of": 8, "global_frames": [128 * r, 128 * r + 128]}
    assert res["frames_checksummed"] == 128 and res["checksums_match_golden"] is True
    assert res["roofline"]["scattered_batch"]["same_checksums_as_contiguous"] is True
    assert res["n_gpus"] == 1 and "other_workloads" not in res


@pytest.mark.gpu
def test_a_hung_rank_ends_the_self_launched_run_quickly():
    """Rank 1 never reaches the rendezvous (test knob); rank 0 waits for it inside init_process_group.  The launcher must
    give up after --rank-timeout, say which ranks were alive, and exit non-zero -- in well under 30 s here."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update({"ZJ_BENCH_SAME_GPU": "1", "ZJ_BENCH_TEST_HANG_RANK": "1"})
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--shard-frames", "16",
                        "--rank-timeout", "8"], capture_output=True, timeout=300, env=env)
    assert r.returncode != 0 and time.monotonic() - t0 < 30
    assert b"still running" in r.stderr and not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_single_rank_line_has_the_contract_fields():
    res = _bench("--steps", "8", "--warmup", "1", "--min-untimed", "1", "--shard-frames", "16", "--no-cpu-baseline",
                 "--no-live-traffic")
    assert res["n_gpus"] == 1 and res["checksums_match_golden"] is True and res["frames_checksummed"] == 16
    rf = res["roofline"]
    assert rf["bound"] == "hbm" and rf["algorithmic_bytes_per_launch"] == 16 * 4096 * 4096 * 6
    assert rf["single_frame_launch"]["kernel_ms"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["kernel_launches_timed"] >= 100                       # whatever --steps says
    dc = rf["dense_control"]                                        # no sparsity shortcut: a dense input decodes at the same rate
    assert dc["kernel_ms"] > 0 and abs(dc["kernel_ms"] - rf["kernel_ms"]) < 0.1 * rf["kernel_ms"], (dc, rf["kernel_ms"])
    assert res["per_rank_ms"] == [res["ms_per_step"]] or abs(res["per_rank_ms"][0] - res["ms_per_step"]) < 0.01
    e = res["e2e_pinned"]
    assert "error" not in e, e
    assert e["megapixels_per_s"] > 1000 and e["last_frame_matches_golden"] is True
    assert e["pcie_probe"]["duplex_gbs_per_direction"] > 5 and 0 < e["frac_of_duplex_ceiling"] < 1.25, e
    assert rf["scattered_batch"] is None                              # a 16-frame shard has no 16 non-adjacent frames
    sbc = rf["same_box_copy"]                                         # the yardstick beside the pin rate: a plain copy on this box
    assert sbc and 2000 < sbc["gbs"] < 8000 and abs(sbc["kernel_vs_copy"] - rf["achieved"] / sbc["gbs"]) < 2e-3
    # configs[2] / configs[3] in the driver's line
    ow = res["other_workloads"]
    for name, bpp in (("444-rgb", 9.0), ("444-gray", 3.0), ("422-rgb", 7.0), ("440-rgb", 7.0)):
        assert "error" not in ow[name], ow[name]
        assert ow[name]["bytes_per_px"] == bpp and ow[name]["kernel_ms"] > 0 and ow[name]["matches_other_variant"] is True
        assert abs(ow[name]["frac"] - 16 * 4096 * 4096 * bpp / (ow[name]["kernel_ms"] * 1e-3) / 1e9 / 8000.0) < 2e-3
    # the reference's medium image size in its tight layout and with its rows at a multiple of 128 bytes (zj_frame_desc.out_pitch)
    for name in ("420-rgb-2500x1786", "420-rgb-2500x1786-pitch128"):
        assert "error" not in ow[name], ow[name]
        assert ow[name]["matches_other_variant"] is True and ow[name]["width"] == 2500 and ow[name]["kernel_ms"] > 0
    assert ow["420-rgb-2500x1786-pitch128"]["out_pitch"] == 7552 and ow["420-rgb-2500x1786-pitch128"]["rows_match_tight_layout"] is True
    rfiles = res["reference_files"]
    for name, prog in (("test-baseline.jpg", False), ("test-progressive.jpg", True)):
        assert "error" not in rfiles[name], rfiles[name]
        assert rfiles[name]["sha256_matches_golden"] is True and rfiles[name]["progressive"] is prog
        assert rfiles[name]["host_entropy_ms"] > 0 and rfiles[name]["gpu_pixels_ms"] > 0
        assert rfiles[name]["blocks"] == 97200 and rfiles[name]["planes"] == "pinned" and rfiles[name]["host_threads"] == 1
        assert rfiles[name]["decode_buffer_matches"] is True and rfiles[name]["decode_buffer_stages_apart_matches"] is True
        assert rfiles[name]["decode_buffer_ms"] > 0 and rfiles[name]["decode_buffer_ms_stages_apart"] > 0
        assert abs(rfiles[name]["ns_per_block"] - rfiles[name]["host_entropy_ms"] * 1e6 / 97200) < 0.2
        assert rfiles[name]["four_threads_match"] is True and rfiles[name]["host_entropy_ms_4_threads"] > 0
        # neither file is entered at four points: progressive scans never are, and the baseline file is 73 KB (the attempt
        # starts at 96 KB of scan; it is all but flat besides -- from_files' 4096 x 4096 file is the case this is for)
        assert rfiles[name]["mcus_decoded_in_parallel"] == 0
    # round 6: the baseline walker no longer loses to the ten-scan progressive file on the same thread (it did, 17.8 vs 3.6 ms,
    # while its blocks left through non-temporal stores: profiles/r06_feeder_ab.txt)
    assert rfiles["test-baseline.jpg"]["host_entropy_ms"] <= rfiles["test-progressive.jpg"]["host_entropy_ms"]
    # the reference's own benchmark (benches/decode.rs on benches/images/speed_bench*.jpg), the whole decode_buffer call
    rb = res["reference_bench"]
    assert "error" not in rb, rb
    for label, pub in (("speed_bench.jpg -> RGB", 62.246), ("speed_bench.jpg -> GRAYSCALE", 45.598), ("speed_bench_hv_subsampling.jpg -> RGB", 52.175)):
        e = rb[label]
        assert "error" not in e, e
        assert e["sha256_matches_golden"] is True and e["published_ms"] == pub and (e["width"], e["height"]) == (7680, 4320)
        assert 0 < e["steady_state_ms"] <= e["as_the_reference_does_ms"] and e["host_threads"] == 4
        assert e["mcus_decoded_in_parallel"] > 100000      # (no restart markers: the scan is entered at four points)
    ff = res["from_files"]
    assert "error" not in ff, ff
    for mode in ("cpu_entropy", "gpu_entropy", "cpu_entropy_4_threads", "cpu_entropy_16_threads"):
        assert ff[mode]["planes"] == "pinned" and ff[mode]["host_threads"] == {"cpu_entropy_4_threads": 4, "cpu_entropy_16_threads": 16}.get(mode, 1) and ff[mode]["blocks"] == 393216
    assert ff["cpu_entropy_4_threads"]["mcus_decoded_in_parallel"] > 60000 and ff["cpu_entropy"]["mcus_decoded_in_parallel"] == 0
    assert ff["cpu_entropy_4_threads"]["host_ms_per_file"] < ff["cpu_entropy"]["host_ms_per_file"]
    assert ff["cpu_entropy_16_threads"]["mcus_decoded_in_parallel"] > 60000 and ff["cpu_entropy_16_threads"]["host_threads"] == 16
    assert ff["cpu_entropy_16_threads"]["host_ms_per_file"] < ff["cpu_entropy"]["host_ms_per_file"]
    assert ff["cpu_entropy"]["host_ms_per_file"] <= 35.0, ff["cpu_entropy"]
    numa = rf["per_rank_numa"]
    assert len(numa) == 1 and (numa[0]["device_node"] < 0 or (numa[0]["bound"] and numa[0]["thread_node"] == numa[0]["device_node"]))


@pytest.mark.gpu
def test_two_ranks_under_torchrun_the_drivers_launch_shape():
    """The driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`: env:// rendezvous instead of the self-launcher's port file.  Two ranks on the
    box's one GPU (ZJ_BENCH_SAME_GPU=1, gloo): the line, the golden checksums, the CPU baseline in the same run (rank 0 times
    it while rank 1 is parked on torchrun's own store)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["ZJ_BENCH_SAME_GPU"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--min-untimed", "1", "--shard-frames", "16", "--no-live-traffic", "--no-single-frame", "--no-dense-control"],
                       capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["frames_checksummed"] == 32 and res["checksums_match_golden"] is True
    assert res["cpu_baseline"]["value"] > 100 and len(res["roofline"]["per_rank_kernel_ms"]) == 2
