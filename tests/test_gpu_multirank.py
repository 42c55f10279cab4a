"""The N > 1 code path on ONE GPU: two ranks (gloo rendezvous, both on device 0) push the HIP path -- not the oracle -- through
scatter_streams -> pv_process_batch_device -> gather_streams, and the gathered result equals the single-process result bit for bit
(streams are independent processors: /root/reference/src/phase-vocoder.js:49-50,71).  Also: plain `python bench.py --gpus 2` on a box with
one GPU degrades to one measured replica and says so."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import phaze_amd, signals as S
from phaze_amd import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
fft, hop, T, nch, nstreams = 1024, 256, 40, 2, int(sys.argv[3])
dev = torch.device("cuda", 0)
if rank == 0:
    x_all = torch.from_numpy(np.stack([np.stack([S.make_signal("tonal", c, T * hop, stream=s) for c in range(nch)]) for s in range(nstreams)]))
else:
    x_all = torch.empty((0, nch, T * hop), dtype=torch.float32)
x = shard.scatter_streams(x_all, nstreams, dist)                 # host tensors over gloo (RCCL carries device tensors the same way)
n = x.shape[0]
xd = x.reshape(n * nch, T * hop).to(dev).contiguous()
yd = torch.empty_like(xd)
pitch = torch.stack([torch.full((T,), 0.8 + 0.1 * (s % 7), dtype=torch.float32) for s in shard.stream_partition(nstreams, world, rank)]).to(dev) if n else None
if n:
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=n * nch, max_hops=1, device_id=0)
    pv.process_batch_device(xd.data_ptr(), yd.data_ptr(), n * nch, T, T * hop, pitch.data_ptr(), T, nch)
    pv.synchronize(); pv.close()
y_all = shard.gather_streams(yd.cpu().reshape(n, nch, T * hop), nstreams, dist)
if rank == 0:
    np.save(sys.argv[2], y_all.numpy())
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("nstreams", [4, 5])
def test_two_ranks_on_one_gpu_equal_one_process(tmp_path, nstreams):
    import phaze_amd
    import signals as S
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "gathered.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", str(29700 + nstreams), str(script), ROOT, str(out), str(nstreams)], check=True, env=env, timeout=600)
    got = np.load(out)
    fft, hop, T, nch = 1024, 256, 40, 2
    x = np.stack([np.stack([S.make_signal("tonal", c, T * hop, stream=s) for c in range(nch)]) for s in range(nstreams)])
    p = np.stack([np.full(T, 0.8 + 0.1 * (s % 7), np.float32) for s in range(nstreams)])
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nstreams * nch, max_hops=T)
    ref = pv.process_batch(x.reshape(nstreams * nch, T * hop), p, channels_per_stream=nch).reshape(nstreams, nch, T * hop)
    pv.close()
    assert np.array_equal(got, ref)


def test_bench_gpus_flag_degrades_explicitly_on_a_smaller_box():
    import torch
    ndev = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ndev + 1), "--steps", "2", "--warmup", "1", "--hops", "4096",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["n_gpus"] == ndev and j["requested_gpus"] == ndev + 1 and j["replicas_measured"] == ndev


def test_bench_time_shard_span_is_bit_exact():
    """bench.py --time-shard: a span deep inside ONE stream started from {input tail, acc = 0, cursor} R - 1 hops early equals the same hops of a
    handle that reached them by processing a long lead-in (rank 3 of 4 simulated on this GPU)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--time-shard", "--simulate-shard", "3/4", "--steps", "2", "--warmup", "1",
                        "--hops", "65536", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["span_starts_bit_exact_vs_processed_lead_in"] is True and j["config"]["rank0_span"] == [49152, 65536] and j["config"]["halo_hops"] == 3


def test_rccl_branch_runs_on_hardware_at_world_size_one():
    """The driver's multi-GPU launch form with ONE rank: torch.distributed.run -> bench.py -> init_process_group("nccl") (= RCCL on ROCm) ->
    barrier, all_reduce(MAX) of the timing on a device tensor, and scatter_streams / gather_streams of the step's input (--scatter-gather).
    A one-GPU box cannot show scaling, but every collective of the N > 1 path executes through RCCL here."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29741",
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--hops", "4096", "--no-extras", "--no-cpu-baseline",
                        "--scatter-gather"], capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["n_gpus"] == 1 and j["dist_backend"] == "nccl"
    assert np.isfinite(j["scatter_gather_ms"]) and j["scatter_gather_ms"] > 0
    assert j["parity_rms_vs_oracle"] < 2e-7 and j["roofline"]["regions"] == 5


def test_two_bench_ranks_on_one_gpu_run_the_multi_rank_line_end_to_end():
    """What the driver launches on an 8-GPU node, with TWO ranks squeezed onto ONE device (HIP_VISIBLE_DEVICES=0, LOCAL_RANK % ndev) and the gloo backend
    (RCCL refuses two ranks on one device; `--dist-backend gloo` exists for this test only): rank-local seeds, barrier + MAX-over-ranks timing,
    aggregate_rate over both ranks, the RAGGED send / recv branch of scatter_streams / gather_streams (3 streams over 2 ranks), and the line's own
    statement that the two ranks shared one device (`replicas_measured` 1, no scaling claimed)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HIP_VISIBLE_DEVICES="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29743",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--hops", "8192", "--repeats", "2", "--no-extras", "--no-cpu-baseline",
                        "--dist-backend", "gloo", "--scatter-gather", "--sg-streams", "3"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                            # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["dist_backend"] == "gloo" and j["ranks"] == 2 and j["n_gpus"] == 1 and j["requested_gpus"] == 2 and j["replicas_measured"] == 1
    assert "no scaling curve" in j["note_gpus"]
    assert j["scatter_gather"]["branch"].startswith("send / recv") and j["scatter_gather"]["round_trip_intact"] is True and j["scatter_gather"]["rank0_streams"] == 2
    # whole-job value = frames of BOTH ranks / slowest rank's time (shard.aggregate_rate)
    assert abs(j["value"] - 2 * 8192 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-6
    assert j["parity_rms_vs_oracle"] < 2e-7 and j["scaling"] == "weak"
