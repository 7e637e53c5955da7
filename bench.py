"""bench.py — RIPAgent.act() throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --gpus 8 ...                     (self-spawns 8 ranks under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): K=4 ensemble, algorithm "WCM" (as coded), N=128 candidate plans,
10 Adam steps, 200x200xC BEV, synthetic observations (SURVEY.md §8d distribution), random-init weights
(`oatomobile_amd.weights.synthetic_state_dict`), bf16 MobileNetV2 encoder (bf16 MFMA, fp32 accumulate) + fp32 flow /
search — the precision configs[2] names; `--encoder-dtype fp32` is the 1e-4-parity mode (reported as `fp32_parity`).

A *step* = the whole unit of SURVEY.md §8(d), R1..R11, over one batch of `--obs-batch` observations: the float32
observations start in PINNED HOST memory (what R1's host preparation leaves behind), their H2D copy runs on a copy
stream double-buffered under the previous step's kernels, then transform, K encoders + merger, plan search, candidate
selection with the [30,3] float64 interpolation (R11 on the device) and the copy of the [B,30,3] float64 plans back
to pinned host memory.  Every observation is one `RIPAgent.act()` call: value = obs_batch * steps * n_gpus / time.
`hbm_resident` is the same step on observations already in HBM without R11 (R2..R10: last round's headline).

Multi-GPU (one process per GPU, RCCL).  `--gpus N` with NO `--mode` (what the driver runs) times the
observation-parallel replicas for `value` AND, in the same invocation, the two compositions that need a collective, and
prints ONE line: `candidate_parallel` (north_star's "candidate throughput": N candidates per rank, one RCCL all-gather
of (best loss, plan) per step), `model_parallel` (when K % N == 0: K / N models per rank, one all-gather of the per-model
(posterior, d posterior / dy) block per Adam step, the K-aggregation of rip/agent.py:109-127 after the gather), each with
its plan difference against ONE rank holding everything, and `rccl` = {backend, ranks_seen, librccl_mapped}; the run
aborts unless every rank was seen by a collective.  An explicit `--mode` runs one composition alone:
  replay      observation-parallel replicas, no data-path collective                            -> "weak"
  candidates  every rank searches `--candidates` latent starts of the SAME observations (N_total = N * world), one
              all-gather of (best loss, plan) per step                                          -> "weak"
  models      `--models` K split over the ranks, gradient-mode model parallelism: per Adam step ONE all-gather of
              the (q_k, dq_k/dy) block (BASELINE configs[3]: --models 8 --candidates 512)       -> "strong"

The JSON line also carries
  roofline      — the dominant kernel (plan search): executed MFMA flops (exact instruction count of this launch
                  configuration, from the kernel's own per-step selection trace) / the dense-f16 MFMA peak (`frac`; the
                  kernel is instruction-issue bound: `bound` = "issue"), `matrix_pipe_busy` = the fraction of the launch
                  the matrix pipe is busy with this instruction mix; the §8(d) contract-formula figure is reported
                  separately (it prices work the kernel legitimately skips)
  cpu_baseline  — the CPU oracle (oracle/reference_cpu.py, "port") timed on this box's host cores
  online        — `agent(observation)` one call at a time, host numpy in -> host numpy out (H2D + D2H inside)
  hbm_resident  — the step without R1 / R11: observations resident in HBM, [B,4,2] plans copied back
  backend / world_size_seen — what torch.distributed reports (a mismatch with --gpus aborts the run)
  scoring_only  — rip/agent.py:109-127 scoring mode (no gradient search): encode + K x N scores + aggregation
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

# SURVEY.md §8(d) / BASELINE.md §5 work-per-unit figures
FLOW_MAC_PER_STEP = 14848  # GRU + head, one time step
ENC_ACT_ELEMS = 1466229 + 1465488  # layer-wise activation elements read + written per image (C = 2)
ENC_WEIGHT_ELEMS = 2370336 + 17056
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA (MI355X_MICROARCH.md; the headline 5 PF includes 2:1 sparsity)
PEAK_FP32_VECTOR_TFLOPS = 157.3  # packed fp32 FMA on the vector ALUs
# flop / matrix-pipe cycles per MFMA instruction (MI355X_MICROARCH.md: v_mfma_f32_16x16x32_f16 issues every 16 cycles
# per SIMD = the 2.5 PFLOP/s dense f16 rate, v_mfma_f32_16x16x4_f32 every 32 = the 157.3 TFLOP/s fp32 rate)
MFMA_F16 = (16 * 16 * 32 * 2, 16)
MFMA_F32 = (16 * 16 * 4 * 2, 32)
PEAK_CLOCK_HZ = 2.4e9
N_SIMD = 256 * 4
# observations (= act() calls) per step and GPU.  Round 6: 512 -> 2048.  Every kernel of the step has a fixed part per launch
# (persistent workgroups' prologues, launch tails, the ~25 launch boundaries): 127.7 k calls/s at 512 observations per step,
# 132.3 k at 1024, 134.5 k at 2048 (profiles/r6/batch_sweep_v1.txt); the line carries the 512-observation step beside it.
DEFAULT_OBS_BATCH = 2048
KERNEL_NAMES = {1: "search_kernel (wave-per-chain, fp32 VALU)", 
                3: "search_phase_kernel (fp32 MFMA, phase-sequential)", 4: "search_split_kernel (split-f16 MFMA, phase-sequential)"}


def search_plan(lib, h, B, N):
  """rip_search_plan: which kernel / workgroup shape the launch uses and its MFMA instruction counts per pass (the
  numbers live next to the kernels, flow_split.hip / flow_phase.hip; SQ_INSTS_MFMA in profiles/ is their check)."""
  import ctypes
  from oatomobile_amd import _lib
  out = (ctypes.c_int32 * 10)()
  _lib.check(lib.rip_search_plan(h, B, N, ctypes.cast(out, ctypes.c_void_p), 10))
  v = list(out)
  return {"kernel": v[0], "waves_per_workgroup": v[1], "pass": (v[2], v[3]), "adj_inv": (v[4], v[5]), "adj_f0": (v[6], v[7]),
          "prefix": (v[8], v[9])}


def synth_batch(rng, B, C, G=10):
  lidar = (rng.integers(0, 6, size=(B, 200, 200, C)) / 5.0) * (rng.random((B, 200, 200, C)) < 0.12)
  vec = np.c_[rng.normal(0, 3.0, size=(B, 3)), (rng.random((B, 1)) < 0.2), rng.integers(0, 4, size=(B, 1))]
  goal = np.cumsum(np.abs(rng.normal(size=(B, G, 2))) * 2.0, axis=1)
  return lidar.astype(np.float32), vec.astype(np.float32), goal.astype(np.float32)


def _measured(args):
  """profiles/measured.json: HBM-side bytes per launch from the round's rocprofv3 PMC passes (FETCH_SIZE x 2 — the
  gfx950 correction of MI355X_MICROARCH.md — + WRITE_SIZE, separate passes), written by tools/prof_round.sh next to
  the CSV summaries it condenses.  Used only when it was taken at this launch configuration (observations scale the
  bytes linearly); otherwise the line says so instead of quoting a stale constant."""
  try:
    with open(os.path.join(ROOT, "profiles", "measured.json")) as f:
      m = json.load(f)
    c = m["config"]
    same = (c["models"] == args.models and c["candidates"] == args.candidates and c["search_steps"] == args.search_steps and
            c["channels"] == args.channels and c["encoder_dtype"] == args.encoder_dtype and c["algorithm"] == args.algorithm)
    if not same:
      return None
    m["scale"] = args.obs_batch / float(c["obs_batch"])
    return m
  except (OSError, KeyError, ValueError):
    return None


def _encoder_roofline(enc_ms, layerwise_bytes, B, K, C, enc_dtype, measured=None):
  """The encoder stage against its three ceilings.  Work per (model, observation) from the architecture (arch.py):
  pointwise convs 68.63 M MAC on the matrix cores, depthwise + stem (4.64 + 0.72 C) M fp32 FMA on the vector ALUs;
  HBM bytes MEASURED (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE over all encoder kernels of one step, read from
  profiles/measured.json and scaled by the observation count)."""
  t = enc_ms * 1e-3
  pw_flops = 2.0 * 68.627e6 * B * K
  valu_fma = (4.641e6 + 0.720e6 * C) * B * K
  line = {"ms_per_step": enc_ms,
          "mfma_TFLOPs": pw_flops / t / 1e12, "mfma_frac": pw_flops / t / 1e12 / (PEAK_BF16_TFLOPS if enc_dtype == "bf16" else PEAK_FP32_TFLOPS),
          "valu_TFMAs": valu_fma / t / 1e12, "valu_frac": 2.0 * valu_fma / t / 1e12 / PEAK_FP32_VECTOR_TFLOPS,
          "layerwise_GBps": layerwise_bytes / t / 1e9,
          "note": "transform + stem + MobileNetV2 features (%s) + classifier + merger.  `layerwise_GBps` = SURVEY §8d "
                  "bytes_pre + bytes_enc (every layer's input and output through HBM) / time: what the UNFUSED network "
                  "would have to move, not what the fused kernels move." % enc_dtype}
  if measured is not None and measured.get("encoder", {}).get("traffic_bytes"):
    meas = measured["encoder"]["traffic_bytes"] * measured["scale"]
    line.update({"measured_hbm_bytes": meas, "measured_GBps": meas / t / 1e9, "frac_hbm": meas / t / 1e9 / PEAK_HBM_GBS,
                 "measured_source": measured.get("source")})
    line["note"] += ("  `measured_*`: HBM bytes from the PMC passes named in `measured_source` (the fused blocks keep the "
                     "expanded tensors in LDS, so this is far below the layer-wise count).")
  else:
    line["measured_hbm_bytes"] = None
  return line


def _encoder_kernel_times(lib, h_obj, lidar, B, K, C, enc, reps=5):
  """The encoder block by block from the event timeline (a regression shows up in the driver's run
  without rocprof): `rip_encode_tap_k` with a NULL destination runs the launch sequence of the SAME (B, K) selection up to
  conv layer i and stops; the difference of two consecutive stops is the time of the kernel(s) between them, whose
  names come from the handle's kernel log.  The tail (classifier + merger) is the full encode minus the last stop."""
  from oatomobile_amd import _lib, arch, transform_visual
  h = h_obj.raw
  vis = transform_visual(lidar, channels_last=True)
  L = len(arch.conv_layers(C))
  h_obj.set_option(_lib.OPT_KERNEL_LOG, 1)
  st = torch.cuda.current_stream()

  def timed_us(fn):
    fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
      a.record(st)
      fn()
      b.record(st)
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3

  out, prev_t, prev_n = [], 0.0, 0
  try:
    for i in range(L):
      rc = lib.rip_encode_tap_k(h, _lib.ptr(vis), B, 0, K, _lib.ENC_DTYPES[enc], i, None, 0, _lib.current_stream())
      if rc == _lib.RIP_EINVAL:
        continue  # interior to a fused block
      _lib.check(rc)
      lines = h_obj.kernel_log()
      log = [l.split(" ")[0] for l in lines]
      t = timed_us(lambda: _lib.check(lib.rip_encode_tap_k(h, _lib.ptr(vis), B, 0, K, _lib.ENC_DTYPES[enc], i, None, 0, _lib.current_stream())))
      out.append({"through_layer": i, "kernels": log[prev_n:], "us": round(t - prev_t, 1),
                  "launch": [" ".join(l.split(" ")[1:]) for l in lines[prev_n:]]})
      prev_t, prev_n = t, len(log)
    vec = torch.zeros(B, 5, device=vis.device)
    z = torch.empty(K, B, 64, device=vis.device)
    enc_fn = lambda: _lib.check(lib.rip_encode(h, _lib.ptr(vis), _lib.ptr(vec), B, 0, K, _lib.ENC_DTYPES[enc], _lib.ptr(z), None, _lib.current_stream()))
    enc_fn()
    tail_kernels = [l.split(" ")[0] for l in h_obj.kernel_log()][prev_n:]
    # the tail = whole encode minus the stop after the last conv layer, both timed back to back with the log switched off
    # (the stops above are tens of milliseconds old by now, and a 40 us difference of two 1.4 ms timings is sensitive to it)
    h_obj.set_option(_lib.OPT_KERNEL_LOG, 0)
    last = L - 1
    t_last = timed_us(lambda: _lib.check(lib.rip_encode_tap_k(h, _lib.ptr(vis), B, 0, K, _lib.ENC_DTYPES[enc], last, None, 0, _lib.current_stream())))
    full = timed_us(enc_fn)
    out.append({"through_layer": "tail", "kernels": tail_kernels, "us": round(full - t_last, 1), "whole_encode_us": round(full, 1)})
  finally:
    h_obj.set_option(_lib.OPT_KERNEL_LOG, 0)
  return out


# ------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle) — a subprocess with a hard timeout
# ------------------------------------------------------------------------------------------------------------
def _cpu_baseline_worker(argv):
  """Fresh process, no GPU context: the oracle's whole act() on the host cores, three variants."""
  K, N, C, algo, nsteps, seconds, threads = int(argv[0]), int(argv[1]), int(argv[2]), argv[3], int(argv[4]), float(argv[5]), int(argv[6])
  from oatomobile_amd import weights
  from oracle import reference_cpu as O
  seeds = [100 + k for k in range(K)]
  models = [O.OracleImitativeModel.from_numpy_state_dict(weights.synthetic_state_dict(s, C), C) for s in seeds]
  x0 = np.random.default_rng(0).standard_normal((N, 4, 2)).astype(np.float32)
  x0[0] = 0.0
  x0 = torch.from_numpy(x0)
  lidar, vec, goal = synth_batch(np.random.default_rng(2), 1, C)
  goal3 = np.c_[goal[0], np.zeros((goal.shape[1], 1), np.float32)]

  def run(nthreads, as_written, budget):
    torch.set_num_threads(nthreads)

    def one():
      O.rip_call(models, lidar[0], vec[0, :3], vec[0, 3], vec[0, 4], goal3, x0=x0, algorithm=algo, num_steps=nsteps,
                 as_written=as_written)

    for _ in range(warm):
      one()
    t0 = time.perf_counter()
    n = 0
    while True:
      one()
      n += 1
      dt = time.perf_counter() - t0
      if n >= calls or dt > budget:
        break
    return {"n": n, "dt": dt, "threads": nthreads}

  # BASELINE.md §4 protocol on the variant the speed-up is quoted against (fair, one thread: the fastest setting of
  # this batch-1 workload): 50 warm-up calls, 200 timed calls; the other variants are bounded samples
  warm, calls = 50, 200
  fair1 = run(1, False, 6.0 * seconds)
  warm, calls = 3, 200
  out = {"fair_1thread": fair1, "fair": run(threads, False, seconds / 2.0), "as_written": run(threads, True, seconds / 2.0)}
  print(json.dumps(out))


def cpu_baseline(args):
  """Bounded sample of the same workload on the CPU oracle ("port")."""
  from oatomobile_amd.replay import effective_cpus
  avail = effective_cpus()  # affinity mask capped by the cgroup CPU quota (the bench container: 16 of 256 threads)
  threads = avail  # all host cores this process may use; one thread is timed beside it
  cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(args.models), str(args.candidates),
         str(args.channels), args.algorithm, str(args.search_steps), str(args.cpu_seconds), str(threads)]
  env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
    env.pop(k, None)
  base = {"unit": "calls/s", "cores": threads, "kind": "port"}
  try:
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=max(240.0, 20 * args.cpu_seconds), env=env, cwd=ROOT)
    r = json.loads(out.stdout.strip().splitlines()[-1])
  except Exception as e:  # timeout / crash: report, never hang
    base.update({"value": None, "sample": "cpu baseline failed: %r" % (e,)})
    return base
  f = r["fair"]
  f1 = r["fair_1thread"]
  if f1["n"] / f1["dt"] > f["n"] / f["dt"]:  # the oracle's ops are small: one thread can beat the OpenMP team
    f = f1
  base.update({
      "value": f["n"] / f["dt"],
      "cores": f["threads"],
      "sample": "%d sequential act() calls after 50 warm-ups (K=%d, N=%d, %d Adam steps; encoders under no_grad = the "
                "'fair' variant) of oracle/reference_cpu.py (PyTorch CPU), %d thread(s) of %d available host cores (the "
                "faster of 1 and %d threads; os.cpu_count() = %d), %.1f s" %
                (f["n"], args.models, args.candidates, args.search_steps, f["threads"], avail, threads,
                 os.cpu_count() or 0, f["dt"]),
      "variants": {
          "fair_%dthreads" % threads: r["fair"]["n"] / r["fair"]["dt"],
          "fair_1thread": r["fair_1thread"]["n"] / r["fair_1thread"]["dt"],
          "as_written_%dthreads" % threads: r["as_written"]["n"] / r["as_written"]["dt"],
          "note": "as_written = autograd graph kept through the K encoders and back-propagated every Adam step, like "
                  "rip/agent.py:93,129 (same outputs); calls/s",
      },
  })
  return base


# ------------------------------------------------------------------------------------------------------------
def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _self_spawn(args):
  """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
  n = torch.cuda.device_count()
  if n < args.gpus and os.environ.get("RIP_BENCH_SHARE_GPU") != "1":
    raise SystemExit("bench.py: --gpus %d requested but only %d GPU(s) are visible" % (args.gpus, n))
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
  raise SystemExit(subprocess.call(cmd, env=env))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps; value = the median one")
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--obs-batch", type=int, default=DEFAULT_OBS_BATCH, help="observations (= act() calls) per step per GPU")
  ap.add_argument("--encoder-dtype", default="bf16", choices=["bf16", "fp32"],
                  help="MobileNetV2 encoder arithmetic; BASELINE configs[2] names bf16 encoder + fp32 flow")
  ap.add_argument("--models", type=int, default=4)
  ap.add_argument("--candidates", type=int, default=128)
  ap.add_argument("--channels", type=int, default=2, help="BEV channels (reference sensor: 2; BASELINE.json text: 4)")
  ap.add_argument("--algorithm", default="WCM")
  ap.add_argument("--search-steps", type=int, default=10)
  ap.add_argument("--mode", default=None, choices=["replay", "candidates", "models"],
                  help="default: replay; with --gpus N > 1 and no --mode the candidate- / model-parallel compositions run too")
  ap.add_argument("--cpu-seconds", type=float, default=5.0)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-extras", action="store_true", help="skip the secondary lines (online, pcie, scoring, fp32, C=4)")
  ap.add_argument("--online-calls", type=int, default=1000)
  if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-worker":
    return _cpu_baseline_worker(sys.argv[2:])
  args = ap.parse_args()
  explicit_mode = args.mode is not None
  if args.mode is None:
    args.mode = "replay"

  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    return _self_spawn(args)
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus:
    raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
  assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (the product has no CPU path)"
  # development / test hook: RIP_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and RIP_BENCH_BACKEND=gloo replaces RCCL
  # (which refuses two ranks on one device), so the world > 1 code path can be exercised on a one-GPU box
  # (tests/test_gpu_parity.py::test_bench_two_ranks_share_one_gpu); the numbers of such a run mean nothing.
  dev_index = 0 if os.environ.get("RIP_BENCH_SHARE_GPU") == "1" else local_rank
  backend = os.environ.get("RIP_BENCH_BACKEND", "nccl")
  torch.cuda.set_device(dev_index)
  dev = torch.device("cuda", dev_index)
  dist = None
  # a process group exists when there is more than one rank, and also for ONE rank of the --mode candidates / models
  # compositions (RIP_DIST_ALWAYS_COLLECTIVE=1 then sends the one-rank group through RCCL: the collective code path
  # of oatomobile_amd/distributed.py executes on a one-GPU box)
  if world > 1 or args.mode != "replay":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
      os.environ.setdefault("MASTER_PORT", str(_free_port()))
      os.environ.setdefault("RIP_DIST_ALWAYS_COLLECTIVE", "1")
    if backend == "nccl":
      dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    else:
      dist.init_process_group(backend=backend, rank=rank, world_size=world)
  # what torch.distributed itself reports goes into every line; a silent fallback or a rank mismatch aborts here
  backend_seen = dist.get_backend() if dist is not None else "none (single process, no process group)"
  world_seen = dist.get_world_size() if dist is not None else 1
  if world_seen != args.gpus or (dist is not None and backend_seen != backend):
    raise SystemExit("bench.py: torch.distributed reports backend %r / world size %d, expected %r / %d" %
                     (backend_seen, world_seen, backend, args.gpus))
  args.backend_seen, args.world_seen = backend_seen, world_seen

  import __graft_entry__ as entry
  if rank == 0:
    entry.build()
  if dist is not None:
    dist.barrier()
  from oatomobile_amd import ImitativeModel, RIPAgent, _lib

  def sync_all():
    if dist is not None and world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  def timed(step_fn, steps, warmup, events=None):
    """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks."""
    for i in range(warmup):
      step_fn(i, None)
    sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
      step_fn(i, events[i] if events is not None else None)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None and world > 1:
      t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      elapsed = float(t.item())
    return elapsed

  K, N, B, C, S = args.models, args.candidates, args.obs_batch, args.channels, args.search_steps
  lib = _lib.load()
  algo = _lib.ALGORITHMS[args.algorithm]
  seeds = [100 + k for k in range(K)]

  if args.mode != "replay":
    line = _bench_parallel_mode(args, args.mode, rank, world, dev, dist, timed, args.steps, args.warmup)
    if rank == 0:
      print(json.dumps(line))
    dist.barrier()
    dist.destroy_process_group()
    return

  models = [ImitativeModel.synthetic(s, in_channels=C, max_batch=1) for s in seeds]
  agent = RIPAgent(None, algorithm=args.algorithm, models=models, num_candidates=N, num_steps=S, max_batch=B, seed=0,
                   device=dev, encoder_dtype=args.encoder_dtype)
  enc_dtype = _lib.ENC_DTYPES[args.encoder_dtype]
  h = agent._handle.raw

  # distinct observation batches per rank and per step parity: float32 in PINNED HOST memory (the state R1's host
  # preparation, rip/agent.py:59-75, leaves an observation in); two device slots, a copy stream
  rng = np.random.default_rng(1000 + rank)
  host_batches = [synth_batch(rng, B, C) for _ in range(2)]
  pinned = [tuple(torch.from_numpy(a).pin_memory() for a in hb) for hb in host_batches]
  slots = [tuple(torch.empty_like(t, device=dev) for t in pinned[0]) for _ in range(2)]
  x0 = agent._x0(B)
  z = torch.empty(K, B, 64, device=dev)
  plan = torch.empty(B, 4, 2, device=dev)
  plan30 = torch.empty(B, 30, 3, device=dev, dtype=torch.float64)
  loss = torch.empty(B, N, device=dev)
  plan_host = torch.empty(B, 4, 2).pin_memory()
  plan30_host = [torch.empty(B, 30, 3, dtype=torch.float64).pin_memory() for _ in range(2)]
  G = pinned[0][2].shape[1]
  stream = torch.cuda.current_stream(dev)
  copy_stream = torch.cuda.Stream(device=dev)
  ready = [torch.cuda.Event() for _ in range(2)]
  freed = [torch.cuda.Event() for _ in range(2)]

  def upload(i):
    """H2D of step i's observations into slot i & 1, on the copy stream, once the slot's previous user is done."""
    j = i & 1
    with torch.cuda.stream(copy_stream):
      copy_stream.wait_event(freed[j])
      for d, src in zip(slots[j], pinned[j]):
        d.copy_(src, non_blocking=True)
      ready[j].record(copy_stream)

  def encode_search(lidar, vec, goal, ev=None, handle=h, zz=z, enc=enc_dtype, interp=None):
    s = _lib.current_stream(dev)
    if ev is not None:
      ev[0].record(stream)
    _lib.check(lib.rip_encode_raw(handle, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), B, 0, K, enc, _lib.ptr(zz), s))
    if ev is not None:
      ev[1].record(stream)
    _lib.check(lib.rip_search(handle, _lib.ptr(zz), _lib.ptr(goal), _lib.ptr(x0), B, N, G, algo, S, 0.1, 1.0,
                              _lib.ptr(plan), None, _lib.ptr(loss), None, None, None, None, s))
    if ev is not None:
      ev[2].record(stream)
    if interp is None:
      plan_host.copy_(plan, non_blocking=True)  # the reference's D2H (rip/agent.py:139)
    else:  # R11 (rip/agent.py:141-151) on the device, then the D2H of what act() hands to the controller
      _lib.check(lib.rip_interpolate_plans(_lib.ptr(plan), B, _lib.ptr(plan30, torch.float64), s))
      interp.copy_(plan30, non_blocking=True)

  tick = [0]  # steps since prime(): warm-up and timed steps are ONE pipeline (slot parity carries over)

  def make_unit_step(enc):
    def step(_, ev):
      i = tick[0]
      tick[0] += 1
      j = i & 1
      upload(i + 1)  # the next step's observations travel under this step's kernels
      stream.wait_event(ready[j])
      encode_search(*slots[j], ev=ev, enc=enc, interp=plan30_host[j])
      freed[j].record(stream)
    return step

  def prime():
    torch.cuda.synchronize()
    tick[0] = 0
    for j in range(2):
      freed[j].record(stream)
    upload(0)

  prime()
  # `--repeats` timed regions of EXACTLY `--steps` steps each (barrier + synchronize on both sides of every one; the warm-up
  # runs once, in front of the first): `value` is the MEDIAN region — one 20-step region is 0.08 s, and single samples of
  # it spread by ~0.6 % — min / max are reported beside it.
  unit_step = make_unit_step(enc_dtype)
  regions = []
  for r in range(max(1, args.repeats)):
    ev_r = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    regions.append((timed(unit_step, args.steps, args.warmup if r == 0 else 0, ev_r), ev_r))
  order = sorted(range(len(regions)), key=lambda i: regions[i][0])
  elapsed, events = regions[order[len(order) // 2]]
  region_rates = [B * args.steps * world / el for el, _ in regions]
  assert torch.isfinite(plan).all() and bool(np.isfinite(plan30_host[0].numpy()).all())
  # the [30,3] float64 plans on the host are the reference's R11 of the device's [4,2] plans, bit for bit
  from oatomobile_amd.agents import interpolate_plan
  last = (tick[0] - 1) & 1
  r11_ok = all(np.array_equal(plan30_host[last].numpy()[i], interpolate_plan(plan.cpu().numpy()[i])) for i in (0, B // 2, B - 1))
  enc_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
  search_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
  calls = B * args.steps * world
  value = calls / elapsed
  h2d_bytes = sum(t.numel() * 4 for t in pinned[0])

  # the same step without R1 / R11 (observations resident in HBM, [B,4,2] plans back): last round's `value`
  batches = [tuple(torch.from_numpy(a).to(dev) for a in hb) for hb in host_batches]
  hbm_steps = max(4, args.steps // 2)
  hbm_el = timed(lambda i, ev: encode_search(*batches[i & 1]), hbm_steps, 2)
  hbm_line = {"calls_per_s": B * hbm_steps * world / hbm_el, "ms_per_step": 1e3 * hbm_el / hbm_steps,
              "note": "R2..R10 on observations already in HBM, [B,4,2] fp32 plans copied back: no H2D, no R11 (round 2's "
                      "headline definition)"}

  # the same resident step at 512 observations (rounds 2-5 ran `value` at 512 observations per step)
  hbm512 = None
  if B > 512 and not args.no_extras:  # (an extra: under rocprofv3 its launches would be averaged into the headline's kernels)
    b5 = 512
    small = [tuple(t[:b5] for t in bt) for bt in batches]
    x0_5, z5 = agent._x0(b5), torch.empty(K, b5, 64, device=dev)
    plan5, loss5 = torch.empty(b5, 4, 2, device=dev), torch.empty(b5, N, device=dev)

    def step512(i, ev):
      lidar5, vec5, goal5 = small[i & 1]
      s5 = _lib.current_stream(dev)
      _lib.check(lib.rip_encode_raw(h, _lib.ptr(lidar5), 1, 200, 200, _lib.ptr(vec5), b5, 0, K, enc_dtype, _lib.ptr(z5), s5))
      _lib.check(lib.rip_search(h, _lib.ptr(z5), _lib.ptr(goal5), _lib.ptr(x0_5), b5, N, G, algo, S, 0.1, 1.0, _lib.ptr(plan5), None,
                                _lib.ptr(loss5), None, None, None, None, s5))
      plan_host[:b5].copy_(plan5, non_blocking=True)

    n5 = max(8, args.steps)
    el5 = timed(step512, n5, 3)
    hbm512 = {"calls_per_s": b5 * n5 * world / el5, "ms_per_step": 1e3 * el5 / n5, "obs_per_step": b5,
              "note": "`hbm_resident` at 512 observations per step (the step size of rounds 2-5): every kernel's fixed part per "
                      "launch weighs four times as much"}

  extras = {}
  plan_info = search_plan(lib, h, B, N)
  use_mfma = plan_info["kernel"] in (3, 4)  # a phase-sequential MFMA kernel (flow_split.hip / flow_phase.hip)
  exec_flops = pipe_cycles = useful_flops = None
  if rank == 0 and use_mfma:
    # exact executed-MFMA count of this launch: per 16-candidate block and Adam step the kernel runs F_0, K-1 inverses,
    # the adjoint of F_0, and the adjoint of inverse k iff model k is the running arg-best (WCM / BCM) of some candidate
    # of the block when it is reached — read from the kernel's own per-step posterior trace
    tp = torch.empty(S, K, B, N, device=dev)
    lidar, vec, goal = batches[0]
    _lib.check(lib.rip_search(h, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(x0), B, N, G, algo, S, 0.1, 1.0, None, None,
                              _lib.ptr(loss), None, _lib.ptr(tp), None, None, _lib.current_stream(dev)))
    torch.cuda.synchronize()
    blocks16 = B * N / 16.0
    if args.algorithm == "MA" or K == 1:
      adj_passes = float(S * (K - 1) * blocks16)
      laid_out = adj_passes
    else:
      sign = 1.0 if args.algorithm == "WCM" else -1.0
      run = (sign * tp).cummax(dim=1).values  # running best over models 0..k
      take = (sign * tp[:, 1:]) > run[:, :-1]  # [S,K-1,B,N]: strictly better than every earlier model
      laid_out = float(take.view(S, K - 1, B, N // 16, 16).any(-1).sum().item())  # with the candidates as laid out
      # what the timed (trace-free) launch really executes: the kernel's own counter (rip_search_stats)
      import ctypes
      cnt = ctypes.c_uint64(0)
      _lib.check(lib.rip_search_stats(h, None, 1))
      _lib.check(lib.rip_search(h, _lib.ptr(z), _lib.ptr(goal), _lib.ptr(x0), B, N, G, algo, S, 0.1, 1.0, _lib.ptr(plan), None,
                                _lib.ptr(loss), None, None, None, None, _lib.current_stream(dev)))
      _lib.check(lib.rip_search_stats(h, ctypes.byref(cnt), 1))
      adj_passes = float(cnt.value)
    extras["adjoint_inverse_passes_as_laid_out"] = laid_out
    def count(info):  # (f16, fp32) MFMA instructions of the whole launch
      return tuple(blocks16 * ((S + 1) * info["pass"][i] + S * info["adj_f0"][i] + S * (K - 1) * info["pass"][i]) +
                   adj_passes * info["adj_inv"][i] + B * K * info["prefix"][i] for i in (0, 1))
    n16, n32 = count(plan_info)
    exec_flops = n16 * MFMA_F16[0] + n32 * MFMA_F32[0]
    pipe_cycles = n16 * MFMA_F16[1] + n32 * MFMA_F32[1]  # matrix-pipe issue cycles, summed over the SIMDs
    extras["mfma_instructions"] = {"f16_16x16x32": n16, "f32_16x16x4": n32}
    # `useful`: a two-term product (three f16 MFMAs) counted once; the round-6 forward step's input / bias k-steps (12 f16
    # MFMAs of which 16 rows x 3 K x 16 candidates are arithmetic) and its W2 block (3 MFMAs for 4 rows x 32 K x 16) at the
    # flops they stand for, not at the 16 x 32 x 16 the instruction executes
    useful_flops = n16 / 3.0 * MFMA_F16[0] + n32 * MFMA_F32[0]
    if plan_info["kernel"] == 4 and plan_info["pass"][1] == 0:
      steps_fwd = blocks16 * 3 * ((S + 1) + S * (K - 1)) + B * K  # forward / inverse GRU + head steps of the launch (+ prefix steps)
      useful_flops -= steps_fwd * (15 / 3.0 * MFMA_F16[0] - (12 * 2 * 16 * 3 * 16 + 2 * 4 * 32 * 16))
    extras["waves_per_workgroup"] = plan_info["waves_per_workgroup"]
    if plan_info["kernel"] == 4:
      # the same launch on the fp32-MFMA kernel (flow_phase.hip), for comparison with the fp32 floor of round 2
      lib.rip_set_option(h, 0, 3)
      ref = search_plan(lib, h, B, N)
      lib.rip_set_option(h, 0, 0)
      extras["fp32_kernel_flops_same_launch"] = count(ref)[1] * MFMA_F32[0]
    extras["adjoint_inverse_passes_executed"] = adj_passes
    extras["adjoint_inverse_passes_possible"] = float(S * (K - 1) * blocks16)

  # ---------------- secondary lines (rank 0, N = 1 only; untimed by the driver's contract) ----------------
  online = scoring = fp32_line = c4_line = train_line = replay_line = pipelined = strict_line = None
  if rank == 0 and world == 1 and not args.no_extras:
    def extra(fn, *a):
      """A secondary line must never cost the headline: a failure becomes {"error": ...} in its place."""
      try:
        return fn(*a)
      except Exception as exc:  # noqa: BLE001 -- reported in the line, the timed result above is already complete
        torch.cuda.synchronize()
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}

    def fp32_step():
      e32 = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(5)]
      prime()
      el = timed(make_unit_step(0), 5, 2, e32)
      # plan-level effect of the bf16 encoder on THIS batch: the fp32-encoder plans of the last step's slot against the
      # bf16-encoder plans of the same observations
      j = (tick[0] - 1) & 1
      p32 = plan.clone()
      encode_search(*slots[j], enc=enc_dtype)
      dev_m = (plan - p32).norm(dim=(1, 2)) / 2.0  # RMS-like: metres per waypoint coordinate pair
      scale = float(p32.abs().max())
      return {"calls_per_s": B * 5 / el, "ms_per_step": 1e3 * el / 5,
              "encoder_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in e32])),
              "bf16_vs_fp32_plan_deviation_m": {"max_abs": float((plan - p32).abs().max()), "mean_abs": float((plan - p32).abs().mean()),
                                                "plan_scale_m": scale},
              "encoder_kernels": "fp32 activations; stem + features.1 / features.2-7 / features.8-17 / features.18 + pool as fused kernels "
                                 "whose pointwise convolutions run on the binary16 matrix pipe with two-term operands (fp32-grade: z "
                                 "against the fp32 oracle 1.65e-5, the true-fp32 layer-wise kernels 1.68e-5), depthwise and stem "
                                 "fp32 on the vector unit; classifier and merger true fp32",
              "note": "the same whole-unit step with the fp32 encoder: the mode in which z, plans and log-probs hold the "
                      "1e-4 parity contract.  `bf16_vs_fp32_plan_deviation_m`: what the bf16 encoder of `value` (the "
                      "precision BASELINE configs[2] names) moves the winning plans by on this batch"}

    def two_handles():
      """Two handles on two streams, each running encode -> search on its own resident batch: the encoder's launch
      tails and the search fill each other's gaps.  A deployment option for batch replay, NOT `value` (whose kernels
      run one after the other so that the event times and the rocprofv3 averages describe single kernels)."""
      import ctypes
      other = RIPAgent(None, algorithm=args.algorithm, models=models, num_candidates=N, num_steps=S, max_batch=B, seed=0,
                       device=dev, encoder_dtype=args.encoder_dtype)
      z2, plan2, loss2 = torch.empty_like(z), torch.empty_like(plan), torch.empty_like(loss)
      s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

      def act(hh, batch, zz, pp, ll, st):
        sp = ctypes.c_void_p(st.cuda_stream)
        lidar_, vec_, goal_ = batch
        _lib.check(lib.rip_encode_raw(hh, _lib.ptr(lidar_), 1, lidar_.shape[1], lidar_.shape[2], _lib.ptr(vec_), B, 0, K,
                                      _lib.ENC_DTYPES[args.encoder_dtype], _lib.ptr(zz), sp))
        _lib.check(lib.rip_search(hh, _lib.ptr(zz), _lib.ptr(goal_), _lib.ptr(x0), B, N, G, algo, S, 0.1, 1.0, _lib.ptr(pp), None,
                                  _lib.ptr(ll), None, None, None, None, sp))

      def pair(i, ev):
        act(h, batches[0], z, plan, loss, s1)
        act(other._handle.raw, batches[1], z2, plan2, loss2, s2)

      n = max(4, args.steps // 4)
      el = timed(pair, n, 2)
      return {"calls_per_s": 2 * B * n / el, "ms_per_pair_of_steps": 1e3 * el / n,
              "note": "two handles x two streams, observations resident in HBM, [B,4,2] plans left on the device; compare "
                      "with hbm_resident.calls_per_s (one handle, one stream)"}

    def strict_fp32_search():
      """The headline step with the plan search on TRUE fp32 operands (flow_phase.hip: v_mfma_f32_16x16x4_f32, option 3):
      what BASELINE configs[2]'s "fp32 flow" costs when it is read strictly — the price of the two-term binary16
      operands `value` runs on is `value` / this."""
      e3 = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(5)]
      _lib.check(lib.rip_set_option(h, _lib.OPT_SEARCH_KERNEL, _lib.SEARCH_KERNELS["phase"]))
      try:
        prime()
        el = timed(make_unit_step(enc_dtype), 5, 2, e3)
        kern = search_plan(lib, h, B, N)["kernel"]
      finally:
        _lib.check(lib.rip_set_option(h, _lib.OPT_SEARCH_KERNEL, _lib.SEARCH_KERNELS["auto"]))
      return {"calls_per_s": B * 5 / el, "ms_per_step": 1e3 * el / 5,
              "ms_per_launch": float(np.mean([e[1].elapsed_time(e[2]) for e in e3])), "kernel": KERNEL_NAMES.get(kern, str(kern)),
              "note": "the same whole-unit step (R1..R11, %s encoder) with the search on fp32 MFMA operands (24 significant "
                      "bits, the reference's arithmetic) instead of two-term binary16 (22 bits)" % args.encoder_dtype}

    online = extra(_bench_online, args, models, dev, host_batches[0])
    strict_line = extra(strict_fp32_search)
    pipelined = extra(two_handles)
    scoring = extra(_bench_scoring, args, lib, h, batches, z, enc_dtype, dev, timed, K, N, B, G, algo)
    if args.encoder_dtype == "bf16":
      fp32_line = extra(fp32_step)
    if C == 2:
      c4_line = extra(_bench_c4, args, dev, timed, seeds)
    train_line = extra(_bench_train, args, dev, timed)
    replay_line = extra(_bench_replay, args, agent, dev, min(B, 1024), C)  # (10 000 observations: ten batches of 1024, the last one ragged)
  # ---------------- N > 1 without --mode: the compositions that need a collective, same invocation ----------------
  par_lines, rccl = {}, None
  if world > 1 and not explicit_mode:
    rccl = _rccl_seen(dist, dev, world)
    if rccl["ranks_seen"] != world or rccl["all_reduce_of_ones"] != float(world):
      raise SystemExit("bench.py: the collectives saw %d ranks (all-reduce of ones = %g), expected %d" %
                       (rccl["ranks_seen"], rccl["all_reduce_of_ones"], world))
    del agent
    torch.cuda.empty_cache()
    sub_steps, sub_warm = max(4, args.steps // 2), 2
    for mode in ["candidates"] + (["models"] if K % world == 0 else []):
      try:
        line = _bench_parallel_mode(args, mode, rank, world, dev, dist, timed, sub_steps, sub_warm)
      except Exception as exc:  # noqa: BLE001 -- every rank raises or none does (same code path); the headline stands
        torch.cuda.synchronize()
        line = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
      if rank == 0:
        if "error" in line:
          par_lines[mode] = line
        else:
          par_lines[mode] = {"calls_per_s": line["value"], "candidate_plans_per_s": line["candidate_plans_per_s"],
                             "ms_per_step": line["ms_per_step"], "collectives_per_step": line["collectives_per_step"],
                             "max_abs_plan_diff_vs_single_gpu": line["check"]["max_abs_plan_diff_vs_single_gpu"],
                             "exchange": line.get("exchange"),
                             "scaling": line["scaling"], "candidates_total": line["config"]["candidates"],
                             "parallelism": line["config"]["parallelism"], "steps": sub_steps, "warmup": sub_warm}
      torch.cuda.empty_cache()
  if rank == 0:
    measured = _measured(args)
    flow_flops = 2.0 * 3.0 * (1 + K) * 4 * FLOW_MAC_PER_STEP * N * S * B  # SURVEY §8(d) flops_flow(grad)
    sb = 2 if args.encoder_dtype == "bf16" else 4  # bytes per encoder element (SURVEY §8d `s`)
    enc_bytes = B * (200 * 200 * C * 4 + 100 * 100 * C * 4) + K * (B * (ENC_ACT_ELEMS + 10000 * (C - 2)) * sb + ENC_WEIGHT_ELEMS * sb)
    bytes_act = enc_bytes / B + K * 64 * 4 + N * 8 * 4 * 2 + 32  # §8(d) bytes_pre + bytes_enc (weights amortised) + bytes_flow
    blocks16 = B * N / 16.0
    exec_tf = exec_flops / (search_ms * 1e-3) / 1e12 if exec_flops else None
    # the matrix pipe running THIS instruction mix back to back: flops / (issue cycles / (SIMDs x 2.4 GHz))
    mix_peak = exec_flops / (pipe_cycles / (N_SIMD * PEAK_CLOCK_HZ)) / 1e12 if exec_flops else PEAK_FP32_TFLOPS
    roof = {
        "kernel": "%s: per 16-candidate wave F_0 + %d inverses + adjoints + Adam, operands in LDS, %d steps in one launch" %
                  (KERNEL_NAMES[plan_info["kernel"]], K - 1, S),
        "bound": "issue",
        "achieved": exec_tf,
        "peak": PEAK_BF16_TFLOPS,
        "unit": "TFLOP/s",
        "frac": exec_tf / PEAK_BF16_TFLOPS if exec_tf else None,
        "frac_of_dense_f16_peak": exec_tf / PEAK_BF16_TFLOPS if exec_tf else None,
        # the split's overhead is not an achievement: every hi/lo product is three MFMAs — counted ONCE here
        "useful_frac": useful_flops / (search_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS if useful_flops else None,
        "matrix_pipe_busy": exec_tf / mix_peak if exec_tf else None,
        "fp32_equivalent_tflops": extras.pop("fp32_kernel_flops_same_launch") / (search_ms * 1e-3) / 1e12
                                  if "fp32_kernel_flops_same_launch" in extras else None,
        # L2 <-> fabric bytes per launch: the adjoint tape, written once and read back once per 16-candidate block,
        # model and Adam step (F_0: 3 steps, inverses: 2 steps, the third stays in registers; a step is 12 or 16 rows
        # of 1 KiB -- r, z, gh_n, hprev; n is recomputed -- + 256 B of ReLU mask); rocprofv3 FETCH_SIZE (x2 gfx950
        # correction) + WRITE_SIZE (profiles/)
        "traffic": measured["search"]["traffic_bytes"] * measured["scale"] if (measured and use_mfma and measured.get("search")) else None,
        "traffic_source": measured.get("source") if measured else "no PMC profile at this launch configuration in profiles/measured.json",
        "ms_per_launch": search_ms,
        "contract_tflops": flow_flops / (search_ms * 1e-3) / 1e12,
        "contract_over_peak": flow_flops / (search_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
        "whole_act_hbm_frac": bytes_act * value / world / 1e9 / PEAK_HBM_GBS,
        "whole_act_GBps": bytes_act * value / world / 1e9,
        # what the memory-side counters saw per step (encoder + search launches: FETCH_SIZE x 2 + WRITE_SIZE, separate
        # --pmc passes) / the step time / 8 TB/s — beside `whole_act_hbm_frac`, which prices the LAYER-WISE bytes of
        # SURVEY 8(d) that the fused kernels never move
        "measured_hbm_frac": ((measured["encoder"]["traffic_bytes"] + measured["search"]["traffic_bytes"]) * measured["scale"] /
                              (elapsed / args.steps) / 1e9 / PEAK_HBM_GBS)
                             if (measured and measured.get("search") and measured.get("encoder", {}).get("traffic_bytes")) else None,
        "note": "`achieved`: MFMA flops this launch EXECUTES (v_mfma_f32_16x16x32_f16 = 16384 flop, v_mfma_f32_16x16x4_f32 "
                "= 2048; instruction counts per pass from rip_search_plan, passes from the kernel's per-step selection "
                "trace: an inverse's adjoint only runs for blocks where some candidate selects that model) over the mean "
                "launch time from HIP events on the launch stream; checked against rocprofv3 SQ_INSTS_MFMA (profiles/).  "
                "`peak` = the dense f16 MFMA peak (2.5 PFLOP/s), `frac` = achieved / peak (every flop priced at the f16 "
                "rate); `useful_frac` counts a two-term product ONCE (its three f16 MFMAs as one): the split's overhead is not "
                "priced as achievement.  `bound` = \"issue\": the kernel runs one wave per SIMD at 256 + 209 registers and is limited by the "
                "instructions it issues between matrix instructions (gate math, operand splits), not by the matrix pipe: "
                "`matrix_pipe_busy` = the time the pipe needs for this launch's instruction mix back to back (f16 MFMAs one "
                "per 16 cycles and SIMD, fp32 MFMAs one per 32) / the launch time.  `fp32_equivalent_tflops`: the flops the fp32-MFMA kernel of round 2 executes for the same "
                "launch / this launch's time — above 157.3 means past that kernel's floor.  `contract_*`: "
                "SURVEY.md §8(d) flops_flow(grad) = 2*3*(1+K)*T*14848*N*steps per act x obs_batch / launch time -- NOT "
                "a utilisation: the formula prices the candidate-independent step-0 prefix, model 0's inverse "
                "(inverse_0(F_0(x)) == x) and K adjoints per candidate, none of which the algorithm needs, so it can "
                "exceed the peak.  `whole_act_*`: §8(d) algorithmic bytes per act (weights amortised over the batch) x "
                "calls/s vs 8 TB/s.  `traffic`: the adjoint tape.",
        "encoder": _encoder_roofline(enc_ms, enc_bytes, B, K, C, args.encoder_dtype, measured),
    }
    if not args.no_extras:
      try:
        roof["encoder"]["kernels"] = _encoder_kernel_times(lib, agent._handle, batches[0][0], B, K, C, args.encoder_dtype)
        roof["encoder"]["kernels_note"] = ("event-timed, median of 5: the launch sequence stopped after each block (rip_encode_tap_k, no "
                                           "copy); `us` = difference of consecutive stops; transform_kernel is not in it")
      except Exception as exc:  # noqa: BLE001
        roof["encoder"]["kernels"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    roof.update(extras)
    out = {
        "metric": "RIPAgent.act() calls/sec (K=%d, %d plans, 200x200 BEV)" % (K, N),
        "value": value,
        "unit": "calls/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "repeats": {"n": len(region_rates), "value_is": "median of n timed regions of `steps` steps each",
                    "calls_per_s": [round(v, 1) for v in region_rates], "min": min(region_rates), "max": max(region_rates)},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("bf16 encoder (bf16 storage, fp32 accumulate) + " if args.encoder_dtype == "bf16" else "fp32 encoder + ") +
                 "flow/search: fp32 accumulate, GRU/head contractions as two-term binary16 (22-bit) operands on f16 MFMA "
                 "(the waypoint / bias terms as three-term, 33-bit operands)",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: RIPAgent K=%d %s, N=%d candidate plans, %d Adam steps, 200x200x%d BEV, "
                               "%s encoder + fp32 flow" % (K, args.algorithm, N, S, C, args.encoder_dtype),
                   "obs_per_step_per_gpu": B, "models": K, "candidates": N, "bev_channels": C,
                   "parallelism": "observation-parallel replicas x%d (no data-path collective)" % world},
        "roofline": roof,
        "unit_of_work": {"covers": "R1..R11: pinned-host float32 observations -> H2D (copy stream, double-buffered) -> transform -> "
                           "K encoders + merger -> plan search -> selection + [30,3] float64 interpolation -> D2H to pinned host",
                         "h2d_MB_per_step": h2d_bytes / 1e6, "d2h_KB_per_step": B * 30 * 3 * 8 / 1e3,
                         "r11_bit_identical_to_reference_arithmetic": bool(r11_ok)},
        "hbm_resident": hbm_line,
        "hbm_resident_512": hbm512,
        "two_handles_two_streams": pipelined,
        "backend": args.backend_seen,
        "world_size_seen": args.world_seen,
        "candidate_parallel": par_lines.get("candidates"),
        "model_parallel": par_lines.get("models") if world > 1 else None,
        "rccl": rccl,
        "online": online,
        "scoring_only": scoring,
        "fp32_parity": fp32_line,
        "strict_fp32_search": strict_line,
        "bev_c4": c4_line,
        "train_step": train_line,
        "replay": replay_line,
    }
    if world == 1 and not args.no_cpu_baseline:
      try:
        out["cpu_baseline"] = cpu_baseline(args)
      except Exception as exc:  # noqa: BLE001
        out["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    print(json.dumps(out))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------
def _bench_online(args, models, dev, host_batch):
  """The reference's usage pattern: one `agent(observation)` per tick — host numpy dict in, [30,3] numpy out, with
  the H2D of the 320 KB BEV, every kernel and the D2H + sync inside each call."""
  from oatomobile_amd import RIPAgent
  lidar, vec, goal = host_batch
  obs = [dict(lidar=lidar[i], velocity=vec[i, :3], is_at_traffic_light=vec[i, 3], traffic_light_state=vec[i, 4],
              goal=np.c_[goal[i], np.zeros((goal.shape[1], 1), np.float32)]) for i in range(8)]
  res = {}
  # the agent's own default encoder is fp32 (the parity mode); at one observation it is also the faster one (fewer,
  # simpler launches: 283 vs 302 us), so the headline of this line is the class used as it comes, the bench step's
  # encoder (`--encoder-dtype`, bf16) is reported next to it
  for name, graph, enc in (("graph", True, "fp32"), ("eager", False, "fp32"), ("graph_" + args.encoder_dtype, True, args.encoder_dtype)):
    if name in res:
      continue
    a1 = RIPAgent(None, algorithm=args.algorithm, models=models, num_candidates=args.candidates,
                  num_steps=args.search_steps, max_batch=1, seed=0, device=dev, encoder_dtype=enc, graph=graph)
    n = args.online_calls if graph else max(20, args.online_calls // 5)
    for i in range(50):  # 50 warm-ups, >= 1000 timed calls
      a1(dict(obs[i % 8]))
    lat = []
    t0 = time.perf_counter()
    for i in range(n):
      t1 = time.perf_counter()
      a1(dict(obs[i % 8]))
      lat.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    lat = np.sort(np.asarray(lat)) * 1e6
    captured = any(st["graph"] is not None for st in a1._online.values())
    res[name] = {"calls_per_s": n / dt, "p50_us": float(lat[len(lat) // 2]), "p99_us": float(lat[min(len(lat) - 1, int(0.99 * len(lat)))]),
                 "mean_us": float(lat.mean()), "hipgraph": captured, "encoder": enc}
  out = {"calls_per_s": res["graph"]["calls_per_s"], "latency_us": res["graph"]["mean_us"], "p50_us": res["graph"]["p50_us"],
         "p99_us": res["graph"]["p99_us"], "hipgraph_captured": res["graph"]["hipgraph"], "encoder": "fp32 (RIPAgent default)",
         "eager": res["eager"],
         "pattern": "agent(observation): B=1 sequential, host numpy observation in, [30,3] numpy plan out; pinned "
                    "staging + H2D + transform + K encoders + search + D2H + stream sync inside every call"}
  if "graph_" + args.encoder_dtype in res:
    out["with_" + args.encoder_dtype + "_encoder"] = res["graph_" + args.encoder_dtype]
  return out


def _bench_replay(args, agent, dev, B, C):
  """BASELINE configs[4] shape on one GPU: cached `.npz` datums (the reference's on-disk schema) decoded by worker
  processes into shared-memory batches (oatomobile_amd/replay.py:DatumBatches), uploaded and planned batch by batch.
  Steady state: the clock starts when the first batch has been planned (the workers' start-up is seconds)."""
  import shutil, tempfile
  from oatomobile_amd import replay
  nfiles, repeats = 256, 16
  workers = max(1, min(48, replay.effective_cpus() - 2))
  tmp = tempfile.mkdtemp(prefix="rip_replay_")
  try:
    ep = replay.Episode(tmp, "ep")
    rng = np.random.default_rng(77)
    lidar, vec, goal = synth_batch(rng, nfiles, C)
    for i in range(nfiles):
      fut = np.cumsum(np.abs(rng.normal(size=(80, 3))) * 0.4, axis=0).astype(np.float32)
      ep.append(lidar=lidar[i], velocity=vec[i, :3], is_at_traffic_light=vec[i, 3], traffic_light_state=vec[i, 4],
                player_future=fut)
    files = ep.files() * repeats
    t_inline0 = time.perf_counter()
    for f in files[:64]:
      replay.load_datum(f)
    inline_rate = 64 / (time.perf_counter() - t_inline0)
    n_done, t0 = 0, None
    for lid, v, g in replay.DatumBatches(files, B, workers=workers, prefetch=2, channels=C):
      plan = agent.plan_batch(lid.to(dev, non_blocking=True), v.to(dev, non_blocking=True), g.to(dev, non_blocking=True))
      plan.cpu()
      if t0 is None:
        t0 = time.perf_counter()
      else:
        n_done += lid.shape[0]
    el = time.perf_counter() - t0
    # BASELINE configs[4] at its size from the packed cache (replay.pack_cache: uint8 codes + float table, expanded in
    # the transform kernel): the 256 datums packed once, tiled to 10 000 observations, replayed with the [30,3] plans
    # the packing rate is taken over the 4096-file list (the 256 files 16 times: one worker process per CPU has
    # seconds of work, as with a real episode directory); the first 256 rows are the 256 distinct datums
    t0 = time.perf_counter()
    small = replay.pack_cache(files, os.path.join(tmp, "cache4k"))
    pack_rate = len(files) / (time.perf_counter() - t0)
    n10k = 10000
    big_dir = os.path.join(tmp, "cache10k")
    os.makedirs(big_dir)
    idx = np.arange(n10k) % nfiles
    codes = np.lib.format.open_memmap(os.path.join(big_dir, "codes.npy"), mode="w+", dtype=np.uint8,
                                      shape=(n10k,) + small.codes.shape[1:])
    for i0 in range(0, n10k, nfiles):
      codes[i0:i0 + nfiles] = small.codes[:min(nfiles, n10k - i0)]
    codes.flush()
    del codes
    np.save(os.path.join(big_dir, "lut.npy"), small.lut)
    np.save(os.path.join(big_dir, "vec.npy"), small.vec[idx])
    np.save(os.path.join(big_dir, "goal.npy"), small.goal[idx])
    cache = replay.PackedCache(big_dir)
    replay.replay_cache(agent, cache, B, interpolate=True, end=2 * B)  # warm-up (page cache, pinned buffers)
    t0 = time.perf_counter()
    plans1 = replay.replay_cache(agent, cache, B, interpolate=True)
    one_el = time.perf_counter() - t0
    # two handles on two streams (even / odd batches): one batch's encoder beside the other batch's search — legitimate
    # for batch replay; `value` stays single-stream
    replay.replay_cache(agent, cache, B, interpolate=True, end=4 * B, streams=2)  # warm-up (the twin handle, its scratch)
    t0 = time.perf_counter()
    plans = replay.replay_cache(agent, cache, B, interpolate=True, streams=2)
    cache_el = time.perf_counter() - t0
    same = bool(np.array_equal(plans[:nfiles], plans[nfiles:2 * nfiles]) and np.array_equal(plans, plans1))  # the tiling repeats: so must the plans
    cache_line = {"observations_per_s": n10k / cache_el, "streams": 2, "single_stream_observations_per_s": n10k / one_el,
                  "observations": n10k, "batch": B, "seconds": cache_el,
                  "bytes_per_observation": int(np.prod(cache.codes.shape[1:])) + 4 * (5 + 2 * cache.goal.shape[1]),
                  "pack_datums_per_s": pack_rate, "pack_datums": len(files), "pack_processes": replay.effective_cpus(),
                  "repeats_consistent": same,
                  "note": "10 000 observations from the packed cache -> pinned staging -> H2D -> rip_encode_raw_u8 + search "
                          "+ R11 -> [30,3] float64 plans on the host; one process, no decode workers.  `observations_per_s`: even / "
                          "odd batches on two handles x two streams (replay_cache(streams=2), bit-identical plans: "
                          "`repeats_consistent`); `single_stream_observations_per_s`: one handle, one stream"}
    return {"observations_per_s": n_done / el, "decode_processes": workers, "files": len(files), "batch": B,
            "inline_decode_per_s": inline_rate, "packed_cache": cache_line,
            "note": "np.load (zipfile + zlib) of compressed 200x200x%d datums in %d worker processes -> shared-memory "
                    "batch -> H2D -> act(); one process decodes %.0f datums/s, which is what bounds a single-process "
                    "replay; this process may use %d CPUs (affinity / cgroup quota) of the host's %d" %
                    (C, workers, inline_rate, replay.effective_cpus(), os.cpu_count() or 0)}
  finally:
    shutil.rmtree(tmp, ignore_errors=True)


def _bench_scoring(args, lib, h, batches, z, enc_dtype, dev, timed, K, N, B, G, algo):
  """Scoring mode (rip/agent.py:109-127 without the gradient search): K x N log-posteriors of N given plans per
  observation, ensemble aggregation and arg-best."""
  from oatomobile_amd import _lib
  rng = np.random.default_rng(5)
  plans = torch.from_numpy(np.cumsum(np.abs(rng.normal(size=(B, N, 4, 2))) * 2, axis=2).astype(np.float32)).to(dev)
  Sm = torch.empty(K, B, N, device=dev)
  lo = torch.empty(B, N, device=dev)
  best = torch.empty(B, device=dev, dtype=torch.int32)
  best_host = torch.empty(B, dtype=torch.int32).pin_memory()

  def step(i, ev):
    lidar, vec, goal = batches[i & 1]
    s = _lib.current_stream(dev)
    _lib.check(lib.rip_encode_raw(h, _lib.ptr(lidar), 1, 200, 200, _lib.ptr(vec), B, 0, K, enc_dtype, _lib.ptr(z), s))
    _lib.check(lib.rip_score(h, 0, K, _lib.ptr(z), _lib.ptr(plans), _lib.ptr(goal), B, N, G, 1.0, _lib.ptr(Sm), s))
    _lib.check(lib.rip_aggregate_scores(_lib.ptr(Sm), K, B, N, algo, _lib.ptr(lo), _lib.ptr(best, torch.int32), s))
    best_host.copy_(best, non_blocking=True)

  steps = max(4, args.steps // 2)
  el = timed(step, steps, 2)
  return {"calls_per_s": B * steps / el, "ms_per_step": 1e3 * el / steps,
          "note": "encode + [K=%d, N=%d] score matrix (rip_score) + ensemble aggregation / arg-best per observation" % (K, N)}


def _bench_c4(args, dev, timed, seeds):
  """BASELINE.json quotes a 200x200x4 BEV (the reference sensor has 2 channels): the same step at C = 4."""
  from oatomobile_amd import ImitativeModel, RIPAgent
  B = min(args.obs_batch, 256)
  models = [ImitativeModel.synthetic(s, in_channels=4, max_batch=1) for s in seeds]
  agent = RIPAgent(None, algorithm=args.algorithm, models=models, num_candidates=args.candidates,
                   num_steps=args.search_steps, max_batch=B, seed=0, device=dev, encoder_dtype=args.encoder_dtype)
  lidar, vec, goal = (torch.from_numpy(a).to(dev) for a in synth_batch(np.random.default_rng(44), B, 4))

  def step(i, ev):
    agent.plan_batch(lidar, vec, goal)

  el = timed(step, 5, 2)
  return {"calls_per_s": B * 5 / el, "ms_per_step": 1e3 * el / 5, "obs_per_step": B, "bev_channels": 4}


def _bench_train(args, dev, timed):
  """the DIM training step (dim/train.py:175-213: train-mode forward, backward, Adam) on one model,
  fp32, batch 128 — the first correct path of that row, reported so that its cost is on record."""
  from oatomobile_amd import DIMTrainer, ImitativeModel, transform_visual
  B = 128
  model = ImitativeModel.synthetic(7, in_channels=args.channels, max_batch=1).to(dev)
  tr = DIMTrainer(model, lr=1e-3, max_batch=B, device=dev)
  lidar, vec, goal = synth_batch(np.random.default_rng(77), B, args.channels)
  batch = dict(visual_features=transform_visual(torch.from_numpy(lidar).to(dev), channels_last=True),
               velocity=torch.from_numpy(vec[:, :3].copy()).to(dev), is_at_traffic_light=torch.from_numpy(vec[:, 3:4].copy()).to(dev),
               traffic_light_state=torch.from_numpy(vec[:, 4:5].copy()).to(dev),
               player_future=torch.from_numpy(np.cumsum(np.abs(np.random.default_rng(78).normal(size=(B, 4, 2))), axis=1).astype(np.float32)).to(dev))
  losses = []

  def step(i, ev):
    losses.append(tr.train_step(batch))

  el = timed(step, 6, 2)
  l = [float(x) for x in losses]
  tr.close()
  return {"observations_per_s": B * 6 / el, "ms_per_step": 1e3 * el / 6, "batch": B, "loss_first_last": [l[0], l[-1]],
          "note": "one ImitativeModel, fp32, BatchNorm batch statistics + dropout, loss.backward, Adam(1e-3); same batch "
                  "every step (the loss falls)"}


def _rccl_seen(dist, dev, world):
  """What the collectives themselves saw: every rank contributes its index to an all-gather and a one to an all-reduce
  on DEVICE tensors (RCCL when the backend is nccl); `librccl_mapped` from /proc/self/maps."""
  where = dev if dist.get_backend() == "nccl" else torch.device("cpu")  # (the one-GPU test hook runs gloo: host tensors)
  mine = torch.tensor([dist.get_rank()], device=where, dtype=torch.int64)
  got = [torch.empty_like(mine) for _ in range(world)]
  dist.all_gather(got, mine)
  ones = torch.ones(1, device=where)
  dist.all_reduce(ones)
  torch.cuda.synchronize()
  with open("/proc/self/maps") as f:
    mapped = any("librccl" in line for line in f)
  return {"backend": dist.get_backend(), "ranks_seen": len({int(t.item()) for t in got}), "all_reduce_of_ones": float(ones.item()),
          "librccl_mapped": mapped}


def _bench_parallel_mode(args, mode, rank, world, dev, dist, timed, steps, warmup):
  """`candidates` | `models`: the compositions of oatomobile_amd/distributed.py over RCCL.  Every rank
  calls this; rank 0 gets the result line (a dict), the others None."""
  from oatomobile_amd import ImitativeModel
  from oatomobile_amd import distributed as D
  K, N, B, C, S = args.models, args.candidates, args.obs_batch, args.channels, args.search_steps
  rng = np.random.default_rng(1000)  # the SAME observations on every rank in both modes
  lidar, vec, goal = (torch.from_numpy(a).to(dev) for a in synth_batch(rng, B, C))
  if mode == "candidates":
    models = [ImitativeModel.synthetic(100 + k, in_channels=C, max_batch=1) for k in range(K)]
    cp = D.CandidateParallelRIP(models, N * world, algorithm=args.algorithm, num_steps=S, seed=0, max_batch=B, device=dev,
                                encoder_dtype=args.encoder_dtype)
    out_host = torch.empty(B, 4, 2).pin_memory()

    def step(i, ev):
      plan, idx, best = cp(lidar, vec, goal)
      out_host.copy_(plan, non_blocking=True)

    par = "candidate-parallel: %d candidates per rank x %d ranks (N_total = %d), one all-gather of (best loss, plan) per step" % (N, world, N * world)
    scaling, n_total, collectives = "weak", N * world, 1
  else:
    kb, ke = D.shard_range(K, rank, world)
    models = [ImitativeModel.synthetic(100 + k, in_channels=C, max_batch=1) for k in range(kb, ke)]
    flow0 = None if kb == 0 else ImitativeModel.synthetic(100, in_channels=C, max_batch=1)
    mp = D.ModelParallelRIP(models, K, flow0=flow0, num_candidates=N, algorithm=args.algorithm, num_steps=S, seed=0,
                            max_batch=B, device=dev)
    out_host = torch.empty(B, 4, 2).pin_memory()

    def step(i, ev):
      plan, best, lb = mp(lidar, vec, goal)
      out_host.copy_(plan, non_blocking=True)

    par = ("model-parallel (gradient mode): K = %d models over %d ranks, one all-gather of z_0 per call and one of the "
           "[K_local,B,N,9] (posterior, d posterior / dy) block per Adam step; the K-aggregation (rip/agent.py:109-127) "
           "runs on every rank after the gather" % (K, world))
    scaling, n_total, collectives = "strong", N, S + 1
  elapsed = timed(step, steps, warmup)
  # the distributed result against the world-1 composition of the same search on rank 0 (same candidate stream)
  if mode == "candidates":
    plan_d = cp(lidar, vec, goal)[0]
  else:
    plan_d = mp(lidar, vec, goal)[0]
  if rank != 0:
    return None
  if mode == "candidates":
    one = D.CandidateParallelRIP(models, N * world, algorithm=args.algorithm, num_steps=S, seed=0, max_batch=B, device=dev,
                                 encoder_dtype=args.encoder_dtype, rank=0, world=1)
    plan_1 = one(lidar, vec, goal)[0]
  else:
    full = [ImitativeModel.synthetic(100 + k, in_channels=C, max_batch=1) for k in range(K)]
    one = D.ModelParallelRIP(full, K, num_candidates=N, algorithm=args.algorithm, num_steps=S, seed=0, max_batch=B,
                             device=dev, rank=0, world=1)
    plan_1 = one(lidar, vec, goal)[0]
  check = {"max_abs_plan_diff_vs_single_gpu": float((plan_d - plan_1).abs().max().item()),
           "note": "plans of the %d-rank run vs ONE rank holding everything, same observations and latent starts" % world}

  def local_ms(fn, reps=max(3, steps // 2)):  # rank 0 alone, no collective inside: what the exchange adds is ms_per_step minus this
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
      fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps

  if mode == "candidates":
    compute_ms = local_ms(lambda: cp.local_search(lidar, vec, goal))
    exchange = {"rank_compute_ms_per_step": compute_ms, "exchange_ms_per_step": 1e3 * elapsed / steps - compute_ms,
                "note": "rank 0's own share of the candidates (encoders + search, no collective, no selection) timed alone; the "
                        "difference to ms_per_step is the all-gather of [B, 10] floats per rank, the arg-min over ranks and "
                        "waiting for the slowest rank"}
  else:
    exchange = {"single_rank_ms_per_step": local_ms(lambda: one(lidar, vec, goal)),
                "note": "ONE rank holding all K models (no collective) on the same observations: ms_per_step of the %d-rank run "
                        "above this divided by %d is what the %d all-gathers per call cost" % (world, world, collectives)}
  calls = B * steps
  return {
      "check": check,
      "metric": "RIPAgent.act() calls/sec (K=%d, %d plans, 200x200 BEV)" % (K, n_total),
      "value": calls / elapsed, "unit": "calls/s", "n_gpus": world, "steps": steps, "warmup": warmup,
      "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
      "dtype": "%s encoder + flow/search: fp32 accumulate, GRU/head contractions as two-term binary16 operands on f16 MFMA" % args.encoder_dtype,
      "data": "synthetic",
      "candidate_plans_per_s": calls / elapsed * n_total, "collectives_per_step": collectives, "exchange": exchange,
      "backend": args.backend_seen, "world_size_seen": args.world_seen,
      "config": {"workload": "RIPAgent K=%d %s, N=%d candidate plans, %d Adam steps, 200x200x%d BEV" % (K, args.algorithm, n_total, S, C),
                 "obs_per_step": B, "models": K, "candidates": n_total, "bev_channels": C, "mode": mode, "parallelism": par}}


if __name__ == "__main__":
  main()
