#!/usr/bin/env python
"""Headline benchmark: mel-frames/sec of one Tacotron2 train step (forward + backward + TF-Adam) at
per-GPU batch 32 x (128 tokens, 800 mel frames), fp32, synthetic data (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W          (N > 1: spawns one rank per GPU itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
  `roofline`: the location-sensitive-attention step (the kernel north_star names).  In the persistent decoder launch it is a STAGE of
      every step, not a launch, so it is timed with device timestamps inside the kernel (s_memrealtime, 100 MHz) by the launch's
      profiling instantiation - a separate binary of the same source, run for one extra untimed step; mean over 256 workgroups x 801
      steps.  `frac` ends the stage where the context store is issued; `frac_incl_outbound_handoff` adds the consumer-side wait for that
      context (the flight time of the outbound hand-off), the stricter reading.  The launch-per-step attention kernel is measured beside
      it with HIP events around every launch (`roofline.launch_per_step`).  `traffic` comes from the committed rocprofv3 PMC pass named
      in `traffic_source`; `traffic_stale` says whether the kernel sources have changed since that pass was taken.
  `config.fp32_contractions` / `f32_input_mfma_everywhere`: how the hoisted fp32 contractions are evaluated (exact three-way bf16 split, six
      products per fp32 product on the bf16 matrix cores, fp32 accuracy - csrc/gemm_split.inc), and the same train step timed for 5 more
      steps with every contraction on the f32-input MFMA instead (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain), N = 1 only.
  `cpu_baseline`: the oracle (torch-CPU fp32 restatement of the same graph) timed on this host at the FULL workload: thread sweep on a
      short prefix, then 1 warm-up + 3 timed full train steps at the best thread count, median (falls back to a truncated sample only
      when a full step would not fit the time limit, and says so).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # kernel-argument blocks in device memory (the runtime's default here; see multi_speaker_tts_amd/__init__.py)

import numpy as np
import torch

B_PER_GPU, T_ENC, L_MEL = 32, 128, 800
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM-side traffic per launch comes from the committed rocprofv3 PMC passes of THIS kernel version (separate --pmc FETCH_SIZE /
# --pmc WRITE_SIZE runs with --kernel-trace only, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; collected by
# tools/profile_round.sh, aggregated by tools/pmc_summary.py).  No file / no row -> traffic is reported as null.
def _latest_pmc_csv():
    """The newest round's committed FETCH/WRITE pass (profiles/rNN_pmc_fetch_write_per_kernel.csv)."""
    import glob
    import re
    best, best_n = os.path.join(ROOT, "profiles", "r02_pmc_fetch_write_per_kernel.csv"), -1
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write_per_kernel.csv")):
        m = re.match(r"r(\d+)_pmc_fetch_write_per_kernel\.csv$", os.path.basename(f))
        if m and int(m.group(1)) > best_n:
            best, best_n = f, int(m.group(1))
    return best


PMC_TRAFFIC_CSV = _latest_pmc_csv()


def _sha16(paths):
    import hashlib
    h = hashlib.sha256()
    for f in paths:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def persist_source_sha16():
    """Hash of the sources the persistent decoder kernels are compiled from: a PMC pass is only as good as the kernel it measured."""
    c = os.path.join(ROOT, "multi_speaker_tts_amd", "csrc")
    return _sha16([os.path.join(c, f) for f in ("persist.hip", "persist_bwd.hip", "persist_common.h", "persist_fwd_parts.h", "common.h")])


def pmc_collected_for():
    """The kernel-source hash recorded next to the committed PMC pass when it was taken (tools/profile_round.sh), or None."""
    f = PMC_TRAFFIC_CSV.replace("_pmc_fetch_write_per_kernel.csv", "_pmc_kernel_source_sha16.txt")
    try:
        return open(f).read().strip()
    except OSError:
        return None


def pmc_traffic_bytes(kernel_prefix):
    """2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes) per launch of the first kernel whose name contains `kernel_prefix`, or None."""
    import csv
    if not os.path.exists(PMC_TRAFFIC_CSV):
        return None
    for r in csv.DictReader(open(PMC_TRAFFIC_CSV)):
        if kernel_prefix in r["kernel"]:
            try:
                return (float(r["avg_FETCH_SIZE_KB_x2_gfx950_correction"]) + float(r["avg_WRITE_SIZE_KB"])) * 1024.0
            except (KeyError, ValueError):
                return None
    return None


def synthetic_batch(dims, B, Te, L, seed, rank, device):
    """fixed-length random tokens / clipped-normal mels / whole-tensor-normalised speaker embeddings."""
    g = np.random.default_rng(seed + rank)
    tok = g.integers(2, dims.n_tok, size=(B, Te)).astype(np.int32)
    tok[:, 0] = 0
    tok[:, -1] = 1
    mel = np.clip(g.normal(0, 1.5, size=(B, L, dims.n_mel)), -4, 4).astype(np.float32)
    spk = g.normal(0, 1, size=(B, dims.spk))
    spk = (spk / np.sqrt((spk ** 2).sum())).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(device).contiguous()
    return {"Token": t(tok), "Token_Length": t(np.full(B, Te, np.int32)), "Mel": t(mel),
            "Mel_Length": t(np.full(B, L, np.int32)), "Speaker_Embedding": t(spk)}


def contraction_replay(eng, batch, w, config3):
    """Every GEMM-entry-point call of one train step, recorded at the ctypes boundary and replayed on its own (HIP events, 5 repeats per distinct
    call, as tools/gemm_step_profile.py): their summed time per step and rate.  The fp32 contractions run as six bf16 products per fp32 product,
    so their ceiling is the bf16 matrix-core peak / 6 (417 TFLOP/s-equivalent at the 2.5 PF spec; the chip runs them power-limited - ~1 375 W, shader
    clock ~1.95 GHz, profiles/r05_gemm_stamps_and_clock.txt - which puts the attainable ceiling nearer 340); config 3's single-product bf16 contractions
    are priced against the bf16 peak itself."""
    import ctypes as C
    from collections import OrderedDict
    from multi_speaker_tts_amd import lib
    import multi_speaker_tts_amd.engine as E
    calls, real = [], lib.call

    def spy(name, *a):
        if name in ("mstts_gemm_f32", "mstts_gemm_bf16"):
            cp = lib.GemmDesc()
            C.memmove(C.byref(cp), C.byref(a[0]._obj), C.sizeof(lib.GemmDesc))
            calls.append((name, cp))
        return real(name, *a)
    lib.call = E.call = spy
    try:
        eng.forward(batch, w)
        eng.loss_and_backward(w)
        torch.cuda.synchronize()
    finally:
        lib.call = E.call = real
    groups = OrderedDict()
    for name, d in calls:
        groups.setdefault((name, d.M, d.N, d.K, d.trans_a, d.trans_b, d.win_T, d.win_C, d.split_k, d.batch, d.accumulate, d.act), []).append(d)
    total_us, total_flop = 0.0, 0.0
    for key, ds in groups.items():
        d = ds[0]
        real(key[0], C.byref(d))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            real(key[0], C.byref(d))
        e1.record()
        torch.cuda.synchronize()
        total_us += e0.elapsed_time(e1) * 1e3 / 5 * len(ds)
        total_flop += 2.0 * d.M * d.N * d.K * max(1, d.batch) * len(ds)
    tf = total_flop / total_us * 1e-6
    out = {"calls_per_step": len(calls), "ms_per_step": total_us * 1e-3, "tflops_equivalent": tf,
           "note": "replayed call by call outside the step (HIP events); accumulating calls re-add onto live gradients - results of this pass are not used"}
    if config3:
        out["fraction_of_bf16_mfma_peak"] = tf / 2500.0
    else:
        out["contractions_fraction_of_six_product_ceiling"] = tf / (2500.0 / 6.0)
        out["contractions_fraction_of_fp32_mfma_peak"] = tf / 157.3
    return out


def _cpu_baseline_worker(threads, budget_s):
    """SURVEY 8(d) protocol: the oracle train step (fwd + autograd bwd + TF-Adam) at the FULL workload - batch 32 x (128 tokens, 800
    mel frames) - 1 warm-up + 3 timed steps, median, on the thread count that a short sweep finds fastest (threads = the sweep's
    candidates, comma separated).  budget_s > 0: the truncated sample of earlier rounds instead (first frames of the 800, extrapolated;
    only for boxes where the full protocol does not fit)."""
    from oracle import model as OM, train as OT
    d = OM.Dims()
    params = OM.init_params(d, 1234)

    def run(L):
        batch = OT.synthetic_batch(d, B_PER_GPU, T_ENC, L, seed=1234)
        masks = OT.make_masks(d, B_PER_GPU, T_ENC, L + 1, True, seed=OT.step_seed(1234, 0))
        t0 = time.perf_counter()
        OT.train_step(params, None, d, batch, masks, 0, dtype=torch.float32)
        return time.perf_counter() - t0

    cands = [int(x) for x in str(threads).split(",")]
    sweep = {}
    for th in cands:                                        # 20-frame sample per candidate (seconds each), after a 2-frame warm-up
        torch.set_num_threads(th)
        run(2)
        sweep[th] = B_PER_GPU * 20 / run(20)
        if sweep[th] < 0.7 * max(sweep.values()):           # more threads have stopped paying: do not try the larger counts
            break
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    if budget_s <= 0:                                       # the full protocol must fit the harness: a host that needs more than ~2 minutes
        probe = run(100)                                    # per full step (estimated from 100 frames) gets the truncated sample instead
        if probe * L_MEL / 100.0 > 110.0:
            budget_s = 60.0
    if budget_s > 0:
        L = 100
        t = run(L)
        if t < 0.5 * budget_s:
            L = int(min(L_MEL, L * budget_s / t))
            t = run(L)
        times = [t]
        proto = "1 un-warmed step, truncated to the first %d of %d frames (time budget %.0f s)" % (L, L_MEL, budget_s)
    else:
        L = L_MEL
        run(L)                                              # warm-up at full size
        times = sorted(run(L) for _ in range(3))
        t = times[1]
        proto = "1 warm-up + 3 timed full steps, median"
    out = {"value": B_PER_GPU * L / t, "unit": "mel-frames/s", "cores": best, "kind": "port",
           "sample": "train step (fwd+bwd+TF-Adam) of the torch-CPU fp32 oracle at batch %d x (%d tokens, %d of %d mel frames) on %d of %d host threads: %s; %.1f s per step"
                     % (B_PER_GPU, T_ENC, L, L_MEL, best, os.cpu_count(), proto, t),
           "sample_frames": L, "full_frames": L_MEL, "extrapolation_factor": L_MEL / float(L), "sample_seconds": t, "step_seconds": times,
           "thread_sweep_frames_per_s": {str(k): v for k, v in sweep.items()}}
    out["config1"] = _cpu_config1(d, params)
    return out


def _cpu_config1(d, params):
    """BASELINE configs[0]: one utterance, 64 tokens -> 400 mel frames free-running forward of the oracle (speaker
    embedding given, Taco1 vocoder included) and a 100-iteration Griffin-Lim on the [400, 1025] spectrogram (host NumPy)."""
    from oracle import model as OM, train as OT
    from multi_speaker_tts_amd import Audio
    import dataclasses
    dd = dataclasses.replace(d, max_inf=399)
    p = dict(params)
    pb = np.array(p["decoder/decoder/linear_projection/dense/bias"], np.float32).copy()
    pb[-1] = -20.0                                      # never stop early: exactly max_inf + 1 = 400 steps
    p["decoder/decoder/linear_projection/dense/bias"] = pb
    g = np.random.default_rng(7)
    tok = g.integers(2, dd.n_tok, size=(1, 64)).astype(np.int32)
    tok[:, 0], tok[:, -1] = 0, 1
    spk = g.normal(0, 1, (1, dd.spk)); spk = (spk / np.sqrt((spk ** 2).sum())).astype(np.float32)
    batch = {"Token": torch.tensor(tok), "Token_Length": torch.tensor([64], dtype=torch.int32), "Mel": torch.zeros(1, 1, dd.n_mel),
             "Mel_Length": torch.zeros(1, dtype=torch.int32), "Speaker_Embedding": torch.tensor(spk)}
    masks = OT.make_masks(dd, 1, 64, 400, False, seed=5)
    pt = OM.to_torch(p, dtype=torch.float32)
    t0 = time.perf_counter()
    with torch.no_grad():
        out = OM.forward(pt, dd, batch, False, masks, with_vocoder=True)
    t_fwd = time.perf_counter() - t0
    frames = int(out["Linear"].shape[1])
    spec = np.clip(out["Spectrogram"][0].numpy().astype(np.float64), 0.0, 1.0)
    t0 = time.perf_counter()
    wav = Audio.Griffin_Lim(spec, rng=np.random.RandomState(0))
    t_gl = time.perf_counter() - t0
    return {"workload": "1 utterance, 64 tokens -> %d mel frames, oracle forward (fp32, incl. Taco1 vocoder) + 100-iteration Griffin-Lim on [%d,1025]" % (frames, frames),
            "forward_seconds": t_fwd, "griffin_lim_seconds": t_gl, "wav_samples": int(wav.shape[0]),
            "mel_frames_per_s_forward": frames / t_fwd, "mel_frames_per_s_end_to_end": frames / (t_fwd + t_gl)}


def cpu_baseline(budget_s=0.0, timeout_s=900):
    """Run the worker in a subprocess with a hard time limit.  Thread candidates: 16 / 32 / 64 / 128 host threads, ascending, the sweep
    stops once a count is clearly slower than a smaller one (a 256-thread host thrashes torch's intra-op pool on the loop's small
    GEMMs: with every host thread the first full-protocol attempt of this round did not finish in 25 minutes)."""
    import subprocess
    n = os.cpu_count() or 1
    cands = sorted({min(n, c) for c in (16, 32, 64, 128)})
    code = "import json,bench;print('CPUBASE'+json.dumps(bench._cpu_baseline_worker('%s',%f)))" % (",".join(map(str, cands)), budget_s)
    env = dict(os.environ)
    env.pop("OMP_NUM_THREADS", None); env.pop("MKL_NUM_THREADS", None)
    try:
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout_s).stdout
        for line in out.splitlines():
            if line.startswith("CPUBASE"):
                return json.loads(line[len("CPUBASE"):])
    except subprocess.TimeoutExpired:
        pass
    return {"value": None, "unit": "mel-frames/s", "cores": max(cands), "kind": "port", "sample": "oracle did not finish within %d s" % timeout_s}


def self_launch(n, cmd=None, grace_s=5.0, poll_s=0.1):
    """`python bench.py --gpus N` without a launcher: spawn N copies of this script (or of `cmd`, the tests' stand-in), one per GPU,
    with the environment torch.distributed.run would export (rendezvous on 127.0.0.1), and WATCH them: all ranks are polled together,
    and the first one that exits non-zero (died in init_process_group, fell out of a collective, out of memory ...) ends the job - the
    other ranks, which would sit in their next collective until the backend's own time-out (30 minutes), get SIGTERM, `grace_s`
    seconds later SIGKILL, and the launcher exits with the failed rank's code.  Returns 0 when every rank exited 0."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = cmd or ([sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd, env=env))

    def stop_all():
        for p in procs:                                     # exactly the processes started above, by handle - never by pattern
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + grace_s
        for p in procs:
            try:
                p.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    def on_signal(signum, _frame):                          # the launcher itself is told to stop: do not orphan the ranks
        stop_all()
        raise SystemExit(128 + signum)
    old = {s: signal.signal(s, on_signal) for s in (signal.SIGTERM, signal.SIGINT)}
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                sys.stderr.write("bench.py launcher: rank %d exited with code %d; stopping the other %d rank(s)\n" % (r, c, n - 1))
                stop_all()
                raise SystemExit(c if c > 0 else 128 - c)   # (a negative code is the signal that killed the rank)
            if all(c == 0 for c in codes):
                return 0
            time.sleep(poll_s)
    finally:
        for s, h in old.items():
            signal.signal(s, h)


def main():
    global B_PER_GPU, T_ENC
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-surface", action="store_true", help="skip the supplementary Tacotron2.Train_Step-through-the-Feeder measurement (train_surface)")
    ap.add_argument("--no-overlap", action="store_true", help="gradient all-reduce behind the backward pass instead of overlapped with it (A/B switch)")
    ap.add_argument("--cpu-budget", type=float, default=0.0, help="seconds for the host baseline; 0 = the full protocol (1 warm-up + 3 timed full steps)")
    ap.add_argument("--frames", type=int, default=L_MEL, help=argparse.SUPPRESS)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help=argparse.SUPPRESS)       # supplementary measurements only: never the headline
    ap.add_argument("--tokens", type=int, default=T_ENC, help=argparse.SUPPRESS)          # ... e.g. a batch padded to 160 tokens (the 256-position persistent kernels)
    ap.add_argument("--recurrent-dtype", default="f32", choices=("f32", "bf16"), help=argparse.SUPPRESS)   # bf16 recurrent products only
    ap.add_argument("--force-bf16-recurrent", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--config3-f32-loops", action="store_true", help=argparse.SUPPRESS)          # A/B: config 3 with exact fp32 inside the persistent loops (round 4's form)
    ap.add_argument("--force-allreduce", action="store_true", help=argparse.SUPPRESS)          # 1 GPU: a one-rank RCCL group, every collective really issued (mechanics check, never a headline)
    ap.add_argument("--no-gemm-tail-split", action="store_true", help=argparse.SUPPRESS)      # A/B: every GEMM tile whole (mstts_gemm_tail_split(0))
    ap.add_argument("--config3", action="store_true", help="BASELINE config 3 arithmetic (bf16 operands everywhere, fp32 master/accumulate); never the headline")
    args = ap.parse_args()
    headline_batch = args.batch == B_PER_GPU and args.tokens == T_ENC
    B_PER_GPU = args.batch
    T_ENC = args.tokens

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world == 1 and args.gpus > 1:
        # not started by torch.distributed.run: launch the ranks ourselves (one process per GPU, rank 0's JSON line passes through)
        return self_launch(args.gpus)
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # the library is loaded (and, should its sources have changed, rebuilt behind a file lock - minutes, one rank at a time) BEFORE the
    # process group exists: nothing with a collective time-out is waiting for a rank that is still compiling
    from multi_speaker_tts_amd import lib as _lib_early
    _lib_early.load()
    dist = None
    if world > 1 or args.force_allreduce:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        t_init = time.perf_counter()
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=datetime.timedelta(seconds=300))
        # start-up self-check, BEFORE anything is timed: one tiny sum over the ranks must come back as world * (world + 1) / 2 on every
        # rank - a job whose communicator is not the N ranks it was asked for stops here, in seconds, with a message (stderr: stdout
        # carries the one JSON line)
        chk = torch.tensor([float(rank + 1)], dtype=torch.float64, device=device)
        dist.all_reduce(chk)
        torch.cuda.synchronize()
        ok = dist.get_world_size() == world and abs(float(chk.item()) - world * (world + 1) / 2.0) < 1e-9
        if rank == 0 or not ok:
            sys.stderr.write("bench.py start-up check: rccl_ranks=%d (asked for %d), all-reduce %s, communicator up after %.1f s\n"
                             % (dist.get_world_size(), world, "ok" if ok else "WRONG (%r)" % float(chk.item()), time.perf_counter() - t_init))
            sys.stderr.flush()
        if not ok:
            raise SystemExit(3)

    from multi_speaker_tts_amd import lib
    from multi_speaker_tts_amd.engine import TrainEngine
    from multi_speaker_tts_amd.params import Dims
    from multi_speaker_tts_amd.dist import GradAllReduce

    dims = Dims()
    L = args.frames
    if args.no_gemm_tail_split:
        lib.call("mstts_gemm_tail_split", 0)
    if args.config3:
        # config 3 = bf16 operands with fp32 master / accumulate, in EVERY product of the step: the hoisted contractions (gemm_bf16) and - round 5 -
        # the recurrent products inside the two persistent decoder loops (their bf16 instantiations, v_mfma_f32_16x16x32_bf16).
        # --config3-f32-loops: round 4's form (bf16 hoisted contractions, exact fp32 inside the loops) for the A/B;
        # --force-bf16-recurrent together with MSTTS_PERSIST_BF16=0: the bf16 launch-per-step loops of rounds 1-2
        args.recurrent_dtype = "f32" if (args.config3_f32_loops and not args.force_bf16_recurrent) else "bf16"
    eng = TrainEngine(dims, device=device, seed=1234, rank=rank, world=world, recurrent_dtype=args.recurrent_dtype,
                      gemm_dtype="bf16" if args.config3 else "f32")
    batch = synthetic_batch(dims, B_PER_GPU, T_ENC, L, 1234, rank, device)
    # config 3: bf16 gradient message with fp32 accumulation on receipt; the fp32 headline keeps the fp32 all-reduce
    reducer = GradAllReduce(eng.params.grad, world, comm_dtype="bf16" if args.config3 else "f32", overlap=not args.no_overlap, trace=True,
                            **({"force": True} if args.force_allreduce else {})) if (world > 1 or args.force_allreduce) else None
    eng.trace_events = reducer is not None

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.train_step(batch, all_reduce=reducer)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step(batch, all_reduce=reducer)
    sync()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [1e3 * elapsed / args.steps]
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        every = [torch.zeros_like(tt) for _ in range(dist.get_world_size())]
        dist.all_gather(every, tt)
        per_rank_ms = [1e3 * float(t.item()) / args.steps for t in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B_PER_GPU * L / (elapsed / args.steps)

    out = {"metric": "mel-frames/sec (train step) at batch 32x(128 tok,800 mel)" if headline_batch else
                     "mel-frames/sec (train step) at per-GPU batch %d x %d tokens (supplementary, not the BASELINE configuration)" % (B_PER_GPU, T_ENC), "value": value, "unit": "mel-frames/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
           "ms_per_step_per_rank": per_rank_ms, "ms_per_step_min_over_ranks": min(per_rank_ms), "ms_per_step_max_over_ranks": max(per_rank_ms),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": ("bf16 operands in every hoisted contraction, f32 accumulate + f32 master, f32 inside the persistent decoder loops (BASELINE config 3 arithmetic, not the headline)"
                     if args.config3 and args.recurrent_dtype == "f32" else "f32" if args.recurrent_dtype == "f32" else
                     ("bf16 operands in every product of the step incl. the recurrent ones (%s), f32 accumulate + f32 master (BASELINE config 3 arithmetic, not the headline)"
                      % ("persistent decoder loops on v_mfma_f32_16x16x32_bf16" if getattr(eng, "persist_bf16", False) and eng.persist else "launch-per-step bf16 loops")) if args.config3 else
                     "f32 + bf16 recurrent products (not the headline)"), "data": "synthetic",
           "config": {"workload": "BASELINE.json configs[%d]: Tacotron2 train step (fwd+bwd+TF-Adam), per-GPU batch %d x (%d tokens, %d mel frames), random speaker embeddings, %s"
                                  % (2 if args.config3 else 1, B_PER_GPU, T_ENC, L, "bf16 operands with fp32 master / accumulate" if args.config3 else "fp32"),
                      "global_batch": world * B_PER_GPU, "parallelism": "dp%d" % world}}
    if not args.config3:
        # How the fp32 contractions are evaluated (csrc/gemm_split.inc), and the same steps with every contraction on the f32-input MFMA
        # (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain) for comparison - a short untimed-warmup + 5-step leg on rank 0 at N = 1.
        out["config"]["fp32_contractions"] = ("fp32 operands and accumulators; every operand element split EXACTLY into three bf16 terms, six bf16 x bf16 "
                                              "products per fp32 product on v_mfma_f32_32x32x16_bf16 (dropped terms <= 2^-26 relative, below fp32 unit roundoff; "
                                              "error vs fp64 equal to the f32-input MFMA kernel's: tests/test_gpu_ops.py::test_gemm_split_is_fp32_accurate); "
                                              "MSTTS_GEMM_SPLIT3=0 or mstts_gemm_split3(0) selects v_mfma_f32_32x32x2_f32 everywhere")
        if world == 1 and not args.no_roofline:
            lib.call("mstts_gemm_split3", 0)
            eng.exact_f32_products = True            # ... and inside the persistent forward loop (its two on-chain products are split products otherwise)
            for _ in range(2):
                eng.train_step(batch, all_reduce=reducer)
            sync()
            t1 = time.perf_counter()
            for _ in range(5):
                eng.train_step(batch, all_reduce=reducer)
            sync()
            out["f32_input_mfma_everywhere"] = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / 5, "steps": 5}
            out["f32_input_mfma_everywhere"]["value"] = B_PER_GPU * L / (out["f32_input_mfma_everywhere"]["ms_per_step"] * 1e-3)
            out["f32_input_mfma_everywhere"]["covers"] = "every hoisted contraction AND every product inside the persistent decoder loops (forward launch in its pre == NULL form)"
            lib.call("mstts_gemm_split3", 1)
            eng.exact_f32_products = False
    # health counters of the persistent launches: in a multi-rank job every rank's, not rank 0's (a rank that falls back every step drags
    # the whole job - the step time is the slowest rank's - and must not hide behind a healthy rank 0): sum and maximum over the ranks
    counters = {"decoder_forward": eng.persist_fallbacks, "decoder_bptt": eng.persist_bwd_fallbacks, "encoder_bilstm": eng.persist_enc_fallbacks,
                "non_persistent_plans": eng.non_persistent_plans, "persist_disabled_steps": eng.persist_disabled_steps, "collective_redos": eng.collective_redos}
    per_rank_max = dict(counters)
    if dist is not None:
        exposed = reducer.exposed_ms()
        trace = reducer.first_piece_trace(eng.bptt_end_event) if eng.bptt_end_event is not None else None
        vec = torch.tensor([float(v) for v in counters.values()] + [exposed] + (list(trace[:2]) if trace else [0.0, 0.0]), dtype=torch.float64, device=device)
        vmax, vsum = vec.clone(), vec.clone()
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vsum, op=dist.ReduceOp.SUM)
        n = len(counters)
        per_rank_max = {k: int(vmax[i].item()) for i, k in enumerate(counters)}
        counters = {k: int(vsum[i].item()) for i, k in enumerate(counters)}
        out["rccl_ranks"] = dist.get_world_size()
        out["allreduce"] = {"overlapped_with_backward": not args.no_overlap, "exposed_ms_per_step": float(vmax[n].item()),
                            "exposed_ms_per_step_mean_over_ranks": float(vsum[n].item()) / dist.get_world_size(),
                            "message": "bf16, fp32 accumulate" if args.config3 else "fp32", "bytes_per_rank": eng.params.grad.numel() * (2 if args.config3 else 4)}
        if trace:
            # last timed step, maximum over the ranks: the first range (postnet, announced right behind the persistent BPTT launch on the compute
            # stream) and the end of its first piece, both relative to the END of persist_bwd_kernel.  A piece that ends about
            # bytes / link rate after its announcement ran under the hoisted weight-gradient products; one that ends near the end of the
            # backward pass queued behind them.
            out["allreduce"]["first_collective_start_vs_bptt_end_ms"] = float(vmax[n + 1].item())
            out["allreduce"]["first_piece_end_vs_bptt_end_ms"] = float(vmax[n + 2].item())
            out["allreduce"]["first_piece_bytes"] = int(trace[2]) * (2 if args.config3 else 4)

    plan0 = eng.plan(B_PER_GPU, T_ENC, L)
    out["persistent_launches"] = {"decoder_forward": bool(getattr(plan0, "persist", False)),
                                  "decoder_bptt": bool(getattr(plan0, "persist_bwd", False)),
                                  "encoder_bilstm": bool(getattr(plan0, "persist_enc", False)),
                                  "fallbacks": {k: counters[k] for k in ("decoder_forward", "decoder_bptt", "encoder_bilstm")},
                                  "non_persistent_plans": counters["non_persistent_plans"], "persist_disabled_steps": counters["persist_disabled_steps"],
                                  "collective_redos": counters["collective_redos"],
                                  "scope": "sum over all %d ranks" % world, "max_over_ranks": per_rank_max}
    if rank == 0 and not args.no_roofline:
        w = eng.plan(B_PER_GPU, T_ENC, L)
        S = L + 1
        lb = lib.load()
        M, A, H = dims.mem, dims.att, dims.dec_lstm
        # attention step, algorithmic bytes per launch / stage: B x (keys + values + cum r/w + alignment write)
        att_bytes = B_PER_GPU * (T_ENC * A * 4 + T_ENC * M * 4 + 3 * T_ENC * 4)

        def probe(kinds, backward):
            """HIP events bracketing every launch of one kernel kind inside the launch-per-step loops, one extra untimed step per kind.
            An event-to-event interval contains the event packets' own processing; the empty bracket recorded right behind each launch
            pays that twice, so half of it is subtracted (reproduces the rocprofv3 kernel-trace average within 3 %)."""
            res = {}
            for name, kind in kinds.items():
                lb.mstts_probe_begin(kind, S)
                eng.forward(batch, w)
                if backward:
                    eng.loss_and_backward(w)
                torch.cuda.synchronize()
                tot, emp = ctypes.c_double(0.0), ctypes.c_double(0.0)
                n = lb.mstts_probe_result(ctypes.byref(tot), ctypes.byref(emp))
                if n:
                    res[name] = {"avg_us": 1e3 * tot.value / n - 0.5 * 1e3 * emp.value / n, "event_bracket_us": 1e3 * tot.value / n,
                                 "empty_bracket_us": 1e3 * emp.value / n}
            lb.mstts_probe_begin(0, 0)
            return res

        persistent = bool(getattr(w, "persist", False))
        stage_names = ["loop top + prenet rows of cell 0 (in the shadow of the context hand-off)", "wait ctx", "cell0 ctx product + publish", "wait partials0", "cell0 update + publish",
                       "wait m0", "cell1 m0 product + publish", "shadow: stage h0 + h0 half 1", "wait partials1", "cell1 update + publish",
                       "shadow: h0 half 2", "wait m1 row", "query + partial energies + publish", "shadow: stage h1 + h1 product", "wait energies",
                       "softmax + context + publish"]
        stage_order = [0, 1, 2, 13, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15]      # execution order (stamp 13 sits behind stage 2)
        if persistent:
            # in-kernel stage timing: the persistent launch stamps the 100 MHz wall clock (s_memrealtime) at 16 points of every
            # step in every workgroup and sums the intervals; one extra, untimed step with the stamping instantiation
            eng.persist_stamps = torch.zeros(256 * 16, dtype=torch.int64, device=device)
            eng.forward(batch, w)
            torch.cuda.synchronize()
            ticks = eng.persist_stamps.view(256, 16).double().cpu().numpy()
            eng.persist_stamps = None
            per_step_us = ticks.mean(axis=0) * 0.01 / S                 # 10 ns per tick
            frame_us = float(per_step_us.sum())
            # the attention STAGE: from the moment the cell-1 outputs (m1) leave their producers to the moment the context store has been
            # issued - the m1 hand-off, the 16 query units, the partial energies, the energy hand-off, softmax, context (stamps 11, 12,
            # 14, 15) and the half-product that runs in the shadow of the m1 hand-off (stamp 10).  The stricter reading also charges the
            # outbound hand-off: the time the consumers of that context wait for it at the top of the next step (stamp 1, "wait ctx").
            att_us = float(per_step_us[10:13].sum() + per_step_us[14:16].sum())
            strict_us = att_us + float(per_step_us[1])
            ach = att_bytes / (att_us * 1e-6) / 1e9
            out["roofline"] = {"kernel": "persist_fwd_kernel, attention stage of one decoder step (m1 hand-off, query units, partial energies, energy hand-off, softmax, context; B=32): keys / values stay on chip",
                               "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "frac_incl_outbound_handoff": att_bytes / (strict_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_launch_us_incl_outbound_handoff": strict_us,
                               "traffic": (pmc_traffic_bytes("persist_fwd_kernel") or 0) / S or None,
                               "traffic_note": "HBM-side bytes of the WHOLE persistent launch per decoder step (committed rocprofv3 PMC pass, FETCH_SIZE x 2 + WRITE_SIZE, / %d steps): history written for BPTT, hand-off rings and operands - the attention stage's keys / values are read once per sequence" % S,
                               "traffic_source": os.path.relpath(PMC_TRAFFIC_CSV, ROOT), "traffic_source_sha16": _sha16([PMC_TRAFFIC_CSV]) if os.path.exists(PMC_TRAFFIC_CSV) else None,
                               "traffic_collected_for_kernel_sources": pmc_collected_for(), "kernel_sources_now": persist_source_sha16(),
                               "traffic_stale": pmc_collected_for() != persist_source_sha16(),
                               "algorithmic_bytes_per_launch": att_bytes, "avg_launch_us": att_us,
                               "timing": "s_memrealtime stamps inside the launch (its PROF template instantiation: a different binary from the timed one, same source, "
                                         "stamping overhead < 1 %% of the frame), one extra untimed step, mean over 256 workgroups x %d steps" % S,
                               "stage_us": {stage_names[i]: float(per_step_us[i]) for i in stage_order}, "frame_us": frame_us,
                               "attention_compute_only_us": float(per_step_us[12] + per_step_us[15]),
                               "persistent_fallbacks": eng.persist_fallbacks}
            # the launch-per-step loop it replaced, measured the old way for the record
            w.persist = False
            fwd = probe({"lsa_step_fwd": 1, "cell0_gemm_fwd": 3, "cell1_gemm_fwd": 4}, False)
            w.persist = True
            if "lsa_step_fwd" in fwd:
                a2 = att_bytes / (fwd["lsa_step_fwd"]["avg_us"] * 1e-6) / 1e9
                out["roofline"]["launch_per_step"] = {"kernel": "lsa_step_kernel (query projection inside)", "avg_launch_us": fwd["lsa_step_fwd"]["avg_us"],
                                                      "achieved": a2, "frac": a2 / HBM_PEAK_GBS}
        else:
            fwd = probe({"lsa_step_fwd": 1, "lsa_context_fwd": 2, "cell0_gemm_fwd": 3, "cell1_gemm_fwd": 4}, False)
            att_us = fwd["lsa_step_fwd"]["avg_us"] + fwd.get("lsa_context_fwd", {"avg_us": 0.0})["avg_us"]
            ach = att_bytes / (att_us * 1e-6) / 1e9
            fused_query = bool(eng.fuse_query and lb.mstts_lsa_step_q_supported(T_ENC, M, H))
            out["roofline"] = {"kernel": "lsa_step_kernel (one decoder step, B=%d%s)" % (B_PER_GPU, ", query projection inside the launch" if fused_query else ""),
                               "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": pmc_traffic_bytes("lsa_step_kernel") if (L == L_MEL and world == 1) else None,
                               "traffic_source": os.path.relpath(PMC_TRAFFIC_CSV, ROOT), "algorithmic_bytes_per_launch": att_bytes,
                               "avg_launch_us": att_us, "event_bracket_us": fwd["lsa_step_fwd"]["event_bracket_us"],
                               "empty_bracket_us": fwd["lsa_step_fwd"]["empty_bracket_us"]}
            if fused_query:         # as launched the kernel also reads the query kernel and the cell-1 output rows
                out["roofline"]["as_launched_bytes"] = att_bytes + H * A * 4 + B_PER_GPU * H * 4 + B_PER_GPU * A * 4
        if bool(getattr(w, "persist_bwd", False)):
            bnames = ["step rows + tanh terms", "wait d_ctx partials", "d_ctx, values . d_ctx, publish partial d_alignment", "wait d_alignment",
                      "softmax backward, energy / query gradients, publish d_m1", "shadow: G of the step before", "wait d_m1 (+ d_h1)",
                      "cell-1 update backward + publish", "d[g1] product, rows on the chain (fetch + MFMA + reduce + publish)",
                      "shadow: d[g1] product, recurrent-state rows", "wait d_m0 (+ d_h0)", "cell-0 update backward + publish",
                      "d[g0] product, rows on the chain", "shadow: d[g0] product, recurrent-state rows", "next step's operand requests + barrier",
                      "loop top: window padding, d_ctx request, barrier"]
            eng.persist_bwd_stamps = torch.zeros(256 * 16, dtype=torch.int64, device=device)
            eng.forward(batch, w)
            eng.loss_and_backward(w)
            torch.cuda.synchronize()
            bt = eng.persist_bwd_stamps.view(256, 16).double().cpu().numpy().mean(axis=0) * 0.01 / S
            eng.persist_bwd_stamps = None
            out["bptt_persistent"] = {"kernel": "persist_bwd_kernel", "frame_us": float(bt.sum()), "fallbacks": eng.persist_bwd_fallbacks,
                                      "traffic_per_step": (pmc_traffic_bytes("persist_bwd_kernel") or 0) / S or None,
                                      "attention_backward_stage_us": float(bt[0:6].sum()),
                                      "stage_us": {n: float(v) for n, v in zip(bnames, bt) if n != "-"}}
        bwd = probe({"lsa_step_bwd": 5, "lsa_denergy_bwd": 6, "cell0_dgemm_bwd": 7, "cell1_dgemm_bwd": 8}, True)
        wb = 2 if args.recurrent_dtype == "bf16" else 4          # the recurrent kernels stream bf16 copies in that mode
        w0 = (M + H) * 4 * H * wb
        w1 = 2 * H * 4 * H * wb
        extra = []
        allk = dict(fwd); allk.update(bwd)
        for nm, byts in (("cell0_gemm_fwd", w0), ("cell1_gemm_fwd", w1), ("cell0_dgemm_bwd", w0), ("cell1_dgemm_bwd", w1)):
            if nm not in allk:
                continue
            a = byts / (allk[nm]["avg_us"] * 1e-6) / 1e9
            extra.append({"kernel": nm + (" (launch-per-step loop, not on the timed path)" if persistent and nm.endswith("fwd") else ""),
                          "bound": "weight-stream (Infinity Cache / HBM)", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": a / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": byts, "avg_launch_us": allk[nm]["avg_us"]})
        out["roofline_other"] = extra
        out["kernel_avg_us"] = {k: v["avg_us"] for k, v in allk.items()}
        out["step_flops_fraction_of_fp32_mfma_peak"] = (4.047e12 * L / L_MEL) / (ms_per_step * 1e-3) / 157.3e12
        out["hoisted_contractions"] = contraction_replay(eng, batch, w, args.config3)

    if rank == 0 and world == 1 and headline_batch and not args.config3 and not args.no_surface and not args.no_roofline:
        # Supplementary, never the headline: the reference's user trains through Tacotron2.Train() on length-bucketed batches of a different
        # shape every step (Feeder.py:111-124, MSTTS_SV.py:266-273).  50 such steps through the real Feeder and the drop-in class, beside the
        # same batches device-resident through the engine (tools/train_surface_bench.py).
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import train_surface_bench
            del eng, batch
            torch.cuda.empty_cache()
            out["train_surface"] = train_surface_bench.run(steps=50, warmup=3, device=str(device))
        except Exception as e:          # (a supplementary leg must not cost the run its headline line)
            out["train_surface"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(budget_s=args.cpu_budget)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio (its own buffer, flushed whenever), so the group
    # is torn down and every C stream flushed first
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
