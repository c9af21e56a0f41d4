#!/usr/bin/env python
"""Benchmark of the SAR-SSL pretraining step on MI355X (BASELINE.json metric: pretrain segments/s).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the whole hot path over one batch of synthetic input that is already resident in HBM (int16 PCM,
64 two-channel 4.112 s @ 16 kHz segments per GPU): fused STFT front-end -> masks -> two MC-Conformer encoders -> decoder ->
masked MSE -> hand-written backward -> (bucketed RCCL all-reduce overlapped with backward) -> fused Adam.  Numeric mode (--precision):
'hybrid' by default since round 6 - fp16 CNN stem, f32 residual stream through the Conformer blocks / decoder with f32 activations and
weights contracted as fp16 pairs on the matrix cores, bf16 backward - the mode that meets the 1e-3 per-bin tolerance against the
reference; 'fp16' (fp16 forward / bf16 backward) is timed next to it in the same line (`fast_mode`).  f32 accumulation; dropout active
(training mode), nothing cached or skipped.

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  * `roofline`: measured live with events on the launch stream around every BN-prologue launch of the dominant kernel
    (conv3x3_fwd_pp_kernel<false>, the forward 3x3 convolution with a stored 64-channel input), the effective shader clock of those
    launches (in-kernel s_memtime / s_memrealtime probe), and the sibling launches under their own labels;
  * `cpu_baseline`: the CPU oracle (oracle/sarssl_oracle.py, the validated restatement of the reference) timed on this host;
  * `knobs`: the resolved state of every environment switch that selects a compute path (a line is only comparable with another one
    under the same knobs; ablation switches that skip work do not exist in the product and any SARSSL_ABLATE* variable is refused);
  * `parity_class`: what the timed numeric mode is pinned to against the reference (tests/);
  * `product_loop` (N = 1): the same step driven the way run_pretrain.py drives it - PCM-16 WAV files -> dataset.PcmSegmentLoader
    (native reader thread + pinned upload) -> STFTLearner.pretrain_epoch (captured step, STFT inside the replay);
  * `dist` (N > 1): backend, RCCL version, ranks, bytes per gradient bucket, the event-timed all-reduce of each bucket and `overlap_check`
    (the CNN-stem backward window the exchange hides under against the sum of the buckets issued in front of it);
  * `fast_mode` / `tolerance_mode` (N = 1): a short timed run of the other 16-bit mode on the same box and batches.
`--via-learner` makes the product loop the line's primary value; `--workload config5` times BASELINE config 5 (4-microphone 10 s
segments: 3 microphone pairs per segment, T = 624 frames).
"""
import argparse
import json
import os
import random
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import sarssl_boot  # noqa: E402,F401

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
FLOP_PER_SEG_STEP = 86.5e9         # 3 x 28.84 GFLOP forward contractions (2-ch segment, T = 256)
ATTN_CORE_FLOP = 1.51e9            # of which the score / PV products (quadratic in T): 3 x (0.201 + 3 x 0.101) GFLOP
NSAMPLE = 65792
DEFAULT_PRECISION = "hybrid"
WORKLOADS = {
    # name: (samples per segment, microphones, frames T, pairs per segment)
    "config2": (65792, 2, 256, 1),
    "config5": (160000, 4, 624, 3),
}


def flop_per_segment(workload):
    _, _, T, pairs = WORKLOADS[workload]
    r = T / 256.0
    return pairs * ((FLOP_PER_SEG_STEP - ATTN_CORE_FLOP) * r + ATTN_CORE_FLOP * r * r)


def flop_executed_per_segment(workload, compact=True, hybrid=False):
    """FLOPs the step actually executes per segment.  `flop_per_segment` is the REFERENCE's op count (every frame through every layer).
    Since round 5 a training step runs the decoder and the row-wise tail of each encoder's last block (second feed-forward module + closing
    LayerNorm) on the masked frames only - half of them (exact: the loss reads nothing else) - so those products execute half
    their reference FLOPs, forward and backward.  hybrid: additionally returns the matrix-core FLOPs ISSUED - the forward Linear layers
    contract fp16 pairs (hi hi + lo hi + hi lo: 3x an f32 activation's product, 2x an fp16 activation's), same algorithmic work."""
    _, _, T, pairs = WORKLOADS[workload]
    rows = float(T)
    dec = 2.0 * rows * (768 * 3072 + 3072 * 1024)                       # decoder forward, all frames
    tail = 2.0 * rows * (2 * 512 * 2048 + 2 * 256 * 1024)               # last blocks' second feed-forward module (d = 512 and d = 256), forward
    skipped = 3.0 * 0.5 * (dec + tail) if compact else 0.0              # forward + two backward products each, half the rows
    executed = flop_per_segment(workload) - pairs * skipped
    if not hybrid:
        return executed, executed
    half = 0.5 if compact else 1.0

    def blk(d, last):        # extra forward matrix-core FLOPs of one Conformer block in the hybrid mode (segments beyond the first)
        ffn = 2.0 * rows * d * 4 * d
        first_ffn = 2 * ffn + 1 * ffn                                                        # ffn1 x3 (+2), ffn2 x2 (+1)
        second_ffn = (2 * ffn + 1 * ffn) * (half if last else 1.0)
        attn = 2 * 2.0 * rows * d * 3 * d + 2 * 2.0 * rows * d * d                          # q/k/v x3 (+2), output projection x3 (+2: f32 context)
        conv = 2 * 2.0 * rows * d * 2 * d + 1 * 2.0 * rows * d * d                          # pointwise 1 x3 (+2), pointwise 2 x2 (+1)
        return first_ffn + second_ffn + attn + conv
    extra = blk(512, True) + 2 * blk(256, False) + blk(256, True)
    extra += half * (2 * 2.0 * rows * 768 * 3072 + 1 * 2.0 * rows * 3072 * 1024)             # decoder: layer 1 x3, layer 2 x2
    extra += 2 * 2.0 * rows * 1024 * (512 + 256)                                             # frame-patch products x3 (pair in, pair weights)
    return executed, executed + pairs * extra


def cpu_baseline(B=8):
    """CPU oracle train step (fp32, B=8: the reference's own CPU-runnable shape) on this host's cores: 3 warm-up + 10
    timed full steps at 8 intra-op threads = the reference's own cap (code/run_pretrain.py:19-24), median; and a second, shorter
    sample at min(32, physical cores) threads when the host has more than 8."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import recipes
    import sarssl_oracle as orc
    from sar_ssl_amd import synth
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest.json")))["pretrain"]
    sig = torch.from_numpy(synth.make_batch(0, B))
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 8
    except Exception:
        phys = os.cpu_count() or 8

    def sample(nthr, nwarm, nstep):
        torch.set_num_threads(nthr)
        sd = recipes.recipe_state_dict(man, 0)
        state = {}
        random.seed(1)
        for _ in range(nwarm):                                     # oneDNN primitive creation, allocator warm-up
            orc.train_step(sig, sd, state, 1e-3)
        times = []
        for _ in range(nstep):
            t0 = time.time()
            orc.train_step(sig, sd, state, 1e-3)
            times.append(time.time() - t0)
        return float(np.median(times)), min(times), max(times)

    nthr = min(8, os.cpu_count() or 8)
    med, lo, hi = sample(nthr, 3, 10)
    out = {"value": round(B / med, 3), "unit": "segments/s", "cores": nthr, "kind": "port",
           "sample": "median of 10 full train steps (STFT+fwd+bwd+Adam) of the CPU oracle, fp32, batch %d, after 3 warm-up steps; "
                     "min %.2f / max %.2f s per step" % (B, lo, hi), "host_physical_cores": phys}
    more = min(32, phys)
    if more > nthr:
        # (one intra-op thread per core of a 256-core host was measured at 0.04 segments/s - oversubscribed oneDNN/OpenMP; the second
        #  sample is bounded to 32 threads)
        med2, lo2, hi2 = sample(more, 2, 5)
        out["more_threads"] = {"value": round(B / med2, 3), "cores": more,
                               "sample": "median of 5 steps after 2 warm-up steps, same workload; min %.2f / max %.2f s per step" % (lo2, hi2)}
    return out


def _smi_sample():
    """One reading of rocm-smi's power / clock report (the container's only card): (socket power W, power cap W, sclk MHz) or None."""
    import re
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout[r.stdout.index("{"):])
        c = d[sorted(d)[0]]
        num = lambda v: float(re.sub(r"[^0-9.]", "", str(v)) or "nan")
        pw = next((num(v) for k, v in c.items() if "Power (W)" in k and "Max" not in k), float("nan"))
        cap = next((num(v) for k, v in c.items() if "Max Graphics Package Power" in k), float("nan"))
        sclk = next((num(v) for k, v in c.items() if k.startswith("sclk clock speed")), float("nan"))
        return pw, cap, sclk
    except Exception:
        return None


def telemetry(run_steps, nsteps=150, sample=True):
    """Socket power and shader clock WHILE the step runs (round-4 verdict: the in-kernel probe read 1.6 GHz of 2.4 inside the step, with
    no power / clock telemetry next to it): a thread polls rocm-smi (a child process: it never touches this process's HIP state) while
    the main thread keeps replaying the step for `seconds` - in a loop of its own AFTER the timed region, so the timing is not disturbed."""
    import threading
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            v = _smi_sample()
            if v is not None:
                samples.append(v)

    # (a FIXED number of steps: under data parallelism every rank runs this loop - the steps contain collectives - and only rank 0 samples)
    idle = _smi_sample() if sample else None
    th = threading.Thread(target=poll, daemon=True)
    if sample:
        th.start()
    n = 0
    while n < nsteps:
        run_steps(10)
        torch.cuda.synchronize()
        n += 10
    stop.set()
    if sample:
        th.join(timeout=15)
    else:
        return None
    if not samples:
        return {"available": False, "note": "rocm-smi gave no reading on this box"}
    pw = [p for p, _, _ in samples if p == p]
    ck = [c for _, _, c in samples if c == c]
    cap = next((c for _, c, _ in samples if c == c), None)
    out = {"available": True, "source": "rocm-smi --showpower --showclocks --showmaxpower, polled while %d further steps run (after the timed region)" % n,
           "samples": len(samples), "power_cap_w": cap,
           "socket_power_w_avg": round(sum(pw) / len(pw), 1) if pw else None, "socket_power_w_max": max(pw) if pw else None,
           "sclk_mhz_avg": round(sum(ck) / len(ck), 1) if ck else None, "sclk_mhz_min": min(ck) if ck else None, "sclk_mhz_max": max(ck) if ck else None,
           "idle_before": {"socket_power_w": idle[0], "sclk_mhz": idle[2]} if idle else None}
    if pw and cap:
        out["power_frac_of_cap"] = round(out["socket_power_w_avg"] / cap, 3)
    return out


FAMILY_OF = (("conv3x3", ("sarssl_conv3x3_",)),
             ("gemm", ("sarssl_gemm", "sarssl_splitk_reduce", "sarssl_colsum", "sarssl_fp8_", "sarssl_ffn")),
             ("attention_core", ("sarssl_relpos_attn", "sarssl_relshift", "sarssl_bias2", "sarssl_axpby", "sarssl_softmax")),
             ("stem_hbm_passes", ("sarssl_stem_", "sarssl_cl_", "sarssl_bn_", "sarssl_mask_inputs", "sarssl_conv_taps", "sarssl_patch_", "sarssl_f64_",
                                  "sarssl_stft")),
             ("norm_dwconv_loss_adam", ("sarssl_layernorm", "sarssl_ln_", "sarssl_dwglu", "sarssl_dwconv", "sarssl_glu", "sarssl_act_bwd",
                                        "sarssl_masked_mse", "sarssl_adam", "sarssl_cast", "sarssl_step_", "sarssl_zero_arena")))


def families(prof, nsteps, npix_b, T, nseg, pairs, flop_per_seg):
    """Per-family kernel time of ONE step from the event-bracketed C-ABI calls of `nsteps` single-stream eager steps (each launch has
    the GPU to itself: comparable with rocprofv3's per-kernel statistics; the two-stream step is shorter than the sum).  For the two
    MFMA-bound families also the algorithmic FLOPs and the fraction of the dense bf16 / fp16 MFMA peak over ALL their launches
    (round-3 verdict: the line named the fastest convolution variant only)."""
    fam = {}
    names = {}
    for k, (n, ms) in prof.items():
        if not k.startswith("call:"):
            continue
        sym = k[5:]
        f = next((name for name, pre in FAMILY_OF if sym.startswith(pre)), "other")
        a = fam.setdefault(f, [0, 0.0])
        a[0] += n; a[1] += ms
        names[sym] = names.get(sym, 0.0) + ms
    if not fam:
        return None
    out = {f: {"launches_per_step": round(n / nsteps, 1), "ms_per_step": round(ms / nsteps, 4)} for f, (n, ms) in fam.items()}
    conv_flop = 12 * 2.0 * npix_b * 256 * T * 64 * 576                       # 2 encoders x 2 layers x (forward + data + weight gradient)
    attn_flop = ATTN_CORE_FLOP * (T / 256.0) ** 2 * nseg * pairs
    gemm_flop = flop_per_seg * nseg - conv_flop - attn_flop                  # every other contraction of the step
    for f, flop in (("conv3x3", conv_flop), ("gemm", gemm_flop)):
        if f in out and out[f]["ms_per_step"] > 0:
            tf = flop / (out[f]["ms_per_step"] * 1e-3) / 1e12
            out[f].update({"flop_per_step": flop, "achieved_tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / PEAK_BF16_TFLOPS, 4)})
    top = sorted(names.items(), key=lambda kv: -kv[1])[:5]
    out["largest_entry_points_ms_per_step"] = {k: round(v / nsteps, 4) for k, v in top}
    out["sum_ms_per_step"] = round(sum(ms for _, ms in fam.values()) / nsteps, 3)
    out["note"] = "single-stream eager steps, events around every C-ABI call; the gemm family includes its split-K folds / column sums"
    return out


def product_loop(dev, batch, precision, nseg=2048, epochs=2):
    """The step as run_pretrain.py drives it: `nseg` PCM-16 WAV segments on /dev/shm -> dataset.PcmSegmentLoader (native reader thread,
    pinned int16 batches uploaded on a copy stream) -> STFTLearner.pretrain_epoch (captured step, STFT front-end inside the replay, Adam
    re-created per epoch).  One untimed epoch (graph capture), then `epochs` timed epochs incl. their end-of-epoch synchronisation."""
    from sar_ssl_amd import dataset, learner as L, model, runtime, synth
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="sarssl_bench_", dir=base)
    try:
        uniq = synth.to_pcm16(synth.make_batch(7000, 16))
        for i in range(nseg):
            dataset.write_wav_pcm16(os.path.join(d, "%d.wav" % i), np.roll(uniq[i % 16], 997 * (i // 16), axis=0))
        files = dataset.segment_files(d)
        loader = dataset.PcmSegmentLoader(files, batch, NSAMPLE, 2, shuffle=True, seed=3, device=dev, nthreads=8, drop_last=True)
        torch.manual_seed(4321)
        net = model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device=dev)
        lrn = L.STFTLearner(net, win_len=512, win_shift_ratio=0.5, nfft=512, fre_used_ratio=1, fs=16000, task=None, ch_mode="M")
        lrn.cuda()
        if precision in ("fp16", "hybrid", "bf16", "fp8"):
            lrn.amp()
        runtime.set_precision(precision)
        random.seed(99)
        loader.set_epoch(0)
        lrn.pretrain_epoch(loader, lr=1e-3, epoch=0)                # capture + first replays
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for ep in range(1, epochs + 1):
            loader.set_epoch(ep)
            last = lrn.pretrain_epoch(loader, lr=1e-3, epoch=ep)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nstep = epochs * len(loader)
        g = lrn.__dict__.get("_step_graph")
        return {"value": round(nstep * batch / dt, 2), "unit": "segments/s", "ms_per_step": round(1e3 * dt / nstep, 3), "steps": nstep,
                "epochs": epochs, "segments_on_disk": nseg, "final_epoch_loss": round(float(last[0]), 5),
                "step_mode": "hipGraph replay, STFT inside the replay" if g is not None and g._plan is not None else "eager launches",
                "path": "PCM-16 WAV files (%s) -> dataset.PcmSegmentLoader (8 reader threads, pinned upload) -> "
                        "STFTLearner.pretrain_epoch" % ("/dev/shm" if base else "tmp dir")}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def other_mode_line(precision, dev, T, pcms, pcm, batch, nsteps=40):
    """Short timed run of the captured training step in another numeric mode (fresh model, same resident batches): capture + 3 warm-up
    replays + `nsteps` timed replays."""
    from sar_ssl_amd import model, parity, runtime
    from sar_ssl_amd.graph import PretrainStepGraph
    runtime.set_precision(precision)
    torch.manual_seed(1234)
    random.seed(1234)
    runtime.RT.manual_seed(1234)
    net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev).to(dev).train()
    flat = runtime.FlatParams(net)
    g = PretrainStepGraph(net, flat, None, lr=1e-3)
    k = [0]

    def step():
        pcm.copy_(pcms[k[0] % len(pcms)])
        k[0] += 1
        return g.step(pcm=pcm, static=True)[0]
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        last = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gate = parity.GATES[precision]
    res = {"precision": precision, "value": round(nsteps * batch / dt, 2), "unit": "segments/s", "ms_per_step": round(1e3 * dt / nsteps, 3),
           "steps": nsteps, "final_loss": round(float(last), 5), "per_bin_max_gate": gate["per_bin_max"],
           "per_bin_max_measured_eval_train": gate.get("measured", {}).get("per_bin_max"),
           "meets_north_star_per_bin": gate["per_bin_max"] <= 1e-3}
    del g, net, flat
    torch.cuda.empty_cache()
    return res


def launch_probe(args, json_fd):
    """Launcher self-test (no GPU, no kernels): the ranks `--gpus N` started form a process group (gloo unless SARSSL_DIST_BACKEND says
    otherwise), all-reduce their rank numbers, and rank 0 prints the launch-related fields of the benchmark line.  This is NOT a
    benchmark and runs none of the product path - it pins the launch convention on a box without GPUs."""
    import torch.distributed as dist
    from sar_ssl_amd import dist as sdist
    rank, world, local = sdist.init_from_env(backend=os.environ.get("SARSSL_DIST_BACKEND") or "gloo")
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the process group has %d rank(s)" % (args.gpus, world))
    t = torch.tensor([float(rank)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        out = {"launch_probe": True, "n_gpus": world, "dist": {"world": world, "backend": dist.get_backend() if world > 1 else None,
                                                               "rank_sum": float(t[0])},
               "self_launched": os.environ.get("SARSSL_SELF_LAUNCHED") == "1", "local_rank": local}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if os.environ.get("SARSSL_PROBE_FAIL_RANK") == str(rank):          # (test hook: a rank that dies must fail the whole command)
        sys.exit(7)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: a > 2 s timed region at ~11 ms per step)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="segments per GPU (default 64; config5: 16 four-microphone segments = 48 pairs)")
    # (round 5: "fp8" is no longer a bench choice - the e4m3 Linear GEMMs are 1.2-1.65x faster than the bf16 kernels, the just-in-time
    #  quantisation makes the step 7-12 % SLOWER and the ceiling with fused quantisation is +4-6 % (NOTES.md 4.5): the mode stays in
    #  runtime.set_precision as a parity-tested experiment, tests/test_gpu_fp8.py, and is not advertised as a configuration to time)
    # default: the mode that meets north_star's 1e-3 per-bin tolerance against the reference (round-5 verdict) - 'hybrid': fp16 CNN stem,
    # f32 residual stream, fp16-pair matrix-core products; 'fp16' (fp16 forward / bf16 backward: 1.2e-3 per bin in train mode) is timed
    # next to it in the same line (`fast_mode`)
    ap.add_argument("--precision", default=DEFAULT_PRECISION, choices=["hybrid", "fp16", "bf16", "fp32", "fp32_1pass"])
    ap.add_argument("--no-other-mode", action="store_true", help="skip the short timed run of the other 16-bit mode (fast_mode / tolerance_mode block)")
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-product-loop", action="store_true")
    ap.add_argument("--via-learner", action="store_true", help="primary value = the product loop (WAV files -> loader -> Learner.pretrain_epoch)")
    ap.add_argument("--eager", action="store_true", help="enqueue the step launch by launch instead of replaying the captured HIP graph")
    ap.add_argument("--graph", action="store_true", help="(default since round 5) replay the captured step also when data-parallel")
    ap.add_argument("--launch-probe", action="store_true",
                    help="launcher self-test: start the ranks, form the process group, agree on a value, print the line's launch fields "
                         "and exit without running the step (no GPU needed; tests/test_launch_cpu.py)")
    args = ap.parse_args()
    # `python bench.py --gpus N` starts its own N ranks (one process per GPU over RCCL) - the reference's multi-GPU form is one command
    # too (code/run_pretrain.py:204-205).  This happens BEFORE anything touches the GPU: the parent only waits, forwards rank 0's single
    # line (rank 0 inherits stdout) and exits with the worst rank's code.  Under torchrun (WORLD_SIZE set) nothing is spawned.
    from sar_ssl_amd import launch
    if args.gpus > 1 and not launch.launched():
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    # stdout carries exactly ONE line, the JSON: everything else that writes to file descriptor 1 - RCCL prints a version banner there
    # from C, buffered until exit - is sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    bad = sorted(k for k in os.environ if k.startswith("SARSSL_ABLATE"))
    if bad:
        sys.exit("bench.py: refusing to run with %s set - a step with work switched off is not a benchmark" % ", ".join(bad))

    if args.launch_probe:
        return launch_probe(args, json_fd)
    from sar_ssl_amd import dist as sdist, engine, hip, model, parity, runtime, synth, _lib
    nsample, nmic, T, pairs = WORKLOADS[args.workload]
    batch = args.batch if args.batch is not None else (64 if args.workload == "config2" else 16)
    # the rank's GPU is selected BEFORE the process group exists (RCCL binds a communicator to the current device at its first collective)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)      # (single-GPU functional tests run 2 ranks on one device over gloo)
    torch.cuda.set_device(local)
    rank, world, _ = sdist.init_from_env()
    if world != args.gpus:
        # never print an n_gpus line for a world that is not the one asked for (round-4 verdict: `--gpus 8` run plain printed n_gpus 1)
        sys.exit("bench.py: --gpus %d but the process group has %d rank(s) - run `python bench.py --gpus N` (starts its own ranks) or "
                 "torchrun --nproc-per-node N bench.py --gpus N" % (args.gpus, world))
    dp = sdist.exchanging()             # more than one rank - or SARSSL_DIST_FORCE=1: the data-parallel step with a process group of ONE rank
    dev = torch.device("cuda", local)
    runtime.set_precision(args.precision)
    torch.manual_seed(1234)
    random.seed(1234 + rank)
    runtime.RT.manual_seed(1234 + rank)

    net = model.SARSSL(sig_shape=(256, T, 2, 2), pretrain=True, device=dev).to(dev).train()
    flat = runtime.FlatParams(net)
    sdist.broadcast_parameters(flat)
    reducer = sdist.FlatGradAllReduce(net, flat)
    opt = runtime.FusedAdam(flat, lr=1e-3)
    opt.zero_grad()

    # synthetic structured segments, int16 PCM resident in HBM: NRES distinct batches per rank (16 unique segments each, circularly
    # shifted copies), visited round-robin - every step copies its batch into the step's input buffer on the device (17 MB at batch 64)
    NRES = 4
    pcms = []
    for r in range(NRES):
        uniq = synth.make_batch(1000 * rank + 100 * r, min(16, batch), nsample=nsample, nch=nmic)
        segs = np.stack([np.roll(uniq[i % len(uniq)], 997 * (i // len(uniq)), axis=0) for i in range(batch)], axis=0)
        pcms.append(torch.from_numpy(synth.to_pcm16(segs)).to(dev))
    pcm = pcms[0].clone()                                           # the step's input buffer
    step_no = [0]
    from_pcm_in_graph = nmic == 2                                   # (the captured step runs the 2-microphone front-end itself)

    def next_batch():
        pcm.copy_(pcms[step_no[0] % NRES])
        step_no[0] += 1

    def step_eager():
        next_batch()
        x = hip.stft_frontend(pcm)
        loss, diff, _ = net(x)
        loss.backward()
        g = reducer.finish()
        opt.step(grad_scale=g)
        opt.zero_grad()
        return loss

    # the training step of the product path (learner.pretrain_epoch): STFT front-end + masks + forward + backward + (all-reduce) + Adam
    # captured into HIP graph(s) and replayed - same kernels, same order, one graph launch per step instead of ~450 launches from Python.
    # Data parallel: the segmented replay (four graphs with the collectives in between; bit-equal to the eager data-parallel step at world
    # 2 and 4) is the default as well - the eager step needs 8-9 ms of host time per 11 ms step and eight ranks share one host; --eager
    # opts out.
    use_graph = not args.eager
    graph = None
    if use_graph:
        from sar_ssl_amd.graph import PretrainStepGraph
        graph = PretrainStepGraph(net, flat, reducer, lr=1e-3)

    state = {"graph": graph}

    def step():
        g = state["graph"]
        if g is None:
            return step_eager()
        try:
            next_batch()
            if from_pcm_in_graph:
                return g.step(pcm=pcm, static=True)[0]  # `pcm` IS the graph's input buffer (refilled above from the resident batches)
            return g.step(x=hip.stft_frontend(pcm))[0]  # >2 microphones: pairing front-end launched in front of the replay
        except Exception as e:                          # (never seen; a failed capture must not cost the whole run its number)
            if g._plan is not None:
                raise
            print("bench: graph capture failed (%r) - falling back to the launch-by-launch step" % (e,), file=sys.stderr, flush=True)
            state["graph"] = None
            opt.zero_grad()
            return step_eager()

    def barrier():
        if dp:
            torch.distributed.barrier()

    if state["graph"] is not None:
        step()                  # set-up, not a warm-up step: the first call captures the graph(s) - also with --warmup 0 the timed region replays
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    graph = state["graph"]
    state_graph = graph
    clk = torch.zeros(20, dtype=torch.int64, device=dev)       # clock probe of the convolution launches (csrc/conv3x3.hip)
    if graph is None:
        hip.profile_start()
    n0 = _lib.ncalls
    t0 = time.perf_counter()
    first = None
    for _ in range(args.steps):
        last = step()
        if first is None:
            first = last.detach().clone()              # (device-side copy: the step's result buffer is overwritten by the next replay)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    calls_per_step = (_lib.ncalls - n0) / max(args.steps, 1)
    # host work per step: the time the host needs to ENQUEUE one step into an idle GPU queue (no waiting behind earlier work: the
    # per-step uploads of masks / the batch copy block only as long as the device is busy).  The loop time above contains that waiting.
    enq = []
    for _ in range(5):
        torch.cuda.synchronize()
        te = time.perf_counter()
        step()
        enq.append(time.perf_counter() - te)
    torch.cuda.synchronize()
    host_enqueue_ms = 1e3 * sorted(enq)[len(enq) // 2]
    tele = None
    if not os.environ.get("SARSSL_BENCH_NO_TELEMETRY"):
        tele = telemetry(lambda k: [step() for _ in range(k)], sample=(rank == 0))
    prof2 = {}
    hip.conv_clock_probe(clk)
    if graph is None:
        prof = hip.profile_stop()
        step_eager()                                           # (one more step with the clock probe attached)
        torch.cuda.synchronize()
    else:
        # per-kernel durations: a graph replay offers no host-side events around single launches, so the event-bracketed launches
        # are taken from eager steps of the SAME training run right after the timed region (same shapes, same kernels, the two
        # encoder streams still overlapping); profiles/ holds the rocprofv3 kernel trace of the graph replays themselves
        opt.m, opt.v, opt.step_count = graph.m, graph.v, graph.nsteps
        for _ in range(2):
            step_eager()
        torch.cuda.synchronize()
        hip.profile_start()
        for _ in range(6):
            step_eager()
        prof2 = hip.profile_stop()                         # two encoder streams: durations stretched by the co-running stream
        os.environ["SARSSL_TWO_STREAMS"] = "0"             # one stream: each launch has the GPU to itself, as in rocprof's
        step_eager()                                       # per-kernel statistics of a single-stream run
        torch.cuda.synchronize()
        import gc
        gc.collect()
        gc.disable()                                       # (a collection pause inside an event interval reads as a slow launch)
        hip.profile_start(all_calls=True)                  # + every C-ABI call under its entry point's name: the step's time by family
        for _ in range(6):
            hip.gpu_runway(40.0)                           # the profiled step is host-bound: keep the GPU queue fed (hip.gpu_runway)
            step_eager()
            torch.cuda.synchronize()
        prof = hip.profile_stop()
        gc.enable()
        os.environ["SARSSL_TWO_STREAMS"] = "1"
    wall_khz = _lib.lib().sarssl_wall_clock_khz()
    clk_step = clk.cpu().numpy().reshape(5, 4).copy()
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dp:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el[0])
    loss_val = float(last.detach())
    first_loss_val = float(first.reshape(-1)[0]) if first is not None else None
    assert np.isfinite(loss_val), "non-finite loss"

    def eff_ghz(slot):
        """effective shader clock of the last probed launch of a variant: shader-clock ticks / constant-rate ticks x that rate"""
        s0, r0, s1, r1 = (int(v) for v in slot)
        if wall_khz <= 0 or r1 <= r0 or s1 <= s0:
            return None
        return round((s1 - s0) / (r1 - r0) * wall_khz * 1e-6, 3)

    dist_info = None
    if dp:
        dist_info = reducer.describe()
        if world == 1:
            dist_info["note"] = "SARSSL_DIST_FORCE=1: one rank - every all-reduce is a copy; this run exercises the RCCL code path, it is not a scaling point"
        ms = reducer.time_buckets(iters=5)
        t = torch.tensor([ms.get(b["name"], 0.0) for b in dist_info["buckets"]], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        for b, v in zip(dist_info["buckets"], t.cpu().tolist()):
            b["allreduce_ms_alone"] = round(v, 4)
            b["bus_GBps_alone"] = round(2.0 * (world - 1) / world * b["bytes"] / (v * 1e-3) / 1e9, 1) if v > 0 else None
        # overlap check (round-5 verdict: the first multi-GPU run has to document itself): the window the exchange is meant to hide under -
        # the CNN-stem backward, event-timed between the "stem_bwd_begin" and "stems" stage hooks of launch-by-launch steps - against the
        # sum of the three buckets issued in front of it, each timed alone above.  hidden_if_alone_times_hold: the buckets fit the window.
        marks = {}

        def timed_hook(name):
            if name in ("stem_bwd_begin", "stems"):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.setdefault(name, []).append(ev)
            reducer._on_stage(name)
        try:
            net.set_backward_stage_hook(timed_hook)
            opt.m, opt.v, opt.step_count = (state_graph.m, state_graph.v, state_graph.nsteps) if state_graph is not None else (opt.m, opt.v, opt.step_count)
            for _ in range(4):
                step_eager()
            torch.cuda.synchronize()
            wins = [a.elapsed_time(b) for a, b in zip(marks.get("stem_bwd_begin", []), marks.get("stems", []))][1:]
            win = torch.tensor([sorted(wins)[len(wins) // 2] if wins else 0.0], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(win, op=torch.distributed.ReduceOp.MIN)
            early = sum(b["allreduce_ms_alone"] for b in dist_info["buckets"] if b["name"] != "stems")
            dist_info["overlap_check"] = {"stem_backward_window_ms_min_over_ranks": round(float(win[0]), 3),
                                          "allreduce_ms_alone_of_the_buckets_issued_before_it": round(early, 3),
                                          "hidden_if_alone_times_hold": bool(early <= float(win[0])),
                                          "note": "eager launch-by-launch steps, events on the origin stream at the stage hooks; the buckets share "
                                                  "the node's xGMI links with nothing else but each other"}
        except Exception as e:                             # (diagnostics only)
            dist_info["overlap_check"] = {"error": repr(e)}
        finally:
            net.set_backward_stage_hook(reducer._on_stage)
        dist_info["steps_seen_by_reducer"] = reducer.nsteps
        dist_info["overlap"] = "buckets issued from backward-stage hooks (decoder -> spat -> spec before the CNN-stem backward, stems after it)"

    npix_b = batch * pairs
    hip.conv_clock_probe(None)

    out = None
    if rank == 0:
        segs_total = batch * world * args.steps
        value = segs_total / elapsed
        n, ms = prof.get("conv3x3_fwd:bn_prologue", (0, 0.0))          # forward 3x3 convolutions with a stored input (BN+ReLU prologue): 2 per step
        nc1, msc1 = prof.get("conv3x3_fwd_c1", (0, 0.0))               # the other 2: input formed from the stem's 4-channel input while staging
        nr1, msr1 = prof.get("conv3x3_dgrad_c1red", (0, 0.0))          # data gradient consumed in its epilogue (nothing stored)
        nd, msd = prof.get("conv3x3_fwd:identity", (0, 0.0))           # plain data-gradient launches of the same kernel
        nb, msb = prof.get("conv3x3_dgrad_bnred", (0, 0.0))            # <true> variant: data gradient + BatchNorm-backward sums
        flop_per_launch = 2.0 * npix_b * 256 * T * 64 * 576              # one 3x3 64->64 conv over B' x 256 x T pixels
        achieved = (flop_per_launch / (ms / n * 1e-3)) / 1e12 if n else 0.0
        nw, msw = prof.get("conv3x3_wgrad_kernel", (0, 0.0))
        if os.environ.get("SARSSL_BENCH_DEBUG"):
            for k in sorted(prof):
                if "conv" in k:
                    print("prof", k, prof[k][0], round(prof[k][1] / max(prof[k][0], 1), 4), file=sys.stderr)
        # HBM traffic / MFMA-busy per launch from the committed PMC passes (same kernels, same shape; tools/prof_counters.py: separate
        # rocprofv3 --pmc runs, FETCH_SIZE x 2 on gfx950).  tag -> variant of the 3x3 family
        PMC_TAGS = {20: "fwd_bn_prologue", 24: "fwd_from_4ch_input", 22: "dgrad_bnred", 26: "dgrad_consumed_in_epilogue", 23: "wgrad", 25: "wgrad_from_4ch_input"}
        traffic, mfma_busy, pmc_file, pmc_var = None, None, None, {}
        for name in ("r06_kernel_counters.json", "r05_kernel_counters.json", "r04_kernel_counters.json", "r03_kernel_counters.json", "r02_kernel_counters.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if traffic is None and args.workload == "config2" and batch == 64 and os.path.exists(pmc):
                for e in json.load(open(pmc)).get("kernels", []):
                    if e.get("tag") in PMC_TAGS and e.get("kernel", "").startswith("conv3x3_") and "hbm_read_mb" in e and "hbm_write_mb" in e:
                        pmc_var[PMC_TAGS[e["tag"]]] = {"traffic": (e["hbm_read_mb"] + e["hbm_write_mb"]) * 1e6, "mfma_busy": e.get("mfma_busy"),
                                                       "pmc_pass_avg_us": e.get("avg_us")}
                if "fwd_bn_prologue" in pmc_var:
                    traffic, mfma_busy, pmc_file = pmc_var["fwd_bn_prologue"]["traffic"], pmc_var["fwd_bn_prologue"]["mfma_busy"], name
        fps = flop_per_segment(args.workload)
        fexec = flop_executed_per_segment(args.workload, compact=bool(engine._DEC_MASKED and engine._TAIL_MASKED), hybrid=args.precision == "hybrid")
        seg_desc = "%dch %.3fs@16kHz segments (%d microphone pair%s, T = %d frames)" % (nmic, nsample / 16000.0, pairs, "" if pairs == 1 else "s", T)
        out = {
            "metric": "pretrain_segments_per_sec", "value": round(value, 2), "unit": "segments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp16": "fp16 forward / bf16 backward (16-bit MFMA operands, f32 accumulate)",
                      "hybrid": "fp16 stem + f32 residual stream (fp16-pair MFMA operands: hi hi + lo hi + hi lo), bf16 backward", "bf16": "bf16", "fp32": "f32(split-bf16 MFMA)", "fp32_1pass": "f32 storage, single-pass bf16 MFMA",
                      "fp8": "bf16 storage, fp8(e4m3) Linear GEMMs"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "SAR-SSL MC-Conformer cross-channel-reconstruction pretrain step (STFT+mask+fwd+bwd+Adam), "
                                   "%s, batch %d per GPU, dropout on" % (seg_desc, batch),
                       "baseline_config": args.workload, "global_batch": batch * world, "segment_samples": nsample, "parallelism": "dp%d" % world},
            "roofline": {"bound": None,
                         "kernel": "conv3x3_fwd_pp_kernel<false>, BN+ReLU-prologue launches (the forward 3x3 convolutions; events around "
                                   "exactly these launches inside training steps" +
                                   (", the other encoder's stream running concurrently" if graph is None else
                                    "; a graph replay has no per-launch host events, so: 6 eager single-stream steps of the same run "
                                    "right after the timed replays - two_stream_avg_ms = the same with both encoder streams") + ")",
                         "achieved": round(achieved, 1),
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": pmc_file, "mfma_busy": mfma_busy, "launches": n,
                         "avg_ms": round(ms / n, 4) if n else None, "effective_clock_ghz": eff_ghz(clk_step[0]),
                         "flop_per_launch": flop_per_launch,
                         "fwd_from_4ch_input_avg_ms": round(msc1 / nc1, 4) if nc1 else None, "fwd_from_4ch_input_clock_ghz": eff_ghz(clk_step[3]),
                         "dgrad_consumed_in_epilogue_avg_ms": round(msr1 / nr1, 4) if nr1 else None,
                         "dgrad_consumed_in_epilogue_clock_ghz": eff_ghz(clk_step[4]),
                         "dgrad_identity_avg_ms": round(msd / nd, 4) if nd else None, "dgrad_identity_clock_ghz": eff_ghz(clk_step[1]),
                         "dgrad_bnred_avg_ms": round(msb / nb, 4) if nb else None, "dgrad_bnred_clock_ghz": eff_ghz(clk_step[2]),
                         "wgrad_avg_ms": round(msw / nw, 4) if nw else None,
                         "two_stream_avg_ms": (round(prof2["conv3x3_fwd:bn_prologue"][1] / prof2["conv3x3_fwd:bn_prologue"][0], 4)
                                               if prof2.get("conv3x3_fwd:bn_prologue", (0, 0))[0] else None),
                         "nominal_clock_ghz": 2.4,
                         "flop_per_segment_step": fps,
                         # whole-step fraction of the dense 16-bit matrix-core peak: with the REFERENCE's op count (what a user of the
                         # reference would compute), with the FLOPs the step executes (decoder / last-block tails on the masked frames
                         # only) and - hybrid mode - with the matrix-core FLOPs it issues (fp16-pair products)
                         "end_to_end_frac": round(value / world * fps / (PEAK_BF16_TFLOPS * 1e12), 4),
                         "end_to_end_frac_reference_flops": round(value / world * fps / (PEAK_BF16_TFLOPS * 1e12), 4),
                         "flop_per_segment_step_executed": fexec[0],
                         "end_to_end_frac_executed_flops": round(value / world * fexec[0] / (PEAK_BF16_TFLOPS * 1e12), 4),
                         "mfma_flop_per_segment_step_issued": fexec[1],
                         "end_to_end_frac_issued_mfma_flops": round(value / world * fexec[1] / (PEAK_BF16_TFLOPS * 1e12), 4),
                         "families": families(prof, 6, npix_b, T, batch, pairs, fps)},
            # loss of the first and the last step of the timed region (lr 1e-3, dropout on, four resident batches round-robin, set-up and warm-up
            # steps in front): tests/test_gpu_train.py::test_bench_batches_loss_goes_down pins the direction at this batch size
            "first_loss": round(first_loss_val, 5) if first_loss_val is not None else None,
            "final_loss": round(loss_val, 5),
            "step_mode": "eager launches" if graph is None else "hipGraph replay (%d graph(s) per step)" % sum(1 for k, _ in graph._plan if k == "graph"),
            # host_enqueue: host time to issue one step into an idle queue (what must stay below the step time for the host not to be the
            # critical path); host_ms_per_step: wall time of the issue loop per step (includes waiting behind the device in blocking uploads)
            "host_enqueue_ms_per_step": round(host_enqueue_ms, 3),
            "host_ms_per_step": round(1e3 * t_host / args.steps, 3),
            "host_calls_per_step": round(calls_per_step, 1),
            # (SARSSL_STEM_LAST_ALL_CUS only takes effect on one GPU: under data parallelism both stems keep the 7/8 rule, model.py - the N = 1
            #  point of a scaling curve is therefore this build with the knob on, the N > 1 points are with it off: +1.2 % at N = 1)
            "knobs": dict(engine.knobs(), STEM_LAST_ALL_CUS_effective=int(engine._STEM_LAST_ALL_CUS and not dp)),
            "parity_class": parity.parity_class(args.precision),
        }
        # ---- both axes of the roofline (round-4 verdict): arithmetic intensity of the named kernel = algorithmic FLOPs / measured HBM
        # bytes against the ridge 2500 TFLOP/s / 8 TB/s = 312.5 FLOP/B decides `bound`; `frac` is the fraction of THAT bound's peak,
        # mfma_frac / hbm_frac carry both; `variants` does the same for every launch variant of the 3x3 family
        rf = out["roofline"]
        RIDGE = PEAK_BF16_TFLOPS * 1e12 / 8.0e12
        avg_s = (ms / n * 1e-3) if n else None
        rf["mfma_achieved_tflops"], rf["mfma_frac"] = rf["achieved"], rf["frac"]
        rf["ridge_flop_per_byte"] = RIDGE
        if traffic and avg_s:
            ai = flop_per_launch / traffic
            rf["arithmetic_intensity_flop_per_byte"] = round(ai, 1)
            rf["hbm_achieved_gbps"] = round(traffic / avg_s / 1e9, 1)
            rf["hbm_frac"] = round(traffic / avg_s / 8.0e12, 4)
            rf["bound"] = "hbm" if ai < RIDGE else "mfma"
            if rf["bound"] == "hbm":
                rf["achieved"], rf["peak"], rf["unit"], rf["frac"] = rf["hbm_achieved_gbps"], 8000.0, "GB/s", rf["hbm_frac"]
            rf["bound_note"] = ("arithmetic intensity %.0f FLOP/B vs ridge %.1f: %s side of the ridge (both fractions are within 10 %% of each "
                                "other: the launch sits at the ridge, limited by neither peak but by its per-tile schedule)" %
                                (ai, RIDGE, "bandwidth" if ai < RIDGE else "compute"))
        else:
            rf["bound"] = "mfma"
            rf["bound_note"] = "no PMC traffic for this workload / batch: arithmetic intensity not measured, MFMA fraction reported"
        live = {"fwd_bn_prologue": (n, ms), "fwd_from_4ch_input": (nc1, msc1), "dgrad_bnred": (nb, msb), "dgrad_consumed_in_epilogue": (nr1, msr1),
                "wgrad": (nw, msw), "wgrad_from_4ch_input": (nw, msw)}
        variants = {}
        for vname, (cnt, tot) in live.items():
            if not cnt:
                continue
            a_s = tot / cnt * 1e-3
            v = {"avg_ms": round(tot / cnt, 4), "mfma_tflops": round(flop_per_launch / a_s / 1e12, 1),
                 "mfma_frac": round(flop_per_launch / a_s / 1e12 / PEAK_BF16_TFLOPS, 4)}
            pv = pmc_var.get(vname)
            if pv:
                v.update({"traffic": pv["traffic"], "hbm_gbps": round(pv["traffic"] / a_s / 1e9, 1), "hbm_frac": round(pv["traffic"] / a_s / 8.0e12, 4),
                          "flop_per_byte": round(flop_per_launch / pv["traffic"], 1), "bound": "hbm" if flop_per_launch / pv["traffic"] < RIDGE else "mfma",
                          "mfma_busy_pmc": pv["mfma_busy"], "pmc_pass_avg_us": pv["pmc_pass_avg_us"]})
            variants[vname] = v
        if "wgrad" in variants and "wgrad_from_4ch_input" in variants:
            variants["wgrad"]["note"] = variants["wgrad_from_4ch_input"]["note"] = "live average over BOTH weight-gradient variants (one label)"
        rf["variants"] = variants
        # clock: in-kernel probe (effective_clock_ghz) next to the power / sclk telemetry of the running step - the part clocks to its power
        # budget (MI355X_MICROARCH.md "DVFS give-back"): matrix-heavy launches run below the 2.4 GHz the peak is quoted at
        rf["telemetry_during_step"] = tele
        if rf.get("effective_clock_ghz"):
            rf["peak_at_effective_clock_tflops"] = round(PEAK_BF16_TFLOPS * rf["effective_clock_ghz"] / 2.4, 1)
            rf["mfma_frac_at_effective_clock"] = round(rf["mfma_achieved_tflops"] / rf["peak_at_effective_clock_tflops"], 4)
        if dist_info is not None:
            out["dist"] = dist_info
    del graph, state
    if world == 1 and rank == 0 and args.precision in ("hybrid", "fp16") and not args.no_other_mode and not args.eager and from_pcm_in_graph:
        # the other 16-bit mode on the same box, same batches, same captured step: `fast_mode` (fp16 forward / bf16 backward) next to a
        # hybrid headline, `tolerance_mode` (hybrid) next to an fp16 one - so that one line carries the price of the per-bin tolerance
        other = "fp16" if args.precision == "hybrid" else "hybrid"
        try:
            out["fast_mode" if other == "fp16" else "tolerance_mode"] = other_mode_line(other, dev, T, pcms, pcm, batch, nsteps=max(40, min(60, args.steps)))
        except Exception as e:                           # (must not cost the run its headline)
            out["fast_mode" if other == "fp16" else "tolerance_mode"] = {"error": repr(e)}
        runtime.set_precision(args.precision)
    if world == 1:
        want_loop = (not args.no_product_loop or args.via_learner) and args.workload == "config2" and not dp
        if want_loop:
            opt = None
            net._stage_hook = None
            del net, flat, reducer, pcm
            torch.cuda.empty_cache()
            loop = product_loop(dev, batch, args.precision, nseg=32 * batch, epochs=2 if not args.via_learner else max(2, args.steps // 32))
            out["product_loop"] = loop
            if args.via_learner:
                out["resident_batch_replay"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "steps": out["steps"]}
                out["value"], out["ms_per_step"], out["steps"] = loop["value"], loop["ms_per_step"], loop["steps"]
                out["step_mode"] = loop["step_mode"]
                out["config"]["workload"] += "; driven through " + loop["path"]
                out["roofline"]["end_to_end_frac"] = round(loop["value"] * flop_per_segment(args.workload) / (PEAK_BF16_TFLOPS * 1e12), 4)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    try:
        reducer.close()                              # (native exchange: ncclCommDestroy; detaches the stage hook)
    except NameError:
        pass                                         # (deleted above in front of the product loop)
    if dp:
        torch.distributed.barrier()                 # rank 0 may still be printing / timing the isolated kernel
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
