#!/usr/bin/env python
"""Benchmark of the SAR-SSL pretraining step on MI355X (BASELINE.json metric: pretrain segments/s).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the whole hot path over one batch of synthetic input that is already resident in HBM (int16 PCM,
64 two-channel 4.112 s @ 16 kHz segments per GPU): fused STFT front-end -> masks -> two MC-Conformer encoders -> decoder ->
masked MSE -> hand-written backward -> (bucketed RCCL all-reduce overlapped with backward) -> fused Adam.  bf16 storage/MFMA,
f32 accumulation; dropout active (training mode), nothing cached or skipped.

Rank 0 prints ONE JSON line.  `roofline` is measured live with events on the launch stream around every BN-prologue launch of the
dominant kernel (conv3x3_fwd_pp_kernel<false>, the ping-pong schedule of the 3x3 convolution: the forward convolution per encoder that
reads a stored 64-channel input; the other forward convolution forms its input from the 4-channel stem input while staging (C1IN
variant of the same kernel), and it and the data-gradient launches are timed under their own labels);
`cpu_baseline` times the CPU oracle (oracle/sarssl_oracle.py, the validated restatement of the reference) on this host.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import sarssl_boot  # noqa: E402,F401

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
FLOP_PER_SEG_STEP = 86.5e9         # 3 x 28.84 GFLOP forward contractions
NSAMPLE = 65792


def cpu_baseline(nwarm=3, nstep=8, B=8):
    """CPU oracle train step (fp32, B=8: the reference's own CPU-runnable shape) on this host's cores: 3 warm-up + 8 timed full
    steps, median.  8 intra-op threads = the reference's own cap (code/run_pretrain.py:19-24)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import recipes
    import sarssl_oracle as orc
    from sar_ssl_amd import synth
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest.json")))["pretrain"]
    sig = torch.from_numpy(synth.make_batch(0, B))
    ncore = os.cpu_count() or 8
    nthr = min(8, ncore)
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
    med = float(np.median(times))
    # (one intra-op thread per core of a 256-core host was measured at 0.04 segments/s - oversubscribed oneDNN/OpenMP - and took
    #  ten minutes; the bounded sample is the 8-thread run only)
    return {"value": round(B / med, 3), "unit": "segments/s", "cores": nthr, "kind": "port",
            "sample": "median of %d full train steps (STFT+fwd+bwd+Adam) of the CPU oracle, fp32, batch %d, after %d warm-up steps; "
                      "min %.2f / max %.2f s per step" % (nstep, B, nwarm, min(times), max(times))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="segments per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="enqueue the step launch by launch instead of replaying the captured HIP graph")
    args = ap.parse_args()

    from sar_ssl_amd import dist as sdist, hip, model, runtime, synth, _lib
    # the rank's GPU is selected BEFORE the process group exists (RCCL binds a communicator to the current device at its first collective)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)      # (single-GPU functional tests run 2 ranks on one device over gloo)
    torch.cuda.set_device(local)
    rank, world, _ = sdist.init_from_env()
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node == --gpus"
    dev = torch.device("cuda", local)
    runtime.set_precision(args.precision)
    torch.manual_seed(1234)
    random.seed(1234 + rank)
    runtime.RT.manual_seed(1234 + rank)

    net = model.SARSSL(sig_shape=(256, 256, 2, 2), pretrain=True, device=dev).to(dev).train()
    flat = runtime.FlatParams(net)
    sdist.broadcast_parameters(flat)
    reducer = sdist.FlatGradAllReduce(net, flat)
    opt = runtime.FusedAdam(flat, lr=1e-3)
    opt.zero_grad()

    # synthetic structured segments, int16 PCM resident in HBM (16 unique segments per rank, circularly shifted copies)
    uniq = synth.make_batch(1000 * rank, 16)
    segs = np.stack([np.roll(uniq[i % 16], 997 * (i // 16), axis=0) for i in range(args.batch)], axis=0)
    pcm = torch.from_numpy(synth.to_pcm16(segs)).to(dev)

    def step_eager():
        x = hip.stft_frontend(pcm)
        loss, diff, _ = net(x)
        loss.backward()
        g = reducer.finish()
        opt.step(grad_scale=g)
        opt.zero_grad()
        return loss

    # the training step of the product path (learner.pretrain_epoch): STFT front-end + masks + forward + backward + (all-reduce) + Adam
    # captured into HIP graph(s) and replayed - same kernels, same order, one graph launch per step (per bucket boundary when
    # data-parallel) instead of ~450 launches from Python
    graph = None
    if not args.eager:
        from sar_ssl_amd.graph import PretrainStepGraph
        graph = PretrainStepGraph(net, flat, reducer, lr=1e-3)

    state = {"graph": graph}

    def step():
        g = state["graph"]
        if g is None:
            return step_eager()
        try:
            return g.step(pcm=pcm, static=True)[0]      # the resident batch IS the graph's input buffer: no per-step copy
        except Exception as e:                          # (never seen; a failed capture must not cost the whole run its number)
            if g._plan is not None:
                raise
            print("bench: graph capture failed (%r) - falling back to the launch-by-launch step" % (e,), file=sys.stderr, flush=True)
            state["graph"] = None
            opt.zero_grad()
            return step_eager()

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    graph = state["graph"]
    if graph is None:
        hip.profile_start()
    n0 = _lib.ncalls
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    calls_per_step = (_lib.ncalls - n0) / max(args.steps, 1)
    if graph is None:
        prof = hip.profile_stop()
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
        hip.profile_start()
        for _ in range(6):
            step_eager()
        prof = hip.profile_stop()
        os.environ["SARSSL_TWO_STREAMS"] = "1"
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el[0])
    loss_val = float(last.detach())
    assert np.isfinite(loss_val), "non-finite loss"

    # the same kernel on the same shape with nothing else on the GPU (in the step its launches share the device with the other
    # encoder's stream, which stretches the event-bracketed durations used for `roofline.achieved`)
    iso_ms = None
    if rank == 0 and args.precision != "fp32":
        xi = torch.randn((args.batch, 256, 256, 64), device=dev).to(torch.bfloat16)
        wi = (torch.randn((9, 64, 64), device=dev) * 0.05).to(torch.bfloat16)
        sci, shi = torch.ones(64, device=dev), torch.zeros(64, device=dev)
        for _ in range(3):
            hip.conv3x3_fwd(xi, wi, sci, shi, want_stats=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            hip.conv3x3_fwd(xi, wi, sci, shi, want_stats=True)   # exactly the in-step variant: BN+ReLU prologue + statistics epilogue
        e1.record()
        torch.cuda.synchronize()
        iso_ms = e0.elapsed_time(e1) / 10
        del xi

    if rank == 0:
        segs_total = args.batch * world * args.steps
        value = segs_total / elapsed
        n, ms = prof.get("conv3x3_fwd:bn_prologue", (0, 0.0))          # forward 3x3 convolutions with a stored input (BN+ReLU prologue): 2 per step
        nc1, msc1 = prof.get("conv3x3_fwd_c1", (0, 0.0))               # the other 2: input formed from the stem's 4-channel input while staging
        nr1, msr1 = prof.get("conv3x3_dgrad_c1red", (0, 0.0))          # data gradient consumed in its epilogue (nothing stored)
        nd, msd = prof.get("conv3x3_fwd:identity", (0, 0.0))           # plain data-gradient launches of the same kernel
        nb, msb = prof.get("conv3x3_dgrad_bnred", (0, 0.0))            # <true> variant: data gradient + BatchNorm-backward sums
        flop_per_launch = 2.0 * args.batch * 65536 * 64 * 576            # one 3x3 64->64 conv over B x 256 x 256 pixels
        achieved = (flop_per_launch / (ms / n * 1e-3)) / 1e12 if n else 0.0
        nw, msw = prof.get("conv3x3_wgrad_kernel", (0, 0.0))
        traffic, mfma_busy = None, None      # per launch, from the committed PMC passes (same kernel, same shape; tools/prof_counters.py)
        for name in ("r02_kernel_counters.json",):
            pmc = os.path.join(ROOT, "profiles", name)
            if args.batch == 64 and os.path.exists(pmc):
                for e in json.load(open(pmc)).get("kernels", []):
                    if e.get("tag") == 20 and e.get("kernel", "").startswith("conv3x3_fwd_pp_kernel"):
                        if "hbm_read_mb" in e and "hbm_write_mb" in e:
                            traffic = (e["hbm_read_mb"] + e["hbm_write_mb"]) * 1e6
                        mfma_busy = e.get("mfma_busy")
        out = {
            "metric": "pretrain_segments_per_sec", "value": round(value, 2), "unit": "segments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp32": "f32(split-bf16 MFMA)", "fp8": "bf16 storage, fp8(e4m3) Linear GEMMs"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "SAR-SSL MC-Conformer cross-channel-reconstruction pretrain step (STFT+mask+fwd+bwd+Adam), "
                                   "2ch 4.112s@16kHz segments, batch %d per GPU, dropout on" % args.batch,
                       "global_batch": args.batch * world, "segment_samples": NSAMPLE, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma",
                         "kernel": "conv3x3_fwd_pp_kernel<false>, BN+ReLU-prologue launches (the forward 3x3 convolutions; events around "
                                   "exactly these launches inside training steps" +
                                   (", the other encoder's stream running concurrently" if graph is None else
                                    "; a graph replay has no per-launch host events, so: 6 eager single-stream steps of the same run "
                                    "right after the timed replays - two_stream_avg_ms = the same with both encoder streams") + ")",
                         "achieved": round(achieved, 1),
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         "traffic": traffic, "mfma_busy": mfma_busy, "launches": n, "avg_ms": round(ms / n, 4) if n else None,
                         "flop_per_launch": flop_per_launch,
                         "fwd_from_4ch_input_avg_ms": round(msc1 / nc1, 4) if nc1 else None,
                         "dgrad_consumed_in_epilogue_avg_ms": round(msr1 / nr1, 4) if nr1 else None,
                         "dgrad_identity_avg_ms": round(msd / nd, 4) if nd else None,
                         "dgrad_bnred_avg_ms": round(msb / nb, 4) if nb else None,
                         "wgrad_avg_ms": round(msw / nw, 4) if nw else None,
                         "two_stream_avg_ms": (round(prof2["conv3x3_fwd:bn_prologue"][1] / prof2["conv3x3_fwd:bn_prologue"][0], 4)
                                               if graph is not None and prof2.get("conv3x3_fwd:bn_prologue", (0, 0))[0] else None),
                         "isolated_avg_ms": round(iso_ms, 4) if iso_ms else None,
                         "isolated_achieved": round(flop_per_launch / (iso_ms * 1e-3) / 1e12, 1) if iso_ms else None,
                         "end_to_end_frac": round(value / world * FLOP_PER_SEG_STEP / (PEAK_BF16_TFLOPS * 1e12), 4)},
            "final_loss": round(loss_val, 5),
            "step_mode": "eager launches" if graph is None else "hipGraph replay (%d graph(s) per step)" % sum(1 for k, _ in graph._plan if k == "graph"),
            "host_ms_per_step": round(1e3 * t_host / args.steps, 3),
            "host_calls_per_step": round(calls_per_step, 1),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()                 # rank 0 may still be printing / timing the isolated kernel
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
