"""Pretraining entry point with the reference's command line (code/run_pretrain.py):

    python run_pretrain.py --pretrain --simu-exp --gpu-id 0,                      # one GPU
    python run_pretrain.py --pretrain --simu-exp --gpu-id 0,1,2,3,4,5,6,7 [--use-amp]       # eight GPUs, the reference's own form
    torchrun --nproc-per-node 8 run_pretrain.py --pretrain --simu-exp --gpu-id 0,1,2,3,4,5,6,7 [--use-amp]     # same thing

Only the ``--pretrain --simu-exp`` branch (fixed pre-generated simulated segments) is implemented - the path BASELINE.json
names.  Multi-GPU = one process per GPU over RCCL, not DataParallel (code/learner.py:25-31): with more than one id in ``--gpu-id``
and no launcher in the environment this entry point starts its own ranks (launch.py) before anything touches the GPU, like the
reference's single command (code/run_pretrain.py:204-205); per-epoch scalars go to a JSONL log (tensorboardX is optional and absent
here).
"""
import json
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import sarssl_boot  # noqa: E402,F401


def main(argv=None):
    from sar_ssl_amd.opt import opt_pretrain
    opts = opt_pretrain()
    args = opts.parse(argv)
    dirs = opts.dir()
    from sar_ssl_amd import launch
    gpu_ids = launch.parse_gpu_ids(args.gpu_id)
    if not launch.launched():
        if len(gpu_ids) > 1 and not args.no_cuda and not args.test:
            # one rank per listed GPU; this process only waits for them (it has not initialised HIP and never will)
            raise SystemExit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), len(gpu_ids),
                                                gpu_ids=gpu_ids))
        os.environ["HIP_VISIBLE_DEVICES"] = ",".join(gpu_ids[:1]) if gpu_ids else ""

    import torch
    from sar_ssl_amd import dataset as at_dataset, learner as at_learner, model as at_model, dist as sdist
    from sar_ssl_amd.common.utils import set_seed, set_random_seed, create_learning_rate_schedule, get_nparams, save_config_to_file

    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("run_pretrain.py (sar_ssl_amd) needs an MI355X GPU: the HIP path has no CPU fallback")
    if not (args.pretrain or args.test) or not args.simu_exp:
        raise SystemExit("only `--pretrain --simu-exp` and `--test --simu-exp` are implemented on this path")
    rank, world, local = sdist.init_from_env()
    device = torch.device("cuda", local)
    set_seed(args.seed)
    if rank == 0:
        os.makedirs(dirs["log_pretrain"], exist_ok=True)
        save_config_to_file([{k: v for k, v in args.__dict__.items()}, dirs], os.path.join(dirs["log_pretrain"], "config.json"))

    fs, T = args.acoustic_setting["fs"], args.acoustic_setting["T"]
    seeds = {"train": int(args.seed + 4e8), "val": int(args.seed + 1e8), "test": int(args.seed + 1)}
    win_len, nfft, win_shift_ratio, fre_used_ratio = 512, 512, 0.5, 1                       # code/run_pretrain.py:67-72
    nf = nfft // 2
    nt = int((T * fs - win_len * (1 - win_shift_ratio)) / (win_len * win_shift_ratio))
    net = at_model.SARSSL(sig_shape=(nf, nt, 2, 2), pretrain=True, device=device)
    nparam, nparam_sum = get_nparams(net, param_key_list=["spec_encoder", "spat_encoder", "decoder"])
    if rank == 0:
        print(f"T: {T:.3f}, nt: {nt}, nf: {nf}; # Parameters (M): {nparam_sum:.2f}")

    if args.test:
        return run_test(args, dirs, net, device, fs, (win_len, win_shift_ratio, nfft, fre_used_ratio))

    data_num = {"train": 5120 * 100, "val": 4000 * 2}
    native = os.environ.get("SARSSL_NATIVE_LOADER", "1") != "0"
    sampler = None
    if native:
        # threaded int16 reader -> pinned buffers -> copy stream (dataset.PcmSegmentLoader); same file listing, same
        # rank-strided shuffling as DataLoader + DistributedSampler below
        nsample = int(T * fs)
        f_train = at_dataset.segment_files(dirs["micsig_simu_pretrain"])[: data_num["train"]]
        f_val = at_dataset.segment_files(dirs["micsig_simu_preval"])[: data_num["val"]]
        dl_train = at_dataset.PcmSegmentLoader(f_train, args.bs[0], nsample, 2, fs=fs, shuffle=True, seed=args.seed, rank=rank, world=world,
                                               drop_last=(world > 1), device=device, nthreads=max(1, args.workers))
        dl_val = at_dataset.PcmSegmentLoader(f_val, args.bs[1], nsample, 2, fs=fs, device=device, nthreads=max(1, args.workers))
    else:
        ds_train = at_dataset.FixMicSigDataset(data_dir=dirs["micsig_simu_pretrain"], load_anno=False, load_dp=False, fs=fs,
                                               dataset_sz=data_num["train"], transforms=None, raw_pcm=True)
        ds_val = at_dataset.FixMicSigDataset(data_dir=dirs["micsig_simu_preval"], load_anno=False, load_dp=False, fs=fs,
                                             dataset_sz=data_num["val"], transforms=None, raw_pcm=True)
        kwargs = {"num_workers": args.workers, "pin_memory": True}
        sampler = torch.utils.data.distributed.DistributedSampler(ds_train, num_replicas=world, rank=rank, shuffle=True,
                                                                  seed=args.seed) if world > 1 else None
        dl_train = torch.utils.data.DataLoader(ds_train, batch_size=args.bs[0], shuffle=(sampler is None), sampler=sampler,
                                               drop_last=(world > 1), **kwargs)
        dl_val = torch.utils.data.DataLoader(ds_val, batch_size=args.bs[1], shuffle=False, **kwargs)

    learner = at_learner.STFTLearner(net, win_len=win_len, win_shift_ratio=win_shift_ratio, nfft=nfft,
                                     fre_used_ratio=fre_used_ratio, fs=fs, task=None, ch_mode="M")
    if world > 1:
        learner.mul_gpu()
    learner.cuda()
    if args.use_amp:
        learner.amp()
    if args.checkpoint_start:
        learner.resume_checkpoint(checkpoints_dir=dirs["log_pretrain"], from_latest=True, as_all_state=True)
    lr_schedule = create_learning_rate_schedule(total_steps=args.nepoch, base=args.lr, decay_type="cosine", warmup_steps=1,
                                                linear_end=1e-6)
    log = open(os.path.join(dirs["log_pretrain"], "scalars.jsonl"), "a") if rank == 0 else None
    for epoch in range(learner.start_epoch, args.nepoch + 1):
        lr = float(lr_schedule(epoch))
        set_random_seed(seeds["train"] + epoch + 1000003 * rank)                              # per-rank mask / dropout streams
        if sampler is not None:
            sampler.set_epoch(epoch)
        if native:
            dl_train.set_epoch(epoch)
        loss_train, diff_train, _ = learner.pretrain_epoch(dl_train, lr=lr, epoch=epoch, return_diff=True)
        set_random_seed(seeds["val"])
        if world > 1:                                       # all ranks validate (and rank 0 saves) rank 0's BatchNorm statistics
            sdist.broadcast_buffers(net)
        loss_val, diff_val, _ = learner.pretest_epoch(dl_val, return_diff=True)
        if world > 1:                                       # one decision for early stopping / best epoch on every rank
            loss_val, diff_val = sdist.agree([loss_val, diff_val], device=device)
        stop_flag, is_best = learner.early_stopping(current_score=-loss_val, patience=100)
        learner.save_checkpoint(epoch=epoch, checkpoints_dir=dirs["log_pretrain"], is_best_epoch=is_best, save_extra_hist=True)
        if rank == 0:
            rec = {"epoch": epoch, "lr": lr, "loss_train": loss_train, "diff_train": diff_train, "loss_val": loss_val,
                   "diff_val": diff_val, "nparam_M": nparam_sum}
            print(json.dumps(rec), flush=True)
            log.write(json.dumps(rec) + "\n"); log.flush()
        if stop_flag:
            break
    if rank == 0:
        print("\nPre-Training finished\n")


def run_test(args, dirs, net, device, fs, stft_cfg):
    """`--test` (code/run_pretrain.py:405-483): best checkpoint -> loss on the pretest set ('all'), or per-instance export of
    mask / prediction / target spectrograms (.mat) and reconstructed waveforms (.wav through the HIP inverse STFT) ('ins')."""
    import numpy as np
    import scipy.io
    import torch
    from sar_ssl_amd import dataset as at_dataset, learner as at_learner
    from sar_ssl_amd.common.utils import set_seed
    win_len, win_shift_ratio, nfft, fre_used_ratio = stft_cfg
    learner = at_learner.STFTLearner(net, win_len=win_len, win_shift_ratio=win_shift_ratio, nfft=nfft, fre_used_ratio=fre_used_ratio,
                                     fs=fs, task=None, ch_mode="M")
    learner.cuda()
    if args.use_amp:
        learner.amp()
    epoch = learner.load_checkpoint_best(checkpoints_dir=dirs["log_pretrain"], as_all_state=True)
    kwargs = {"num_workers": args.workers, "pin_memory": True}
    if args.test_mode == "all":
        set_seed(args.seed)
        ds = at_dataset.FixMicSigDataset(data_dir=dirs["micsig_simu_pretest"], load_anno=False, load_dp=False, fs=fs, dataset_sz=4000,
                                         transforms=None)
        dl = torch.utils.data.DataLoader(ds, batch_size=args.bs[2], shuffle=False, **kwargs)
        loss_test, diff_test, _ = learner.pretest_epoch(dl, return_diff=True)
        print("Test loss: {:.4f}".format(loss_test))
        print(json.dumps({"epoch": epoch, "loss_test": loss_test, "diff_test": diff_test}), flush=True)
        return
    for dir_pretest in dirs["micsig_simu_pretest_ins"]:
        set_seed(args.seed)
        ds = at_dataset.FixMicSigDataset(data_dir=dirs["micsig_simu_pretest_ins"], load_anno=False, load_dp=True, dataset_sz=None, fs=fs,
                                         transforms=None)
        dl = torch.utils.data.DataLoader(ds, batch_size=len(ds), shuffle=False, **kwargs)
        loss_test, diff_test, vis, res = learner.pretest_epoch(dl, return_diff=True, return_eval=True)
        name = dir_pretest.split("/")[-1]
        print(name, "Test loss: {:.4f}".format(loss_test))
        data_path = dirs["log_pretrain"] + "/test_result/"
        os.makedirs(data_path, exist_ok=True)
        rt = dir_pretest.split("T")[-1]
        for ins_idx in range(len(ds)):
            for key, sig in (("pred", res["sig_pred"]), ("tar", res["sig_tar"])):
                pcm = (sig[ins_idx].clamp(-1, 1) * 32767.0).round().to(torch.int16).cpu().numpy()
                at_dataset.write_wav_pcm16(data_path + "rt" + rt + "_ins" + str(ins_idx) + "_epoch" + str(epoch) + "_test_" + key + ".wav", pcm, fs)
        scipy.io.savemat(data_path + "rt" + rt + "_ins" + "_epoch" + str(epoch) + "_test.mat",
                         {"mask": vis["mask"].cpu().numpy(), "pred": vis["pred"].cpu().numpy(), "tar": vis["tar"].cpu().numpy(),
                          "pesq": res["pesq"].cpu().numpy(), "pesq_mask_ch": res["pesq_mask_ch"].cpu().numpy(),
                          "mse": float(res["mse"]), "mse_mask": float(res["mse_mask"]), "mse_mask_ch": float(res["mse_mask_ch"])})


if __name__ == "__main__":
    main()
