"""Parity class of every numeric mode: the gates that tests/ ASSERT against the reference's golden vectors on MI355X, in one place so
that bench.py's `parity_class` prints exactly what is asserted (round-3 verdict: the line claimed 1e-2 per bin while the test gate
was 3e-2).  north_star: "within 1e-3 relative on the reconstruction loss and per-bin magnitudes ... 100-step recon-loss curve within
1e-3 of reference".

Quantities (fixture F3, tests/test_gpu_model.py::test_fullsize_forward_backward; F5 / F12, tests/test_gpu_train.py):
  loss          |loss / loss_ref - 1|, full-size forward, eval and train mode
  per_bin_max   max over F3's 2 048 sampled `pred` bins of |pred - pred_ref| / max|pred_ref|
  per_bin_rms   rms of the same deviations / max|pred_ref|
  grad_norm     worst per-parameter gradient L2 norm, relative
  bn_running    BatchNorm running statistics after one train-mode forward
  curve100      max relative deviation of the 100-step loss curve (dropout off, replayed masks)
`measured` = what the MI355X run of round 4 measured (eval / train), for the reader; gates sit at <= 5x measured and never above the
class the mode claims.

Round 5, per_bin_max of the timed mode: it sits AT north_star's 1e-3, not inside it - 9.1e-4 in eval mode, 1.21e-3 in train mode (maximum
of 2 048 bins whose rms deviation is 2-3e-4 of range), gated at 1.5e-3 and stated so in `parity_class`.  A two-pass fp16 split of a SUBSET of
the contractions does not change that (profiles/r05_operand_rounding_study.txt: decoder in f32 - eval 6.5e-4 but train 1.12e-3; every
Conformer Linear in f32 operands AND f32 storage - 6.3e-4 / 7.7e-4; with 16-bit activation storage no family subset moves the train-mode
rms of 3.1e-4): the deviation is the fp16 rounding of the STORED activations between layers, amplified by the train-mode BatchNorm
statistics, not the operand rounding of a few products.  Inside 1e-3 with margin costs f32 storage (`fp32`, 61 ms).  What the timed mode
claims: loss / curve / rms inside 1e-3, per-bin maximum 1.5e-3.
"""

GATES = {
    # round 6 - fp16 CNN stem + f32 residual stream in the Conformer blocks / decoder (f32 activations and weights as fp16 pairs on the
    # matrix cores): the mode that meets north_star's 1e-3 per bin at 16-bit matrix-core speed
    # grad_norm: worst parameter overall - always one whose gradient is a near-cancelling sum formed by the bf16 backward pass (BatchNorm
    # betas of the stem, the attention module's u / v biases: measured 0.7-1.1e-2 on F3 / F14, same as the fp16 mode's); grad_norm_body: the
    # parameters outside the stem and those biases, whose gradients flow along the f32 stream (measured 2.1e-3 on F14 / F3 eval, 3.4e-3 F3 train, <= 4.5e-3 on F15; fp16 mode: 1.7e-2)
    "hybrid": dict(loss=1e-3, per_bin_max=1e-3, per_bin_rms=3e-4, grad_norm=1.5e-2, grad_norm_body=5e-3, bn_running=5e-4, curve100=1e-3,
                   measured=dict(loss=(1.1e-6, 1.0e-6), per_bin_max=(3.3e-4, 8.5e-4), per_bin_rms=(7.6e-5, 2.2e-4), grad_norm=(1.03e-2, 9.6e-3),
                                 grad_norm_body=(2.1e-3, 3.4e-3), curve100=1.9e-4,
                                 b64_f13=dict(per_bin_max=(2.6e-4, 6.5e-4), per_bin_rms=(7.0e-5, 1.6e-4)))),
    # fp16 forward / bf16 backward - the mode bench.py times since round 4
    "fp16": dict(loss=1e-3, per_bin_max=1.5e-3, per_bin_rms=5e-4, grad_norm=2.5e-2, bn_running=5e-4, curve100=1e-3,
                 measured=dict(loss=(6.1e-6, 9.6e-7), per_bin_max=(9.1e-4, 1.21e-3), grad_norm=(1.5e-2, 7.9e-3), curve100=1.9e-4)),
    # bf16 throughout - the timed mode of rounds 1-3
    "bf16": dict(loss=2e-3, per_bin_max=3e-2, per_bin_rms=1e-2, grad_norm=6e-2, bn_running=5e-3, curve100=1e-3,
                 measured=dict(loss=(3.8e-4, 1.2e-4), per_bin_max=(5.8e-3, 1.1e-2), grad_norm=(2.3e-2, 3.5e-2), curve100=1.6e-4)),
    # f32 storage, every contraction as three split-bf16 MFMA passes
    "fp32": dict(loss=1e-3, per_bin_max=1e-3, per_bin_rms=1e-3, grad_norm=5e-3, bn_running=1e-4, curve100=1e-3,
                 measured=dict(loss=(2.3e-7, 2.4e-7), per_bin_max=(9.7e-6, 1.5e-5), grad_norm=(3.5e-4, 3.9e-4), curve100=1.9e-4)),
    # f32 storage, one bf16 MFMA pass (round-3 experiment)
    "fp32_1pass": dict(loss=2e-3, per_bin_max=3e-2, per_bin_rms=1e-2, grad_norm=6e-2, bn_running=5e-3, curve100=None, measured=dict()),
}


def parity_class(precision):
    """What bench.py prints for the timed mode."""
    g = GATES[precision]
    return {"loss_vs_reference": g["loss"], "loss_curve_100_steps": g["curve100"], "per_bin_pred_of_range_max": g["per_bin_max"],
            "per_bin_pred_of_range_rms": g["per_bin_rms"], "per_parameter_grad_norm": g["grad_norm"], "measured_eval_train": g["measured"],
            "north_star": "1e-3 on the loss, the per-bin magnitudes and the 100-step curve",
            "meets_north_star": ({"loss": True, "curve100": True, "per_bin_rms": True,
                                  "per_bin_max": "NO in train mode: measured 9.1e-4 (eval) / 1.21e-3 (train) of range, gated at 1.5e-3 - the maximum over "
                                                 "2 048 bins of rounding noise with rms 2-3e-4; the fp32 mode measures 1.5e-5"}
                                 if g["per_bin_max"] > 1e-3 else {"loss": True, "curve100": g["curve100"] is not None, "per_bin_rms": True, "per_bin_max": True}),
            "pinned_by": "tests/test_gpu_model.py::test_fullsize_forward_backward (F3), tests/test_gpu_train.py (F5, F12), "
                         "tests/test_gpu_graph.py (B = 64 captured step vs the fp32 mode, F13 reference forward at B = 64)"}
