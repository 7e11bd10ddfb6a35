"""MC-dropout helpers of the reference's `utils_extra` on the HIP path (src/utils_extra.py:119-244).

  mc_infer(driver, image, T=10)        serving-level twin: T serve() calls stacked            (:119-139)
  mc_eval(mc_model, images, config)    model-level twin used by eval.py:108-112 / train_lib   (:142-198)
  stack_mcpred / get_mcuncert          stacking and mean / population std over the sample axis (:201-244)

`mc_eval` on an `efficientdet_keras.EfficientDetNet` of this package runs the T stochastic passes as ONE
launch sequence with the sample axis explicit (weights read once, everything upstream of the first dropout
site computed once per image) instead of T model calls; the result has the reference's structure - per
level [T,N,h,w,ch] for a head whose (or the global) rate is non-zero, else the deterministic output.
"""
import numpy as np


def mc_infer(driver, image, T=10):
    """T calls of `driver.serve(image)`, each output stacked on a new axis 0 (src/utils_extra.py:119-139)."""
    runs = [driver.serve(image) for _ in range(T)]
    return [np.stack([np.asarray(r[i]) for r in runs], axis=0) for i in range(len(runs[0]))]


def stack_mcpred(output):
    """[[level-0 samples], ..., [level-4 samples]] -> five arrays stacked on axis 0 (:201-217)."""
    return [np.stack([np.asarray(x) for x in lvl], axis=0) for lvl in output]


def get_mcuncert(output):
    """(mean, population std) over axis 0 of each of the five stacked arrays (:220-244); sequential float32 sums,
    the order the HIP aggregate kernel and the oracle use."""
    means, stds = [], []
    for x in output:
        x = np.asarray(x, dtype=np.float32)
        acc = x[0].copy()
        for t in range(1, x.shape[0]):
            acc = acc + x[t]
        m = acc / np.float32(x.shape[0])
        v = np.zeros_like(m)
        for t in range(x.shape[0]):
            d = x[t] - m
            v = v + d * d
        means.append(m)
        stds.append(np.sqrt(v / np.float32(x.shape[0])))
    return means, stds


def mc_eval(mc_model, images, config):
    """[cls_outputs, box_outputs] with the MC samples stacked (src/utils_extra.py:142-198)."""
    if hasattr(mc_model, "mc_forward"):           # this package's EfficientDetNet: one run, explicit sample axis
        return mc_model.mc_forward(images)
    # any other callable with the reference's model signature: the reference's loop
    stack_c = bool(config.mc_classheadrate or config.mc_dropoutrate)
    stack_b = bool(config.mc_boxheadrate or config.mc_dropoutrate)
    cls_runs, box_runs = [], []
    cls = box = None
    for _ in range(config.mc_dropoutsamp):
        cls, box = mc_model(images, training=False)
        cls_runs.append(cls)
        box_runs.append(box)
    cls_concat = stack_mcpred([[r[l] for r in cls_runs] for l in range(len(cls))]) if stack_c else cls
    box_concat = stack_mcpred([[r[l] for r in box_runs] for l in range(len(box))]) if stack_b else box
    return [cls_concat, box_concat]
