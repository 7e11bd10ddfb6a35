"""MI355X-native hot path of uncertainty-detection-autolabeling.

Import this package as ``uda_amd`` (see ``uda_amd/__init__.py``).
Modules:
  hparams_config  mirror of the reference's config object for this path
  dataset_data    dataset letter -> label map / raw image shape
  weights         reference-named weight sets (random init per reference initialisers)
  plan            network topology -> flat op list + packed weight blob for the C-ABI
  capi            ctypes binding of include/uda_hip.h (csrc/libuda_hip.so)
  infer_lib       ServingDriver-shaped boundary (serve / predict / benchmark)
  dist            image sharding + detection gather (torch.distributed / RCCL)
"""
