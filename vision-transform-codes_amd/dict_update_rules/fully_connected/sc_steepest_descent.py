"""
Steepest-descent dictionary update for fully-connected sparse coding on MI355X.

Drop-in for vision_transform_codes/dict_update_rules/fully_connected/
sc_steepest_descent.py:9-41.  In a data-parallel run (vtc_hip.parallel.enable)
the gradient sum is all-reduced over ranks before it is applied.
"""
from dict_update_rules.fully_connected import _common


def run(images, dictionary, codes, stepsize=0.001, num_iters=1,
        normalize_dictionary=True):
  """
  D <- D - stepsize * C^T (C D - X) / b, then unit-norm rows; num_iters times.

  images (b, n), dictionary (s, n) [updated IN PLACE], codes (b, s), all
  float32 tensors on a HIP device.  Returns None.
  """
  _common.descend(images, dictionary, codes, stepsize, num_iters,
                  normalize_dictionary)
