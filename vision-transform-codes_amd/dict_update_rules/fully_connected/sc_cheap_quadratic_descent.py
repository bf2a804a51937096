"""
Hessian-diagonal-scaled dictionary update for fully-connected sparse coding.

Drop-in for vision_transform_codes/dict_update_rules/fully_connected/
sc_cheap_quadratic_descent.py:11-48: the scaled gradient of atom i is divided
by (hessian_diagonal[i] + lowest_code_val) before it is subtracted.
"""
from dict_update_rules.fully_connected import _common


def run(images, dictionary, codes, hessian_diagonal, stepsize=0.001,
        num_iters=1, lowest_code_val=0.001, normalize_dictionary=True):
  """
  images (b, n), dictionary (s, n) [updated IN PLACE], codes (b, s),
  hessian_diagonal (s,) [read only; the trainer maintains its EMA].
  Returns None.
  """
  _common.descend(images, dictionary, codes, stepsize, num_iters,
                  normalize_dictionary, hessian_diagonal=hessian_diagonal,
                  lowest_code_val=lowest_code_val)
