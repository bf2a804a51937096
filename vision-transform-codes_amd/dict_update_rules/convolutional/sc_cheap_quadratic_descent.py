"""
Hessian-diagonal-scaled dictionary update for convolutional sparse coding.

Drop-in for vision_transform_codes/dict_update_rules/convolutional/
sc_cheap_quadratic_descent.py:14-79: each kernel's gradient is divided by
(hessian_diagonal + lowest_code_val) before the global rescale.
"""
from dict_update_rules.convolutional import _common


def run(images_padded, dictionary, codes, hessian_diagonal, kernel_stride,
        padding_dims, stepsize=0.001, num_iters=1, lowest_code_val=0.001,
        normalize_dictionary=True):
  """See sc_steepest_descent.run; hessian_diagonal (s,) is read only."""
  _common.descend(images_padded, dictionary, codes, kernel_stride,
                  padding_dims, stepsize, num_iters, normalize_dictionary,
                  hessian_diagonal=hessian_diagonal,
                  lowest_code_val=lowest_code_val)
