"""
Steepest-descent dictionary update for convolutional sparse coding on MI355X.

Drop-in for vision_transform_codes/dict_update_rules/convolutional/
sc_steepest_descent.py:12-72.
"""
from dict_update_rules.convolutional import _common


def run(images_padded, dictionary, codes, kernel_stride, padding_dims,
        stepsize=0.001, num_iters=1, normalize_dictionary=True):
  """
  images_padded (b, c, h, w), dictionary (s, c, kh, kw) [updated IN PLACE],
  codes (b, s, code_h, code_w).  The gradient is rescaled to the Frobenius
  norm of the dictionary before the step.  Returns None.
  """
  _common.descend(images_padded, dictionary, codes, kernel_stride,
                  padding_dims, stepsize, num_iters, normalize_dictionary)
