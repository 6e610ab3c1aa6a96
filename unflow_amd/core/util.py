"""Mirror of src/e2eflow/core/util.py:21-26."""
from ..ops import downsample as downsample_ops


def downsample(tensor, num):
    """core/util.py:21-26.  The reference falls back to tf.image.resize_area for odd sizes; that path is not
    reached by the training configurations (384x512, 768x1024) and is not implemented."""
    _, height, width, _ = tensor.shape
    if height % num == 0 and width % num == 0:
        return downsample_ops(tensor, num)
    raise NotImplementedError("resize_area fallback for sizes not divisible by %d" % num)
