class DepthwiseConv2D(object):
    """Marker for checkpoint custom_objects (make_submission.py:9,66); the depthwise convolution is
    csrc/dwconv.hip."""
