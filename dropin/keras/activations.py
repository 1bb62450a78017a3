def softmax(x, axis=-1):
    """Marker used in checkpoint custom_objects (make_submission.py:69); softmaxes run inside the tail kernels."""
    raise NotImplementedError("softmax is fused into the network tail kernels")
