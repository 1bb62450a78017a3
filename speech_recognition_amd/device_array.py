"""Device-resident batch containers handed from the AudioProcessor generator to the Model.

The reference's generators yield float64 NumPy arrays (input_data.py:450-451) that Keras casts to
f32 and copies host->device every step.  Here batches are produced on the GPU and stay
there: `DeviceArray` wraps the tensor plus the event that marks it ready on the producer stream.
Anything that treats it as an array (np.asarray, iteration, slicing) gets the reference's float64
host view on demand, so reference-style host code keeps working.
"""
import numpy as np
import torch


class DeviceArray(object):
    __array_priority__ = 100

    def __init__(self, tensor, ready_event=None):
        self.tensor = tensor
        self.ready_event = ready_event

    # -- array-ish protocol ---------------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self.tensor.shape)

    @property
    def ndim(self):
        return self.tensor.dim()

    @property
    def dtype(self):
        return np.dtype(np.float64)   # what np.asarray() yields, like the reference's np.zeros buffers

    def __len__(self):
        return self.tensor.shape[0]

    def wait(self, stream=None):
        """Make `stream` (default: current) wait until the producer has finished writing."""
        if self.ready_event is not None:
            (stream or torch.cuda.current_stream()).wait_event(self.ready_event)
        return self.tensor

    def numpy(self):
        if self.ready_event is not None:
            self.ready_event.synchronize()
        return self.tensor.detach().cpu().numpy().astype(np.float64)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, idx):
        return DeviceArray(self.tensor[idx], self.ready_event)

    def __iter__(self):
        return iter(self.numpy())

    def __mul__(self, k):     # 1.2 * X style TTA in reference-like host code (make_submission.py:128)
        self.wait()
        return DeviceArray(self.tensor * float(k), None)

    __rmul__ = __mul__

    def __repr__(self):
        return "DeviceArray(shape=%s, device=%s)" % (self.shape, self.tensor.device)


class Labels(np.ndarray):
    """float64 one-hot label matrix (reference input_data.py:451) that also carries its f32 device
    copy so train_on_batch does not re-upload it."""

    def __new__(cls, host, device_tensor=None, ready_event=None):
        obj = np.asarray(host, dtype=np.float64).view(cls)
        obj.device_tensor = device_tensor
        obj.ready_event = ready_event
        return obj

    def __array_finalize__(self, obj):
        self.device_tensor = getattr(obj, 'device_tensor', None) if obj is not None and obj.shape == self.shape else None
        self.ready_event = getattr(obj, 'ready_event', None) if self.device_tensor is not None else None


def as_device_f32(x, device, stream=None):
    """DeviceArray / torch tensor / numpy -> f32 CUDA tensor usable on `stream`."""
    if isinstance(x, DeviceArray) or (isinstance(x, Labels) and x.device_tensor is not None):
        # The tensor was allocated and written on the producer's stream.  Make the consumer stream wait
        # for the producer, and tell the caching allocator about the second stream so the block is not
        # handed back to the producer while consumer kernels that read it are still queued.
        s = stream if stream is not None else torch.cuda.current_stream(device)
        t = x.tensor if isinstance(x, DeviceArray) else x.device_tensor
        if x.ready_event is not None:
            s.wait_event(x.ready_event)
        t.record_stream(s)
        return t
    if torch.is_tensor(x):
        return x.to(device=device, dtype=torch.float32).contiguous()
    a = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
    return torch.from_numpy(a).to(device, non_blocking=False)
