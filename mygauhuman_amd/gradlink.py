"""Gradients that meet INSIDE a kernel instead of in autograd's accumulation.

In one render() frame three tensors receive two gradients each, and autograd adds every such pair with a kernel of its own
(3 x ~4.7 us of a 0.68 ms frame):
  * the posed positions: from the rasterizer (dL_dmeans3D) and from the attribute kernel (view direction, covariance);
  * the raw quaternion leaf: directly from the attribute kernel (rot_cov) and through the normalisation of the activations;
  * the albedo activation: it is also the roughness (scene/gaussian_model.py:197-199).
The third is an aliasing rule of the attribute kernel (attributes.py).  For the first two the Function whose backward runs FIRST
parks its gradient here and returns None for that input, and the one that runs LATER hands the parked tensor to its kernel as an
accumulation input (gsr_frame_attributes_backward_acc / gsr_model_activations_backward_acc).

A FrameLink lives for one render() call.  A gradient is only ever parked when, at FORWARD time, the later Function has registered
that it will run and consume it (same storage, gradient required); anything else keeps autograd's own accumulation.
"""
import contextlib

_CURRENT = None


class FrameLink:
    __slots__ = ("attr_means_ptr", "means_grad", "act_rot_in_ptr", "act_rot_out_ptr", "rot_grad")

    def __init__(self):
        self.attr_means_ptr = None   # data_ptr of the means3D the attribute kernel of this frame consumes (it requires grad)
        self.means_grad = None       # parked by the rasterizer's backward
        self.act_rot_in_ptr = None   # data_ptr of the raw quaternion the activations of this frame normalise (it requires grad)
        self.act_rot_out_ptr = None  # ... and of the normalised quaternion they return
        self.rot_grad = None         # parked by the attribute kernel's backward


def current():
    return _CURRENT


@contextlib.contextmanager
def frame_link(enabled=True):
    """with frame_link() as link: ... the forward of one frame ...   (link is None when disabled)"""
    global _CURRENT
    prev, _CURRENT = _CURRENT, (FrameLink() if enabled else None)
    try:
        yield _CURRENT
    finally:
        _CURRENT = prev
