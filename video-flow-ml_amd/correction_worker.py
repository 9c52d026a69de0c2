"""Flow quality map: how well frame 2, warped back along a flow field, matches frame 1.

API mirror of the one GPU function of reference correction_worker.py, `generate_quality_frame_gpu` (:175-208): same
arguments and result (numpy uint8 [H,W,3]: green = match above the threshold, red = below, full red = the vector
leaves the image; fields at a cached LOD's resolution are resized inside).  The work is one HIP kernel
(`vfml_flow_quality_map`) instead of ~25 torch ops; `quality_frame_resident` is the same for inputs that already
live in HBM.  The phase-correlation / template-matching correction search of that file is OpenCV on the host and is
not part of this build."""
import numpy as np
import torch


def quality_frame_resident(frame1, frame2, flow, good_quality_threshold):
    """Device tensors in (uint8 [H,W,3] x2, float32 [fh,fw,2]), device uint8 [H,W,3] out."""
    from vfml import hip
    return hip.flow_quality_map(frame1, frame2, flow, good_quality_threshold)


def generate_quality_frame_gpu(frame1, frame2, flow, device, good_quality_threshold):
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"generate_quality_frame_gpu: device {device} - the quality map is a HIP kernel, there is no "
                           "CPU path in this build")
    f1 = torch.from_numpy(np.ascontiguousarray(frame1)).to(device)
    f2 = torch.from_numpy(np.ascontiguousarray(frame2)).to(device)
    fl = torch.from_numpy(np.ascontiguousarray(flow, dtype=np.float32)).to(device)
    return quality_frame_resident(f1, f2, fl, good_quality_threshold).cpu().numpy()
