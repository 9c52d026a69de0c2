"""DeviceManager — same contract as reference config/device_manager.py:9-84.

'auto' | 'cuda' | 'cpu'  ->  'cuda' or 'cpu' (string, cached on first call).  On ROCm builds of
PyTorch the HIP device is addressed as 'cuda', so the user-visible name does not change on MI355X.
A 'cuda' request without a GPU degrades to 'cpu' with the reference's warning (:31-36); the flow
engine itself then refuses to load (it has no CPU path) — see processing/videoflow_core.py here.
"""
from typing import Any, Dict

import torch


class DeviceManager:
    def __init__(self):
        self._device = None
        self._device_info = None

    def get_device(self, device_preference: str = "auto") -> str:
        if self._device is None:
            have_gpu = torch.cuda.is_available()
            if device_preference == "cpu":
                self._device = "cpu"
            elif device_preference == "cuda" and not have_gpu:
                print("Warning: CUDA requested but not available, falling back to CPU")
                self._device = "cpu"
            else:
                self._device = "cuda" if have_gpu else "cpu"
        return self._device

    def get_device_info(self) -> Dict[str, Any]:
        if self._device_info is None:
            dev = self.get_device()
            info = {"device": dev, "cuda_available": torch.cuda.is_available()}
            if dev == "cuda" and info["cuda_available"]:
                props = torch.cuda.get_device_properties(0)
                gb = props.total_memory / (1024 ** 3)
                info.update(gpu_name=torch.cuda.get_device_name(0), gpu_memory_gb=gb,
                            gpu_memory_formatted=f"{gb:.1f} GB")
            self._device_info = info
        return self._device_info

    def print_device_info(self):
        info = self.get_device_info()
        print(f"CUDA available: {info['cuda_available']}")
        if info["device"] == "cuda":
            print(f"GPU: {info['gpu_name']}")
            print(f"GPU Memory: {info['gpu_memory_formatted']}")
        else:
            print("Using CPU for processing")

    def reset(self):
        self._device = None
        self._device_info = None
