import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import felz_ref as FR
from unscene3d_amd import _lib
print("count before anything:", _lib.lib.usc_device_count(), _lib.last_error())
ref = FR.reference_module()
print("after ref module:", _lib.lib.usc_device_count(), _lib.last_error())
from unscene3d_amd import felzenszwalb_cpp as FZ
try:
    FZ.merge_host(np.array([0, 7], np.int32), np.array([1, 2], np.int32), np.zeros(2, np.float32), 3, 0.005, 20)
except RuntimeError as e:
    print("expected:", e)
print("after failing merge:", _lib.lib.usc_device_count(), _lib.last_error())
import torch
print(torch.cuda.is_available())
print("after torch:", _lib.lib.usc_device_count(), _lib.last_error())
