"""carparkingmaps_amd -- MI355X (gfx950) implementation of the CarParkingMaps HMM traffic-flow
sampler path (initializestates -> solveinitialvalueproblem -> resampling -> zone x hour
histogram, fed by the p_drive / p_dest tables), behind the reference's own call surface.

The compute lives in csrc/libcpm_hip.so (hand-written HIP, C ABI in include/cpm.h).  There is
no CPU fallback; importing the package works without a GPU, computing does not.
"""
from . import _lib
from ._lib import (CPM_KERNEL_AUTO, CPM_KERNEL_CAR, CPM_KERNEL_ZONE_GROUPED, CPM_KERNEL_ZONE_LDS,
                   CpmError)
from .sampler import Sampler, device_count, device_info
from .reference_api import (DeviceArray, Params, averagedrivingtime, correctparameters, createdatamatrix, createpdestin,
                            createpdrive, createresultsdirectory, initializestates, invalidate, params, processgeodata, release,
                            resampling, run_dataset, saveparameters, saveresults, solveinitialvalueproblem,
                            zone_hour_counts)

__all__ = [
    "Sampler", "device_count", "device_info", "CpmError", "CPM_KERNEL_AUTO", "CPM_KERNEL_CAR",
    "CPM_KERNEL_ZONE_LDS", "CPM_KERNEL_ZONE_GROUPED", "Params", "params", "createpdrive", "createpdestin", "initializestates",
    "solveinitialvalueproblem", "resampling", "averagedrivingtime", "correctparameters", "saveresults",
    "zone_hour_counts", "run_dataset", "release", "createdatamatrix", "processgeodata", "createresultsdirectory",
    "saveparameters", "DeviceArray", "invalidate",
]
